"""Operation layer (shortint ops, radix integers) with the bootstraps evaluated on the CPU by the
oracle: covers the host logic that surrounds the GPU path. The LWE dimension is shortened so a
bootstrap costs milliseconds; the algorithms do not depend on it."""
import numpy as np

import ops_common as OC
from concrete_amd import shortint_ops as SO


def test_degree_helpers_match_exhaustive_search():
    for a in range(16):
        for b in range(16):
            assert SO.degree_after_bitand(a, b) == min(a, b)
            assert SO.degree_after_bitor(a, b) == max(max(a, b) | i for i in range(min(a, b) + 1))
            assert SO.degree_after_bitxor(a, b) == max(max(a, b) ^ i for i in range(min(a, b) + 1))
    assert SO.degree_after_left_shift(3, 1, 4) == 2 and SO.degree_after_left_shift(3, 2, 16) == 12


def test_shortint_operation_layer_on_oracle():
    cks, sks = OC.make_keys()
    OC.check_shortint_ops(cks, sks)


def test_deferred_bootstraps_share_one_launch():
    cks, sks = OC.make_keys()
    OC.check_deferred_is_one_launch(cks, sks)


def test_radix_layer_on_oracle():
    cks, sks = OC.make_keys()
    OC.check_radix_ops(cks, sks, count=5)


def test_gathered_operations_share_launches():
    cks, sks = OC.make_keys()
    OC.check_gathered_operations(cks, sks)


def test_crt_layer_on_oracle():
    cks, sks = OC.make_keys()
    OC.check_crt_ops(cks, sks, count=5)


def test_crt_operation_is_one_launch():
    """The bootstraps of a CRT operation over all residues and all integers are one launch."""
    from concrete_amd import integer as INT

    cks, sks = OC.make_keys()
    icks, isks = INT.ClientKey.from_shortint(cks), INT.ServerKey(sks)
    a, b = icks.encrypt_crt_many([1, 2, 5], [2, 3]), icks.encrypt_crt_many([5, 4, 3], [2, 3])
    before_l, before_b = sks.key.launches, sks.key.bootstraps
    isks.unchecked_crt_mul_assign(a, b)
    assert sks.key.launches == before_l + 1 and sks.key.bootstraps == before_b + 2 * 3
    assert icks.decrypt_crt(a).tolist() == [5, 2, 3]


def test_i_crt_matches_direct_search():
    from concrete_amd.integer_crt import i_crt

    for basis in ([2, 3], [2, 5, 7], [3, 7, 13], [2, 3, 5]):
        whole = int(np.prod(basis))
        for v in range(whole):
            assert i_crt(basis, [v % m for m in basis]) == v


def test_radix_layer_on_oracle_reference_schedule():
    cks, sks = OC.make_keys()
    OC.check_radix_ops(cks, sks, reference_schedule=True)


def test_reference_schedule_issues_the_reference_bootstraps():
    """smart_mul of two fresh 4-block integers under the reference's schedule: both operands fully
    propagated (radix_parallel/mul.rs:319-322: 2 x 4 blocks x 2 bootstraps), the 16 block products
    (mul.rs:157-181) and the propagations of the sum tree; the default schedule leaves out every
    bootstrap of a block that has no carry and the carry of a last block."""
    from concrete_amd import integer as INT

    cks, sks = OC.make_keys()
    icks = INT.RadixClientKey.__new__(INT.RadixClientKey)
    icks.key, icks.num_blocks = cks, 4
    counts = {}
    for ref in (False, True):
        isks = INT.ServerKey(sks, reference_schedule=ref)
        a, b = icks.encrypt_many([13, 200, 255]), icks.encrypt_many([7, 99, 255])
        before = sks.key.bootstraps
        r = isks.smart_mul(a, b)
        counts[ref] = (sks.key.bootstraps - before) // 3
        assert np.array_equal(icks.decrypt(r), np.array([13 * 7, 200 * 99, 255 * 255]) % 256)
    print("bootstraps per smart_mul of fresh FheUint8 operands:", counts)
    # 48 = 16 (both operands) + 16 (block products) + 16 (the last level of the sum tree adds two
    # integers of block degree 10, beyond max_degree 11: both are fully propagated; the first
    # level's sums, 5 + 5, need none)
    assert counts == {False: 22, True: 48}, counts


def test_radix_multiplication_launch_count():
    """All 16 block products of a 4-block multiplication are one launch, whatever the batch."""
    from concrete_amd import integer as INT

    cks, sks = OC.make_keys()
    icks = INT.RadixClientKey.__new__(INT.RadixClientKey)
    icks.key, icks.num_blocks = cks, 4
    isks = INT.ServerKey(sks)
    a, b = icks.encrypt_many([13, 200, 255]), icks.encrypt_many([7, 99, 255])
    before_l, before_b = sks.key.launches, sks.key.bootstraps
    r = isks.unchecked_mul(a, b)
    assert sks.key.launches == before_l + 1 and sks.key.bootstraps == before_b + 16 * 3
    assert np.array_equal(icks.decrypt(r), np.array([13 * 7, 200 * 99, 255 * 255]) % 256)
