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
