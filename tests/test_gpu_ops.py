"""Operation layer on the GPU, full PARAM_MESSAGE_2_CARRY_2 keys: the shortint operations and
FheUint8-shaped radix integers (4 blocks of 2 bits, the FheUint8 benchmark configuration) driven through the C ABI,
on host batches and on device-resident batches (linear part: camd_lwe_axpby_device)."""
import numpy as np
import pytest

import ops_common as OC

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu_keys(gpu_ctx):
    return OC.make_keys(n_override=None, backend="gpu", ctx=gpu_ctx)


def attributed(sks, tally):
    """Every wrong output the tally let through has to come from a bootstrap that also fails on the
    CPU oracle for the same input (the parameter set's own failure rate), never from the GPU path."""
    audit = sks.key
    before = len(audit.failed)

    def finish():
        checked, failed, own = audit.attribute()
        print(f"bootstraps audited so far: {checked}; failed {failed} (expected {checked * 2.0 ** -13.9:.1f} "
              f"at the parameter set's rate); not shared with the CPU oracle: {own}")
        assert own == 0, "a bootstrap failed on the GPU and not on the oracle"
        assert tally.bad == 0 or len(audit.failed) > before, "wrong outputs without a failed bootstrap"

    return finish


@pytest.mark.parametrize("device", [False, True])
def test_shortint_operation_layer(gpu_keys, device):
    cks, sks = gpu_keys
    tally = OC.Tally(per_check=1, total=2)
    finish = attributed(sks, tally)
    OC.check_shortint_ops(cks, sks, device=device, tally=tally)
    finish()


@pytest.mark.parametrize("reference_schedule", [False, True])
@pytest.mark.parametrize("device", [False, True])
def test_radix_fheuint8(gpu_keys, device, reference_schedule):
    """Once with the bootstraps that cannot change a decryption left out (the default), once with
    exactly the reference's (radix_parallel/mod.rs:44-56: two per block in every propagation step)."""
    cks, sks = gpu_keys
    tally = OC.Tally(per_check=2, total=8)
    finish = attributed(sks, tally)
    OC.check_radix_ops(cks, sks, count=48, device=device, tally=tally, reference_schedule=reference_schedule)
    finish()


@pytest.mark.parametrize("device", [False, True])
def test_gathered_operations(gpu_keys, device):
    """A multiplication, an addition with carries and a bitwise operation of different integers in
    one ``gather``: launches shared round by round, same decryptions."""
    cks, sks = gpu_keys
    tally = OC.Tally(per_check=2, total=4)
    finish = attributed(sks, tally)
    OC.check_gathered_operations(cks, sks, count=32, device=device, tally=tally)
    finish()


@pytest.mark.parametrize("device", [False, True])
def test_crt_integers(gpu_keys, device):
    """concrete-integer's CRT representation (basis [2, 3] for 2_2, crt/tests.rs:25-33)."""
    cks, sks = gpu_keys
    OC.check_crt_ops(cks, sks, count=48, device=device, rounds=10, tally=OC.Tally(per_check=2, total=8))


def test_lwe_axpby_strided_views(gpu_ctx):
    """The leveled kernel against numpy, on strided block views of a [count][blocks][words] batch."""
    import torch

    rs = np.random.RandomState(5)
    count, blocks, words = 37, 3, 2049
    x = rs.randint(0, 2 ** 63, size=(count, blocks, words), dtype=np.int64)
    y = rs.randint(0, 2 ** 63, size=(count, blocks, words), dtype=np.int64)
    dx, dy = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    out = torch.zeros_like(dx)
    alpha, beta, body = 5, (1 << 64) - 1, 0x123456789ABCDEF0
    gpu_ctx.lwe_axpby_device(out[:, 1, :], dx[:, 1, :], alpha, dy[:, 2, :], beta, body, 64, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    want = x[:, 1, :].astype(np.uint64) * np.uint64(alpha) + y[:, 2, :].astype(np.uint64) * np.uint64(beta)
    want[:, -1] += np.uint64(body)
    got = out.cpu().numpy().view(np.uint64)
    assert np.array_equal(got[:, 1, :], want)
    assert not got[:, 0, :].any() and not got[:, 2, :].any()
    # 32-bit torus, in place, no second operand
    z = rs.randint(0, 2 ** 31, size=(5, 587), dtype=np.int32)
    dz = torch.from_numpy(z).cuda()
    gpu_ctx.lwe_axpby_device(dz, dz, (1 << 32) - 1, None, 0, 1 << 29, 32, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    want32 = (np.uint32(0) - z.astype(np.uint32)).astype(np.uint32)
    want32[:, -1] += np.uint32(1 << 29)
    assert np.array_equal(dz.cpu().numpy().view(np.uint32), want32)
