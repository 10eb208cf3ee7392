"""GPU tests of the entry points that take a linear combination as the input of an operation
(camd_axpby_ks_pbs_batch_device, camd_axpby_pbs_ks_batch_device): the result must equal, word for
word, camd_lwe_axpby_device followed by the plain entry point - the same arithmetic, formed while
the first kernel reads instead of in a pass of its own. The plain entry points are what the parity
tests compare with the oracle. The key material is random: only the equality of two paths through
the same kernels is looked at here, so neither keys nor ciphertexts need to decrypt.

Reference operations this stands for: the shift-and-add ahead of a bivariate bootstrap
(concrete-shortint/src/engine/server_side/mod.rs:182-195) and a gate's combination ahead of
bootstrap_keyswitch (concrete-boolean/src/engine/mod.rs:364-405)."""
import os

import numpy as np
import pytest

import concrete_amd.parameters as P
from concrete_amd._ffi import CAMD_PBS_AUTO, CAMD_PBS_BLOCK, CAMD_PBS_GENERIC, CamdError

pytestmark = pytest.mark.gpu


def random_key(ctx, params, bits, variant, seed):
    from concrete_amd.backend import DeviceServerKey

    n, k, N = params.lwe_dimension, params.glwe_dimension, params.polynomial_size
    dtype = np.uint64 if bits == 64 else np.uint32
    rs = np.random.RandomState(seed)
    hi = 2 ** (bits - 1)
    bsk = rs.randint(0, hi, size=(n, params.pbs_level, k + 1, k + 1, N), dtype=dtype) * dtype(2)
    ksk = rs.randint(0, hi, size=(k * N, params.ks_level, n + 1), dtype=dtype) * dtype(2)
    key = DeviceServerKey.from_standard_keys(ctx, params, bsk, ksk, variant=variant, torus_bits=bits)
    for t in range(3):
        lut = np.zeros((k + 1) * N, dtype=dtype)
        lut[k * N:] = rs.randint(0, hi, size=N, dtype=dtype)
        key.upload_lut(lut)
    return key


def random_rows(count, words, bits, seed, pad=0):
    """[count][words] view of a [count][words + pad] tensor: rows `words + pad` apart."""
    import torch

    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    if bits == 64:
        full = torch.randint(-2**63, 2**63 - 1, (count, words + pad), dtype=torch.int64, device="cuda", generator=g)
    else:
        full = torch.randint(-2**31, 2**31 - 1, (count, words + pad), dtype=torch.int32, device="cuda", generator=g)
    return full[:, :words]


def both_ways(key, order, bits, x, alpha, y, beta, body_add, ids):
    """(combined entry point, linear pass + plain entry point) on the same inputs."""
    import torch

    stream = torch.cuda.current_stream().cuda_stream
    count, words = x.shape
    fused = torch.empty((count, words), dtype=x.dtype, device="cuda")
    plain = torch.empty_like(fused)
    mixed = torch.empty_like(fused)
    if order == "ks_pbs":
        key.axpby_ks_pbs_device(x, alpha, y, beta, body_add, ids, fused, stream)
        key.ctx.lwe_axpby_device(mixed, x, alpha, y, beta, body_add, bits, stream)
        key.ks_pbs_device(mixed, ids, plain, count, stream)
    else:
        key.axpby_pbs_ks_device(x, alpha, y, beta, body_add, ids, fused, stream)
        key.ctx.lwe_axpby_device(mixed, x, alpha, y, beta, body_add, bits, stream)
        key.pbs_ks_device(mixed, ids, plain, count, stream)
    key.check(stream)
    return fused.cpu().numpy(), plain.cpu().numpy()


def lut_ids(count):
    import torch

    return torch.from_numpy((np.arange(count) % 3).astype(np.int32)).cuda()


@pytest.mark.parametrize("keyswitch", ["matrix cores", "f64 fma"])
def test_combination_read_by_the_keyswitch_2_2(gpu_ctx, keyswitch):
    """Shortint order at PARAM_MESSAGE_2_CARRY_2 (the wave kernel follows the keyswitch): both
    keyswitch kernels, strided operands, one operand, the output aliasing an operand."""
    import torch

    p = P.PARAM_MESSAGE_2_CARRY_2
    key = random_key(gpu_ctx, p, 64, CAMD_PBS_AUTO, 5)
    words = p.glwe_dimension * p.polynomial_size + 1
    old = os.environ.get("CAMD_KS_MFMA_MIN")
    if keyswitch == "f64 fma":
        os.environ["CAMD_KS_MFMA_MIN"] = "0"
    try:
        for count, pad in ((300, 7), (3, 0)):
            x = random_rows(count, words, 64, 100 + count, pad)
            y = random_rows(count, words, 64, 200 + count, 2 * pad)
            fused, plain = both_ways(key, "ks_pbs", 64, x, 5, y, 1, 0, lut_ids(count))
            assert np.array_equal(fused, plain)
            fused, plain = both_ways(key, "ks_pbs", 64, x, -1, None, 0, 3 << 59, None)
            assert np.array_equal(fused, plain)
        # in place: the output is the first operand
        x = random_rows(64, words, 64, 7).contiguous()
        y = random_rows(64, words, 64, 8)
        want = both_ways(key, "ks_pbs", 64, x, 4, y, 1, 1 << 40, None)[1]
        key.axpby_ks_pbs_device(x, 4, y, 1, 1 << 40, None, x, torch.cuda.current_stream().cuda_stream)
        assert np.array_equal(x.cpu().numpy(), want)
    finally:
        if old is None:
            os.environ.pop("CAMD_KS_MFMA_MIN", None)
        else:
            os.environ["CAMD_KS_MFMA_MIN"] = old


@pytest.mark.parametrize("pname,variant", [("DEFAULT_PARAMETERS", CAMD_PBS_AUTO), ("DEFAULT_PARAMETERS", CAMD_PBS_BLOCK),
                                           ("DEFAULT_PARAMETERS", CAMD_PBS_GENERIC), ("TFHE_LIB_PARAMETERS", CAMD_PBS_AUTO)])
def test_combination_read_by_the_bootstrap_boolean(gpu_ctx, pname, variant):
    """Boolean order (32-bit torus): the block and generic kernels read the combination; the wave
    kernels (N = 512 in both of its schedules, N = 1024) read plain batches only, so the library
    writes the combination out for them inside the call."""
    p = getattr(P, pname)
    key = random_key(gpu_ctx, p, 32, variant, 9)
    words = p.lwe_dimension + 1
    counts = ((1500, 3), (5, 0)) if variant == CAMD_PBS_AUTO else ((40, 1),)
    for count, pad in counts:
        x = random_rows(count, words, 32, 300 + count, pad)
        y = random_rows(count, words, 32, 400 + count, pad)
        for alpha, beta, plain_add in ((-1, -1, 1 << 29), (2, 2, 1 << 30)):     # NAND, XOR
            fused, plain = both_ways(key, "pbs_ks", 32, x, alpha, y, beta, plain_add, lut_ids(count))
            assert np.array_equal(fused, plain)
        fused, plain = both_ways(key, "pbs_ks", 32, x, 1, None, 0, 0, None)
        assert np.array_equal(fused, plain)


def test_combination_read_by_the_bootstrap_shortint_order_swapped(gpu_ctx):
    """pbs_ks on a 64-bit key whose bootstrap kernel is the N = 1024 wave kernel (written-out path)
    and on its block kernel (reads the combination)."""
    p = P.PARAM_MESSAGE_2_CARRY_2
    for variant in (CAMD_PBS_AUTO, CAMD_PBS_BLOCK):
        key = random_key(gpu_ctx, p, 64, variant, 21)
        words = p.lwe_dimension + 1
        x = random_rows(70, words, 64, 31, 5)
        y = random_rows(70, words, 64, 32, 0)
        fused, plain = both_ways(key, "pbs_ks", 64, x, 3, y, -2, 1 << 61, lut_ids(70))
        assert np.array_equal(fused, plain)


def test_combination_rejects_short_strides(gpu_ctx):
    import torch

    p = P.DEFAULT_PARAMETERS
    key = random_key(gpu_ctx, p, 32, CAMD_PBS_AUTO, 9)
    words = p.lwe_dimension + 1
    x = random_rows(4, words, 32, 1)
    out = torch.empty((4, words), dtype=torch.int32, device="cuda")
    with pytest.raises(ValueError):
        key.axpby_pbs_ks_device(x[:, :-1], 1, None, 0, 0, None, out, 0)
    from concrete_amd import _ffi
    import ctypes as C

    rc = _ffi.lib().camd_axpby_pbs_ks_batch_device(key._h, C.c_void_p(x.data_ptr()), words - 1, 1, None, 0, 0, 0, None,
                                                   C.c_void_p(out.data_ptr()), 4, None)
    assert rc != 0
    with pytest.raises(CamdError):
        _ffi.check(rc)
