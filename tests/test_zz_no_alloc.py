"""The entry point that allocates nothing, camd_ks_pbs_batch_device_noalloc, against
camd_ks_pbs_batch_device: same inputs, same words out, on two calls in a row (the second with a
smaller batch, so that whatever the first call left in the fixed scratch would show)."""
import ctypes as C

import numpy as np
import pytest

import concrete_amd.parameters as P
from concrete_amd import _ffi
from concrete_amd.backend import CAMD_PBS_BLOCK, CAMD_PBS_WAVE, DeviceServerKey
from oracle import cpu_oracle as O
from conftest import get_keys

pytestmark = pytest.mark.gpu


def test_no_alloc_entry_point_matches_the_allocating_one(gpu_ctx):
    import torch

    p = P.PARAM_MESSAGE_2_CARRY_2
    ks = get_keys(p, 64, n_override=24)
    o = ks.o
    dk = DeviceServerKey.from_standard_keys(gpu_ctx, ks.device_params(), ks.bsk, ks.ksk, variant=CAMD_PBS_WAVE, torus_bits=64)
    luts = [o.generate_accumulator(f) for f in (lambda x: x % 4, lambda x: x // 4)]
    for l in luts:
        dk.upload_lut(l)
    delta = O.shortint_delta(p)
    base = np.stack([o.lwe_encrypt(ks.rng, ks.glwe, (i % 16) * delta, ks.lwe_std) for i in range(64)])
    stream = torch.cuda.current_stream().cuda_stream
    # first call: more ciphertexts than one round of persistent waves (the kernels' counters are
    # in use); second call: a smaller batch right behind it; third: a batch small enough for the
    # schedule that gives a ciphertext k+1 waves (at most one ciphertext per CU)
    for count in (2500, 300, 40):
        rs = np.random.RandomState(count)
        cts = base[rs.randint(0, 64, size=count)].copy()
        cts[:, :8] += rs.randint(0, 2 ** 40, size=(count, 8)).astype(np.uint64)   # distinct masks, same message
        ids = (np.arange(count) % 2).astype(np.int32)
        d_in = torch.from_numpy(cts.view(np.int64)).cuda()
        d_ids = torch.from_numpy(ids).cuda()
        want = torch.zeros_like(d_in)
        got = torch.zeros_like(d_in)
        dk.ks_pbs_device(d_in, d_ids, want, count, stream)
        dk.ks_pbs_device_noalloc(d_in, d_ids, got, count, stream)
        dk.check(stream)
        # The call allocates nothing: the device's free memory (hipMemGetInfo) is the same before
        # and after. Measured on a repeat of the call, because the HIP runtime itself sets up a
        # kernel's private-segment memory the first time that kernel is launched in a process.
        again = torch.zeros_like(d_in)
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info()[0]
        dk.ks_pbs_device_noalloc(d_in, d_ids, again, count, stream)
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free_before, "the call changed the device's free memory"
        assert torch.equal(again, want), count
        assert torch.equal(got, want), count
        # and the words are right, not merely equal: a sample against the oracle's decryption
        host = got.cpu().numpy().view(np.uint64)
        ref = o.ks_pbs_batch(ks.ksk, ks.fbsk, np.concatenate(luts), ids[:32].astype(np.uint32), cts[:32], threads=4)
        for i in range(32):
            assert O.shortint_decode(p, o.lwe_phase(ks.glwe, host[i])) == O.shortint_decode(p, o.lwe_phase(ks.glwe, ref[i]))
    # beyond the fixed scratch: refused before anything is launched
    too_many = (16 << 20) // ((24 + 1) * 8) + 1
    rc = _ffi.lib().camd_ks_pbs_batch_device_noalloc(dk._h, C.c_void_p(d_in.data_ptr()), None, C.c_void_p(got.data_ptr()),
                                                     too_many, C.c_void_p(stream))
    assert rc == -1 and b"too large" in _ffi.lib().camd_last_error()
    dk.close()
    # families whose scratch grows with the batch are refused as well
    dkb = DeviceServerKey.from_standard_keys(gpu_ctx, ks.device_params(), ks.bsk, ks.ksk, variant=CAMD_PBS_BLOCK, torus_bits=64)
    dkb.upload_lut(luts[0])
    rc = _ffi.lib().camd_ks_pbs_batch_device_noalloc(dkb._h, C.c_void_p(d_in.data_ptr()), None, C.c_void_p(got.data_ptr()),
                                                     4, C.c_void_p(stream))
    assert rc == -1
    dkb.close()
