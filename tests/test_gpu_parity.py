"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle.

Bars
  * integer steps (keyswitch, modulus switch, decomposition, rotation, sample extraction,
    bootstrap of trivial ciphertexts): bit-exact, every kernel family.
  * workgroup-per-ciphertext family (CAMD_PBS_GENERIC): performs the same sequence of IEEE-754
    double operations as the oracle, so whole output ciphertexts are compared bit for bit.
  * wave-per-ciphertext family (CAMD_PBS_WAVE): a different FFT factorisation, so f64 rounding
    differs in the last bits. A rounding difference of one unit in one CMUX flips some gadget
    digits of the next CMUX, after which the two masks are unrelated although both
    ciphertexts encrypt the same value; ciphertext words are therefore not comparable. The
    bar is the one the reference's own tests use (decrypt equality, for every ciphertext and
    look-up table) plus a bound on the phase error: |phase - expected| of the GPU output must
    stay below PHASE_TOL, and its variance within VARIANCE_RATIO of the oracle's.
"""
import numpy as np
import pytest

import concrete_amd.parameters as P
from concrete_amd.backend import CAMD_PBS_GENERIC, CAMD_PBS_WAVE, DeviceServerKey
from oracle import cpu_oracle as O
from conftest import get_keys

pytestmark = pytest.mark.gpu

# 2_2 output noise is about 2^-15.5 of the torus (std); half a message box is 2^-6.
PHASE_TOL = 2.0 ** -9
VARIANCE_RATIO = 1.5


def torus_dist(a, b, bits):
    d = np.ascontiguousarray(a - b)  # wraps
    s = d.view(np.int64 if bits == 64 else np.int32)
    return np.abs(s.astype(np.float64)) / 2.0 ** bits


def make_device_key(ctx, ks, variant):
    return DeviceServerKey.from_standard_keys(ctx, ks.device_params(), ks.bsk, ks.ksk, variant=variant, torus_bits=ks.bits)


def encrypt_batch(ks, msgs):
    p = ks.params
    delta = O.shortint_delta(p)
    return np.stack([ks.o.lwe_encrypt(ks.rng, ks.glwe, int(m) * delta, ks.lwe_std) for m in msgs])


@pytest.mark.parametrize("variant", [CAMD_PBS_GENERIC, CAMD_PBS_WAVE])
def test_keyswitch_bit_exact_2_2(gpu_ctx, keys_2_2, variant):
    ks = keys_2_2
    dk = make_device_key(gpu_ctx, ks, variant)
    rs = np.random.RandomState(1)
    cts = rs.randint(0, 2 ** 63, size=(37, ks.o.big + 1), dtype=np.uint64) * np.uint64(2) + rs.randint(0, 2, size=(37, ks.o.big + 1)).astype(np.uint64)
    cts[3, :] = 0                       # all-zero ciphertext
    cts[4, :-1] = 0                     # trivial ciphertext
    cts[5, :] = np.uint64(2 ** 64 - 1)  # every mask element rounds up and wraps
    got = dk.ks(cts)
    want = np.stack([ks.o.keyswitch(ks.ksk, c) for c in cts])
    assert got.dtype == np.uint64 and np.array_equal(got, want)


@pytest.mark.parametrize("count", [1, 17, 300, 1100])
def test_keyswitch_small_batches_split_over_key_rows(gpu_ctx, keys_2_2, count):
    """Below about a thousand workgroups the keyswitch also splits the key rows over workgroups
    (16, 16, 13 and 3 ranges for these counts, the last range shorter than the others) and sums
    the ranges with atomic adds: wrapping integer sums, so still bit for bit the oracle's."""
    ks = keys_2_2
    dk = make_device_key(gpu_ctx, ks, CAMD_PBS_WAVE)
    rs = np.random.RandomState(100 + count)
    cts = rs.randint(0, 2 ** 63, size=(count, ks.o.big + 1), dtype=np.uint64) * np.uint64(2) + rs.randint(0, 2, size=(count, ks.o.big + 1)).astype(np.uint64)
    got = dk.ks(cts)
    pick = np.unique(np.concatenate([np.arange(min(count, 3)), np.arange(max(count - 3, 0), count), rs.randint(0, count, size=24)]))
    want = np.stack([ks.o.keyswitch(ks.ksk, cts[i]) for i in pick])
    assert np.array_equal(got[pick], want)
    assert np.array_equal(dk.ks(cts), got)      # and the same on a second call, whatever the arrival order


@pytest.mark.parametrize("variant", [CAMD_PBS_GENERIC, CAMD_PBS_WAVE])
def test_trivial_bootstrap_bit_exact_2_2(gpu_ctx, keys_2_2, variant):
    """Mask = 0 skips every CMUX: the output is the rotated look-up table, sample-extracted."""
    ks = keys_2_2
    dk = make_device_key(gpu_ctx, ks, variant)
    luts = [ks.o.generate_accumulator(f) for f in (lambda x: x % 4, lambda x: x // 4, lambda x: (2 * x) % 16)]
    ids = [dk.upload_lut(l) for l in luts]
    assert ids == [0, 1, 2]
    rs = np.random.RandomState(2)
    count = 48
    cts = np.zeros((count, ks.n + 1), dtype=np.uint64)
    cts[:, -1] = rs.randint(0, 2 ** 63, size=count, dtype=np.uint64) * np.uint64(2)
    cts[0, -1] = 0
    cts[1, -1] = np.uint64(2 ** 64 - 1)   # modulus switch returns 2N
    lut_ids = (np.arange(count) % 3).astype(np.uint32)
    got = dk.pbs(cts, lut_ids)
    want = np.stack([ks.o.bootstrap(ks.fbsk, luts[lut_ids[i]], cts[i]) for i in range(count)])
    assert np.array_equal(got, want)


@pytest.mark.parametrize("variant", [CAMD_PBS_GENERIC, CAMD_PBS_WAVE])
def test_ks_pbs_parity_2_2(gpu_ctx, keys_2_2, variant):
    ks = keys_2_2
    p = ks.params
    dk = make_device_key(gpu_ctx, ks, variant)
    fs = [lambda x: x % 4, lambda x: x // 4, lambda x: (x * x) % 16]
    luts = [ks.o.generate_accumulator(f) for f in fs]
    for l in luts:
        dk.upload_lut(l)
    count = 48
    msgs = np.arange(count) % 16
    lut_ids = ((np.arange(count) // 16) % 3).astype(np.uint32)
    cts = encrypt_batch(ks, msgs)
    got = dk.ks_pbs(cts, lut_ids)
    want = ks.o.ks_pbs_batch(ks.ksk, ks.fbsk, np.concatenate(luts), lut_ids, cts, threads=8)
    if dk.variant == CAMD_PBS_GENERIC:
        assert np.array_equal(got, want)
    delta = O.shortint_delta(p)
    err_g, err_w = [], []
    for i in range(count):
        expect = fs[lut_ids[i]](int(msgs[i]))
        pg, pw = ks.o.lwe_phase(ks.glwe, got[i]), ks.o.lwe_phase(ks.glwe, want[i])
        assert O.shortint_decode(p, pg) == O.shortint_decode(p, pw) == expect
        err_g.append(((pg - expect * delta + 2 ** 63) % 2 ** 64 - 2 ** 63) / 2.0 ** 64)
        err_w.append(((pw - expect * delta + 2 ** 63) % 2 ** 64 - 2 ** 63) / 2.0 ** 64)
    err_g, err_w = np.array(err_g), np.array(err_w)
    assert np.abs(err_g).max() < PHASE_TOL
    assert err_g.var() < VARIANCE_RATIO * err_w.var() + 1e-30


def gadget_variance(p):
    """Predicted output variance of one bootstrap in torus^2 units from the gadget decomposition
    (Chillotti-Ligier-Orfila-Tap 2021, binary keys): key noise amplified by the digits, plus the
    rounding of the decomposition; the two 1/q^2 terms of the formula are below 1e-30 and dropped."""
    n, k, N, L = p.lwe_dimension, p.glwe_dimension, p.polynomial_size, p.pbs_level
    B = 2.0 ** p.pbs_base_log
    key_term = n * L * (k + 1) * N * ((B * B + 2.0) / 12.0) * p.glwe_modular_std_dev ** 2
    rounding_term = n * (1.0 / (24.0 * B ** (2 * L))) * (1.0 + k * N / 2.0)
    return key_term + rounding_term


def fft_variance(p, o):
    """Variance the f64 transforms add to the output phase. One digit-polynomial x key-polynomial
    product carries a rounding error e per coefficient (measured here against the exact integer
    product, about 2^-26 of the torus for 2^22-sized digits at N = 1024); a CMUX sums (k+1)L such
    products per output polynomial, and decryption multiplies the mask errors by the binary secret
    key, so each CMUX - whatever its key bit - adds (k+1)L (1 + kN/2) var(e) to the phase."""
    n, k, N, L = p.lwe_dimension, p.glwe_dimension, p.polynomial_size, p.pbs_level
    if o.bits == 32:
        # 32-bit torus: the rounding of every CMUX's result to 2^-32 (variance 2^-64 / 12 per
        # coefficient) is far above what the f64 transforms leave of products of 8-bit digits
        return n * (1.0 + k * N / 2.0) * 2.0 ** -64 / 12.0
    half = 1 << (p.pbs_base_log - 1)
    rs = np.random.RandomState(5)
    M = N // 2
    errs = []
    for _ in range(4):
        a = rs.randint(0, 2 ** 63, N, dtype=np.uint64) * np.uint64(2) + rs.randint(0, 2, N).astype(np.uint64)
        b = rs.randint(-half, half, N).astype(np.int64)
        fa, fb = o.fft_forward_torus(a), o.fft_forward_integer(b.astype(np.uint64))
        prod = np.concatenate([fa[:M] * fb[:M] - fa[M:] * fb[M:], fa[:M] * fb[M:] + fa[M:] * fb[:M]])
        got = o.fft_add_backward_torus(prod, np.zeros(N, dtype=np.uint64))
        errs.append((got - O.negacyclic_schoolbook(a, b)).view(np.int64).astype(np.float64) / 2.0 ** 64)
    return n * (k + 1) * L * (1.0 + k * N / 2.0) * np.concatenate(errs).var()


def check_noise_variance(bits, count, sub, bound, gadget, got, want, key, nearest_of, label, ratio_bar=(0.75, 1.15)):
    """The three assertions of the noise acceptance bar, shared by every wave instance:
    gadget < variance < 1.2 x predicted for the GPU (all `count` outputs) and for the oracle (the
    first `sub`), the two variances within `ratio_bar` of each other on the same inputs, and no
    bias. `nearest_of(phase)` returns the plaintext the phase decodes to (same dtype), `key` the
    secret key the outputs are under. Returns the decoded plaintexts of both."""
    from concrete_amd import client as CL

    word = np.uint64 if bits == 64 else np.uint32
    signed = np.int64 if bits == 64 else np.int32

    def phase_error(out):
        ph = CL.lwe_phase(bits, key, out).astype(word)
        nearest = nearest_of(ph)
        return (ph - nearest).view(signed).astype(np.float64) / 2.0 ** bits, nearest

    err_g, dec_g = phase_error(got)
    err_w, dec_w = phase_error(want)
    print(f"{label}: variance gpu {err_g.var():.4e} (subset {err_g[:sub].var():.4e}), oracle {err_w.var():.4e}, "
          f"predicted {bound:.4e} (gadget {gadget:.4e})")
    assert gadget < err_g.var() < 1.2 * bound, (label, err_g.var(), bound)
    assert gadget < err_w.var() < 1.2 * bound, (label, err_w.var(), bound)
    ratio = err_g[:sub].var() / err_w.var()
    assert ratio_bar[0] < ratio < ratio_bar[1], (label, ratio)
    assert abs(err_g.mean()) < 4 * np.sqrt(bound / count), (label, err_g.mean())   # no bias from the fast rounding
    return dec_g, dec_w


# every shortint set whose default kernel is a wave instance: N = 1024 (one schedule, four sets that
# differ in the moduli only: 2_2 stands for them) and the three N = 512 sets, which differ in n and
# in the keyswitch decomposition (concrete-shortint/src/parameters/mod.rs:215-304)
WAVE_SHORTINT_SETS = ["PARAM_MESSAGE_2_CARRY_2", "PARAM_MESSAGE_1_CARRY_2", "PARAM_MESSAGE_2_CARRY_1", "PARAM_MESSAGE_3_CARRY_0"]


@pytest.mark.parametrize("pname", WAVE_SHORTINT_SETS)
def test_output_noise_variance(gpu_ctx, pname):
    """Acceptance bar for the fast kernels, for every shortint set CAMD_PBS_AUTO sends to a wave
    instance: the variance of the output phase error stays within the predicted figure (gadget
    noise + the f64 transforms' rounding, the tolerance the path is allowed) and does not exceed
    the oracle's (measured on a 4096-ciphertext subset of the same inputs) beyond sampling error;
    decryptions agree ciphertext by ciphertext. 2^14 bootstraps at 2_2, 2^13 at the N = 512 sets."""
    from concrete_amd import client as CL
    from concrete_amd.backend import CAMD_PBS_AUTO

    p = getattr(P, pname)
    ks = get_keys(p, 64)
    dk = make_device_key(gpu_ctx, ks, CAMD_PBS_AUTO)
    assert dk.variant == CAMD_PBS_WAVE
    msg = p.message_modulus
    lut = ks.o.generate_accumulator(lambda x: x % msg)
    dk.upload_lut(lut)
    rng = CL.Rng(99)
    count, sub = (1 << 14 if p.polynomial_size == 1024 else 1 << 13), 1 << 12
    space = p.message_modulus * p.carry_modulus
    delta = O.shortint_delta(p)
    msgs = np.arange(count) % msg
    cts = CL.lwe_encrypt(rng, 64, ks.glwe, [int(m) * delta for m in msgs], ks.lwe_std)
    got = dk.ks_pbs(cts)
    want = ks.o.ks_pbs_batch(ks.ksk, ks.fbsk, lut, None, cts[:sub], threads=16)
    gadget = gadget_variance(p)
    bound = gadget + fft_variance(p, ks.o)
    nearest = lambda ph: ((ph + np.uint64(delta // 2)) // np.uint64(delta)) * np.uint64(delta)   # wraps like the torus
    near_g, near_w = check_noise_variance(64, count, sub, bound, gadget, got, want, ks.glwe, nearest, pname)
    dec_g, dec_w = (near_g // np.uint64(delta)) % np.uint64(space), (near_w // np.uint64(delta)) % np.uint64(space)
    # identical decryption wherever the oracle was run; wrong boxes only at the parameter set's own rate
    assert np.array_equal(dec_g[:sub], dec_w)
    assert np.count_nonzero(dec_g != msgs.astype(np.uint64)) <= 8       # expectation 2^14 * 2^-13.9 = 1.1
    dk.close()


@pytest.mark.parametrize("params,pname", [(P.DEFAULT_PARAMETERS, "DEFAULT_PARAMETERS"), (P.TFHE_LIB_PARAMETERS, "TFHE_LIB_PARAMETERS")])
def test_output_noise_variance_boolean(gpu_ctx, params, pname):
    """The same bar for the two boolean sets (concrete-boolean/src/parameters/mod.rs:83-111), whose
    default kernels are wave instances too: 2^13 NAND inputs (the gate's linear combination of two
    fresh encryptions, engine/mod.rs:388-397) are bootstrapped; the phase of each output under the
    GLWE key is compared with +-2^29 (lib.rs:76-82). The oracle runs on the first 2^12."""
    from concurrent.futures import ThreadPoolExecutor
    from concrete_amd import client as CL
    from concrete_amd.backend import CAMD_PBS_AUTO

    p = params
    ks = get_keys(p, 32)
    o = ks.o
    dk = make_device_key(gpu_ctx, ks, CAMD_PBS_AUTO)
    assert dk.variant == CAMD_PBS_WAVE
    acc = o.boolean_accumulator()
    dk.upload_lut(acc)
    rng = CL.Rng(77)
    count, sub = 1 << 13, 1 << 12
    T, F = 1 << 29, 7 << 29
    a_bits, b_bits = (np.arange(count) & 1).astype(bool), ((np.arange(count) >> 1) & 1).astype(bool)
    ca = CL.lwe_encrypt(rng, 32, ks.small, np.where(a_bits, T, F), p.lwe_modular_std_dev)
    cb = CL.lwe_encrypt(rng, 32, ks.small, np.where(b_bits, T, F), p.lwe_modular_std_dev)
    cts = (np.uint32(0) - (ca + cb)).astype(np.uint32)
    cts[:, -1] += np.uint32(T)
    got = dk.pbs(cts)
    with ThreadPoolExecutor(16) as pool:       # the oracle call releases the interpreter lock
        want = np.stack(list(pool.map(lambda c: o.bootstrap(ks.fbsk, acc, c), cts[:sub])))
    gadget = gadget_variance(p)
    bound = gadget + fft_variance(p, o)
    nearest = lambda ph: np.where(ph < np.uint32(1 << 31), np.uint32(T), np.uint32(F)).astype(np.uint32)
    near_g, near_w = check_noise_variance(32, count, sub, bound, gadget, got, want, ks.glwe, nearest, pname)
    nand = ~(a_bits & b_bits)
    assert np.array_equal(near_g[:sub], near_w)
    assert np.array_equal(near_g == np.uint32(T), nand)      # failure probability 2^-25 / 2^-165 per gate
    dk.close()


def test_wave_persistent_rounds_and_tail(gpu_ctx, keys_2_2):
    """More ciphertexts than resident waves, not a multiple of anything: every wave runs two
    rounds with a rendezvous in between, a few run a third, partial one. Same decryptions as the
    oracle for each ciphertext, whatever its position."""
    import torch

    ks = keys_2_2
    p = ks.params
    dk = make_device_key(gpu_ctx, ks, CAMD_PBS_WAVE)
    lut = ks.o.generate_accumulator(lambda x: (3 * x) % 4)
    dk.upload_lut(lut)
    from concrete_amd import client as CL

    cus = torch.cuda.get_device_properties(0).multi_processor_count
    count = 2 * 4 * cus + 7
    rng = CL.Rng(4242)
    delta = O.shortint_delta(p)
    msgs = np.arange(count) % 16
    cts = CL.lwe_encrypt(rng, 64, ks.glwe, [int(m) * delta for m in msgs], ks.lwe_std)
    got = dk.ks_pbs(cts)
    want = ks.o.ks_pbs_batch(ks.ksk, ks.fbsk, lut, None, cts, threads=16)
    dec = lambda out: np.array([O.shortint_decode(p, int(ph)) for ph in CL.lwe_phase(64, ks.glwe, out)])
    assert np.array_equal(dec(got), dec(want))
    assert np.count_nonzero(dec(got) != (3 * msgs) % 4) <= 3   # the parameter set's own failure rate


@pytest.mark.parametrize("variant", [CAMD_PBS_GENERIC, CAMD_PBS_WAVE])
def test_host_buffer_batch_is_sliced_without_changing_results(gpu_ctx, keys_2_2, variant, monkeypatch):
    """A host-buffer batch is cut into slices whose copies overlap the kernels of their
    neighbours. Word for word the same output as one launch over device-resident inputs,
    including the per-ciphertext table ids and a ragged last slice."""
    import torch

    ks = keys_2_2
    dk = make_device_key(gpu_ctx, ks, variant)
    luts = [ks.o.generate_accumulator(f) for f in (lambda x: x % 4, lambda x: x // 4)]
    for l in luts:
        dk.upload_lut(l)
    count = 3 * 16 + 5
    cts = encrypt_batch(ks, np.arange(count) % 16)
    lut_ids = (np.arange(count) % 2).astype(np.uint32)

    d_in = torch.from_numpy(cts.view(np.int64)).cuda()
    d_ids = torch.from_numpy(lut_ids.view(np.int32)).cuda()
    d_out = torch.empty_like(d_in)
    dk.ks_pbs_device(d_in, d_ids, d_out, count, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    whole = d_out.cpu().numpy().view(np.uint64)

    monkeypatch.setenv("CAMD_HOST_CHUNK", "16")      # 3 slices: 16, 16, 21
    sliced = dk.ks_pbs(cts, lut_ids)
    monkeypatch.setenv("CAMD_HOST_CHUNK", "0")       # one slice
    single = dk.ks_pbs(cts, lut_ids)
    assert np.array_equal(sliced, whole) and np.array_equal(single, whole)
    monkeypatch.setenv("CAMD_HOST_CHUNK", "12")      # four slices and more: half slices at both ends (6, 12, 12, 17, 6)
    assert np.array_equal(dk.ks_pbs(cts, lut_ids), whole)
    monkeypatch.setenv("CAMD_HOST_CHUNK", "13")      # 6, 13, 13, 15, 6
    assert np.array_equal(dk.ks_pbs(cts, lut_ids), whole)
    monkeypatch.setenv("CAMD_HOST_CHUNK", "16")
    assert np.array_equal(dk.ks(cts), np.stack([ks.o.keyswitch(ks.ksk, c) for c in cts]))


def test_concurrent_single_ciphertext_callers_share_launches(gpu_ctx, keys_2_2):
    """The reference bootstraps one ciphertext per call, from many threads. Callers that find the
    key busy are served together, as one batch: every caller gets word for word what a lone call
    gives (a lone call and a batch run the same per-ciphertext instruction sequence), a bad
    argument of one caller fails that caller only, and 48 concurrent calls take far less than 48
    calls one after the other."""
    import threading
    import time
    from concrete_amd._ffi import CamdError

    ks = keys_2_2
    dk = make_device_key(gpu_ctx, ks, CAMD_PBS_WAVE)
    luts = [ks.o.generate_accumulator(f) for f in (lambda x: x % 4, lambda x: x // 4)]
    for l in luts:
        dk.upload_lut(l)
    callers = 48
    cts = encrypt_batch(ks, np.arange(callers) % 16)
    lut_ids = (np.arange(callers) % 2).astype(np.uint32)
    t0 = time.time()
    alone = [dk.ks_pbs(cts[i], lut_ids[i:i + 1])[0] for i in range(8)]
    one_call = (time.time() - t0) / 8

    got, errors = [None] * callers, [None] * callers
    gate = threading.Barrier(callers + 1)

    def caller(i):
        gate.wait()
        try:
            if i == 5:
                dk.ks_pbs(cts[i], np.array([7], dtype=np.uint32))       # no such table
            elif i % 3 == 0:
                got[i] = dk.ks(cts[i])[0]                                 # another operation in between
            elif i % 3 == 1:
                got[i] = dk.ks_pbs(cts[i])[0]                             # no ids = table 0
            else:
                got[i] = dk.ks_pbs(cts[i], lut_ids[i:i + 1])[0]
        except CamdError as e:
            errors[i] = str(e)

    threads = [threading.Thread(target=caller, args=(i,)) for i in range(callers)]
    for t in threads:
        t.start()
    gate.wait()
    t0 = time.time()
    for t in threads:
        t.join()
    together = time.time() - t0

    assert errors[5] is not None and "lut id out of range" in errors[5]
    assert all(e is None for i, e in enumerate(errors) if i != 5)
    whole0 = dk.ks_pbs(cts)                    # one batch, table 0
    whole = dk.ks_pbs(cts, lut_ids)
    small = dk.ks(cts)
    for i in range(callers):
        if i == 5:
            continue
        want = small[i] if i % 3 == 0 else whole0[i] if i % 3 == 1 else whole[i]
        assert np.array_equal(got[i], want), i
    for i in range(8):
        assert np.array_equal(alone[i], whole[i])
    print(f"one call {one_call * 1e3:.1f} ms; {callers} concurrent calls {together * 1e3:.1f} ms")
    assert together < 0.5 * callers * one_call


def test_other_wave_schedules_in_their_own_process():
    """The schedule of the wave family is chosen once per process (CAMD_WAVE_SCHEDULE); the first
    (one row at a time, also the instance of the TFHE-lib boolean set) is kept as a selectable variant
    and must meet the same parity bar as the default. Batches smaller than the chip normally go to
    the latency schedule (k+1 waves per ciphertext); CAMD_WAVE_LATENCY=0 sends them to the selected
    schedule, so the small cases are run that way for both."""
    import os
    import subprocess
    import sys

    if os.environ.get("CAMD_WAVE_SCHEDULE") or os.environ.get("CAMD_WAVE_LATENCY"):
        pytest.skip("already running under a non-default schedule")
    here = os.path.abspath(__file__)
    for schedule in ("1", "2"):      # the default is 3 (two waves per SIMD) where it has an instance
        env = dict(os.environ, CAMD_WAVE_SCHEDULE=schedule, CAMD_WAVE_LATENCY="0")
        res = subprocess.run([sys.executable, "-m", "pytest", here, "-q", "-x", "-m", "gpu", "-k",
                              "ks_pbs_parity_2_2 or trivial_bootstrap or persistent_rounds", "-p", "no:cacheprovider"],
                             env=env, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, f"schedule {schedule}:\n" + res.stdout[-3000:] + res.stderr[-2000:]


def _schedule_probe(gpu_ctx, ks):
    dk = make_device_key(gpu_ctx, ks, CAMD_PBS_WAVE)
    for f in (lambda x: x % 4, lambda x: x // 4):
        dk.upload_lut(ks.o.generate_accumulator(f))
    rs = np.random.RandomState(77)
    count = 40
    cts = rs.randint(0, 2 ** 63, size=(count, ks.o.big + 1), dtype=np.uint64) * np.uint64(2) + rs.randint(0, 2, size=(count, ks.o.big + 1)).astype(np.uint64)
    return dk.ks_pbs(cts, (np.arange(count) % 2).astype(np.uint32))


def test_dump_for_schedule_comparison(gpu_ctx, keys_2_2):
    """Helper of the next test: writes this process's outputs where CAMD_DUMP says."""
    import os

    path = os.environ.get("CAMD_DUMP")
    if not path:
        pytest.skip("only run as a child of the schedule comparison")
    np.save(path, _schedule_probe(gpu_ctx, keys_2_2))


def test_latency_schedule_gives_the_words_of_the_first_schedule(gpu_ctx, keys_2_2, tmp_path):
    """k+1 waves per ciphertext perform the products and transforms of the first one-wave schedule
    in the same order (rows outermost, one transform at a time), so a small batch (latency
    schedule) and the same batch under CAMD_WAVE_SCHEDULE=1 agree word for word."""
    import os
    import subprocess
    import sys

    if os.environ.get("CAMD_WAVE_SCHEDULE") or os.environ.get("CAMD_WAVE_LATENCY"):
        pytest.skip("already running under a non-default schedule")
    here = os.path.abspath(__file__)
    got = _schedule_probe(gpu_ctx, keys_2_2)
    path = str(tmp_path / "first.npy")
    env = dict(os.environ, CAMD_WAVE_SCHEDULE="1", CAMD_WAVE_LATENCY="0", CAMD_DUMP=path)
    res = subprocess.run([sys.executable, "-m", "pytest", here, "-q", "-x", "-m", "gpu", "-k", "dump_for_schedule_comparison",
                          "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
    assert np.array_equal(got, np.load(path))


def test_device_batch_then_host_batch_without_sync(gpu_ctx):
    """A device-pointer call returns while its kernels run on the caller's stream; a host-pointer
    call on the same key right behind it uses the same scratch buffers on the context's stream.
    The library orders the two: both results must be right, with no synchronisation in between."""
    import torch

    p = P.PARAM_MESSAGE_2_CARRY_2
    ks = get_keys(p, 64, n_override=48)
    o = ks.o
    dk = make_device_key(gpu_ctx, ks, CAMD_PBS_WAVE)
    lut = o.generate_accumulator(lambda x: x % 4)
    dk.upload_lut(lut)
    big_n, small_n = 2048, 64   # the device batch is still running when the host batch arrives
    rs = np.random.RandomState(3)
    msgs = rs.randint(0, 4, size=big_n)
    cts = encrypt_batch(ks, msgs[:256])
    cts = np.concatenate([cts] * (big_n // 256))
    host_cts = encrypt_batch(ks, (np.arange(small_n) % 4))
    want_dev = o.ks_pbs_batch(ks.ksk, ks.fbsk, lut, None, cts[:256], threads=8)
    d_in = torch.from_numpy(cts.view(np.int64)).cuda()
    d_out = torch.zeros_like(d_in)
    side = torch.cuda.Stream()
    dk.reserve(big_n)
    for _ in range(3):
        d_out.zero_()
        torch.cuda.synchronize()
        dk.ks_pbs_device(d_in, None, d_out, big_n, side.cuda_stream)
        got_host = dk.ks_pbs(host_cts)                       # no synchronisation in between
        side.synchronize()
        got_dev = d_out.cpu().numpy().view(np.uint64)
        for got, ref_cts, count in ((got_dev, cts, big_n), (got_host, host_cts, small_n)):
            for i in range(0, count, 7):
                exp = O.shortint_decode(p, o.lwe_phase(ks.glwe, ref_cts[i])) % 4
                assert O.shortint_decode(p, o.lwe_phase(ks.glwe, got[i])) == exp, i
        # the copies of the 256 distinct inputs all give the same words, and decrypt like the oracle's
        assert np.array_equal(got_dev[:256], got_dev[256:512])
        for i in range(0, 256, 5):
            assert O.shortint_decode(p, o.lwe_phase(ks.glwe, got_dev[i])) == O.shortint_decode(p, o.lwe_phase(ks.glwe, want_dev[i]))
    dk.close()


@pytest.mark.parametrize("variant", [CAMD_PBS_GENERIC, CAMD_PBS_WAVE])
def test_device_call_reports_a_bad_lut_id(gpu_ctx, variant):
    """The host entry points reject a look-up table id that is out of range before launching. A
    device-pointer call cannot: its kernel flags the launch, and the next call on the key fails
    with CAMD_EINVAL instead of silently handing back results computed with table 0."""
    import torch

    p = P.PARAM_MESSAGE_2_CARRY_2
    ks = get_keys(p, 64, n_override=10)
    o = ks.o
    dk = make_device_key(gpu_ctx, ks, variant)
    dk.upload_lut(o.generate_accumulator(lambda x: x % 4))
    dk.upload_lut(o.generate_accumulator(lambda x: x // 4))
    cts = encrypt_batch(ks, [0, 1, 2, 3, 1, 2])
    with pytest.raises(Exception, match="lut id out of range"):
        dk.ks_pbs(cts, np.array([0, 1, 2, 0, 0, 0], dtype=np.uint32))
    d_in = torch.from_numpy(cts.view(np.int64)).cuda()
    d_out = torch.zeros_like(d_in)
    good = torch.tensor([0, 1, 0, 1, 1, 0], dtype=torch.int32).cuda()
    bad = torch.tensor([0, 1, 7, 1, 1, 0], dtype=torch.int32).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    dk.ks_pbs_device(d_in, good, d_out, 6, stream)
    dk.check(stream)                                         # nothing to report
    dk.ks_pbs_device(d_in, bad, d_out, 6, stream)
    with pytest.raises(Exception, match="look-up table id out of range"):
        dk.check(stream)
    dk.check(stream)                                         # the report clears the flag
    dk.ks_pbs_device(d_in, bad, d_out, 6, stream)
    torch.cuda.synchronize()
    with pytest.raises(Exception, match="look-up table id out of range"):
        dk.ks_pbs(cts)                                       # any later call on the key reports it
    assert dk.ks_pbs(cts).shape == cts.shape
    dk.close()


@pytest.mark.parametrize("count", [1, 37, 300, 1024])
def test_keyswitch_on_the_matrix_cores_is_bit_exact_2_2(gpu_ctx, keys_2_2, count, monkeypatch):
    """Batches go to the integer-MFMA keyswitch (balanced byte limbs of the key, i32 accumulation,
    limbs recombined with shifts): the same words as the oracle, for a lone ciphertext (key rows
    split over workgroups, atomics), a ragged tile and several tiles."""
    monkeypatch.setenv("CAMD_KS_MFMA_MIN", "1")
    ks = keys_2_2
    dk = make_device_key(gpu_ctx, ks, CAMD_PBS_WAVE)
    rs = np.random.RandomState(count)
    cts = rs.randint(0, 2 ** 63, size=(count, ks.o.big + 1), dtype=np.uint64) * np.uint64(2) + rs.randint(0, 2, size=(count, ks.o.big + 1)).astype(np.uint64)
    cts[0, :] = np.uint64(2 ** 64 - 1)   # every mask element rounds up and wraps
    if count > 4:
        cts[3, :] = 0
        cts[4, :-1] = 0
    got = dk.ks(cts)
    check = range(count) if count <= 64 else list(range(0, count, 17)) + [255, 256, count - 1]
    for i in check:
        assert np.array_equal(got[i], ks.o.keyswitch(ks.ksk, cts[i])), i
    # and the two keyswitch kernels agree on the whole batch
    monkeypatch.setenv("CAMD_KS_MFMA_MIN", "0")
    assert np.array_equal(dk.ks(cts), got)
    dk.close()
