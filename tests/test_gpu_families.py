"""GPU parity across parameter families and edge cases, through the C ABI.

The two workgroup-per-ciphertext kernels (radix-2 fallback and register-blocked) cover every
parameter set and perform the oracle's exact sequence of IEEE-754 operations, so outputs are
compared bit for bit. The LWE dimension is
shortened (n = 6..24 CMUX iterations instead of 567..991): the kernels are the same, the loop is
shorter, and the oracle finishes in seconds.
"""
import dataclasses

import numpy as np
import pytest

import concrete_amd.parameters as P
from concrete_amd.backend import CAMD_PBS_AUTO, CAMD_PBS_BLOCK, CAMD_PBS_GENERIC, CAMD_PBS_WAVE, DeviceServerKey
from oracle import cpu_oracle as O
from conftest import get_keys

pytestmark = pytest.mark.gpu


def device_key(ctx, ks, variant=CAMD_PBS_GENERIC):
    return DeviceServerKey.from_standard_keys(ctx, ks.device_params(), ks.bsk, ks.ksk, variant=variant, torus_bits=ks.bits)


SHORTINT_FAMILIES = [
    ("PARAM_MESSAGE_1_CARRY_1", 8),    # N=256,  k=6
    ("PARAM_MESSAGE_1_CARRY_2", 8),    # N=512,  k=3
    ("PARAM_MESSAGE_2_CARRY_2", 12),   # N=1024, k=2
    ("PARAM_MESSAGE_2_CARRY_3", 8),    # N=2048, k=1
    ("PARAM_MESSAGE_2_CARRY_4", 6),    # N=4096, k=1, two levels
    ("PARAM_MESSAGE_3_CARRY_4", 6),    # N=8192, k=1, two levels
    ("PARAM_MESSAGE_4_CARRY_4", 6),    # N=16384, k=1, two levels (the largest-polynomial class)
    ("PARAM_MESSAGE_1_CARRY_7", 6),    # N=16384, three levels
]


EXACT_VARIANTS = [CAMD_PBS_GENERIC, CAMD_PBS_BLOCK]


@pytest.mark.parametrize("variant", EXACT_VARIANTS)
@pytest.mark.parametrize("pname,n", SHORTINT_FAMILIES)
def test_ks_pbs_bit_exact_every_family(gpu_ctx, pname, n, variant):
    p = P.SHORTINT_BY_NAME[pname]
    ks = get_keys(p, 64, n_override=n)
    o = ks.o
    dk = device_key(gpu_ctx, ks, variant)
    assert dk.variant == variant
    msg, sup = p.message_modulus, p.message_modulus * p.carry_modulus
    fs = [lambda x: x % msg, lambda x: x // msg]
    luts = [o.generate_accumulator(f) for f in fs]
    for l in luts:
        dk.upload_lut(l)
    delta = O.shortint_delta(p)
    count = 5
    msgs = [(3 * i + 1) % sup for i in range(count)]
    cts = np.stack([o.lwe_encrypt(ks.rng, ks.glwe, m * delta, ks.lwe_std) for m in msgs])
    ids = (np.arange(count) % 2).astype(np.uint32)
    got = dk.ks_pbs(cts, ids)
    want = o.ks_pbs_batch(ks.ksk, ks.fbsk, np.concatenate(luts), ids, cts, threads=4)
    assert np.array_equal(got, want)
    # standalone entry points agree with the fused call
    small = dk.ks(cts)
    assert np.array_equal(small, np.stack([o.keyswitch(ks.ksk, c) for c in cts]))
    assert np.array_equal(dk.pbs(small, ids), want)
    dk.close()


@pytest.mark.parametrize("variant", EXACT_VARIANTS)
@pytest.mark.parametrize("params", [P.DEFAULT_PARAMETERS, P.TFHE_LIB_PARAMETERS])
def test_boolean_pbs_ks_bit_exact(gpu_ctx, params, variant):
    """32-bit torus, bootstrap then keyswitch (concrete-boolean cpu.rs:215-259)."""
    ks = get_keys(params, 32, n_override=16)
    o = ks.o
    dk = device_key(gpu_ctx, ks, variant)
    acc = o.boolean_accumulator()
    dk.upload_lut(acc)
    T, F = 1 << 29, 7 << 29
    rows = []
    for a in (False, True):
        for b in (False, True):
            ca = o.lwe_encrypt(ks.rng, ks.small, T if a else F, params.lwe_modular_std_dev)
            cb = o.lwe_encrypt(ks.rng, ks.small, T if b else F, params.lwe_modular_std_dev)
            buf = (np.uint32(0) - (ca + cb)).astype(np.uint32)
            buf[-1] += np.uint32(T)
            rows.append(buf)
    cts = np.stack(rows)
    got = dk.pbs_ks(cts)
    want = o.pbs_ks_batch(ks.ksk, ks.fbsk, acc, None, cts, threads=2)
    assert got.dtype == np.uint32 and np.array_equal(got, want)
    big = dk.pbs(cts)
    assert np.array_equal(big, np.stack([o.bootstrap(ks.fbsk, acc, c) for c in cts]))
    assert np.array_equal(dk.ks(big), want)
    dk.close()


def test_full_size_boolean_nand_truth_table(gpu_ctx):
    """The single-gate benchmark configuration on the GPU: NAND at DEFAULT_PARAMETERS with the full n = 586, on the
    bit-exact kernel (asked for explicitly; the default for this set is the wave kernel, next test but one)."""
    p = P.DEFAULT_PARAMETERS
    ks = get_keys(p, 32)
    o = ks.o
    dk = device_key(gpu_ctx, ks, CAMD_PBS_BLOCK)
    acc = o.boolean_accumulator()
    dk.upload_lut(acc)
    T, F = 1 << 29, 7 << 29
    rows, want_bits = [], []
    for rep in range(4):
        for a in (False, True):
            for b in (False, True):
                ca = o.lwe_encrypt(ks.rng, ks.small, T if a else F, p.lwe_modular_std_dev)
                cb = o.lwe_encrypt(ks.rng, ks.small, T if b else F, p.lwe_modular_std_dev)
                buf = (np.uint32(0) - (ca + cb)).astype(np.uint32)
                buf[-1] += np.uint32(T)
                rows.append(buf)
                want_bits.append(not (a and b))
    cts = np.stack(rows)
    got = dk.pbs_ks(cts)
    want = o.pbs_ks_batch(ks.ksk, ks.fbsk, acc, None, cts, threads=8)
    assert dk.variant == CAMD_PBS_BLOCK
    assert np.array_equal(got, want)
    for i, bit in enumerate(want_bits):
        assert (o.lwe_phase(ks.small, got[i]) < (1 << 31)) == bit
        assert (o.lwe_phase(ks.small, want[i]) < (1 << 31)) == bit
    dk.close()


@pytest.mark.parametrize("variant", [CAMD_PBS_GENERIC, CAMD_PBS_BLOCK, CAMD_PBS_WAVE])
def test_edge_cases_2_2(gpu_ctx, variant):
    """Empty batch, ragged batch sizes, aliasing of input and output, trivial inputs, lut ids."""
    p = P.PARAM_MESSAGE_2_CARRY_2
    ks = get_keys(p, 64, n_override=10)
    o = ks.o
    dk = device_key(gpu_ctx, ks, variant)
    luts = [o.generate_accumulator(f) for f in (lambda x: x % 4, lambda x: x // 4, lambda x: 15 - x)]
    for l in luts:
        dk.upload_lut(l)
    assert dk.lut_count == 3
    # empty
    assert dk.ks_pbs(np.zeros((0, o.big + 1), dtype=np.uint64)).shape == (0, o.big + 1)
    delta = O.shortint_delta(p)
    for count in (1, 3, 4, 5, 9, 17):
        msgs = [(5 * i + 2) % 16 for i in range(count)]
        cts = np.stack([o.lwe_encrypt(ks.rng, ks.glwe, m * delta, ks.lwe_std) for m in msgs])
        cts[0, :-1] = 0  # one trivial ciphertext in every batch
        ids = (np.arange(count) % 3).astype(np.uint32)
        got = dk.ks_pbs(cts, ids)
        want = o.ks_pbs_batch(ks.ksk, ks.fbsk, np.concatenate(luts), ids, cts, threads=4)
        if variant in EXACT_VARIANTS:
            assert np.array_equal(got, want)
        assert np.array_equal(got[0], want[0])  # trivial input: exact in every family
        for i in range(count):
            assert O.shortint_decode(p, o.lwe_phase(ks.glwe, got[i])) == O.shortint_decode(p, o.lwe_phase(ks.glwe, want[i]))
    # lut id out of range is an error on the host path
    with pytest.raises(Exception):
        dk.ks_pbs(cts, np.full(cts.shape[0], 7, dtype=np.uint32))
    dk.close()


@pytest.mark.parametrize("pname,count", [("PARAM_MESSAGE_2_CARRY_2", 301), ("PARAM_MESSAGE_1_CARRY_2", 521),
                                         ("PARAM_MESSAGE_4_CARRY_4", 5), ("PARAM_MESSAGE_3_CARRY_4", 3),
                                         ("PARAM_MESSAGE_3_CARRY_3", 7), ("PARAM_MESSAGE_2_CARRY_4", 2), ("PARAM_MESSAGE_2_CARRY_3", 9)])
def test_throughput_kernels_zero_mask_elements_and_ragged_batches(gpu_ctx, pname, count):
    """The throughput schedules of the wave family (two waves per SIMD at N = 1024 and N = 512, the
    workgroup-per-ciphertext kernel at N >= 8192) on batches that are no multiple of a workgroup, with
    mask elements that are zero (the CMUX is skipped, as the reference skips it), a ciphertext whose
    mask is all zero (a table look-up: exact) and two look-up tables: bootstrap alone through
    camd_pbs_batch against the oracle's bootstrap, decryption for decryption."""
    p = P.SHORTINT_BY_NAME[pname]
    n = 10
    ks = get_keys(p, 64, n_override=n)
    o = ks.o
    dk = device_key(gpu_ctx, ks, CAMD_PBS_WAVE)
    msg, sup = p.message_modulus, p.message_modulus * p.carry_modulus
    fs = [lambda x: x % msg, lambda x: (sup - 1 - x)]
    luts = [o.generate_accumulator(f) for f in fs]
    for l in luts:
        dk.upload_lut(l)
    delta = O.shortint_delta(p)
    msgs = [(7 * i + 3) % sup for i in range(count)]
    cts = np.stack([o.lwe_encrypt(ks.rng, ks.small, m * delta, ks.lwe_std) for m in msgs])
    cts[0, :-1] = 0                       # a trivial ciphertext
    cts[1, 0] = cts[1, 3] = 0             # zero mask elements, the first one among them
    cts[2 % count, n - 1] = 0             # and the last
    ids = (np.arange(count) % 2).astype(np.uint32)
    got = dk.pbs(cts, ids)
    acc = np.concatenate(luts).reshape(2, -1)
    for i in list(range(min(count, 40))) + [count - 1]:
        want = o.bootstrap(ks.fbsk, acc[ids[i]], cts[i])
        assert O.shortint_decode(p, o.lwe_phase(ks.glwe, got[i])) == O.shortint_decode(p, o.lwe_phase(ks.glwe, want)), i
        if i == 0:
            assert np.array_equal(got[0], want)
    dk.close()


def test_device_pointer_entry_points(gpu_ctx):
    """Device-resident inputs (torch owns the memory), in-place output, explicit stream."""
    import torch

    p = P.PARAM_MESSAGE_2_CARRY_2
    ks = get_keys(p, 64, n_override=10)
    o = ks.o
    dk = device_key(gpu_ctx, ks, CAMD_PBS_GENERIC)
    lut = o.generate_accumulator(lambda x: x % 4)
    dk.upload_lut(lut)
    delta = O.shortint_delta(p)
    cts = np.stack([o.lwe_encrypt(ks.rng, ks.glwe, m * delta, ks.lwe_std) for m in range(16)])
    want = o.ks_pbs_batch(ks.ksk, ks.fbsk, lut, None, cts, threads=4)
    d_in = torch.from_numpy(cts.view(np.int64)).cuda()
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        dk.ks_pbs_device(d_in, None, d_in, 16, stream.cuda_stream)  # output aliases input
    stream.synchronize()
    assert np.array_equal(d_in.cpu().numpy().view(np.uint64), want)
    # blob export / import gives an equal key
    blob = torch.empty(dk.blob_size(), dtype=torch.uint8, device="cuda")
    dk.export_blob(blob)
    torch.cuda.synchronize()
    dk2 = DeviceServerKey.from_device_blob(gpu_ctx, ks.device_params(), blob, dk.variant)
    dk2.upload_lut(lut)
    assert np.array_equal(dk2.ks_pbs(cts), want)
    dk.close()
    dk2.close()


def test_plain_c_caller(gpu_ctx, tmp_path):
    """The C ABI from a C program (tests/c/abi_driver.c): own process, no Python or torch inside.
    N = 256 selects the register-blocked kernel, so the output equals the oracle's bit for bit."""
    import dataclasses
    import os
    import subprocess

    p = P.SHORTINT_BY_NAME["PARAM_MESSAGE_1_CARRY_1"]
    ks = get_keys(p, 64, n_override=8)
    o = ks.o
    lut = o.generate_accumulator(lambda x: (x * 3) % 4)
    delta = O.shortint_delta(p)
    cts = np.stack([o.lwe_encrypt(ks.rng, ks.glwe, (m % 4) * delta, ks.lwe_std) for m in range(11)])
    dp = ks.device_params()
    header = np.array([dp.lwe_dimension, dp.glwe_dimension, dp.polynomial_size, dp.pbs_base_log, dp.pbs_level,
                       dp.ks_base_log, dp.ks_level, 64], dtype=np.uint32)
    src, dst = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(src, "wb") as f:
        f.write(header.tobytes())
        f.write(np.uint64(cts.shape[0]).tobytes())
        for arr in (ks.bsk, ks.ksk, lut, cts):
            f.write(np.ascontiguousarray(arr, dtype=np.uint64).tobytes())
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "c", "abi_driver")
    res = subprocess.run([exe, str(src), str(dst)], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    assert "kernel variant 3" in res.stdout
    got = np.fromfile(dst, dtype=np.uint64).reshape(cts.shape)
    want = o.ks_pbs_batch(ks.ksk, ks.fbsk, lut, None, cts, threads=4)
    assert np.array_equal(got, want)


def test_boolean_wave_kernel_full_size(gpu_ctx):
    """The wave-per-ciphertext instance for the boolean default parameters (what CAMD_PBS_AUTO
    selects for this set), full n = 586, against the oracle:
    every decryption identical, every gate of the truth table right, and ciphertext words within
    the last units of the 32-bit torus (same algorithm, different order of f64 operations)."""
    p = P.DEFAULT_PARAMETERS
    ks = get_keys(p, 32)
    o = ks.o
    dk = device_key(gpu_ctx, ks, CAMD_PBS_AUTO)
    assert dk.variant == CAMD_PBS_WAVE
    acc = o.boolean_accumulator()
    dk.upload_lut(acc)
    T, F = 1 << 29, 7 << 29
    rows, want_bits = [], []
    for rep in range(8):
        for a in (False, True):
            for b in (False, True):
                ca = o.lwe_encrypt(ks.rng, ks.small, T if a else F, p.lwe_modular_std_dev)
                cb = o.lwe_encrypt(ks.rng, ks.small, T if b else F, p.lwe_modular_std_dev)
                buf = (np.uint32(0) - (ca + cb)).astype(np.uint32)
                buf[-1] = np.uint32((int(buf[-1]) + T) & 0xFFFFFFFF)
                rows.append(buf)
                want_bits.append(not (a and b))
    cts = np.stack(rows)
    want_big = np.stack([o.bootstrap(ks.fbsk, acc, c) for c in cts])
    got_big = dk.pbs(cts)
    diff = (got_big.astype(np.int64) - want_big.astype(np.int64) + (1 << 31)) % (1 << 32) - (1 << 31)
    assert np.abs(diff).max() <= 2, np.abs(diff).max()
    got = dk.pbs_ks(cts)
    want = o.pbs_ks_batch(ks.ksk, ks.fbsk, acc, None, cts, threads=8)
    for i, bit in enumerate(want_bits):
        assert (o.lwe_phase(ks.small, got[i]) < (1 << 31)) == bit
        assert (o.lwe_phase(ks.small, want[i]) < (1 << 31)) == bit
    # ragged sizes around the persistent grid (more ciphertexts than one wave per SIMD of every CU)
    many = np.tile(cts, (40, 1))[:1027]
    out = dk.pbs_ks(many)
    assert np.array_equal(out[:32], got) and np.array_equal(out[992:1024], got) and np.array_equal(out[1024:], got[:3])
    dk.close()


def test_tfhe_lib_wave_kernel_full_size(gpu_ctx):
    """The wave instance for the boolean TFHE-lib parameters (N = 1024, k = 1, three levels; on
    request), full n = 630, against the oracle: every decryption identical and every gate of the
    truth table right. A small batch runs on the k+1-wave variant, a large one on the one-wave
    schedule; both accumulate the products in the same order, so they give the same words. Against
    the oracle (another order of f64 operations) words agree to the last units of the 32-bit torus
    unless a last-unit difference tips the rounding of a decomposition (21 of 32 bits kept), after
    which masks differ while decryptions do not: most ciphertexts must be within 2 units."""
    p = P.TFHE_LIB_PARAMETERS
    ks = get_keys(p, 32)
    o = ks.o
    dk = device_key(gpu_ctx, ks, CAMD_PBS_AUTO)
    assert dk.variant == CAMD_PBS_WAVE
    acc = o.boolean_accumulator()
    dk.upload_lut(acc)
    T, F = 1 << 29, 7 << 29
    rows, want_bits = [], []
    for rep in range(8):
        for a in (False, True):
            for b in (False, True):
                ca = o.lwe_encrypt(ks.rng, ks.small, T if a else F, p.lwe_modular_std_dev)
                cb = o.lwe_encrypt(ks.rng, ks.small, T if b else F, p.lwe_modular_std_dev)
                buf = (np.uint32(0) - (ca + cb)).astype(np.uint32)
                buf[-1] = np.uint32((int(buf[-1]) + T) & 0xFFFFFFFF)
                rows.append(buf)
                want_bits.append(not (a and b))
    cts = np.stack(rows)
    want_big = np.stack([o.bootstrap(ks.fbsk, acc, c) for c in cts])
    got_big = dk.pbs(cts)
    diff = (got_big.astype(np.int64) - want_big.astype(np.int64) + (1 << 31)) % (1 << 32) - (1 << 31)
    close = np.abs(diff).max(axis=1) <= 2
    print(f"ciphertexts within 2 units of the oracle: {int(close.sum())} of {len(close)}")
    assert close.sum() >= 0.75 * len(close)
    for i, bit in enumerate(want_bits):
        assert (o.lwe_phase(ks.glwe, got_big[i]) < (1 << 31)) == bit
        assert (o.lwe_phase(ks.glwe, want_big[i]) < (1 << 31)) == bit
    got = dk.pbs_ks(cts)
    want = o.pbs_ks_batch(ks.ksk, ks.fbsk, acc, None, cts, threads=8)
    for i, bit in enumerate(want_bits):
        assert (o.lwe_phase(ks.small, got[i]) < (1 << 31)) == bit
        assert (o.lwe_phase(ks.small, want[i]) < (1 << 31)) == bit
    # more ciphertexts than CUs: the one-wave schedule, ragged last workgroup; same words
    many = np.tile(cts, (40, 1))[:1027]
    out = dk.pbs_ks(many)
    assert np.array_equal(out[:32], got) and np.array_equal(out[992:1024], got) and np.array_equal(out[1024:], got[:3])
    dk.close()


@pytest.mark.parametrize("pname", ["PARAM_MESSAGE_1_CARRY_2", "PARAM_MESSAGE_2_CARRY_1", "PARAM_MESSAGE_3_CARRY_0"])
def test_wave_kernel_small_shortint_sets(gpu_ctx, pname):
    """The N = 512 wave instance on the three small shortint sets (64-bit torus, k = 3, one level),
    selected explicitly, full n, against the oracle: every ciphertext decrypts to the same value as
    the oracle's and to the table's value, for two look-up tables and every message; phase errors
    stay far inside the decoding margin; batch sizes around the persistent grid agree."""
    p = getattr(P, pname)
    ks = get_keys(p, 64)
    o = ks.o
    dk = device_key(gpu_ctx, ks, CAMD_PBS_WAVE)
    assert dk.variant == CAMD_PBS_WAVE
    space = p.message_modulus * p.carry_modulus
    fs = [lambda x: x % p.message_modulus, lambda x: (3 * x + 1) % space]
    luts = [o.generate_accumulator(f) for f in fs]
    for l in luts:
        dk.upload_lut(l)
    count = 4 * space
    msgs = np.arange(count) % space
    ids = ((np.arange(count) // space) % 2).astype(np.uint32)
    delta = O.shortint_delta(p)
    cts = np.stack([o.lwe_encrypt(ks.rng, ks.glwe, int(m) * delta, p.lwe_modular_std_dev) for m in msgs])
    got = dk.ks_pbs(cts, ids)
    want = o.ks_pbs_batch(ks.ksk, ks.fbsk, np.concatenate(luts), ids, cts, threads=8)
    for i in range(count):
        expect = fs[ids[i]](int(msgs[i]))
        pg, pw = o.lwe_phase(ks.glwe, got[i]), o.lwe_phase(ks.glwe, want[i])
        assert O.shortint_decode(p, pg) == O.shortint_decode(p, pw) == expect
        err = ((pg - expect * delta + 2 ** 63) % 2 ** 64 - 2 ** 63) / float(delta)
        assert abs(err) < 0.25, err            # decoding fails at 0.5
    # Large batches run on the throughput schedule, which keeps 32 bits per accumulator coefficient: its
    # words differ from the small-batch variant's, its decryptions do not; two large batches around
    # the persistent grid give each other's words.
    many = np.tile(cts, (1100 // count + 1, 1))[:1027]
    many_ids = np.tile(ids, 1100 // count + 1)[:1027]
    out = dk.ks_pbs(many, many_ids)
    for i in list(range(count)) + [1024, 1025, 1026]:
        assert O.shortint_decode(p, o.lwe_phase(ks.glwe, out[i])) == O.shortint_decode(p, o.lwe_phase(ks.glwe, got[i % count])), i
    assert np.all(out & np.uint64(0xFFFFFFFF) == 0)
    fewer = dk.ks_pbs(many[:700], many_ids[:700])
    assert np.array_equal(fewer, out[:700])
    dk.close()


@pytest.fixture(scope="module")
def full_size_keys():
    """Real keys of the N = 16384 sets at their full LWE dimension, generated once per set for the
    tests below (1 GB of bootstrap key each, from the library's multithreaded host key generation:
    the same algorithm as the oracle's single-threaded one, which tests/test_host_cpu.py compares)."""
    from concrete_amd import client as CL

    made = {}

    def get(pname):
        if pname not in made:
            made.clear()                        # one set at a time: the keys are large
            p = P.SHORTINT_BY_NAME[pname]
            rng = CL.Rng(0xBEEF)
            small = CL.gen_binary_key(rng, 64, p.lwe_dimension)
            glwe = CL.gen_binary_key(rng, 64, p.glwe_dimension * p.polynomial_size)
            bsk = CL.gen_bootstrap_key(rng, p, 64, small, glwe, threads=16)
            ksk = CL.gen_keyswitch_key(rng, p, 64, glwe, small)
            made[pname] = (p, small, glwe, bsk, ksk)
        return made[pname]

    yield get
    made.clear()


@pytest.mark.parametrize("pname,count", [("PARAM_MESSAGE_4_CARRY_4", 16), ("PARAM_MESSAGE_1_CARRY_7", 8)])
def test_largest_polynomial_full_size_bit_exact(gpu_ctx, full_size_keys, pname, count):
    """The largest-polynomial class (N = 16384, two and three decomposition levels) at its FULL
    LWE dimension with real keys: keyswitch + bootstrap through the C ABI with the register-blocked
    kernel must give the oracle's ciphertexts word for word (that kernel's contract) and every
    ciphertext must decrypt to f(m) (the property of concrete-shortint/src/server_key/tests.rs:252-277).
    Parameter sets: concrete-shortint/src/parameters/mod.rs:719-736 and :665-682."""
    p, small, glwe, bsk, ksk = full_size_keys(pname)
    o = O.Oracle(p, 64)
    dk = DeviceServerKey.from_standard_keys(gpu_ctx, p, bsk, ksk, variant=CAMD_PBS_BLOCK)
    assert dk.variant == CAMD_PBS_BLOCK
    fbsk = o.bsk_to_fourier(bsk)
    msg, sup = p.message_modulus, p.message_modulus * p.carry_modulus
    fs = [lambda x: x % msg, lambda x: x // msg]
    luts = [o.generate_accumulator(f) for f in fs]
    for l in luts:
        dk.upload_lut(l)
    delta = O.shortint_delta(p)
    # every message class: both ends of the range, both sides of the padding-bit boundary, carries
    msgs = ([0, 1, msg - 1, msg, msg + 1, sup // 2 - 1, sup // 2, sup // 2 + 1, sup - 2, sup - 1]
            + [(37 * i + 5) % sup for i in range(count)])[:count]
    orng = O.Rng(0xF00D)
    cts = np.stack([o.lwe_encrypt(orng, glwe, m * delta, p.lwe_modular_std_dev) for m in msgs])
    ids = (np.arange(count) % 2).astype(np.uint32)
    got = dk.ks_pbs(cts, ids)
    want = o.ks_pbs_batch(ksk, fbsk, np.concatenate(luts), ids, cts, threads=16)
    assert np.array_equal(got, want)
    for i, m in enumerate(msgs):
        assert O.shortint_decode(p, o.lwe_phase(glwe, got[i])) == fs[i % 2](m), (i, m)
    dk.close()


@pytest.mark.parametrize("pname", ["PARAM_MESSAGE_4_CARRY_4", "PARAM_MESSAGE_1_CARRY_7", "PARAM_MESSAGE_3_CARRY_4", "PARAM_MESSAGE_7_CARRY_0",
                                   "PARAM_MESSAGE_2_CARRY_4", "PARAM_MESSAGE_3_CARRY_3", "PARAM_MESSAGE_2_CARRY_3"])
def test_largest_polynomial_full_size_throughput_kernel(gpu_ctx, full_size_keys, pname):
    """PARAM_MESSAGE_4_CARRY_4 (N = 16384), 1_7 (N = 16384, three levels), 3_4 (N = 8192, two levels),
    7_0 (N = 8192, one level), 2_4 and 3_3 (N = 4096, two levels and one; two ciphertexts per workgroup) and
    2_3 (N = 2048; four per workgroup) at their full LWE dimension through the kernel CAMD_PBS_AUTO selects for them
    (pbs_wave16k.hip): 2048 fresh ciphertexts over the whole message space must decrypt to
    f(m), and the variance of the output phase error must stay within the predicted figure and
    next to the reference algorithm's on the same inputs. The reference values come from the
    register-blocked kernel, which the test above pins to the oracle word for word at this size
    (the oracle itself takes seconds per bootstrap here). The prediction is the gadget noise, the
    f64 transforms' rounding and this kernel's own term: it keeps 32 bits per accumulator
    coefficient, so every CMUX rounds its increment at 2^-32 of the torus (variance 2^-64 / 12
    per coefficient, times 1 + kN/2 through the binary key at decryption). At this size the
    transforms' rounding is most of the output noise, and this kernel's four-step transform
    rounds a little less than the reference's (measured ratio of the variances: 0.75), hence the
    lower bar of the ratio."""
    from concrete_amd import client as CL
    from concrete_amd.backend import CAMD_PBS_AUTO
    from test_gpu_parity import check_noise_variance, fft_variance, gadget_variance

    p, small, glwe, bsk, ksk = full_size_keys(pname)
    o = O.Oracle(p, 64)
    # 1_7 (three levels of 2^11) has so little output noise that the accumulator's rounding is most of it:
    # CAMD_PBS_AUTO keeps the bit-exact kernel there, and the instance is tested as what it is, on request
    three_levels = p.pbs_level == 3
    dk = DeviceServerKey.from_standard_keys(gpu_ctx, p, bsk, ksk, variant=CAMD_PBS_WAVE if three_levels else CAMD_PBS_AUTO)
    assert dk.variant == CAMD_PBS_WAVE
    ref = DeviceServerKey.from_standard_keys(gpu_ctx, p, bsk, ksk, variant=CAMD_PBS_AUTO if three_levels else CAMD_PBS_BLOCK)
    assert ref.variant == CAMD_PBS_BLOCK
    msg, sup = p.message_modulus, p.message_modulus * p.carry_modulus
    lut = o.generate_accumulator(lambda x: x % msg)
    dk.upload_lut(lut)
    ref.upload_lut(lut)
    delta = O.shortint_delta(p)
    count = 2048
    msgs = np.arange(count) % sup
    cts = CL.lwe_encrypt(CL.Rng(77), 64, glwe, [int(m) * delta for m in msgs], p.lwe_modular_std_dev)
    got = dk.ks_pbs(cts)
    want = ref.ks_pbs(cts)
    n, k, N = p.lwe_dimension, p.glwe_dimension, p.polynomial_size
    gadget = gadget_variance(p)
    accumulator_rounding = n * (1.0 + k * N / 2.0) * 2.0 ** -64 / 12.0
    bound = gadget + fft_variance(p, o) + accumulator_rounding
    nearest = lambda ph: ((ph + np.uint64(delta // 2)) // np.uint64(delta)) * np.uint64(delta)
    # the ratio to the reference arithmetic may exceed one by the accumulator's share of the prediction
    upper = 1.15 * bound / (bound - accumulator_rounding)
    near_g, near_w = check_noise_variance(64, count, count, bound, gadget, got, want, glwe, nearest, pname,
                                           ratio_bar=(0.55, upper))
    dec_g, dec_w = (near_g // np.uint64(delta)) % np.uint64(2 * sup), (near_w // np.uint64(delta)) % np.uint64(2 * sup)
    assert np.array_equal(dec_g, dec_w)
    # a ciphertext both kernels decode alike but not to f(m) is the parameter set's own failure rate at
    # work (keyswitch + modulus-switch noise against the half box; 7_0 has no carry space to spare)
    wrong = int(np.count_nonzero(dec_g != (msgs % msg).astype(np.uint64)))
    print(f"{pname}: {wrong} of {count} decode to another message on both kernels")
    assert wrong <= count // 256
    assert np.all(got & np.uint64(0xFFFFFFFF) == 0)       # the 32-bit accumulator: low dwords are zero
    dk.close()
    ref.close()


@pytest.mark.parametrize("n", [1, 7])
@pytest.mark.parametrize("pname", ["PARAM_MESSAGE_4_CARRY_4", "PARAM_MESSAGE_1_CARRY_7", "PARAM_MESSAGE_3_CARRY_4", "PARAM_MESSAGE_7_CARRY_0",
                                   "PARAM_MESSAGE_2_CARRY_4", "PARAM_MESSAGE_3_CARRY_3", "PARAM_MESSAGE_2_CARRY_3"])
def test_wave_kernel_largest_polynomial_short(gpu_ctx, pname, n):
    """The throughput kernel of the N >= 8192 classes (pbs_wave16k.hip: N = 16384 with two and three
    levels, N = 8192 with two levels and with one) on a shortened LWE dimension: one CMUX pins the
    transform, the key layout and the digits against the oracle almost word for word (the f64
    results differ in their last bits only, and nothing has diverged yet); a few CMUXes must
    decrypt like the oracle with a phase error far inside the decoding margin.
    Parameter sets: concrete-shortint/src/parameters/mod.rs:575-808."""
    p = P.SHORTINT_BY_NAME[pname]
    ks = get_keys(p, 64, n_override=n)
    o = ks.o
    dk = device_key(gpu_ctx, ks, CAMD_PBS_WAVE)
    assert dk.variant == CAMD_PBS_WAVE
    msg, sup = p.message_modulus, p.message_modulus * p.carry_modulus
    fs = [lambda x: x % msg, lambda x: x // msg]
    luts = [o.generate_accumulator(f) for f in fs]
    for l in luts:
        dk.upload_lut(l)
    delta = O.shortint_delta(p)
    count = 12
    msgs = [0, 1, msg - 1, msg, sup // 2 - 1, sup // 2, sup - 1] + [(37 * i + 5) % sup for i in range(count - 7)]
    cts = np.stack([o.lwe_encrypt(ks.rng, ks.glwe, m * delta, ks.lwe_std) for m in msgs])
    ids = (np.arange(count) % 2).astype(np.uint32)
    got = dk.ks_pbs(cts, ids)
    want = o.ks_pbs_batch(ks.ksk, ks.fbsk, np.concatenate(luts), ids, cts, threads=8)
    diff = np.abs((got - want).view(np.int64).astype(np.float64)) / 2.0 ** 64
    print(f"n = {n}: largest word difference from the oracle {diff.max():.3e} of the torus")
    # (with a single level a difference of exactly half the base has two digits, -B/2 here and +B/2 in the
    # reference: the same plaintext under unrelated masks, so the words are only compared for two levels)
    if n == 1 and p.pbs_level > 1:
        assert diff.max() < 2.0 ** -28        # rounding of the accumulator to 2^-32 plus the f64 transforms
    for i in range(count):
        pg, pw = o.lwe_phase(ks.glwe, got[i]), o.lwe_phase(ks.glwe, want[i])
        assert O.shortint_decode(p, pg) == O.shortint_decode(p, pw), i
        err = ((pg - pw + 2 ** 63) % 2 ** 64 - 2 ** 63) / float(delta)
        assert abs(err) < 0.05, (i, err)
    dk.close()


@pytest.mark.parametrize("pname,bits", [("PARAM_MESSAGE_1_CARRY_1", 64), ("PARAM_MESSAGE_4_CARRY_4", 64),
                                        ("PARAM_MESSAGE_1_CARRY_7", 64), ("DEFAULT_PARAMETERS", 32), ("TFHE_LIB_PARAMETERS", 32)])
def test_matrix_core_keyswitch_every_torus_and_decomposition(gpu_ctx, pname, bits, monkeypatch):
    """The integer-MFMA keyswitch over both torus widths and the extremes of the keyswitch
    decompositions (base 2^2 with 8 levels ... base 2^4 with 5; input dimensions 1024 ... 16384):
    word for word the oracle's keyswitch and the f64-FMA kernel's."""
    monkeypatch.setenv("CAMD_KS_MFMA_MIN", "1")
    p = P.SHORTINT_BY_NAME[pname] if bits == 64 else getattr(P, pname)
    ks = get_keys(p, bits, n_override=40)
    o = ks.o
    dk = device_key(gpu_ctx, ks, CAMD_PBS_GENERIC)
    dt = np.uint64 if bits == 64 else np.uint32
    rs = np.random.RandomState(9)
    count = 70
    cts = (rs.randint(0, 2 ** 31, size=(count, o.big + 1), dtype=np.int64).astype(dt) * dt(2 ** (bits - 31))
           + rs.randint(0, 2 ** 31, size=(count, o.big + 1), dtype=np.int64).astype(dt))
    cts[1, :] = dt(2 ** bits - 1)
    cts[2, :-1] = 0
    got = dk.ks(cts)
    assert got.dtype == dt
    for i in range(0, count, 3):
        assert np.array_equal(got[i], o.keyswitch(ks.ksk, cts[i])), i
    monkeypatch.setenv("CAMD_KS_MFMA_MIN", "0")
    assert np.array_equal(dk.ks(cts), got)
    dk.close()
