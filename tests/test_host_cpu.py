"""CPU tests of the host side: the C ABI library loads and exports every declared symbol, the
client-side operations agree bit for bit with the oracle, the shortint mirror encodes and decodes
like the reference, and the multi-process sharding/broadcast plumbing works under gloo."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import concrete_amd.parameters as P
from concrete_amd import _ffi, client as CL
from concrete_amd.distributed import shard_bounds
from oracle import cpu_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for h in os.listdir(os.path.join(ROOT, "include")):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(camd_[a-z0-9_]+)\s*\(", text))
    return names


def test_library_exports_every_declared_symbol():
    lib = _ffi.lib()
    declared = declared_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in include/ but not exported"
    # and the Python binding table covers the headers
    assert declared == set(_ffi.SYMBOLS), declared ^ set(_ffi.SYMBOLS)
    assert b"gfx950" in lib.camd_version()


def test_rust_declarations_match_the_headers():
    """rust/concrete-amd-sys/src/lib.rs is generated from the two headers: the committed file must
    be what the generator writes today, every declared function must be in it exactly once with
    the header's argument count, and the wrapper crates may only call functions it declares."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import gen_rust_sys as G

    text = open(G.OUT).read()
    assert text == G.render(), "run python scripts/gen_rust_sys.py"
    decls = G.declarations()
    assert {d[0] for d in decls} == declared_symbols()
    for name, _ret, params in decls:
        m = re.findall(r"pub fn %s\(([^)]*)\)" % name, text)
        assert len(m) == 1, name
        assert len([a for a in m[0].split(",") if a.strip()]) == len(params), name
    # struct layouts: same field order and widths as the C structs
    assert re.search(r"pub struct CamdParams \{\s*pub lwe_dimension: u32,\s*pub glwe_dimension: u32,\s*pub polynomial_size: u32,"
                     r"\s*pub pbs_base_log: u32,\s*pub pbs_level: u32,\s*pub ks_base_log: u32,\s*pub ks_level: u32,"
                     r"\s*pub torus_bits: u32,", text)
    import ctypes

    assert ctypes.sizeof(_ffi.CamdRng) == 4 * 8 + 8 + 4 + 4 + 8 * 8   # CamdRng in lib.rs: [u64;4], u64, u32, u32, [u64;8]
    for wrapper in ("rust/concrete-shortint-amd/src/lib.rs", "rust/concrete-boolean-amd/src/amd.rs"):
        used = set(re.findall(r"\b(camd_[a-z0-9_]+)\s*\(", open(os.path.join(ROOT, wrapper)).read()))
        assert used and used <= {d[0] for d in decls}, used - {d[0] for d in decls}


def test_no_gpu_is_an_error_not_a_fallback():
    """Without a usable device the library reports CAMD_ENODEV (or the device is there)."""
    lib = _ffi.lib()
    h = C.c_void_p()
    rc = lib.camd_context_create(0, C.byref(h))
    if rc != 0:
        assert rc in (-4, -2, -1)
        assert len(lib.camd_last_error()) > 0
        with pytest.raises(_ffi.CamdError):
            _ffi.check(rc)
    else:
        lib.camd_context_destroy(h)


def test_bad_arguments_are_rejected():
    lib = _ffi.lib()
    assert lib.camd_context_create(0, None) == -1
    assert lib.camd_lut_upload(None, None, None) == -1
    assert lib.camd_ks_pbs_batch(None, None, None, None, 4) == -1
    assert lib.camd_ks_pbs_batch(None, None, None, None, 0) == -1
    rng = CL.Rng(1)
    with pytest.raises(_ffi.CamdError):
        _ffi.check(lib.camd_client_gen_binary_key(C.byref(rng.state), 48, 8, None))


@pytest.mark.parametrize("bits,params", [(64, P.PARAM_MESSAGE_2_CARRY_2), (64, P.PARAM_MESSAGE_1_CARRY_1),
                                         (32, P.DEFAULT_PARAMETERS)])
def test_client_ops_equal_oracle(bits, params):
    o = O.Oracle(params, bits)
    r1, r2 = O.Rng(77), CL.Rng(77)
    n = 6
    s1, s2 = o.gen_binary_key(r1, n), CL.gen_binary_key(r2, bits, n)
    big = params.glwe_dimension * params.polynomial_size
    g1, g2 = o.gen_binary_key(r1, big), CL.gen_binary_key(r2, bits, big)
    assert np.array_equal(s1, s2) and np.array_equal(g1, g2)
    assert np.array_equal(o.gen_bsk(r1, s1, g1), CL.gen_bootstrap_key(r2, params, bits, s2, g2, threads=3))
    assert np.array_equal(o.gen_ksk(r1, g1, s1), CL.gen_keyswitch_key(r2, params, bits, g2, s2))
    pt = 5 << (bits - 5)
    c1 = o.lwe_encrypt(r1, g1, pt, params.lwe_modular_std_dev)
    c2 = CL.lwe_encrypt(r2, bits, g2, [pt], params.lwe_modular_std_dev)[0]
    assert np.array_equal(c1, c2)
    assert o.lwe_phase(g1, c1) == int(CL.lwe_phase(bits, g2, c2)[0])


def test_accumulators_equal_oracle():
    p = P.PARAM_MESSAGE_2_CARRY_2
    o = O.Oracle(p, 64)
    for f in (lambda x: x % 4, lambda x: x // 4, lambda x: (x * 3) % 16):
        assert np.array_equal(o.generate_accumulator(f), CL.shortint_accumulator(p, f))
    b = P.DEFAULT_PARAMETERS
    assert np.array_equal(O.Oracle(b, 32).boolean_accumulator(), CL.boolean_accumulator(b))


def test_shortint_client_key_roundtrip():
    """encrypt / decrypt of engine/client_side.rs:45-143, every message and carry value."""
    from concrete_amd.shortint import ClientKey

    p = P.PARAM_MESSAGE_2_CARRY_2
    cks = ClientKey.new(p, seed=3)
    for m in range(16):
        ct = cks.unchecked_encrypt(m)
        assert ct.degree == 15 and ct.ct.shape == (2049,)
        assert cks.decrypt_message_and_carry(ct) == m
        assert cks.decrypt(ct) == m % 4
    ct = cks.encrypt(7)
    assert ct.degree == 3 and cks.decrypt(ct) == 3
    ct = cks.encrypt_with_message_modulus(5, 8)
    assert (ct.message_modulus, ct.carry_modulus, ct.degree) == (8, 2, 7)
    assert cks.decrypt_message_and_carry(ct) == 5
    # decode rounding: half a box below and above
    delta = (1 << 63) // 16
    assert cks.decode(3 * delta + delta // 2 - 1) == 3 and cks.decode(3 * delta + delta // 2) == 4
    assert cks.decode((1 << 64) - 1) == 0 or cks.decode((1 << 64) - 1) == 32


def test_shard_bounds_partition():
    for total in (0, 1, 7, 16384, 131072, 1000003):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


_GLOO_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from concrete_amd.distributed import shard_bounds, broadcast_blob
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
# rank 0 owns the key blob; everyone ends up with the same bytes
meta = torch.zeros(2, dtype=torch.int64)
blob = None
if rank == 0:
    g = torch.Generator().manual_seed(5)
    blob = torch.randint(0, 256, (123457,), dtype=torch.uint8, generator=g)
    meta[0], meta[1] = blob.numel(), 2
blob, meta = broadcast_blob(blob, meta, 0)
g = torch.Generator().manual_seed(5)
want = torch.randint(0, 256, (123457,), dtype=torch.uint8, generator=g)
assert torch.equal(blob, want) and int(meta[1]) == 2
# index sharding of a batch: every rank transforms its own shard, the union is the whole batch
total = 1001
batch = torch.arange(total, dtype=torch.int64)
b, e = shard_bounds(total, world, rank)
mine = batch[b:e] * 3 + 1          # stands in for the per-ciphertext work, no communication
counts = [shard_bounds(total, world, r)[1] - shard_bounds(total, world, r)[0] for r in range(world)]
gathered = [torch.empty(c, dtype=torch.int64) for c in counts]
dist.all_gather(gathered, mine) if len(set(counts)) == 1 else None
if len(set(counts)) != 1:
    # ragged shards: gather through padded tensors
    pad = max(counts)
    padded = torch.zeros(pad, dtype=torch.int64); padded[: e - b] = mine
    outs = [torch.zeros(pad, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(outs, padded)
    gathered = [o[:c] for o, c in zip(outs, counts)]
assert torch.equal(torch.cat(gathered), batch * 3 + 1)
# max-over-ranks timing reduction used by the benchmark
t = torch.tensor([float(rank + 1)], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t) == float(world)
dist.destroy_process_group()
print("ok", rank)
"""


def test_two_process_gloo_plumbing(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29577", str(script), ROOT]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert res.stdout.count("ok") == 2


def _bench(*argv, env=None):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True,
                          timeout=300, env=e)


def test_bench_gpus_flag_starts_that_many_ranks():
    """`bench.py --gpus 2` with no launcher around it starts two ranks itself (the CPU rehearsal
    of the plumbing: gloo, no kernels) and rank 0 reports both."""
    import json

    r = _bench("--gpus", "2", "--plumbing", "--batch", "64")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = lines[0]
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 128 and line["config"]["batch_per_gpu"] == 64
    assert line["ranks_with_identical_key_blob"] == 2 and line["key_broadcast_ms"] is not None
    assert line["plumbing_only"] is True and line["value"] is None
    # counted through the process group, not read from the launcher's environment
    assert line["ranks_reporting"] == 2 and line["backend"] == "gloo"


def test_bench_parent_fails_when_a_rank_fails():
    """A rank other than 0 that dies takes the whole run down: the parent that started the ranks
    exits non-zero and no JSON line is printed (rank 0 never gets through its first collective)."""
    r = _bench("--gpus", "2", "--plumbing", "--batch", "64", env={"CAMD_BENCH_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")], r.stdout
    assert "rank 1 fails on request" in r.stderr


def test_bench_rejects_a_gpus_flag_that_disagrees_with_the_launcher():
    r = _bench("--gpus", "4", "--plumbing", env={"WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode != 0 and "does not match WORLD_SIZE" in r.stderr


def test_wave_kernel_keeps_parked_key_registers_untouched(tmp_path):
    """The wave kernel loads next iteration's key rows into accumulation registers with inline
    assembly, asynchronously: the compiler must never recycle or move those registers (a0..a223)
    between the load and the multiply-accumulate, or it would copy values that have not landed.
    It may use the 32 spare ones as spill space. Checked on the generated code."""
    import re
    import subprocess

    src = os.path.join(ROOT, "concrete_amd", "csrc", "pbs_wave.hip")
    out = tmp_path / "pbs_wave.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", src,
                    "-o", str(out)], check=True, capture_output=True)
    text = out.read_text()
    start = text.index("wave_pbs_kernel_v2")
    body = text[text.index(":", start):text.index("s_endpgm", start)]
    assert "scratch_" not in body and "v_accvgpr_mov_b32" not in body
    parked = 224
    writes = [int(m.group(1)) for m in re.finditer(r"v_accvgpr_write_b32 a(\d+),", body)]
    low = sorted(w for w in writes if w < parked)
    assert low == list(range(0, parked, 4)), low            # only the zero-initialisation of each register quad
    for m in re.finditer(r"ds_read\w* a\[(\d+):", body):
        assert int(m.group(1)) >= parked
    loads = [int(m.group(1)) for m in re.finditer(r"global_load_dwordx4 a\[(\d+):", body)]
    assert sorted(loads) == list(range(0, parked, 4))       # one load site per parked quad


def test_row_step_wave_kernel_keeps_parked_key_registers_untouched(tmp_path):
    """The same invariant for the row-step schedule of the multi-level N = 1024 instance
    (three levels, k = 1: six key polynomials of a step in a0..a191)."""
    import re
    import subprocess

    src = os.path.join(ROOT, "concrete_amd", "csrc", "pbs_wave.hip")
    out = tmp_path / "pbs_wave.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", src,
                    "-o", str(out)], check=True, capture_output=True)
    text = out.read_text()
    labels = [m.start() for m in re.finditer(r"^_ZN\S*wave_pbs_kernel_rows\S*:", text, re.M)]
    assert len(labels) == 1
    body = text[text.index(":", labels[0]):text.index("s_endpgm", labels[0])]
    assert "scratch_" not in body and "v_accvgpr_mov_b32" not in body
    parked = 192
    writes = [int(m.group(1)) for m in re.finditer(r"v_accvgpr_write_b32 a(\d+),", body)]
    assert sorted(w for w in writes if w < parked) == list(range(0, parked, 4))
    for m in re.finditer(r"ds_read\w* a\[(\d+):", body):
        assert int(m.group(1)) >= parked
    loads = [int(m.group(1)) for m in re.finditer(r"global_load_dwordx4 a\[(\d+):", body)]
    assert sorted(loads) == list(range(0, parked, 4))


def test_wave512_kernel_keeps_parked_key_registers_untouched(tmp_path):
    """The same invariant for the N = 512 instance: 14 key slabs of 4 register quads (a0..a223)."""
    import re
    import subprocess

    src = os.path.join(ROOT, "concrete_amd", "csrc", "pbs_wave512.hip")
    out = tmp_path / "pbs_wave512.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", src,
                    "-o", str(out)], check=True, capture_output=True)
    text = out.read_text()
    labels = [m.start() for m in re.finditer(r"^_ZN\S*w410pbs_kernel\S*:", text, re.M)]
    assert len(labels) == 2          # the 32-bit two-level instance and the 64-bit single-level one
    for start in labels:
        body = text[text.index(":", start):text.index("s_endpgm", start)]
        assert "scratch_" not in body and "v_accvgpr_mov_b32" not in body
        parked = 224
        writes = [int(m.group(1)) for m in re.finditer(r"v_accvgpr_write_b32 a(\d+),", body)]
        assert sorted(w for w in writes if w < parked) == list(range(0, parked, 4))
        for m in re.finditer(r"ds_read\w* a\[(\d+):", body):
            assert int(m.group(1)) >= parked
        loads = [int(m.group(1)) for m in re.finditer(r"global_load_dwordx4 a\[(\d+):", body)]
        assert sorted(loads) == list(range(0, parked, 4))


def test_keyswitch_dpp_operand_is_settled(tmp_path):
    """The keyswitch products read their digit through a DPP row broadcast written in inline
    assembly, where the compiler's hazard handling does not look: the broadcast source must not be
    written by either of the two instructions issued before a product that reads it, and the
    products must all be of the broadcast form. Checked on the generated code of both torus widths."""
    import re
    import subprocess

    src = os.path.join(ROOT, "concrete_amd", "csrc", "keyswitch.hip")
    out = tmp_path / "keyswitch.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", src,
                    "-o", str(out)], check=True, capture_output=True)
    lines = [ln.strip() for ln in out.read_text().splitlines()]
    code = [ln for ln in lines if ln and not ln.startswith((";", ".", "//")) and not ln.endswith(":")]
    dpp = re.compile(r"v_fmac_f64_dpp v\[\d+:\d+\], v\[(\d+):(\d+)\], v\[\d+:\d+\] row_newbcast:(\d+) ")
    seen = 0
    for idx, ln in enumerate(code):
        m = dpp.match(ln)
        if not m:
            continue
        seen += 1
        lo, hi = int(m.group(1)), int(m.group(2))
        for prev in code[max(0, idx - 2):idx]:
            if prev.startswith(("v_fmac_f64_dpp", "s_", "ds_read", "global_load")):
                continue   # products write accumulators only; memory results are awaited by s_waitcnt
            dest = re.match(r"v_\w+ v\[?(\d+)(?::(\d+))?\]?,", prev)
            assert dest, prev
            d_lo = int(dest.group(1))
            d_hi = int(dest.group(2)) if dest.group(2) else d_lo
            assert d_hi < lo or d_lo > hi, (prev, ln)
    assert seen == 2 * 16 * 2 + 2 * 16, seen   # two rows x 16 ciphertexts x limbs, u64 and u32 kernels
    assert "v_fma_f64" not in "\n".join(code) and "ds_read_b128" not in "\n".join(code)


def test_headers_are_plain_c_and_the_c_driver_links():
    """include/*.h must be consumable by a C99 compiler (they are what a Rust `-sys` crate or a C
    host binds), and the plain-C driver used by the GPU test must link against the library."""
    import subprocess

    inc = os.path.join(ROOT, "include")
    for h in ("concrete_amd.h", "concrete_amd_client.h"):
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c",
                        os.path.join(inc, h)], check=True, capture_output=True)
    subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "c"), "-s"], check=True, capture_output=True)
    assert os.path.exists(os.path.join(ROOT, "tests", "c", "abi_driver"))


def test_closed_form_of_the_balanced_digit_step():
    """The multi-level wave kernels take a digit as state - (next << BL) with
    next = (state + B/2 - 1 + bit(2 BL - 1 of state)) >> BL (csrc/wave_common.hpp: digit_next32,
    csrc/pbs_wave512.hip: next_digit). That must be the decomposer's recurrence
    (res = state & (B-1); state >>= BL; carry = ((res-1) | state) & res) >> (BL-1); ...), for every
    state the kernels can hold: exhaustively up to 22 bits, sampled above."""
    rng = np.random.RandomState(7)

    def recurrence(s, bl):
        res, st = s & ((1 << bl) - 1), s >> bl
        carry = ((((res - 1) & 0xFFFFFFFF) | st) & res) >> (bl - 1)
        return res - (carry << bl), (st + carry) & 0xFFFFFFFF

    def closed(s, bl):
        nxt = ((s + (1 << (bl - 1)) - 1 + ((s >> (2 * bl - 1)) & 1)) & 0xFFFFFFFF) >> bl
        d = (s - (nxt << bl)) & 0xFFFFFFFF
        return d - (1 << 32) if d >= 1 << 31 else d, nxt

    for bl in (2, 3, 7, 8, 11, 15):
        states = list(range(1 << min(2 * bl + 2, 18))) + [int(x) for x in rng.randint(0, 1 << (32 - bl), size=20000)]
        for s in states:
            assert recurrence(s, bl) == closed(s, bl), (bl, s)
