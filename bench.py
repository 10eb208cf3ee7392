"""Benchmark of the hot path: keyswitch + programmable bootstrap of a batch of LWE ciphertexts.

    python bench.py --gpus N --steps K --warmup W

Workload (the configuration the headline metric is quoted on): concrete-shortint PARAM_MESSAGE_2_CARRY_2, batch 2^14 fresh
ciphertexts per GPU, already resident in HBM. One "step" is one KS+PBS pass over the batch.
With N > 1 every rank owns its own batch of 2^14 (weak scaling); the server key is built on
rank 0 and broadcast once over RCCL; there is no collective in the timed region.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself
(`python -m torch.distributed.run ...` as a child process, before this process makes any GPU
call) and exits with the child's status; started by a launcher (WORLD_SIZE set), --gpus must
equal WORLD_SIZE. When there are more ranks than GPUs (a rehearsal on a one-GPU box) the ranks
share the GPUs round-robin and the process group is gloo. `--plumbing` runs the multi-rank
plumbing only (gloo, no GPU, no kernels: spawn, shards, key-blob broadcast, barriers, reduction).

Prints one JSON line on rank 0 (see the keys below).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import concrete_amd.parameters as P  # noqa: E402
from concrete_amd import client as CL  # noqa: E402
from concrete_amd.backend import CAMD_PBS_AUTO, Context, DeviceServerKey  # noqa: E402
from concrete_amd.shortint import ClientKey  # noqa: E402

# MI355X FP64 vector rate: 256 CUs x 4 SIMDs x 16 FMA lanes x 2 flop x 2.4 GHz. This is the vendor's
# datasheet figure (the FP64 matrix rate is the same number); the microarchitecture guide of this
# image lists no FP64 row, so it is not a measured peak.
FP64_PEAK_TFLOPS = 78.6
SEED = 0xC0FFEE
PROFILES = os.path.join(ROOT, "profiles", "r3")
# shortint workloads: parameter set, default batch per GPU, ciphertexts timed on the host cores
# (10-30 s on 16 threads), the bootstrap kernel CAMD_PBS_AUTO launches for it and the committed
# counter summary of this same command, failure probability of the parameter set per bootstrap
SHORTINT_WORKLOADS = {
    "2_2": ("PARAM_MESSAGE_2_CARRY_2", 1 << 14, 4096, "wave8_pbs_kernel", "pmc_per_dispatch_final.json", 2.0 ** -13.9),
    "4_4": ("PARAM_MESSAGE_4_CARRY_4", 512, 256, "w16::pbs_kernel", "pmc_per_dispatch_4_4_final.json", None),
    "3_3": ("PARAM_MESSAGE_3_CARRY_3", 4096, 128, None, None, None),
}
CSRC = os.path.join(ROOT, "concrete_amd", "csrc")


def kernel_source_hash() -> str:
    """sha256 over the kernel sources: ties a committed counter summary to the code it measured."""
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        if name.endswith((".hip", ".hpp", ".cpp")):
            with open(os.path.join(CSRC, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def measured_traffic(kernel: str, summary: str):
    """(bytes per launch, provenance) of `kernel` from the committed rocprofv3 counter passes of
    this same command (FETCH_SIZE and WRITE_SIZE cannot be read from inside the process). The
    figure is only reported when the summary was taken from the kernel sources as they are now."""
    path = os.path.join(PROFILES, summary)
    try:
        with open(path) as f:
            doc = json.load(f)
        e = doc[kernel]
        prov = {"from_profile": os.path.relpath(path, ROOT), "profile_source_hash": doc.get("source_hash"),
                "source_hash": kernel_source_hash()}
        if doc.get("source_hash") != prov["source_hash"]:
            return None, dict(prov, stale=True)
        prov["counted_fp64_flop_per_launch"] = e.get("counted_fp64_flop")
        return float(e["hbm_fetch_bytes"] + e["hbm_write_bytes"]), prov
    except (OSError, KeyError, ValueError):
        return None, {"from_profile": None}


def flop_per_pbs(p) -> float:
    """Algorithmic flop of one bootstrap: n CMUXes of (k+1)L forward + (k+1) inverse FFTs of
    N/2 points (5 M log2 M each), (k+1)^2 L M complex FMAs (8 flop) and ((k+1)L + (k+1)) M
    twist multiplies (6 flop)."""
    k, L, M, n = p.glwe_dimension, p.pbs_level, p.polynomial_size // 2, p.lwe_dimension
    fft = 5.0 * M * np.log2(M)
    per_cmux = (k + 1) * L * fft + (k + 1) * fft + (k + 1) ** 2 * L * M * 8 + ((k + 1) * L + (k + 1)) * M * 6
    return float(n * per_cmux)


def make_inputs(cks: ClientKey, count: int, device, seed: int):
    """Fresh encryptions of uniform messages, built on the device with wrapping int64 arithmetic."""
    p = cks.parameters
    big = p.glwe_dimension * p.polynomial_size
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    mask = torch.randint(-2 ** 63, 2 ** 63 - 1, (count, big), dtype=torch.int64, device=device, generator=g)
    key = torch.from_numpy(cks.lwe_secret_key.astype(np.int64)).to(device)
    msgs = torch.randint(0, p.message_modulus, (count,), dtype=torch.int64, device=device, generator=g)
    delta = (1 << 63) // (p.message_modulus * p.carry_modulus)
    noise = torch.round(torch.randn(count, dtype=torch.float64, device=device, generator=g)
                        * (p.lwe_modular_std_dev * 2.0 ** 64)).to(torch.int64)
    body = (mask * key).sum(dim=1) + msgs * delta + noise
    cts = torch.cat([mask, body[:, None]], dim=1).contiguous()
    return cts, msgs


def decode_outputs(cks: ClientKey, out: torch.Tensor) -> torch.Tensor:
    """decrypt_message_and_carry on the device (client_side.rs:128-138), for the self-check."""
    p = cks.parameters
    key = torch.from_numpy(cks.lwe_secret_key.astype(np.int64)).to(out.device)
    phase = out[:, -1] - (out[:, :-1] * key).sum(dim=1)
    delta = (1 << 63) // (p.message_modulus * p.carry_modulus)
    rounding = (phase & (delta >> 1)) << 1
    v = phase + rounding
    # unsigned division of the wrapped 64-bit value by delta (a power of two)
    shift = delta.bit_length() - 1
    return (v >> shift) & ((1 << (64 - shift)) - 1)


def cpu_baseline(p, bsk, ksk, lut, cts_host, cores: int):
    """The CPU oracle (a port of the reference's algorithm, not concrete-core itself) timed on
    the host cores on a bounded sample of the same batch."""
    from oracle import cpu_oracle as O

    o = O.Oracle(p, 64)
    fbsk = o.bsk_to_fourier(bsk)
    sample = cts_host
    t0 = time.perf_counter()
    res = o.ks_pbs_batch(ksk, fbsk, lut, None, sample, threads=cores)
    dt = time.perf_counter() - t0
    return {"value": sample.shape[0] / dt, "unit": "bootstraps/s", "cores": cores, "kind": "port",
            "sample": f"{sample.shape[0]} ciphertexts of the same batch, {cores} threads, one ciphertext per thread, "
                      f"{dt:.2f} s wall"}, res


OTHER_WORKLOADS = {
    # name: (parameter set, default batch per GPU, torus bits, order)
    "boolean": ("DEFAULT_PARAMETERS", 1 << 15, 32, "pbs_ks"),
}


class Ranks:
    """Rank layout of this process. `dist` is torch.distributed when a process group is up."""

    def __init__(self, args):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        self.backend = None
        self.device_index = None
        if args.plumbing:
            self.backend = "gloo"
        else:
            ndev = torch.cuda.device_count()   # counting devices does not initialise the GPU
            if ndev == 0:
                raise SystemExit("bench.py: no GPU visible (use --plumbing for the CPU rehearsal of the rank plumbing)")
            # more ranks than GPUs (rehearsal on a one-GPU box): ranks share GPUs, and RCCL refuses
            # two ranks on one device, so the process group is gloo and the key travels via the host
            self.device_index = self.local_rank % ndev
            self.backend = "nccl" if self.world <= ndev else "gloo"
            torch.cuda.set_device(self.device_index)
        # CAMD_BENCH_FORCE_DIST=1 takes the multi-rank code path with a single rank
        if self.world > 1 or os.environ.get("CAMD_BENCH_FORCE_DIST") == "1":
            import torch.distributed as dist

            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            if self.backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", self.device_index))
            else:
                dist.init_process_group(backend="gloo")
            self.dist = dist

    @property
    def device(self):
        return torch.device("cpu") if self.device_index is None else torch.device("cuda", self.device_index)

    def sync(self) -> None:
        if self.device_index is not None:
            torch.cuda.synchronize()

    def barrier_sync(self) -> None:
        """barrier bracketed by device synchronisation, as the timing contract asks"""
        self.sync()
        if self.dist is not None:
            self.dist.barrier()
        self.sync()

    def max_over_ranks(self, x: float) -> float:
        if self.dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=self.device if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, x: int) -> int:
        if self.dist is None:
            return x
        t = torch.tensor([x], dtype=torch.int64, device=self.device if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t)
        return int(t.item())

    def census(self) -> dict:
        """What the process group itself saw, for the JSON line: the number of ranks that took part
        in a collective (not the launcher's WORLD_SIZE), and the backend that carried it."""
        return {"ranks_reporting": self.sum_over_ranks(1),
                "backend": self.backend if self.dist is not None else None}

    def close(self) -> None:
        if self.dist is not None:
            self.dist.destroy_process_group()


_JSON_OUT = None


def claim_stdout() -> None:
    """The JSON line is the only thing this process writes to its standard output: keep a private
    copy of descriptor 1 for it and send everything else that is written there (gloo prints its rank
    connections on it) to standard error."""
    global _JSON_OUT
    if _JSON_OUT is None:
        sys.stdout.flush()
        _JSON_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def emit(line: dict) -> None:
    out = _JSON_OUT if _JSON_OUT is not None else sys.stdout
    out.write(json.dumps(line) + "\n")
    out.flush()


def spawn_ranks(args, argv) -> int:
    """Start `args.gpus` ranks of this script under torch.distributed.run and wait for them. Runs in
    a process that has made no GPU call; rank 0's JSON line reaches stdout through the inherited
    descriptor."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1" if args.plumbing else env.get("OMP_NUM_THREADS", "16"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def plumbing(args) -> None:
    """The multi-rank plumbing with no GPU: process group (gloo), index shards, one broadcast of a
    key-sized blob, barriers around an empty timed region, max-reduce of the time. Checks that
    `--gpus N` really runs N ranks; computes nothing, so `value` is null."""
    from concrete_amd.distributed import broadcast_blob, shard_bounds

    rk = Ranks(args)
    if os.environ.get("CAMD_BENCH_FAIL_RANK") == str(rk.rank):
        # failure injection for the test of the parent's exit status: this rank dies before its
        # first collective, as a rank with a broken GPU would
        raise SystemExit(f"bench.py: rank {rk.rank} fails on request (CAMD_BENCH_FAIL_RANK)")
    count = args.batch
    blob_bytes = 1 << 20   # stand-in for the 94.5 MB key blob
    meta = torch.zeros(2, dtype=torch.int64)
    blob = None
    if rk.rank == 0:
        meta[0], meta[1] = blob_bytes, 2
        blob = torch.arange(blob_bytes, dtype=torch.int64).to(torch.uint8)
    key_broadcast_ms = None
    if rk.dist is not None:
        rk.barrier_sync()
        tb = time.perf_counter()
        blob, meta = broadcast_blob(blob, meta, 0)
        key_broadcast_ms = rk.max_over_ranks((time.perf_counter() - tb) * 1e3)
    checksum = int(blob.to(torch.int64).sum().item())
    lo, hi = shard_bounds(count * rk.world, rk.world, rk.rank)
    rk.barrier_sync()
    t0 = time.perf_counter()
    rk.barrier_sync()
    elapsed = rk.max_over_ranks(time.perf_counter() - t0)
    owned = rk.sum_over_ranks(hi - lo)
    agree = rk.sum_over_ranks(int(checksum == int(torch.arange(blob_bytes, dtype=torch.int64).to(torch.uint8).to(torch.int64).sum())))
    census = rk.census()
    if rk.rank == 0:
        emit({
            "metric": "bootstraps/sec (node), MESSAGE_2_CARRY_2", "value": None, "unit": "bootstraps/s",
            "n_gpus": rk.world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "none",
            "plumbing_only": True, **census,
            "config": {"workload": "rank plumbing only (PARAM_MESSAGE_2_CARRY_2 shards), no kernels",
                       "batch_per_gpu": count, "global_batch": owned},
            "ranks_with_identical_key_blob": agree, "key_broadcast_ms": key_broadcast_ms,
            "barrier_region_ms": elapsed * 1e3})
    rk.close()


def other_workload(args) -> None:
    """The boolean gate bootstrap (PBS then keyswitch): same contract, random key material and
    random ciphertexts, so there is no decryption check and no CPU baseline."""
    rk = Ranks(args)
    world, rank, dist = rk.world, rk.rank, rk.dist
    pname, default_batch, bits, order = OTHER_WORKLOADS[args.workload]
    p = getattr(P, pname)
    count = args.batch if args.batch != 1 << 14 else default_batch
    n, k, N, L = p.lwe_dimension, p.glwe_dimension, p.polynomial_size, p.pbs_level
    dt = np.uint64 if bits == 64 else np.uint32
    rs = np.random.RandomState(SEED & 0xFFFF)

    def rnd(shape):
        return rs.randint(0, 2 ** 31, size=shape, dtype=np.int64).astype(dt) * dt(2)

    ctx = Context(rk.device_index)
    dk = None
    if rank == 0:
        dk = DeviceServerKey.from_standard_keys(ctx, p, rnd((n, L, k + 1, k + 1, N)), rnd((k * N, p.ks_level, n + 1)),
                                                variant=args.variant, torus_bits=bits)
    if dist is not None:
        from concrete_amd.distributed import replicate_server_key

        dk = replicate_server_key(ctx, p, dk if rank == 0 else None, torus_bits=bits)
    lut = np.zeros((k + 1) * N, dtype=dt)
    lut[k * N:] = rnd(N)
    dk.upload_lut(lut)
    device = rk.device
    width = (k * N + 1) if order == "ks_pbs" else (n + 1)
    tdt = torch.int64 if bits == 64 else torch.int32
    cts = torch.randint(-2 ** 31, 2 ** 31 - 1, (count, width), dtype=torch.int64, device=device)
    cts = (cts * 4294967297).to(tdt) if bits == 64 else cts.to(tdt)
    out = torch.empty_like(cts)
    dk.reserve(count)
    stream = torch.cuda.current_stream().cuda_stream
    step = (lambda: dk.ks_pbs_device(cts, None, out, count, stream)) if order == "ks_pbs" else \
        (lambda: dk.pbs_ks_device(cts, None, out, count, stream))
    for _ in range(args.warmup):
        step()
    rk.barrier_sync()
    dk.timing_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    rk.barrier_sync()
    elapsed = rk.max_over_ranks(time.perf_counter() - t0)
    tm = dk.timing_read()
    census = rk.census()
    if rank == 0:
        fpp = flop_per_pbs(p)
        achieved = count * fpp / (tm["pbs_ms"] * 1e-3) / 1e12 if tm["pbs_ms"] > 0 else 0.0
        emit({
            "metric": f"bootstraps/sec (node), {pname}", "value": count * world * args.steps / elapsed,
            "unit": "bootstraps/s", "n_gpus": world, **census, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic (random key material and ciphertexts)",
            "config": {"workload": f"{pname} {'keyswitch+PBS' if order == 'ks_pbs' else 'PBS+keyswitch (gate bootstrap)'}, "
                                   f"batch {count} per GPU, inputs resident in HBM",
                       "batch_per_gpu": count, "global_batch": count * world,
                       "pbs_kernel": {1: "generic", 2: "wave", 3: "block"}.get(dk.variant, str(dk.variant))},
            "roofline": {"bound": "valu_fp64", "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_PEAK_TFLOPS, "traffic": None, "flop_per_bootstrap": fpp,
                         "kernel_ms": tm["pbs_ms"], "keyswitch_kernel_ms": tm["ks_ms"],
                         "note": "peak is the vendor's FP64 vector figure; the guide of this image has no FP64 row"},
            "cpu_baseline": None})
    rk.close()


def fheuint8_workload(args) -> None:
    """FheUint8 (four blocks of PARAM_MESSAGE_2_CARRY_2) multiplications on device-resident batches
    through the radix layer of concrete_amd.integer: one step = one smart_mul of `batch` independent
    pairs per GPU (real keys; every product is decrypted and compared after the timed region).
    Timed twice: under the default schedule, which leaves out the bootstraps that cannot change a
    decryption (`value`), and under the reference's own (every propagation step bootstraps every
    block twice, concrete-integer radix_parallel/mod.rs:44-56; `reference_schedule` in the line)."""
    from concrete_amd import integer as INT

    rk = Ranks(args)
    world, rank = rk.world, rk.rank
    p = P.PARAM_MESSAGE_2_CARRY_2
    pairs = args.batch if args.batch != 1 << 14 else 1024
    ctx = Context(rk.device_index)
    cks = INT.RadixClientKey(p, 4, seed=SEED)          # fixed test seed: the same keys on every rank
    sks = INT.ServerKey.new(cks, ctx=ctx, variant=args.variant, reference_schedule=False)
    rs = np.random.RandomState(1 + rank)
    x = rs.randint(0, 256, size=pairs).astype(np.uint64)
    y = rs.randint(0, 256, size=pairs).astype(np.uint64)
    cx, cy = cks.encrypt_many(x, device=rk.device), cks.encrypt_many(y, device=rk.device)
    dk = sks.key.key

    def timed(reference_schedule: bool) -> dict:
        sks.reference_schedule = reference_schedule

        def step():
            return sks.smart_mul(cx.clone(), cy.clone())

        for _ in range(max(args.warmup, 1)):   # the first call uploads the look-up tables
            step()
        rk.barrier_sync()
        dk.timing_enable(True)
        l0, b0 = dk.launches, dk.bootstraps
        t0 = time.perf_counter()
        for _ in range(args.steps):
            prod = step()
        rk.barrier_sync()
        elapsed = rk.max_over_ranks(time.perf_counter() - t0)
        tm = dk.timing_read()
        dk.timing_enable(False)
        wrong = rk.sum_over_ranks(int(np.count_nonzero(cks.decrypt(prod) != (x * y) % 256)))
        bootstraps = (dk.bootstraps - b0) / args.steps
        return {"value": pairs * world * args.steps / elapsed, "ms_per_step": elapsed / args.steps * 1e3,
                "launches_per_multiplication_batch": (dk.launches - l0) / args.steps,
                "bootstraps_per_multiplication": bootstraps / pairs, "wrong_products": wrong,
                "expected_wrong_products": world * bootstraps * 2.0 ** -13.9,
                "kernel_ms_per_launch": tm["pbs_ms"], "keyswitch_kernel_ms_per_launch": tm["ks_ms"]}

    def timed_addition() -> dict:
        """smart_add of operands whose blocks all hold carries (x + y + y, twice): both operands are
        propagated block by block before the addition, the carry-propagation chain of the configuration."""
        sks.reference_schedule = False

        def step():
            a = sks.unchecked_add(sks.unchecked_add(cx, cy), cy)
            return sks.smart_add(a, a.clone())

        step()
        rk.barrier_sync()
        l0, b0 = dk.launches, dk.bootstraps
        t0 = time.perf_counter()
        for _ in range(args.steps):
            total = step()
        rk.barrier_sync()
        elapsed = rk.max_over_ranks(time.perf_counter() - t0)
        wrong = rk.sum_over_ranks(int(np.count_nonzero(cks.decrypt(total) != (2 * (x + 2 * y)) % 256)))
        bootstraps = (dk.bootstraps - b0) / args.steps
        return {"value": pairs * world * args.steps / elapsed, "unit": "additions/s", "ms_per_step": elapsed / args.steps * 1e3,
                "launches_per_addition_batch": (dk.launches - l0) / args.steps,
                "bootstraps_per_addition": bootstraps / pairs, "wrong_sums": wrong,
                "expected_wrong_sums": world * bootstraps * 2.0 ** -13.9}

    fast = timed(False)
    ref = timed(True)
    addition = timed_addition()
    census = rk.census()
    if rank == 0:
        emit({
            "metric": "FheUint8 multiplications/sec (node), radix of 4 x MESSAGE_2_CARRY_2",
            "value": fast["value"], "unit": "multiplications/s", "n_gpus": world, **census,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": fast["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"concrete-integer smart_mul of {pairs} independent FheUint8 pairs per GPU "
                                   "(fresh operands, device-resident), carry-propagation and block-product "
                                   "bootstraps batched per dependency level",
                       "pairs_per_gpu": pairs, "launches_per_multiplication_batch": fast["launches_per_multiplication_batch"],
                       "bootstraps_per_multiplication": fast["bootstraps_per_multiplication"],
                       "schedule": "default: blocks without a carry are not bootstrapped by a propagation step, "
                                   "nor is the carry of an integer's last block"},
            "addition_with_carry_propagation": addition,
            "reference_schedule": dict(ref, unit="multiplications/s",
                                       note="the same call issuing exactly the reference's bootstraps "
                                            "(radix_parallel/mul.rs:314-333, mod.rs:44-88)"),
            "roofline": {"bound": "valu_fp64", "achieved": None, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": None,
                         "traffic": None, "kernel_ms_per_launch": fast["kernel_ms_per_launch"],
                         "keyswitch_kernel_ms_per_launch": fast["keyswitch_kernel_ms_per_launch"],
                         "note": "the bootstrap kernel of the headline workload at the batch of one dependency level"},
            "cpu_baseline": None, "wrong_products": fast["wrong_products"],
            "expected_wrong_products": fast["expected_wrong_products"]})
    rk.close()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1 << 14, help="ciphertexts per GPU")
    ap.add_argument("--variant", type=int, default=CAMD_PBS_AUTO)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-buffer-pass", action="store_true",
                    help="skip the PCIe-inclusive pass (host buffers in, host buffers out) after the timed region")
    ap.add_argument("--luts", type=int, default=1, help="number of distinct look-up tables spread over the batch")
    ap.add_argument("--plumbing", action="store_true",
                    help="rank plumbing only, on the CPU with gloo: no GPU and no kernels (value is null)")
    ap.add_argument("--workload", default="2_2", choices=sorted(OTHER_WORKLOADS) + sorted(SHORTINT_WORKLOADS) + ["fheuint8"],
                    help="2_2 (default) is the headline configuration; 4_4 and 3_3 run the same measurement (real keys, "
                         "decryption check, CPU baseline) on further parameter sets; boolean times the gate bootstrap "
                         "with random key material")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            # this process has made no GPU call: start the ranks as children and report their status
            raise SystemExit(spawn_ranks(args, sys.argv[1:]))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} does not match WORLD_SIZE={os.environ['WORLD_SIZE']} of the launcher")
    claim_stdout()
    if args.plumbing:
        return plumbing(args)
    if args.workload == "fheuint8":
        return fheuint8_workload(args)
    if args.workload in OTHER_WORKLOADS:
        return other_workload(args)

    rk = Ranks(args)
    world, rank, dist = rk.world, rk.rank, rk.dist
    device = rk.device

    pname, default_batch, cpu_sample, auto_kernel, pmc_summary, failure_rate = SHORTINT_WORKLOADS[args.workload]
    p = getattr(P, pname)
    short_name = pname.replace("PARAM_", "")
    count = args.batch if args.batch != 1 << 14 else default_batch
    ctx = Context(rk.device_index)
    cks = ClientKey(p, seed=SEED)  # secret keys only: cheap and identical on every rank (fixed test seed)

    # ---- server key: built on rank 0, broadcast once (Fourier BSK + KSK in one blob) ----
    bsk = ksk = None
    t_key = time.perf_counter()
    if rank == 0:
        bsk = CL.gen_bootstrap_key(cks._rng, p, 64, cks.lwe_secret_key_after_ks, cks.glwe_secret_key,
                                   threads=min(16, os.cpu_count() or 1))
        ksk = CL.gen_keyswitch_key(cks._rng, p, 64, cks.lwe_secret_key, cks.lwe_secret_key_after_ks)
        dk = DeviceServerKey.from_standard_keys(ctx, p, bsk, ksk, variant=args.variant)
    key_broadcast_ms = None
    if dist is not None:
        from concrete_amd.distributed import replicate_server_key

        rk.barrier_sync()
        tb = time.perf_counter()
        dk = replicate_server_key(ctx, p, dk if rank == 0 else None)
        rk.sync()
        key_broadcast_ms = rk.max_over_ranks((time.perf_counter() - tb) * 1e3)   # the slowest rank's
    key_setup_s = time.perf_counter() - t_key

    lut = CL.shortint_accumulator(p, lambda x: x % p.message_modulus)
    dk.upload_lut(lut)
    # further tables x -> (x + j) % message_modulus, ciphertext i uses table i % luts
    lut_fns = [(lambda x, j=j: (x + j) % p.message_modulus) for j in range(args.luts)]
    for f in lut_fns[1:]:
        dk.upload_lut(CL.shortint_accumulator(p, f))

    cts, msgs = make_inputs(cks, count, device, seed=1 + rank)
    out = torch.empty_like(cts)
    dk.reserve(count)
    stream = torch.cuda.current_stream().cuda_stream

    lut_ids = None
    if args.luts > 1:
        lut_ids = (torch.arange(count, device=device) % args.luts).to(torch.int32)

    def step():
        dk.ks_pbs_device(cts, lut_ids, out, count, stream)

    for _ in range(args.warmup):
        step()
    rk.barrier_sync()
    dk.timing_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    rk.barrier_sync()
    elapsed = rk.max_over_ranks(time.perf_counter() - t0)
    tm = dk.timing_read()
    dk.timing_enable(False)

    # ---- self-check: every output decrypts to f(m) ----
    got = decode_outputs(cks, out)
    expect = msgs % p.message_modulus
    if args.luts > 1:
        expect = (msgs + torch.arange(count, device=device) % args.luts) % p.message_modulus
    bad = torch.nonzero(got != expect).flatten()
    wrong_here = int(bad.numel())
    wrong = rk.sum_over_ranks(wrong_here)
    census = rk.census()

    if rank == 0:
        total = count * world * args.steps
        value = total / elapsed
        fpp = flop_per_pbs(p)
        pbs_s = tm["pbs_ms"] * 1e-3
        achieved = count * fpp / pbs_s / 1e12 if pbs_s > 0 else 0.0
        # the counters of the default schedule (two waves per SIMD); another schedule asked for through
        # CAMD_WAVE_SCHEDULE has no committed summary
        default_schedule = os.environ.get("CAMD_WAVE_SCHEDULE", "3") == "3"
        traffic, provenance = (measured_traffic(auto_kernel, pmc_summary)
                               if auto_kernel and dk.variant == 2 and count == default_batch and default_schedule
                               else (None, {"from_profile": None}))
        key_mb = p.lwe_dimension * p.pbs_level * (p.glwe_dimension + 1) ** 2 * (p.polynomial_size // 2) * 16 / 1e6
        measured_peak = ctx.measure_fp64_fma_rate()   # 5 ms probe: what v_fma_f64 sustains on this GPU right now
        ks_macs = p.glwe_dimension * p.polynomial_size * p.ks_level * (p.lwe_dimension + 1)
        line = {
            "metric": f"bootstraps/sec (node), {short_name}",
            "value": value,
            "unit": "bootstraps/s",
            "n_gpus": world,
            # ranks that answered a collective and the backend that carried it (null backend: one
            # rank, no process group): n_gpus above is the launcher's WORLD_SIZE, this is the group's
            **census,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"concrete-shortint {pname} keyswitch+PBS, batch {count} fresh ciphertexts per GPU, "
                            "inputs resident in HBM, " + (f"one shared look-up table (x % {p.message_modulus})" if args.luts == 1 else
                                                         f"{args.luts} look-up tables interleaved over the batch"),
                "batch_per_gpu": count,
                "global_batch": count * world,
                "parallelism": f"batch sharded over {world} rank(s), one per GPU; server key broadcast once "
                               f"({rk.backend if dist is not None else 'no process group'}), no per-step collective",
                "pbs_kernel": {1: "generic", 2: "wave", 3: "block"}.get(dk.variant, str(dk.variant)),
            },
            "roofline": {
                "bound": "valu_fp64",
                "achieved": achieved,
                "peak": FP64_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / FP64_PEAK_TFLOPS,
                "measured_fp64_fma_peak": measured_peak,
                "frac_of_measured_peak": achieved / measured_peak if measured_peak > 0 else None,
                "traffic": traffic,
                "traffic_provenance": provenance,
                # cross-check of `achieved`: f64 instructions the kernel really executed (rocprofv3
                # SQ_INSTS_VALU_{FMA,ADD,MUL}_F64 of the committed profile, 64 lanes, FMA = 2 flop) over the
                # duration measured in this run; below the algorithmic figure because radix-8 butterflies
                # need fewer operations than the 5 M log2 M convention charges
                "executed_fp64_tflops": (provenance.get("counted_fp64_flop_per_launch") / pbs_s / 1e12
                                         if traffic is not None and provenance.get("counted_fp64_flop_per_launch") and pbs_s > 0 else None),
                "traffic_unit": "bytes per launch past the L2 (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE of this command, "
                                f"from the committed summary; null when the kernel sources changed since); the Fourier "
                                f"key is {key_mb:.1f} MB" + (" and stays in the Infinity Cache, so this is fabric, not HBM, traffic"
                                                             if key_mb < 200 else
                                                             "; most of the figure is the per-ciphertext accumulators (32 bits per "
                                                             "coefficient, 128 KB per ciphertext) leaving and re-entering the L2 once per CMUX"),
                "kernel": "programmable bootstrap (blind rotation), f64 on the vector ALUs",
                "flop_per_bootstrap": fpp,
                "kernel_ms": tm["pbs_ms"],
                "keyswitch_kernel_ms": tm["ks_ms"],
                "keyswitch_int_ops_per_s": (count * ks_macs / (tm["ks_ms"] * 1e-3)) if tm["ks_ms"] > 0 else None,
                "note": "bound: f64 vector ALU (not the matrix cores: per frequency the contraction is a 3x3 complex "
                        "matrix times one vector). peak = 78.6 TFLOP/s is the vendor's FP64 vector figure (256 CUs x 4 "
                        "SIMDs x 16 FMA lanes x 2.4 GHz); the microarchitecture guide of this image has no FP64 row. "
                        "measured_fp64_fma_peak is what independent v_fma_f64 chains sustain on this GPU in this run "
                        "(the chip holds 1.8-2.0 GHz under a dense FP64 load); frac stays the fraction of the datasheet figure. "
                        "keyswitch_int_ops_per_s counts wrapping u64 multiply-adds and is not part of the FP64 fraction",
            },
            # outputs that do not decrypt to f(m). The parameter set itself fails with probability
            # about 2^-13.9 per bootstrap (keyswitch + modulus-switch noise against the half box),
            # i.e. about one ciphertext per 2^14 batch on ANY correct implementation.
            "decrypt_differs_from_cleartext": wrong,
            "expected_failures_of_parameter_set": count * world * failure_rate if failure_rate else None,
            "key_setup_s": key_setup_s,
            "key_broadcast_ms": key_broadcast_ms,
        }
        if not args.no_host_buffer_pass:
            # the same pass through the host-pointer entry point: pageable host buffers in and out,
            # PCIe both ways inside the time. Reported beside `value`, never as `value`.
            host_in = cts.cpu().numpy().view(np.uint64)
            dk.ks_pbs(host_in[: min(2048, count)], None if lut_ids is None else lut_ids[: min(2048, count)].cpu().numpy())  # staging buffers, first touch
            th = time.perf_counter()
            host_out = dk.ks_pbs(host_in, None if lut_ids is None else lut_ids.cpu().numpy())
            dth = time.perf_counter() - th
            line["pcie_inclusive_value"] = count / dth
            line["pcie_inclusive_ms_per_step"] = dth * 1e3
            line["pcie_inclusive_matches_resident"] = bool(np.array_equal(host_out.view(np.int64), out.cpu().numpy()))
        if world == 1 and not args.no_cpu_baseline and args.luts == 1:
            cores = min(16, os.cpu_count() or 1)
            sample = cts[: min(count, cpu_sample)].cpu().numpy().view(np.uint64)
            line["cpu_baseline"], cpu_out = cpu_baseline(p, bsk, ksk, lut, sample, cores)
            # parity on the sample: the GPU and the CPU restatement must decrypt identically
            dec_cpu = cks.decrypt_message_and_carry_batch(cpu_out)
            dec_gpu = got[: sample.shape[0]].cpu().numpy()
            line["parity_vs_cpu_sample"] = {"checked": int(sample.shape[0]),
                                            "decrypt_mismatches": int((dec_cpu != dec_gpu).sum())}
        else:
            line["cpu_baseline"] = None
        if not args.no_cpu_baseline and args.luts == 1:
            # attribution of the ciphertexts that did not decrypt to f(m): the CPU restatement runs
            # on exactly those inputs (rank 0's); a failure it shares is the parameter set's noise
            # budget at work, one it does not share would be the GPU path's own
            shared = None
            if wrong_here:
                from oracle import cpu_oracle as O

                o = O.Oracle(p, 64)
                fb = o.bsk_to_fourier(bsk)
                idx = bad.cpu().numpy()
                cpu_bad = o.ks_pbs_batch(ksk, fb, lut, None, cts[bad].cpu().numpy().view(np.uint64), threads=min(16, len(idx)))
                dec = cks.decrypt_message_and_carry_batch(cpu_bad)
                exp = expect[bad].cpu().numpy()
                shared = int((dec != exp).sum())
                line["failing_indices_rank0"] = [int(i) for i in idx[:16]]
            line["failures_on_rank0"] = wrong_here
            line["failures_shared_with_cpu"] = shared if wrong_here else 0
        emit(line)

    rk.close()


if __name__ == "__main__":
    main()
