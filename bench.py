"""Benchmark of the hot path: keyswitch + programmable bootstrap of a batch of LWE ciphertexts.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): concrete-shortint PARAM_MESSAGE_2_CARRY_2, batch 2^14 fresh
ciphertexts per GPU, already resident in HBM. One "step" is one KS+PBS pass over the batch.
With N > 1 every rank owns its own batch of 2^14 (weak scaling); the server key is built on
rank 0 and broadcast once over RCCL; there is no collective in the timed region.

Prints one JSON line on rank 0 (see the keys below).
"""
from __future__ import annotations

import argparse
import json
import os
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

FP64_PEAK_TFLOPS = 78.6          # MI355X FP64 vector peak (the FP64 matrix peak is the same figure)
SEED = 0xC0FFEE
CPU_SAMPLE = 4096                # ciphertexts timed on the host cores (about 10 s on 16 threads)
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r1", "pmc_per_dispatch_final.json")


def measured_traffic(kernel: str):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 counter passes of this same
    command (FETCH_SIZE and WRITE_SIZE cannot be read from inside the process). None when the
    summary is absent."""
    try:
        with open(PMC_SUMMARY) as f:
            e = json.load(f)[kernel]
        return float(e["hbm_fetch_bytes"] + e["hbm_write_bytes"])
    except (OSError, KeyError, ValueError):
        return None


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
    "4_4": ("PARAM_MESSAGE_4_CARRY_4", 512, 64, "ks_pbs"),
    "3_3": ("PARAM_MESSAGE_3_CARRY_3", 4096, 64, "ks_pbs"),
    "boolean": ("DEFAULT_PARAMETERS", 1 << 15, 32, "pbs_ks"),
}


def other_workload(args) -> None:
    """Further configurations (largest polynomial size, boolean gates): same contract, random key
    material and random ciphertexts, so there is no decryption check and no CPU baseline."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist  # noqa: F811

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    pname, default_batch, bits, order = OTHER_WORKLOADS[args.workload]
    p = getattr(P, pname)
    count = args.batch if args.batch != 1 << 14 else default_batch
    n, k, N, L = p.lwe_dimension, p.glwe_dimension, p.polynomial_size, p.pbs_level
    dt = np.uint64 if bits == 64 else np.uint32
    rs = np.random.RandomState(SEED & 0xFFFF)

    def rnd(shape):
        return rs.randint(0, 2 ** 31, size=shape, dtype=np.int64).astype(dt) * dt(2)

    ctx = Context(local_rank)
    dk = None
    if rank == 0:
        dk = DeviceServerKey.from_standard_keys(ctx, p, rnd((n, L, k + 1, k + 1, N)), rnd((k * N, p.ks_level, n + 1)),
                                                variant=args.variant, torus_bits=bits)
    if world > 1:
        from concrete_amd.distributed import replicate_server_key

        dk = replicate_server_key(ctx, p, dk if rank == 0 else None, torus_bits=bits)
    lut = np.zeros((k + 1) * N, dtype=dt)
    lut[k * N:] = rnd(N)
    dk.upload_lut(lut)
    device = torch.device("cuda", local_rank)
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
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dk.timing_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    tm = dk.timing_read()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        fpp = flop_per_pbs(p)
        achieved = count * fpp / (tm["pbs_ms"] * 1e-3) / 1e12 if tm["pbs_ms"] > 0 else 0.0
        print(json.dumps({
            "metric": f"bootstraps/sec (node), {pname}", "value": count * world * args.steps / elapsed,
            "unit": "bootstraps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic (random key material and ciphertexts)",
            "config": {"workload": f"{pname} {'keyswitch+PBS' if order == 'ks_pbs' else 'PBS+keyswitch (gate bootstrap)'}, "
                                   f"batch {count} per GPU, inputs resident in HBM",
                       "batch_per_gpu": count, "global_batch": count * world,
                       "pbs_kernel": {1: "generic", 2: "wave", 3: "block"}.get(dk.variant, str(dk.variant))},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_PEAK_TFLOPS, "traffic": None, "flop_per_bootstrap": fpp,
                         "kernel_ms": tm["pbs_ms"], "keyswitch_kernel_ms": tm["ks_ms"]},
            "cpu_baseline": None}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1 << 14, help="ciphertexts per GPU")
    ap.add_argument("--variant", type=int, default=CAMD_PBS_AUTO)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--luts", type=int, default=1, help="number of distinct look-up tables spread over the batch")
    ap.add_argument("--workload", default="2_2", choices=sorted(OTHER_WORKLOADS) + ["2_2"],
                    help="2_2 (default) is the headline configuration; the others time further configurations of the "
                         "same path with random key material (timing does not depend on the data)")
    args = ap.parse_args()
    if args.workload != "2_2":
        return other_workload(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # CAMD_BENCH_FORCE_DIST=1 takes the multi-rank code path (RCCL process group, key broadcast,
    # barriers, max-reduce of the time) with a single rank: a rehearsal on a one-GPU box
    use_dist = world > 1 or os.environ.get("CAMD_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist  # noqa: F811

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    p = P.PARAM_MESSAGE_2_CARRY_2
    count = args.batch
    ctx = Context(local_rank)
    cks = ClientKey(p, seed=SEED)  # secret keys only: cheap and identical on every rank

    # ---- server key: built on rank 0, broadcast once (Fourier BSK + KSK in one blob) ----
    bsk = ksk = None
    t_key = time.perf_counter()
    if rank == 0:
        bsk = CL.gen_bootstrap_key(cks._rng, p, 64, cks.lwe_secret_key_after_ks, cks.glwe_secret_key)
        ksk = CL.gen_keyswitch_key(cks._rng, p, 64, cks.lwe_secret_key, cks.lwe_secret_key_after_ks)
        dk = DeviceServerKey.from_standard_keys(ctx, p, bsk, ksk, variant=args.variant)
    key_broadcast_ms = None
    if use_dist:
        from concrete_amd.distributed import replicate_server_key

        torch.cuda.synchronize()
        dist.barrier()
        tb = time.perf_counter()
        dk = replicate_server_key(ctx, p, dk if rank == 0 else None)
        torch.cuda.synchronize()
        key_broadcast_ms = (time.perf_counter() - tb) * 1e3
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
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dk.timing_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    tm = dk.timing_read()
    dk.timing_enable(False)

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- self-check: every output decrypts to f(m) ----
    got = decode_outputs(cks, out)
    expect = msgs % p.message_modulus
    if args.luts > 1:
        expect = (msgs + torch.arange(count, device=device) % args.luts) % p.message_modulus
    wrong = int((got != expect).sum().item())
    if dist is not None:
        w = torch.tensor([wrong], dtype=torch.int64, device=device)
        dist.all_reduce(w)
        wrong = int(w.item())

    if rank == 0:
        total = count * world * args.steps
        value = total / elapsed
        fpp = flop_per_pbs(p)
        pbs_s = tm["pbs_ms"] * 1e-3
        achieved = count * fpp / pbs_s / 1e12 if pbs_s > 0 else 0.0
        line = {
            "metric": "bootstraps/sec (node), MESSAGE_2_CARRY_2",
            "value": value,
            "unit": "bootstraps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "concrete-shortint PARAM_MESSAGE_2_CARRY_2 keyswitch+PBS, batch 2^14 fresh ciphertexts per GPU, "
                            "inputs resident in HBM, " + ("one shared look-up table (x % 4)" if args.luts == 1 else
                                                         f"{args.luts} look-up tables interleaved over the batch"),
                "batch_per_gpu": count,
                "global_batch": count * world,
                "parallelism": f"batch sharded over {world} GPU(s); server key broadcast once, no per-step collective",
                "pbs_kernel": {1: "generic", 2: "wave", 3: "block"}.get(dk.variant, str(dk.variant)),
            },
            "roofline": {
                "bound": "mfma",
                "achieved": achieved,
                "peak": FP64_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / FP64_PEAK_TFLOPS,
                "traffic": measured_traffic("wave_pbs_kernel") if dk.variant == 2 and count == 1 << 14 else None,
                "traffic_unit": "bytes per launch past the L2 (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, "
                                "profiles/r1/pmc_per_dispatch_final.json); the Fourier key is 56.7 MB",
                "kernel": "programmable bootstrap (blind rotation), f64 on the vector ALUs",
                "flop_per_bootstrap": fpp,
                "kernel_ms": tm["pbs_ms"],
                "keyswitch_kernel_ms": tm["ks_ms"],
                "note": "peak is the MI355X FP64 vector rate; the FP64 matrix (MFMA) peak is the same 78.6 TFLOP/s",
            },
            # outputs that do not decrypt to f(m). The parameter set itself fails with probability
            # about 2^-13.9 per bootstrap (keyswitch + modulus-switch noise against the half box),
            # i.e. about one ciphertext per 2^14 batch on ANY correct implementation.
            "decrypt_differs_from_cleartext": wrong,
            "expected_failures_of_parameter_set": count * world * 2.0 ** -13.9,
            "key_setup_s": key_setup_s,
            "key_broadcast_ms": key_broadcast_ms,
        }
        if world == 1 and not args.no_cpu_baseline and args.luts == 1:
            cores = min(16, os.cpu_count() or 1)
            sample = cts[: min(count, CPU_SAMPLE)].cpu().numpy().view(np.uint64)
            line["cpu_baseline"], cpu_out = cpu_baseline(p, bsk, ksk, lut, sample, cores)
            # parity on the sample: the GPU and the CPU restatement must decrypt identically
            dec_cpu = cks.decrypt_message_and_carry_batch(cpu_out)
            dec_gpu = got[: sample.shape[0]].cpu().numpy()
            line["parity_vs_cpu_sample"] = {"checked": int(sample.shape[0]),
                                            "decrypt_mismatches": int((dec_cpu != dec_gpu).sum())}
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)

    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
