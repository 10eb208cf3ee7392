"""Keyswitch-only timing over batch sizes, both kernels (CAMD_KS_MFMA_MIN is read per call)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import concrete_amd.parameters as P
from concrete_amd.backend import Context, DeviceServerKey
pname = sys.argv[1] if len(sys.argv) > 1 else "PARAM_MESSAGE_2_CARRY_2"
bits = 32 if pname in ("DEFAULT_PARAMETERS", "TFHE_LIB_PARAMETERS") else 64
p = getattr(P, pname)
n, k, N = p.lwe_dimension, p.glwe_dimension, p.polynomial_size
dt = np.uint64 if bits == 64 else np.uint32
rs = np.random.RandomState(0)
rnd = lambda shape: rs.randint(0, 2**31, size=shape, dtype=np.int64).astype(dt) * dt(2)
ctx = Context(0)
dk = DeviceServerKey.from_standard_keys(ctx, p, rnd((2, p.pbs_level, k + 1, k + 1, N)) if False else rnd((n, p.pbs_level, k + 1, k + 1, N)),
                                        rnd((k * N, p.ks_level, n + 1)), variant=1, torus_bits=bits)
tdt = torch.int64 if bits == 64 else torch.int32
s = torch.cuda.current_stream().cuda_stream
for count in [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "1,16,64,128,256,1024,4096,16384".split(","))]:
    cts = torch.randint(-2**31, 2**31 - 1, (count, k * N + 1), dtype=torch.int64, device="cuda").to(tdt)
    out = torch.empty((count, n + 1), dtype=tdt, device="cuda")
    row = []
    for mode in ("0", "1"):
        os.environ["CAMD_KS_MFMA_MIN"] = mode
        dk.reserve(count)
        for _ in range(2):
            dk.ks_device(cts, out, count, s)
        torch.cuda.synchronize()
        reps = 10
        t = time.perf_counter()
        for _ in range(reps):
            dk.ks_device(cts, out, count, s)
        torch.cuda.synchronize()
        row.append((time.perf_counter() - t) / reps * 1e3)
    print(f"{pname} count {count}: f64-fma {row[0]:.3f} ms, mfma {row[1]:.3f} ms", flush=True)
