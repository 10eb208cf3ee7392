"""Throughput probe per parameter family: python scripts/family_bench.py <param name> <count> [steps] [variant]
Random keys (timing is data-independent); shortint order KS->PBS for u64 sets, PBS->KS for boolean."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import concrete_amd.parameters as P
from concrete_amd.backend import Context, DeviceServerKey
name = sys.argv[1]
count = int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
variant = int(sys.argv[4]) if len(sys.argv) > 4 else 0
p = getattr(P, name)
boolean = isinstance(p, P.BooleanParameters)
bits = 32 if boolean else 64
dt_np = np.uint32 if boolean else np.uint64
n, k, N, L = p.lwe_dimension, p.glwe_dimension, p.polynomial_size, p.pbs_level
rs = np.random.RandomState(0)
def rnd(shape):
    return rs.randint(0, 2**31, size=shape, dtype=np.int64).astype(dt_np) * dt_np(2)
bsk = rnd((n, L, k + 1, k + 1, N))
ksk = rnd((k * N, p.ks_level, n + 1))
ctx = Context(0)
dk = DeviceServerKey.from_standard_keys(ctx, p, bsk, ksk, variant=variant, torus_bits=bits)
lut = np.zeros((k + 1) * N, dtype=dt_np); lut[k * N:] = rnd(N)
dk.upload_lut(lut)
tdt = torch.int32 if boolean else torch.int64
width = (n + 1) if boolean else (k * N + 1)
cts = torch.randint(-2**31, 2**31 - 1, (count, width), dtype=torch.int64, device='cuda').to(tdt)
if not boolean:
    cts = cts * 4294967297
out = torch.empty_like(cts)
dk.reserve(count)
s = torch.cuda.current_stream().cuda_stream
run = (lambda: dk.pbs_ks_device(cts, None, out, count, s)) if boolean else (lambda: dk.ks_pbs_device(cts, None, out, count, s))
run(); torch.cuda.synchronize()
dk.timing_enable(True)
t = time.time()
for _ in range(steps):
    run()
torch.cuda.synchronize()
dt = (time.time() - t) / steps
tm = dk.timing_read()
M = N // 2
fft = 5.0 * M * np.log2(M)
flop = n * ((k + 1) * L * fft + (k + 1) * fft + (k + 1) ** 2 * L * M * 8 + ((k + 1) * L + (k + 1)) * M * 6)
print(f"{name} variant {dk.variant} count {count}: {dt*1e3:.2f} ms/step -> {count/dt:.0f} PBS/s; kernel pbs {tm['pbs_ms']:.2f} ms ks {tm['ks_ms']:.2f} ms; "
      f"fp64 fraction {count / (tm['pbs_ms']*1e-3) * flop / 78.6e12:.3f}", flush=True)
