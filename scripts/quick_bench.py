"""Throughput probe: KS+PBS batch on device-resident random inputs (timing is data-independent).
QB_PARAMS names the parameter set (default PARAM_MESSAGE_2_CARRY_2)."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import concrete_amd.parameters as P
from concrete_amd.backend import Context, DeviceServerKey
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
p = getattr(P, os.environ.get('QB_PARAMS', 'PARAM_MESSAGE_2_CARRY_2'))
n, k, N = p.lwe_dimension, p.glwe_dimension, p.polynomial_size
rs = np.random.RandomState(0)
bsk = rs.randint(0, 2**63, size=(n, p.pbs_level, k + 1, k + 1, N), dtype=np.uint64) * np.uint64(2)
ksk = rs.randint(0, 2**63, size=(k * N, p.ks_level, n + 1), dtype=np.uint64) * np.uint64(2)
ctx = Context(0)
dk = DeviceServerKey.from_standard_keys(ctx, p, bsk, ksk, variant=variant)
lut = np.zeros((k + 1) * N, dtype=np.uint64); lut[k * N:] = rs.randint(0, 2**63, size=N, dtype=np.uint64)
dk.upload_lut(lut)
cts = torch.randint(-2**63, 2**63 - 1, (count, k * N + 1), dtype=torch.int64, device='cuda')
out = torch.empty_like(cts)
dk.reserve(count)
s = torch.cuda.current_stream().cuda_stream
dk.ks_pbs_device(cts, None, out, count, s); torch.cuda.synchronize()
dk.timing_enable(True)
t = time.time()
for _ in range(steps):
    dk.ks_pbs_device(cts, None, out, count, s)
torch.cuda.synchronize()
dt = (time.time() - t) / steps
tm = dk.timing_read()
print(f"variant {dk.variant} count {count}: {dt*1e3:.2f} ms/step -> {count/dt:.0f} PBS/s; kernel pbs {tm['pbs_ms']:.2f} ms ks {tm['ks_ms']:.2f} ms")
flop = 148829184
print(f"fp64 fraction (78.6 TF): {count/ (tm['pbs_ms']*1e-3) * flop / 78.6e12:.3f}")
