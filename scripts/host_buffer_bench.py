"""PCIe-inclusive rate: the same KS+PBS batch as quick_bench.py, handed over and returned in host
buffers (camd_ks_pbs_batch) instead of device-resident tensors."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import concrete_amd.parameters as P
from concrete_amd.backend import Context, DeviceServerKey
count = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
p = P.PARAM_MESSAGE_2_CARRY_2
n, k, N = p.lwe_dimension, p.glwe_dimension, p.polynomial_size
rs = np.random.RandomState(0)
bsk = rs.randint(0, 2**63, size=(n, p.pbs_level, k + 1, k + 1, N), dtype=np.uint64) * np.uint64(2)
ksk = rs.randint(0, 2**63, size=(k * N, p.ks_level, n + 1), dtype=np.uint64) * np.uint64(2)
ctx = Context(0)
dk = DeviceServerKey.from_standard_keys(ctx, p, bsk, ksk, variant=0)
lut = np.zeros((k + 1) * N, dtype=np.uint64); lut[k * N:] = rs.randint(0, 2**63, size=N, dtype=np.uint64)
dk.upload_lut(lut)
cts = rs.randint(0, 2**63, size=(count, k * N + 1), dtype=np.uint64)
dk.reserve(count)
dk.ks_pbs(cts)
t = time.time()
for _ in range(steps):
    out = dk.ks_pbs(cts)
dt = (time.time() - t) / steps
print(f"host buffers, count {count}: {dt*1e3:.2f} ms/step -> {count/dt:.0f} PBS/s "
      f"({2 * cts.nbytes / 1e6:.0f} MB over PCIe per step)")
