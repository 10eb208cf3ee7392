"""Digest of the KS+PBS outputs of a seeded random batch under seeded random keys:
python scripts/output_hash.py <param name> <count>. Two library builds (CONCRETE_AMD_LIB) whose kernels
compute the same words print the same digest."""
import hashlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import concrete_amd.parameters as P
from concrete_amd.backend import Context, DeviceServerKey
p = getattr(P, sys.argv[1])
count = int(sys.argv[2])
n, k, N = p.lwe_dimension, p.glwe_dimension, p.polynomial_size
rs = np.random.RandomState(3)
bsk = rs.randint(0, 2**63, size=(n, p.pbs_level, k + 1, k + 1, N), dtype=np.uint64) * np.uint64(2)
ksk = rs.randint(0, 2**63, size=(k * N, p.ks_level, n + 1), dtype=np.uint64) * np.uint64(2)
ctx = Context(0)
dk = DeviceServerKey.from_standard_keys(ctx, p, bsk, ksk)
lut = np.zeros((k + 1) * N, dtype=np.uint64); lut[k * N:] = rs.randint(0, 2**63, size=N, dtype=np.uint64)
dk.upload_lut(lut)
g = torch.Generator(device="cuda"); g.manual_seed(5)
cts = torch.randint(-2**63, 2**63 - 1, (count, k * N + 1), dtype=torch.int64, device="cuda", generator=g)
out = torch.empty_like(cts)
dk.ks_pbs_device(cts, None, out, count, torch.cuda.current_stream().cuda_stream)
dk.check(torch.cuda.current_stream().cuda_stream)
print(sys.argv[1], count, hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16], flush=True)
