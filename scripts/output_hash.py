"""Digest of the KS+PBS (shortint sets) or PBS+KS (boolean sets) outputs of a seeded random batch under
seeded random keys: python scripts/output_hash.py <param name> <count>. Two library builds
(CONCRETE_AMD_LIB) whose kernels compute the same words print the same digest."""
import hashlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import concrete_amd.parameters as P
from concrete_amd.backend import Context, DeviceServerKey
p = getattr(P, sys.argv[1])
count = int(sys.argv[2])
n, k, N = p.lwe_dimension, p.glwe_dimension, p.polynomial_size
boolean = isinstance(p, P.BooleanParameters)
bits = 32 if boolean else 64
dt = np.uint32 if boolean else np.uint64
rs = np.random.RandomState(3)
def rnd(shape):
    return rs.randint(0, 2**31, size=shape, dtype=np.int64).astype(dt) * dt(2) + (rs.randint(0, 2**31, size=shape, dtype=np.int64).astype(dt) << dt(bits - 31))
bsk = rnd((n, p.pbs_level, k + 1, k + 1, N))
ksk = rnd((k * N, p.ks_level, n + 1))
ctx = Context(0)
dk = DeviceServerKey.from_standard_keys(ctx, p, bsk, ksk, torus_bits=bits)
lut = np.zeros((k + 1) * N, dtype=dt); lut[k * N:] = rnd(N)
dk.upload_lut(lut)
g = torch.Generator(device="cuda"); g.manual_seed(5)
width = (n + 1) if boolean else (k * N + 1)
if boolean:
    cts = torch.randint(-2**31, 2**31 - 1, (count, width), dtype=torch.int32, device="cuda", generator=g)
else:
    cts = torch.randint(-2**63, 2**63 - 1, (count, width), dtype=torch.int64, device="cuda", generator=g)
out = torch.empty_like(cts)
stream = torch.cuda.current_stream().cuda_stream
if boolean:
    dk.pbs_ks_device(cts, None, out, count, stream)
else:
    dk.ks_pbs_device(cts, None, out, count, stream)
dk.check(stream)
print(sys.argv[1], count, hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16], flush=True)
