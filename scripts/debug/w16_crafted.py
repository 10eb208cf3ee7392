"""One CMUX of the N = 16384 wave kernel with crafted keys (one constant GGSW polynomial at a time)."""
import sys, os, numpy as np, dataclasses
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import concrete_amd.parameters as P
from concrete_amd.backend import Context, DeviceServerKey, CAMD_PBS_WAVE
from oracle import cpu_oracle as O
p = dataclasses.replace(P.PARAM_MESSAGE_4_CARRY_4, lwe_dimension=1)
N = 16384
o = O.Oracle(p, 64)
ctx = Context(0)
rs = np.random.RandomState(3)
ksk = np.zeros((N, p.ks_level, 2), dtype=np.uint64)
lut = np.zeros(2 * N, dtype=np.uint64)
lut[:] = rs.randint(0, 2**63, size=2 * N, dtype=np.uint64) * np.uint64(2)
cts = np.zeros((6, 2), dtype=np.uint64)
cts[:, 0] = np.array([3 << 48, 12345 << 45, 9000 << 49, 17000 << 49, 30000 << 49, (1 << 64) - 5], dtype=np.uint64)     # rotations by small and larger amounts
cts[:, 1] = rs.randint(0, 2**63, size=6, dtype=np.uint64) * np.uint64(2)
for (l, r, c, coef, val) in [(0, 0, 0, 0, 1 << 49), (1, 0, 0, 0, 1 << 34), (0, 1, 1, 0, 1 << 49), (0, 0, 1, 0, 1 << 49),
                             (0, 1, 0, 0, 1 << 49), (0, 0, 0, 5, 1 << 49), (0, 0, 0, 9000, 1 << 49), (0, 0, 0, 0, 1 << 62), (0, 0, 0, 5, 1 << 62), (0, 0, 0, 5, 1 << 56), (1, 0, 0, 5, 1 << 62)]:
    bsk = np.zeros((1, 2, 2, 2, N), dtype=np.uint64)
    bsk[0, l, r, c, coef] = val
    dk = DeviceServerKey.from_standard_keys(ctx, p, bsk, ksk, variant=CAMD_PBS_WAVE)
    dk.upload_lut(lut)
    got = dk.pbs(cts)
    fb = o.bsk_to_fourier(bsk)
    want = np.stack([o.bootstrap(fb, lut, x) for x in cts])
    d = np.abs((got - want).view(np.int64).astype(np.float64)) / 2.0**64
    print("   per ciphertext:", ["%.1e" % d[i].max() for i in range(len(cts))])
    print(f"level {l + 1} row {r} col {c} X^{coef}: largest difference {d.max():.3e}; differing words (> 2^-28): mask {int((d[:, :N] > 2.0**-28).sum())} of {2 * N}, body {int((d[:, N] > 2.0**-28).sum())}")
    dk.close()

# dense keys
for dense in (1, 2):
    bsk = np.zeros((1, 2, 2, 2, N), dtype=np.uint64)
    if dense == 1:
        bsk[0, 0, 0, 0, :] = rs.randint(0, 2**63, size=N, dtype=np.uint64) * np.uint64(2)
    else:
        bsk[:] = rs.randint(0, 2**63, size=bsk.shape, dtype=np.uint64) * np.uint64(2)
    dk = DeviceServerKey.from_standard_keys(ctx, p, bsk, ksk, variant=CAMD_PBS_WAVE)
    dk.upload_lut(lut)
    got = dk.pbs(cts)
    fb = o.bsk_to_fourier(bsk)
    want = np.stack([o.bootstrap(fb, lut, x) for x in cts])
    d = np.abs((got - want).view(np.int64).astype(np.float64)) / 2.0**64
    print("dense", dense, "per ciphertext:", ["%.1e" % d[i].max() for i in range(len(cts))], "median", np.median(d))
