"""Intermediate values of the first CMUX of the N = 16384 kernel (library built with -DW16_DEBUG_KERNELS)."""
import sys, os, ctypes, numpy as np, dataclasses
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import concrete_amd.parameters as P
from concrete_amd.backend import Context, DeviceServerKey, CAMD_PBS_WAVE
p = dataclasses.replace(P.PARAM_MESSAGE_4_CARRY_4, lwe_dimension=1)
N, M = 16384, 8192
ctx = Context(0)
rs = np.random.RandomState(3)
ksk = np.zeros((N, p.ks_level, 2), dtype=np.uint64)
lut = rs.randint(0, 2**63, size=2 * N, dtype=np.uint64) * np.uint64(2)
if os.environ.get('LUT_QUARTER'):
    lut = (lut >> np.uint64(62)) << np.uint64(62)
cts = np.zeros((1, 2), dtype=np.uint64)
A_HAT = 17000
cts[0, 0] = A_HAT << 49
bsk = np.zeros((1, 2, 2, 2, N), dtype=np.uint64)
bsk[0, 0, 0, 0, 0] = 1 << 62
dk = DeviceServerKey.from_standard_keys(ctx, p, bsk, ksk, variant=CAMD_PBS_WAVE)
dk.upload_lut(lut)
got = dk.pbs(cts)[0]
lib = ctypes.CDLL(os.environ["CONCRETE_AMD_LIB"])
def dump(which):
    out = np.empty(M, dtype=np.complex128)
    assert lib.camd_debug_w16_dump(ctypes.c_int(which), ctypes.c_void_p(out.ctypes.data)) == 0
    return out
digits, sums, back = dump(0), dump(1), dump(2)
# expected digits of row 0, level 1 (most significant)
acc = (lut[:N] >> np.uint64(32)).astype(np.int64)
j = np.arange(N)
s = (j - A_HAT) % (2 * N)
rot = np.where(s >= N, -acc[s % N], acc[s % N])
d = ((rot - acc + 2) % (1 << 32))
d = np.where(d >= (1 << 31), d - (1 << 32), d) >> 2          # 30-bit signed
lo = ((d + (1 << 14)) % (1 << 15)) - (1 << 14)
hi = (d - lo) >> 15
want_digits = hi[:M] + 1j * hi[M:]
print("digits: max |diff|", np.abs(digits - want_digits).max(), " max |digit|", np.abs(want_digits).max())
F = np.fft.fft(digits * np.exp(1j * np.pi * np.arange(M) / 16384.0))
lane = np.arange(64)
w = np.arange(8)[:, None, None, None]; ee = np.arange(2)[None, :, None, None]; g = np.arange(8)[None, None, :, None]; l = lane[None, None, None, :]
k = ((2 * w + ee) + 16 * (64 * g + 8 * (l & 7) + (l >> 3))).reshape(-1)
want_sums = F[k] * 0.25 / 8192
print("sums: max abs err %.3e of max %.3e" % (np.abs(sums - want_sums).max(), np.abs(want_sums).max()))
e = np.abs(sums - want_sums)
idx = np.argsort(-e)[:10]
print("  worst at", idx, e[idx])
want_back = 0.25 * digits
print("back: max abs err %.3e of max %.3e" % (np.abs(back - want_back).max(), np.abs(want_back).max()))
dd = digits - want_digits
bad = np.nonzero(np.abs(dd) > 0.5)[0]
print("digit positions off:", len(bad), bad[:32])
for b in bad[:12]:
    print("  j=%d got %r want %r  acc[j]=%#x acc[j+M]=%#x" % (b, digits[b], want_digits[b], acc[b], acc[b + M]))
