"""One CMUX of the N = 16384 kernel against the exact integer products of its own digits
(library built with -DW16_DEBUG_KERNELS)."""
import sys, os, ctypes, numpy as np, dataclasses
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import concrete_amd.parameters as P
from concrete_amd import client as CL
from concrete_amd.backend import Context, DeviceServerKey, CAMD_PBS_WAVE
from oracle import cpu_oracle as O
n = 8
IT = int(os.environ.get("IT", 7))
p = dataclasses.replace(P.PARAM_MESSAGE_4_CARRY_4, lwe_dimension=n)
N, M = 16384, 8192
rng = CL.Rng(0xBEEF)
small = CL.gen_binary_key(rng, 64, n)
glwe = CL.gen_binary_key(rng, 64, N)
bsk = CL.gen_bootstrap_key(rng, p, 64, small, glwe, threads=8)
ksk = CL.gen_keyswitch_key(rng, p, 64, glwe, small)
ctx = Context(0)
dk = DeviceServerKey.from_standard_keys(ctx, p, bsk, ksk, variant=CAMD_PBS_WAVE)
o = O.Oracle(p, 64)
dk.upload_lut(o.generate_accumulator(lambda x: x % 16))
lib = ctypes.CDLL(os.environ["CONCRETE_AMD_LIB"])
assert lib.camd_debug_w16_dump(ctypes.c_int(10 + IT), None) == 0
COL = int(os.environ.get('COL', 0))
assert lib.camd_debug_w16_dump(ctypes.c_int(100 + COL), None) == 0
rs = np.random.RandomState(4)
cts = rs.randint(0, 2**63, size=(1, n + 1), dtype=np.uint64) * np.uint64(2)
dk.pbs(cts)
def dump(which, size):
    out = np.empty(size, dtype=np.complex128)
    assert lib.camd_debug_w16_dump(ctypes.c_int(which), ctypes.c_void_p(out.ctypes.data)) == 0
    return out
digits, back = dump(0, 4 * M).reshape(2, 2, M), dump(2, M)      # [row][level-1][point]
def negacyclic_exact(d, k):
    """d: int64 digits (N), k: uint64 key polynomial (N) -> their negacyclic product mod 2^64 (uint64)"""
    fd = np.fft.rfft(d.astype(np.float64), 2 * N)
    acc = np.zeros(N, dtype=np.uint64)
    for m in range(4):
        limb = ((k >> np.uint64(16 * m)) & np.uint64(0xFFFF)).astype(np.float64)
        full = np.fft.irfft(fd * np.fft.rfft(limb, 2 * N), 2 * N)
        c = np.rint(full[:N] - full[N:]).astype(np.int64).view(np.uint64)
        acc += c << np.uint64(16 * m)
    return acc
t = np.zeros(N, dtype=np.uint64)
for r in range(2):
    for l in range(2):
        d = np.concatenate([digits[r, l].real, digits[r, l].imag])
        assert np.all(d == np.rint(d)) and np.abs(d).max() <= 16384, np.abs(d).max()
        t += negacyclic_exact(d.astype(np.int64), bsk[IT, l, r, COL])
        print("row %d level %d: digit rms %.1f max %d" % (r, l + 1, d.std(), np.abs(d).max()))
gpu = np.concatenate([back.real, back.imag])
print("gpu increments: rms %.3e turns" % gpu.std())
frac = gpu - np.floor(gpu)
want = t.astype(np.float64) / 2.0**64
e = frac - want
e -= np.rint(e)
print("error of the increment: rms %.3e max %.3e turns (2^-32 = %.3e)" % (e.std(), np.abs(e).max(), 2.0**-32))
