"""Transform pieces of the N = 16384 kernel against numpy (needs a library built with -DW16_DEBUG_KERNELS)."""
import ctypes, os, sys, numpy as np
lib = ctypes.CDLL(os.environ["CONCRETE_AMD_LIB"])
f = lib.camd_debug_w16
f.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
M = 8192
def run(mode, x):
    x = np.ascontiguousarray(x, dtype=np.complex128)
    out = np.empty(M, dtype=np.complex128)
    rc = f(mode, x.ctypes.data, out.ctypes.data)
    assert rc == 0, rc
    return out
lane = np.arange(64)
order512 = (64 * np.arange(8)[:, None] + 8 * (lane & 7)[None, :] + (lane >> 3)[None, :]).reshape(-1)   # [g][lane] -> frequency
def report(name, got, want):
    scale = np.abs(want).max()
    err = np.abs(got - want)
    print("%-28s max rel err %.3e  (at %d)  rms rel %.3e" % (name, err.max() / scale, int(err.argmax()), np.sqrt((err**2).mean()) / scale))
    return err / scale
rs = np.random.RandomState(1)
def inputs():
    yield "random", rs.standard_normal(M) + 1j * rs.standard_normal(M)
    for s in (0, 100, 616, 5000):
        x = np.zeros(M, dtype=np.complex128); x[s] = 1.0
        yield "spike %d" % s, x
for name, x in inputs():
    # mode 1: eight independent 512-point forward transforms
    X = np.fft.fft(x.reshape(16, 512), axis=1)
    want = X[:, order512].reshape(-1)
    e = report("fwd512 " + name, run(1, x), want)
    # mode 2: inverse
    spec = X[:, order512].reshape(-1)
    e = report("inv512 " + name, run(2, spec), 512 * x)
    if name.startswith("spike"):
        bad = np.nonzero(e > 1e-12)[0]
        print("   inv512 positions off:", len(bad), bad[:24])
    # mode 3: full forward
    j = np.arange(M)
    F = np.fft.fft(x * np.exp(1j * np.pi * j / 16384.0))
    w = np.arange(8)[:, None, None, None]; ee = np.arange(2)[None, :, None, None]; g = np.arange(8)[None, None, :, None]; l = lane[None, None, None, :]
    k = ((2 * w + ee) + 16 * (64 * g + 8 * (l & 7) + (l >> 3))).reshape(-1)
    e = report("fwd8192 " + name, run(3, x), F[k])
    # mode 4: full inverse
    e = report("inv8192 " + name, run(4, F[k]), 8192 * x)
    if name.startswith("spike"):
        bad = np.nonzero(e > 1e-12)[0]
        print("   inv8192 positions off:", len(bad), bad[:24])
