"""Checks the Fourier key the N = 16384 wave kernel builds on the GPU against numpy's FFT."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dataclasses
import concrete_amd.parameters as P
from concrete_amd.backend import Context, DeviceServerKey, CAMD_PBS_WAVE
p = dataclasses.replace(P.PARAM_MESSAGE_4_CARRY_4, lwe_dimension=1)
N, M = 16384, 8192
rs = np.random.RandomState(1)
bsk = rs.randint(0, 2**63, size=(1, 2, 2, 2, N), dtype=np.uint64) * np.uint64(2) + rs.randint(0, 2, size=(1, 2, 2, 2, N)).astype(np.uint64)
ksk = rs.randint(0, 2**63, size=(N, p.ks_level, 2), dtype=np.uint64)
dk = DeviceServerKey.from_standard_keys(Context(0), p, bsk, ksk, variant=CAMD_PBS_WAVE)
blob = dk.export_blob_host()
f = np.frombuffer(blob[:8 * M * 16].tobytes(), dtype=np.complex128).reshape(8, M)
worst = 0
for poly in range(8):
    c = bsk.reshape(8, N)[poly].view(np.int64).astype(np.float64) * 2.0**-64
    z = (c[:M] + 1j * c[M:]) * np.exp(1j * np.pi * np.arange(M) / N)
    X = np.fft.fft(z) / M
    wave, e, g, lane = np.meshgrid(np.arange(8), np.arange(2), np.arange(8), np.arange(64), indexing="ij")
    k = (2 * wave + e) + 16 * (64 * g + 8 * (lane & 7) + (lane >> 3))
    want = X[k].reshape(-1)
    err = np.abs(f[poly] - want).max()
    worst = max(worst, err)
    if poly == 0:
        print("first values got ", f[0][:3], "\n         want", want[:3], " |X| typical", np.abs(X).mean())
print("largest deviation of the device spectrum:", worst)
