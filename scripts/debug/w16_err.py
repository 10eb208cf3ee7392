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
lut = rs.randint(0, 2**63, size=2 * N, dtype=np.uint64) * np.uint64(2)
lut >>= np.uint64(int(os.environ.get('LUT_SHIFT', '0')))
if os.environ.get('LUT_QUARTER'):
    lut = (lut >> np.uint64(62)) << np.uint64(62)
if os.environ.get('LUT_NOBODY'):
    lut[N:] = 0
if os.environ.get('LUT_SPARSE'):
    lut[8:N] = 0
if os.environ.get('LUT_SPIKE'):
    lut[:] = 0
    lut[int(os.environ['LUT_SPIKE'])] = np.uint64(1) << np.uint64(62)
cts = np.zeros((1, 2), dtype=np.uint64)
cts[0, 0] = 17000 << 49
bsk = np.zeros((1, 2, 2, 2, N), dtype=np.uint64)
bsk[0, 0, 0, 0, 0] = 1 << 62
dk = DeviceServerKey.from_standard_keys(ctx, p, bsk, ksk, variant=CAMD_PBS_WAVE)
dk.upload_lut(lut)
got = dk.pbs(cts)[0]
triv = dk.pbs(np.array([[0, 0]], dtype=np.uint64))[0]          # the table itself, sample-extracted
want = o.bootstrap(o.bsk_to_fourier(bsk), lut, cts[0])
e = (got - want).view(np.int64).astype(np.float64) / 2.0**64 / 0.25      # error in digit units
def dev(x):
    f = (x.view(np.int64).astype(np.float64) / 2.0**64) / 0.25 * float(os.environ.get('SCALE', '1'))
    return np.abs(f - np.round(f)).max()
print("deviation from multiples of 1/4 (in units of 1/4): gpu %.3e oracle %.3e" % (dev(got), dev(want)))
print("error in digit units: mean %.3e std %.3e max %.3e" % (e[:N].mean(), e[:N].std(), np.abs(e[:N]).max()))
print("first 16:", np.array2string(e[:16], precision=2))
idx = np.argsort(-np.abs(e[:N]))[:8]
print("largest at", idx, e[idx])
# by position class
for k in (512, 64, 8):
    print("mean |err| by j mod", k, ":", np.array2string(np.array([np.abs(e[:N][np.arange(N) % k == r]).mean() for r in range(min(k, 8))]), precision=2))
if os.environ.get('LUT_SPIKE'):
    ee = (got.view(np.int64).astype(np.float64) / 2.0**64 / 0.25)[:N]
    dev_ = ee - np.round(ee)
    nz = np.nonzero(np.abs(dev_) > 1e-9)[0]
    print("positions with a deviation:", len(nz), "first", nz[:40])
    print("values", np.array2string(dev_[nz[:12]], precision=3))
    big = np.nonzero(np.round(ee) != 0)[0]
    print("nonzero integer part at", big[:10], np.round(ee)[big[:10]])
