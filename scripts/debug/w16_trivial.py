"""Trivial ciphertext (no CMUX) and one-CMUX checks of the N = 16384 wave kernel against the oracle."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import concrete_amd.parameters as P
from concrete_amd.backend import Context, DeviceServerKey, CAMD_PBS_WAVE
from oracle import cpu_oracle as O
from conftest import get_keys
p = P.PARAM_MESSAGE_4_CARRY_4
ks = get_keys(p, 64, n_override=1)
o = ks.o
dk = DeviceServerKey.from_standard_keys(Context(0), ks.device_params(), ks.bsk, ks.ksk, variant=CAMD_PBS_WAVE)
lut = o.generate_accumulator(lambda x: x % 16)
dk.upload_lut(lut)
rs = np.random.RandomState(3)
cts = np.zeros((4, 2), dtype=np.uint64)
cts[:, 1] = rs.randint(0, 2**63, size=4, dtype=np.uint64) * np.uint64(2)
got = dk.pbs(cts)
want = np.stack([o.bootstrap(ks.fbsk, lut, c) for c in cts])
print("trivial: equal", np.array_equal(got, want), "first words", got[0][:3], want[0][:3], "body", got[0][-1], want[0][-1])
# one CMUX, non-trivial mask
cts[:, 0] = rs.randint(0, 2**63, size=4, dtype=np.uint64) * np.uint64(2) + np.uint64(1)
got = dk.pbs(cts)
want = np.stack([o.bootstrap(ks.fbsk, lut, c) for c in cts])
d = np.abs((got - want).view(np.int64).astype(np.float64)) / 2.0**64
print("one CMUX: largest difference", d.max(), "median", np.median(d))
print(" got ", got[0][:4], "\n want", want[0][:4])
