import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import concrete_amd.parameters as P
from concrete_amd.backend import Context, DeviceServerKey, CAMD_PBS_GENERIC
from oracle import cpu_oracle as O
from conftest import get_keys
ctx = Context(0)
p = P.PARAM_MESSAGE_2_CARRY_2
n = 2
ks = get_keys(p, 64, n_override=n, noise=False)
dk = DeviceServerKey.from_standard_keys(ctx, ks.device_params(), ks.bsk, ks.ksk, variant=int(sys.argv[1]))
lut = ks.o.generate_accumulator(lambda x: x % 4)
dk.upload_lut(lut)
a0, a1, b = 0x1234567890ABCDEF & ~1, 0x0FEDCBA987654320, 0x7777777777777770
cts = np.array([[0, a1, b], [a0, 0, b], [a0, a1, b], [a0, a0, b], [1 << 53, 1 << 53, 0]], dtype=np.uint64)
got = dk.pbs(cts)
want = np.stack([ks.o.bootstrap(ks.fbsk, lut, c) for c in cts])
d = np.ascontiguousarray(got - want).view(np.int64).astype(np.float64) / 2.0**64
print(np.abs(d).max(axis=1))
