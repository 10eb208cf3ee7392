import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import concrete_amd.parameters as P
from concrete_amd.backend import Context, DeviceServerKey, CAMD_PBS_GENERIC, CAMD_PBS_AUTO
from oracle import cpu_oracle as O
from conftest import get_keys
variant = int(sys.argv[1]) if len(sys.argv) > 1 else CAMD_PBS_GENERIC
ctx = Context(0)
p = P.PARAM_MESSAGE_2_CARRY_2
for n in (1, 2, 8):
    ks = get_keys(p, 64, n_override=n, noise=False)
    dk = DeviceServerKey.from_standard_keys(ctx, ks.device_params(), ks.bsk, ks.ksk, variant=variant)
    lut = ks.o.generate_accumulator(lambda x: x % 4)
    dk.upload_lut(lut)
    rs = np.random.RandomState(5)
    cts = rs.randint(0, 2**63, size=(4, n + 1), dtype=np.uint64) * np.uint64(2)
    got = dk.pbs(cts)
    want = np.stack([ks.o.bootstrap(ks.fbsk, lut, c) for c in cts])
    d = np.ascontiguousarray(got - want).view(np.int64).astype(np.float64) / 2.0**64
    print('n', n, 'variant', dk.variant, 'max dist', np.abs(d).max(), 'first words', got[0, :3], want[0, :3])
