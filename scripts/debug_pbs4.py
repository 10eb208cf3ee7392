import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import concrete_amd.parameters as P
from concrete_amd.backend import Context, DeviceServerKey, CAMD_PBS_GENERIC
from oracle import cpu_oracle as O
from conftest import get_keys
ctx = Context(0)
p = P.PARAM_MESSAGE_2_CARRY_2
n = 8
ks = get_keys(p, 64, n_override=n, noise=False)
dk = DeviceServerKey.from_standard_keys(ctx, ks.device_params(), ks.bsk, ks.ksk, variant=int(sys.argv[1]))
lut = ks.o.generate_accumulator(lambda x: x)
dk.upload_lut(lut)
delta = O.shortint_delta(p)
cts = np.stack([ks.o.lwe_encrypt(ks.rng, ks.small, m * delta, 0.0) for m in range(16)])
got = dk.pbs(cts)
want = np.stack([ks.o.bootstrap(ks.fbsk, lut, c) for c in cts])
for i in range(16):
    pg = ks.o.lwe_phase(ks.glwe, got[i]); pw = ks.o.lwe_phase(ks.glwe, want[i])
    print(i, 'gpu', O.shortint_decode(p, pg), hex(pg), 'oracle', O.shortint_decode(p, pw), hex(pw))
