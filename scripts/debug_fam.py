import torch; torch.cuda.init()
import sys, os, dataclasses, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import concrete_amd.parameters as P
from concrete_amd.backend import Context, DeviceServerKey, CAMD_PBS_GENERIC
from oracle import cpu_oracle as O
from conftest import get_keys
ctx = Context(0)
cases = [("4096 L1", P.PARAM_MESSAGE_3_CARRY_3), ("4096 L2", P.PARAM_MESSAGE_2_CARRY_4),
         ("2048 L2", dataclasses.replace(P.PARAM_MESSAGE_2_CARRY_3, pbs_base_log=15, pbs_level=2)),
         ("1024 L2 k2", dataclasses.replace(P.PARAM_MESSAGE_2_CARRY_2, pbs_base_log=10, pbs_level=2))]
for name, p in cases:
    ks = get_keys(p, 64, n_override=3)
    o = ks.o
    dk = DeviceServerKey.from_standard_keys(ctx, ks.device_params(), ks.bsk, ks.ksk, variant=CAMD_PBS_GENERIC)
    lut = o.generate_accumulator(lambda x: x % p.message_modulus)
    dk.upload_lut(lut)
    delta = O.shortint_delta(p)
    cts = np.stack([o.lwe_encrypt(ks.rng, ks.small, m * delta, 0.0) for m in range(3)])
    for nn in (1, 2, 3):
        c2 = cts.copy(); c2[:, nn:ks.n] = 0
        got = dk.pbs(c2); want = np.stack([o.bootstrap(ks.fbsk, lut, c) for c in c2])
        print(name, 'active iterations', nn, 'equal', np.array_equal(got, want), 'mismatching words', int((got != want).sum()))
name, p = cases[0]
ks = get_keys(p, 64, n_override=3); o = ks.o
dk = DeviceServerKey.from_standard_keys(ctx, ks.device_params(), ks.bsk, ks.ksk, variant=CAMD_PBS_GENERIC)
lut = o.generate_accumulator(lambda x: x % p.message_modulus); dk.upload_lut(lut)
cts = np.stack([o.lwe_encrypt(ks.rng, ks.small, m * O.shortint_delta(p), 0.0) for m in range(1)]); cts[:, 1:ks.n] = 0
got = dk.pbs(cts); want = np.stack([o.bootstrap(ks.fbsk, lut, c) for c in cts])
bad = np.nonzero(got[0] != want[0])[0]
print('bad idx', bad[:40], '...', bad[-10:], 'count', len(bad))
d = (got[0][bad] - want[0][bad]).view(np.int64)
print('diffs', d[:20])
import torch
blob = torch.empty(dk.blob_size(), dtype=torch.uint8, device='cuda'); dk.export_blob(blob); torch.cuda.synchronize()
polys = 3 * 4
f = blob[:polys * 2048 * 16].cpu().numpy().view(np.float64).reshape(polys, 2048, 2)
of = ks.fbsk.reshape(polys, 2, 2048)
print('fourier key max diff re', np.abs(f[:, :, 0] - of[:, 0, :]).max(), 'im', np.abs(f[:, :, 1] - of[:, 1, :]).max(), 'nonzero diffs', int((f[:, :, 0] != of[:, 0, :]).sum()))
