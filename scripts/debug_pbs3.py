import sys, os, numpy as np, torch
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
blob = torch.empty(dk.blob_size(), dtype=torch.uint8, device='cuda')
dk.export_blob(blob)
torch.cuda.synchronize()
polys = n * 9
f = blob[:polys * 512 * 16].cpu().numpy().view(np.float64).reshape(polys, 512, 2)
of = ks.fbsk.reshape(polys, 2, 512)
print('re diff', np.abs(f[:, :, 0] - of[:, 0, :]).max(), 'im diff', np.abs(f[:, :, 1] - of[:, 1, :]).max(), 'mag', np.abs(of).max())
for q in range(polys):
    print(q, np.abs(f[q, :, 0] - of[q, 0, :]).max())
