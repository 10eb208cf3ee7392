import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import concrete_amd.parameters as P
from concrete_amd import client as CL
from concrete_amd.backend import Context, DeviceServerKey
from concrete_amd.shortint import ClientKey
import bench
from oracle import cpu_oracle as O
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
p = P.PARAM_MESSAGE_2_CARRY_2
ctx = Context(0)
cks = ClientKey(p, seed=bench.SEED)
bsk = CL.gen_bootstrap_key(cks._rng, p, 64, cks.lwe_secret_key_after_ks, cks.glwe_secret_key)
ksk = CL.gen_keyswitch_key(cks._rng, p, 64, cks.lwe_secret_key, cks.lwe_secret_key_after_ks)
dk = DeviceServerKey.from_standard_keys(ctx, p, bsk, ksk, variant=variant)
lut = CL.shortint_accumulator(p, lambda x: x % 4)
dk.upload_lut(lut)
count = 16384
cts, msgs = bench.make_inputs(cks, count, torch.device('cuda'), seed=1)
out = torch.empty_like(cts)
dk.ks_pbs_device(cts, None, out, count, 0); torch.cuda.synchronize()
got = bench.decode_outputs(cks, out)
bad = torch.nonzero(got != msgs % 4).flatten().cpu().numpy()
print('variant', dk.variant, 'wrong', len(bad), bad)
# input phase error of the wrong ones
hc = cts.cpu().numpy().view(np.uint64); ho = out.cpu().numpy().view(np.uint64)
delta = 1 << 59
o = O.Oracle(p, 64)
fb = o.bsk_to_fourier(bsk)
for i in bad[:6]:
    m = int(msgs[i]); ph_in = int(CL.lwe_phase(64, cks.lwe_secret_key, hc[i])[0])
    e_in = ((ph_in - m * delta + 2**63) % 2**64 - 2**63) / 2.0**64
    small = o.keyswitch(ksk, hc[i])
    ph_s = int(CL.lwe_phase(64, cks.lwe_secret_key_after_ks, small)[0])
    e_s = ((ph_s - m * delta + 2**63) % 2**64 - 2**63) / 2.0**64
    # modswitch phase
    ms = np.array([O.modswitch(int(x), 10) for x in small], dtype=np.int64)
    key = cks.lwe_secret_key_after_ks.astype(np.int64)
    ph_ms = (int(ms[-1]) - int((ms[:-1] * key).sum())) % 2048
    want = o.bootstrap(fb, lut, small)
    d_or = cks.decode(int(CL.lwe_phase(64, cks.lwe_secret_key, want)[0]))
    print(i, 'm', m, 'gpu', int(got[i]), 'oracle', d_or, 'in_err %.2e ks_err %.2e' % (e_in, e_s), 'modswitched phase', ph_ms, 'expected', m * 64, 'box +-32')
