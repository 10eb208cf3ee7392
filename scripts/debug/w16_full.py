"""Full-size 4_4: throughput kernel against the register-blocked kernel on the same inputs."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import concrete_amd.parameters as P
from concrete_amd import client as CL
from concrete_amd.backend import Context, DeviceServerKey, CAMD_PBS_WAVE, CAMD_PBS_BLOCK
from oracle import cpu_oracle as O
p = P.PARAM_MESSAGE_4_CARRY_4
n_over = int(os.environ.get("N_LWE", p.lwe_dimension))
import dataclasses
p = dataclasses.replace(p, lwe_dimension=n_over)
rng = CL.Rng(0xBEEF)
small = CL.gen_binary_key(rng, 64, p.lwe_dimension)
glwe = CL.gen_binary_key(rng, 64, p.glwe_dimension * p.polynomial_size)
bsk = CL.gen_bootstrap_key(rng, p, 64, small, glwe, threads=16)
ksk = CL.gen_keyswitch_key(rng, p, 64, glwe, small)
ctx = Context(0)
dk = DeviceServerKey.from_standard_keys(ctx, p, bsk, ksk, variant=CAMD_PBS_WAVE)
ref = DeviceServerKey.from_standard_keys(ctx, p, bsk, ksk, variant=CAMD_PBS_BLOCK)
o = O.Oracle(p, 64)
lut = o.generate_accumulator(lambda x: x % 16)
dk.upload_lut(lut); ref.upload_lut(lut)
delta = O.shortint_delta(p)
count = int(os.environ.get("COUNT", 512))
msgs = np.arange(count) % 256
cts = CL.lwe_encrypt(CL.Rng(77), 64, glwe, [int(m) * delta for m in msgs], p.lwe_modular_std_dev)
got, want = dk.ks_pbs(cts), ref.ks_pbs(cts)
def perr(out):
    ph = CL.lwe_phase(64, glwe, out).astype(np.uint64)
    near = ((ph + np.uint64(delta // 2)) // np.uint64(delta)) * np.uint64(delta)
    return (ph - near).view(np.int64).astype(np.float64) / 2.0**64
eg, ew = perr(got), perr(want)
print("n =", p.lwe_dimension, "variance wave %.3e block %.3e" % (eg.var(), ew.var()))
print("wave |err| quantiles", np.quantile(np.abs(eg), [0.5, 0.9, 0.99, 1.0]))
print("block |err| quantiles", np.quantile(np.abs(ew), [0.5, 0.9, 0.99, 1.0]))
d = (got - want).view(np.int64).astype(np.float64) / 2.0**64
print("word difference wave-block: std %.3e max %.3e; body std %.3e" % (d[:, :-1].std(), np.abs(d).max(), d[:, -1].std()))
print("per-ciphertext std of word difference (first 8):", d.std(axis=1)[:8])
