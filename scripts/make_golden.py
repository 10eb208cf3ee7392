"""Generates the fixtures under tests/golden/ from the CPU oracle (the reference holds none).

Run from the repository root:  python scripts/make_golden.py
"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import concrete_amd.parameters as P
from oracle import cpu_oracle as O
G = os.path.join(ROOT, "tests", "golden")
os.makedirs(G, exist_ok=True)

p = P.PARAM_MESSAGE_2_CARRY_2
o = O.Oracle(p, 64)
fs = {"id": lambda x: x, "mod4": lambda x: x % 4, "div4": lambda x: x // 4, "double": lambda x: (2 * x) % 16,
      "mul_lsb": lambda x: ((x // 4) * (x % 4)) % 4}
acc = {}
for name, f in fs.items():
    body = o.generate_accumulator(f)[p.glwe_dimension * p.polynomial_size:]
    acc[name] = {"first4": [int(v) for v in body[:4]], "last4": [int(v) for v in body[-4:]],
                 "xor": int(np.bitwise_xor.reduce(body)), "sum": int(body.sum(dtype=np.uint64))}
json.dump(acc, open(os.path.join(G, "accumulators_2_2.json"), "w"), indent=1)

rs = np.random.RandomState(1234)
cases = []
for bits, log_n in ((64, 10), (64, 14), (64, 8), (32, 9), (32, 10)):
    for x in [0, 1, (1 << bits) - 1, 1 << (bits - 1), (1 << (bits - log_n - 2)), (1 << (bits - log_n - 2)) - 1] + \
             [int(v) & ((1 << bits) - 1) for v in rs.randint(0, 2 ** 62, 8) * 4 + 3]:
        cases.append({"x": str(x), "log2_n": log_n, "bits": bits, "out": O.modswitch(x, log_n, bits)})
json.dump({"cases": cases}, open(os.path.join(G, "modswitch.json"), "w"), indent=0)

cases = []
for base_log, level, bits in [(23, 1, 64), (15, 2, 64), (11, 3, 64), (4, 3, 64), (3, 6, 64), (8, 2, 32), (2, 5, 32)]:
    mask = (1 << bits) - 1
    for x in [0, 1, mask, 1 << (bits - 1), (1 << (bits - 1)) - 1] + [int(v) & mask for v in rs.randint(0, 2 ** 62, 6) * 4 + 1]:
        cases.append({"x": str(x), "base_log": base_log, "level": level, "bits": bits,
                      "digits": O.decompose(x, base_log, level, bits)})
json.dump({"cases": cases}, open(os.path.join(G, "decomposer.json"), "w"), indent=0)

q = type("Q", (), dict(lwe_dimension=4, glwe_dimension=1, polynomial_size=256, pbs_base_log=15, pbs_level=2,
                       ks_base_log=4, ks_level=3, message_modulus=2, carry_modulus=2,
                       lwe_modular_std_dev=0.0, glwe_modular_std_dev=0.0))()
oq = O.Oracle(q, 64)
rng = O.Rng(42)
small = oq.gen_binary_key(rng, 4)
glwe = oq.gen_binary_key(rng, 256)
bsk = oq.gen_bsk(rng, small, glwe)
ksk = oq.gen_ksk(rng, glwe, small)
lut = oq.generate_accumulator(lambda x: (x + 1) % 4)
delta = O.shortint_delta(q)
ct_in = np.stack([oq.lwe_encrypt(rng, glwe, m * delta, 0.0) for m in range(4)])
fb = oq.bsk_to_fourier(bsk)
ct_ks = np.stack([oq.keyswitch(ksk, c) for c in ct_in])
ct_out = oq.ks_pbs_batch(ksk, fb, lut, None, ct_in, threads=1)
for i in range(4):
    assert O.shortint_decode(q, oq.lwe_phase(glwe, ct_out[i])) == (i + 1) % 4
np.savez_compressed(os.path.join(G, "tiny_transcript.npz"), bsk=bsk, ksk=ksk, lut=lut, ct_in=ct_in, ct_ks=ct_ks,
                    ct_out=ct_out, small_key=small, glwe_key=glwe)
print("golden fixtures written to", G)
