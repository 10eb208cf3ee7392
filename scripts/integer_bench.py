"""FheUint8 (4 blocks of PARAM_MESSAGE_2_CARRY_2) smart_add and smart_mul on device-resident batches:
python scripts/integer_bench.py [pairs]. Reports integer operations per second and bootstraps per op."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import concrete_amd.parameters as P
from concrete_amd import integer as INT

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
p = P.PARAM_MESSAGE_2_CARRY_2
cks = INT.RadixClientKey(p, 4, seed=11)
t0 = time.time()
sks = INT.ServerKey.new(cks)
print(f"keys: {time.time() - t0:.1f} s", flush=True)
rs = np.random.RandomState(1)
x = rs.randint(0, 256, size=pairs).astype(np.uint64)
y = rs.randint(0, 256, size=pairs).astype(np.uint64)
cx, cy = cks.encrypt_many(x, device="cuda"), cks.encrypt_many(y, device="cuda")
dk = sks.key.key


def timed(name, fn, want, reps=3):
    fn()  # warm-up (uploads the look-up tables)
    torch.cuda.synchronize()
    dk.timing_enable(True)
    t = time.time()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    dt = (time.time() - t) / reps
    tm = dk.timing_read()
    dk.timing_enable(False)
    got = cks.decrypt(r)
    bad = int(np.count_nonzero(got != want))
    launches = tm["pbs_launches"] / reps
    print(f"{name}: {dt * 1e3:.1f} ms per batch of {pairs} -> {pairs / dt:.0f} ops/s; {launches:.0f} bootstrap launches per op, "
          f"kernel time {tm['pbs_ms'] * launches + tm['ks_ms'] * launches:.1f} ms; wrong results {bad}/{pairs}", flush=True)


def add_chain():
    # carries must be propagated before the third addition: one smart_add with full propagation of both operands
    a = sks.unchecked_add(cx, cy)
    a = sks.unchecked_add(a, cy)
    return sks.smart_add(a, a.clone())


timed("smart_add (operands with full carries -> 2 propagations)", add_chain, (6 * ((x + 2 * y) % 256) // 3) % 256 * 0 + (2 * (x + 2 * y)) % 256)
timed("smart_mul (fresh operands)", lambda: sks.smart_mul(cx.clone(), cy.clone()), (x * y) % 256)


def mul_carried():
    # both operands hold carries in every block: the whole propagation chain runs first
    a = sks.unchecked_add(sks.unchecked_add(cx, cy), cy)
    b = sks.unchecked_add(sks.unchecked_add(cy, cx), cx)
    return sks.smart_mul(a, b)


timed("smart_mul (operands with full carries)", mul_carried, ((x + 2 * y) * (y + 2 * x)) % 256)
timed("gather(smart_mul, smart_add with carries, smart_bitxor)",
      lambda: sks.gather([lambda: sks.smart_mul(cx.clone(), cy.clone()), add_chain,
                          lambda: sks.smart_bitxor(cx.clone(), cy.clone())])[0], (x * y) % 256)
timed("smart_bitxor", lambda: sks.smart_bitxor(cx.clone(), cy.clone()), x ^ y)
