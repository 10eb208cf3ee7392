"""Shared bodies of the operation-layer tests (shortint ops, radix integers).

The same checks run twice: on the CPU with the bootstraps evaluated by the oracle (a stand-in
for the device key, so the host logic — linear combinations, degrees, look-up tables, batching,
deferred launches — is covered without a GPU), and on the GPU through the C ABI. Expected values
are the reference's own test oracles: clear arithmetic modulo the message modulus
(concrete-shortint/src/server_key/tests.rs, concrete-integer/src/server_key/radix/tests.rs).
"""
import dataclasses
import itertools

import numpy as np

import concrete_amd.parameters as P
from concrete_amd import client as CL
from concrete_amd import integer as INT
from concrete_amd import shortint as SI


class OracleBackend:
    """CPU stand-in for ``DeviceServerKey``: same calls, bootstraps done by the CPU oracle."""

    def __init__(self, params, bsk, ksk):
        from oracle import cpu_oracle as O

        self.o = O.Oracle(params, 64)
        self.ksk = ksk
        self.fbsk = self.o.bsk_to_fourier(bsk)
        self.luts = []
        self.ctx = None
        self.launches = 0
        self.bootstraps = 0

    def upload_lut(self, glwe):
        self.luts.append(np.ascontiguousarray(glwe, dtype=np.uint64))
        return len(self.luts) - 1

    @property
    def lut_count(self):
        return len(self.luts)

    def ks_pbs(self, cts, ids=None):
        cts = np.atleast_2d(np.ascontiguousarray(cts, dtype=np.uint64))
        self.launches += 1
        self.bootstraps += cts.shape[0]
        ids = np.zeros(cts.shape[0], dtype=np.uint32) if ids is None else np.asarray(ids, dtype=np.uint32)
        return self.o.ks_pbs_batch(self.ksk, self.fbsk, np.concatenate(self.luts), ids, cts, threads=4)


class Tally:
    """Equality checks with a budget for the parameter set's own failure rate.

    PARAM_MESSAGE_2_CARRY_2 mis-decodes a bootstrap with probability about 2^-13.9
    (concrete/docs/getting_started/security_and_cryptography.md:7); the reference's CI simply
    retries failed tests (.config/nextest.toml:10-16). A GPU run of these tests performs some 10^4
    bootstraps, so a few wrong outputs are expected from ANY correct implementation. On the CPU
    stand-in (shortened LWE dimension, negligible noise) the budget is zero."""

    def __init__(self, per_check=0, total=0):
        self.per_check, self.total, self.bad = per_check, total, 0

    def eq(self, got, want, what=""):
        got, want = np.asarray(got), np.asarray(want)
        assert got.shape == want.shape, what
        bad = int(np.count_nonzero(got != want))
        assert bad <= self.per_check, f"{what}: {bad} of {got.size} differ"
        self.bad += bad
        assert self.bad <= self.total, f"{what}: {self.bad} wrong outputs in total"


class AuditedKey:
    """Proxy of a device server key that looks at every bootstrap of every launch: the input and
    the output are decrypted, and an output that is not the table's value at the input's message
    is recorded with its input ciphertext. ``attribute()`` then runs the CPU oracle on exactly
    those inputs: a failure the oracle shares is the parameter set's noise budget at work (the
    keyswitch and the modulus switch are integer steps, identical on both sides, and it is their
    noise that tips a ciphertext into the neighbouring box); one it does not share would be the
    GPU path's own. This is what lets the operation-layer tests tolerate the few wrong outputs a
    2^-13.9 failure rate produces over 10^4 bootstraps without tolerating a wrong kernel."""

    def __init__(self, dk, cks, params, bsk, ksk):
        self._dk, self._cks, self._p, self._bsk, self._ksk = dk, cks, params, bsk, ksk
        self._luts = []
        self.checked = 0
        self.failed = []     # (input ciphertext, table id, decoded input, decoded output)

    def __getattr__(self, name):
        return getattr(self._dk, name)

    def upload_lut(self, glwe):
        i = self._dk.upload_lut(glwe)
        while len(self._luts) <= i:
            self._luts.append(None)
        self._luts[i] = np.array(glwe, dtype=np.uint64, copy=True)
        return i

    def _decode(self, phases):
        p = self._p
        delta = np.uint64((1 << 63) // (p.message_modulus * p.carry_modulus))
        return ((phases + (delta >> np.uint64(1))) // delta) % np.uint64(2 * p.message_modulus * p.carry_modulus)

    def _audit(self, cts_in, ids, cts_out):
        p = self._p
        N, k = p.polynomial_size, p.glwe_dimension
        box = N // (p.message_modulus * p.carry_modulus)
        key = self._cks.lwe_secret_key
        m_in = self._decode(CL.lwe_phase(64, key, cts_in))
        m_out = self._decode(CL.lwe_phase(64, key, cts_out))
        ids = np.zeros(len(m_in), dtype=np.uint32) if ids is None else np.asarray(ids).astype(np.uint32)
        tables = np.stack([self._decode(l[k * N:][::box]) for l in self._luts])          # [lut][message]
        space = p.message_modulus * p.carry_modulus
        # a message with the padding bit set reads the table negated (negacyclic rotation)
        want = np.where(m_in < space, tables[ids, m_in % space], (np.uint64(2 * space) - tables[ids, m_in % space]) % np.uint64(2 * space))
        self.checked += len(m_in)
        for i in np.nonzero(want != m_out)[0]:
            self.failed.append((np.array(cts_in[i], copy=True), int(ids[i]), int(m_in[i]), int(m_out[i])))

    def ks_pbs(self, cts, ids=None):
        cts = np.atleast_2d(np.ascontiguousarray(cts, dtype=np.uint64))
        out = self._dk.ks_pbs(cts, ids)
        self._audit(cts, ids, np.atleast_2d(out))
        return out

    def ks_pbs_device(self, d_in, d_ids, d_out, count, stream):
        import torch

        host_in = d_in[:count].cpu().numpy().view(np.uint64).copy()     # the output may alias the input
        self._dk.ks_pbs_device(d_in, d_ids, d_out, count, stream)
        torch.cuda.synchronize()
        ids = None if d_ids is None else d_ids[:count].cpu().numpy()
        self._audit(host_in, ids, d_out[:count].cpu().numpy().view(np.uint64))

    def attribute(self):
        """(bootstraps checked, failures, failures the CPU oracle does not share)."""
        if not self.failed:
            return self.checked, 0, 0
        from oracle import cpu_oracle as O

        o = O.Oracle(self._p, 64)
        fbsk = o.bsk_to_fourier(self._bsk)
        cts = np.stack([f[0] for f in self.failed])
        ids = np.array([f[1] for f in self.failed], dtype=np.uint32)
        out = o.ks_pbs_batch(self._ksk, fbsk, np.concatenate(self._luts), ids, cts, threads=min(16, len(ids)))
        m_cpu = self._decode(CL.lwe_phase(64, self._cks.lwe_secret_key, out))
        own = sum(1 for f, m in zip(self.failed, m_cpu) if int(m) != f[3])
        return self.checked, len(self.failed), own


_CACHE = {}


def make_keys(pname="PARAM_MESSAGE_2_CARRY_2", n_override=8, backend="oracle", ctx=None, seed=7):
    """(shortint ClientKey, shortint ServerKey) on the oracle stand-in or on the GPU."""
    key = (pname, n_override, backend, seed)
    if key in _CACHE:
        return _CACHE[key]
    p = P.SHORTINT_BY_NAME[pname]
    if n_override:
        p = dataclasses.replace(p, lwe_dimension=n_override)
    cks = SI.ClientKey(p, seed=seed)
    bsk = CL.gen_bootstrap_key(cks._rng, p, 64, cks.lwe_secret_key_after_ks, cks.glwe_secret_key)
    ksk = CL.gen_keyswitch_key(cks._rng, p, 64, cks.lwe_secret_key, cks.lwe_secret_key_after_ks)
    if backend == "oracle":
        sks = SI.ServerKey(OracleBackend(p, bsk, ksk), p)
    else:
        from concrete_amd.backend import Context, DeviceServerKey

        dk = DeviceServerKey.from_standard_keys(ctx or Context(0), p, bsk, ksk, torus_bits=64)
        sks = SI.ServerKey(AuditedKey(dk, cks, p, bsk, ksk), p)
    _CACHE[key] = (cks, sks)
    return cks, sks


def check_shortint_ops(cks, sks, device=False, tally=None):
    """Every operation family on a batch holding all 16 message pairs of 2_2 (or a sample)."""
    t = tally or Tally()
    msg = sks.message_modulus
    pairs = list(itertools.product(range(msg), range(msg)))
    a_clear = np.array([a for a, _ in pairs])
    b_clear = np.array([b for _, b in pairs])

    def fresh():
        a, b = cks.encrypt_many(a_clear), cks.encrypt_many(b_clear)
        return (a.to_device(), b.to_device()) if device else (a, b)

    def dec(ct):
        return np.asarray(cks.decrypt_message_and_carry(ct))

    a, b = fresh()
    t.eq(dec(sks.unchecked_add(a, b)), a_clear + b_clear)
    r = sks.smart_add(a, b)
    t.eq(dec(r) % msg, (a_clear + b_clear) % msg)
    assert r.degree == 2 * (msg - 1)
    # sub / neg: results are correct modulo the message modulus (tests.rs:1125-1160)
    r = sks.unchecked_sub(a, b)
    t.eq(dec(r) % msg, (a_clear - b_clear) % msg)
    r, z = sks.unchecked_neg_with_z(a)
    assert z == msg and r.degree == z
    t.eq(dec(r), z - a_clear)
    t.eq(dec(sks.smart_neg(a)) % msg, (-a_clear) % msg)
    # scalar operations
    for s in (0, 1, 3):
        t.eq(dec(sks.unchecked_scalar_add(a, s)), a_clear + s)
        t.eq(dec(sks.unchecked_scalar_sub(a, s)) % msg, (a_clear - s) % msg)
        t.eq(dec(sks.unchecked_scalar_mul(a, s)), a_clear * s)
        t.eq(dec(sks.smart_scalar_mul(a.clone(), s)) % msg, (a_clear * s) % msg)
    big = sks.unchecked_scalar_mul(a, 4)          # degree 12: the next leveled op must bootstrap
    assert not sks.is_scalar_add_possible(big, 5)
    r = sks.smart_scalar_add(big, 5)
    assert r.degree == msg - 1
    t.eq(dec(r), (4 * a_clear + 5) % msg)
    # multiplications
    a, b = fresh()
    t.eq(dec(sks.unchecked_mul_lsb(a, b)), (a_clear * b_clear) % msg)
    t.eq(dec(sks.unchecked_mul_msb(a, b)), (a_clear * b_clear) // msg)
    t.eq(dec(sks.smart_mul_lsb(a, b)), (a_clear * b_clear) % msg)
    t.eq(dec(sks.unchecked_mul_lsb_small_carry(a, b)) % msg, (a_clear * b_clear) % msg)
    # bitwise
    a, b = fresh()
    t.eq(dec(sks.unchecked_bitand(a, b)), a_clear & b_clear)
    t.eq(dec(sks.unchecked_bitor(a, b)), a_clear | b_clear)
    t.eq(dec(sks.smart_bitxor(a, b)), a_clear ^ b_clear)
    # comparisons
    for name, f in (("greater", np.greater), ("greater_or_equal", np.greater_equal), ("less", np.less),
                    ("less_or_equal", np.less_equal), ("equal", np.equal)):
        r = getattr(sks, "unchecked_" + name)(a, b)
        assert r.degree == 1
        t.eq(dec(r), f(a_clear, b_clear).astype(int), name)
        r = getattr(sks, "smart_scalar_" + name)(a, 2)
        t.eq(dec(r), f(a_clear, 2).astype(int), name)
    # division family
    want = np.array([0 if y == 0 else x // y for x, y in pairs])
    t.eq(dec(sks.unchecked_div(a, b)), want)
    t.eq(dec(sks.unchecked_scalar_div(a, 2)), a_clear // 2)
    t.eq(dec(sks.unchecked_scalar_mod(a, 3)), a_clear % 3)
    t.eq(dec(sks.unchecked_scalar_right_shift(a, 1)), a_clear >> 1)
    t.eq(dec(sks.unchecked_scalar_left_shift(a, 1)), a_clear << 1)
    r = sks.smart_scalar_left_shift(sks.unchecked_scalar_mul(a, 4), 1)
    t.eq(dec(r), (a_clear * 8) % msg)
    # carry / message extraction on a carry-full ciphertext
    full = sks.unchecked_add(sks.unchecked_scalar_mul(a, 4), b)      # 4a + b < 16
    t.eq(dec(sks.carry_extract(full)), a_clear)
    t.eq(dec(sks.message_extract(full)), b_clear)
    # checked forms refuse what would overflow the carry space
    import pytest

    with pytest.raises(SI.CheckError):
        sks.checked_add(full, full)
    with pytest.raises(SI.CheckError):
        sks.checked_mul_lsb(full, b)
    # the CRT flavour of scalar add scales by the KEY's moduli (scalar_add.rs:32-47): on a native-CRT
    # ciphertext (modulus 3, carry 1) adding s moves the body by s/16 of half the torus
    if not device:
        crt = cks.encrypt_native_crt(2, 3)
        before = crt.ct.copy()
        sks.unchecked_scalar_add_assign_crt(crt, 5)
        assert crt.degree == 2 + 5
        delta = (1 << 63) // (sks.message_modulus * sks.carry_modulus)
        assert int(crt.ct[-1]) == (int(before[-1]) + 5 * delta) % (1 << 64)
        np.testing.assert_array_equal(crt.ct[:-1], before[:-1])
        # create_trivial_assign: value reduced mod the message modulus, mask zero, n+1 words
        # (server_side/mod.rs:284-310)
        sks.create_trivial_assign(crt, 7)
        assert crt.degree == 7 % msg and crt.ct.size == sks.parameters.lwe_dimension + 1
        assert int(crt.ct[-1]) == (7 % msg) * delta and not crt.ct[:-1].any()


def check_deferred_is_one_launch(cks, sks):
    """Bootstraps issued inside ``deferred`` become a single launch with per-ciphertext tables."""
    backend = sks.key
    a, b = cks.encrypt(2), cks.encrypt(3)
    before = backend.launches
    with sks.deferred():
        lo = sks.unchecked_mul_lsb(a, b)
        hi = sks.unchecked_mul_msb(a, b)
        eq = sks.unchecked_equal(a, b)
    assert backend.launches == before + 1
    assert (cks.decrypt_message_and_carry(lo), cks.decrypt_message_and_carry(hi), cks.decrypt_message_and_carry(eq)) == (2, 1, 0)


def check_radix_ops(cks, sks, count=6, device=False, seed=3, num_blocks=4, tally=None, reference_schedule=False):
    """FheUint8-shaped integers (4 blocks of 2 bits): the reference's radix tests on a batch.
    ``reference_schedule``: every propagation step bootstraps what the reference's does."""
    t = tally or Tally()
    icks = INT.RadixClientKey.__new__(INT.RadixClientKey)
    icks.key, icks.num_blocks = cks, num_blocks
    isks = INT.ServerKey(sks, reference_schedule=reference_schedule)
    mod = sks.message_modulus ** num_blocks
    rs = np.random.RandomState(seed)
    x = rs.randint(0, mod, size=count).astype(np.uint64)
    y = rs.randint(0, mod, size=count).astype(np.uint64)
    x[0], y[0] = mod - 1, mod - 1           # every carry set
    x[1], y[1] = 0, mod - 1
    dev = "cuda" if device else None

    def enc(v):
        return icks.encrypt_many(v, device=dev)

    def dec(ct):
        return icks.decrypt(ct)

    cx, cy = enc(x), enc(y)
    t.eq(dec(cx), x)
    # add: leveled, then smart adds until a propagation is forced (radix/tests.rs smart_add loop)
    acc = isks.unchecked_add(cx, cy)
    clear = (x + y) % mod
    t.eq(dec(acc), clear)
    for _ in range(4):
        acc = isks.smart_add(acc, cy)
        clear = (clear + y) % mod
        t.eq(dec(acc), clear)
    isks.full_propagate(acc)
    assert all(b.degree == sks.message_modulus - 1 for b in acc.blocks)
    t.eq(dec(acc), clear)
    # neg / sub
    t.eq(dec(isks.unchecked_neg(cx)), (mod - x) % mod)
    t.eq(dec(isks.smart_sub(cx.clone(), cy.clone())), (x - y) % mod)
    # scalar operations
    for s in (0, 1, 77, mod - 1):
        t.eq(dec(isks.unchecked_scalar_add(cx, s)), (x + s) % mod)
        t.eq(dec(isks.smart_scalar_sub(cx.clone(), s)), (x - s) % mod)
    t.eq(dec(isks.smart_small_scalar_mul(cx.clone(), 3)), (x * 3) % mod)
    t.eq(dec(isks.smart_scalar_mul(cx.clone(), 113)), (x * 113) % mod)
    # multiplication: one block, then the full product
    blk = isks.unchecked_block_mul(cx, cy.blocks[1], 1)
    y1 = (y // sks.message_modulus) % sks.message_modulus
    t.eq(dec(blk), (x * y1 * sks.message_modulus) % mod)
    t.eq(dec(isks.unchecked_mul(cx, cy)), (x * y) % mod)
    prod = isks.smart_mul(cx.clone(), cy.clone())
    t.eq(dec(prod), (x * y) % mod)
    prod2 = isks.smart_mul(prod, cx.clone())            # operands with carries
    t.eq(dec(prod2), (x * y * x) % mod)
    # bitwise
    t.eq(dec(isks.smart_bitand(cx.clone(), cy.clone())), x & y)
    t.eq(dec(isks.smart_bitor(cx.clone(), cy.clone())), x | y)
    t.eq(dec(isks.smart_bitxor(acc, cy.clone())), clear ^ y)
    # shifts (radix/tests.rs: the value is shifted inside the modulus)
    for sh in (1, 2, 3, 5):
        t.eq(dec(isks.unchecked_scalar_left_shift(cx.clone(), sh)), (x << np.uint64(sh)) % mod)
        t.eq(dec(isks.unchecked_scalar_right_shift(cx.clone(), sh)), x >> np.uint64(sh))
    # block shifts by the whole width or more leave zeros and keep the block count (radix/scalar_mul.rs:280-292)
    nb = len(cx.blocks)
    for sh in (0, 1, nb, nb + 3):
        left, right = isks.blockshift(cx.clone(), sh), isks.blockshift_right(cx.clone(), sh)
        assert len(left.blocks) == nb and len(right.blocks) == nb
        t.eq(dec(left), (x << np.uint64(2 * min(sh, nb))) % mod if sh < nb else x * 0)
        t.eq(dec(right), x >> np.uint64(2 * min(sh, nb)) if sh < nb else x * 0)
    # single integers go through the same code (reference signatures)
    one = icks.encrypt(201)
    two = icks.encrypt(77)
    if not device:
        assert icks.decrypt(isks.smart_mul(one, two)) == (201 * 77) % mod


def check_gathered_operations(cks, sks, count=5, device=False, seed=9, tally=None):
    """Different operations on different integers inside ``gather``: same decryptions as one after
    the other, and as many launches as the longest of them needs, not the sum."""
    t = tally or Tally()
    icks = INT.RadixClientKey.__new__(INT.RadixClientKey)
    icks.key, icks.num_blocks = cks, 4
    isks = INT.ServerKey(sks)
    mod = sks.message_modulus ** 4
    rs = np.random.RandomState(seed)
    a, b, c, d, e, f = (rs.randint(0, mod, size=count).astype(np.uint64) for _ in range(6))
    dev = "cuda" if device else None
    ca, cb, cc, cd, ce, cf = (icks.encrypt_many(v, device=dev) for v in (a, b, c, d, e, f))
    backend = sks.key

    def carried(x, y):
        s = isks.unchecked_add(x, y)
        return isks.unchecked_add(s, y)      # degrees 9: the next smart operation must propagate

    def launches_of(fn):
        before = backend.launches if hasattr(backend, "launches") else None
        r = fn()
        return r, (backend.launches - before if before is not None else None)

    # fresh operands start without a propagation chain: block products + the one sum level that needs carries
    _, n_mul = launches_of(lambda: isks.smart_mul(ca.clone(), cb.clone()))
    _, n_add = launches_of(lambda: isks.smart_add(carried(cc, cd), carried(cd, cc)))
    _, n_xor = launches_of(lambda: isks.smart_bitxor(ce.clone(), cf.clone()))
    if n_mul is not None:
        assert n_mul <= 6 and n_xor == 1, (n_mul, n_add, n_xor)
    (r_mul, r_add, r_xor), n_all = launches_of(lambda: isks.gather([
        lambda: isks.smart_mul(ca.clone(), cb.clone()),
        lambda: isks.smart_add(carried(cc, cd), carried(cd, cc)),
        lambda: isks.smart_bitxor(ce.clone(), cf.clone()),
    ]))
    t.eq(icks.decrypt(r_mul), (a * b) % mod)
    t.eq(icks.decrypt(r_add), (3 * (c + d)) % mod)
    t.eq(icks.decrypt(r_xor), e ^ f)
    if n_all is not None:
        assert n_all == max(n_mul, n_add, n_xor), (n_all, n_mul, n_add, n_xor)
    # gathers nest: an operation that gathers its own parts joins the rounds of the outer gather
    (n_mul2, n_xor2), n_nested = launches_of(lambda: isks.gather([
        lambda: isks.gather([lambda: isks.smart_mul(ca.clone(), cb.clone()), lambda: isks.smart_mul(cb.clone(), ca.clone())]),
        lambda: isks.smart_bitxor(ce.clone(), cf.clone()),
    ]))
    t.eq(icks.decrypt(n_mul2[0]), (a * b) % mod)
    t.eq(icks.decrypt(n_mul2[1]), (a * b) % mod)
    t.eq(icks.decrypt(n_xor2), e ^ f)
    if n_nested is not None:
        assert n_nested == n_mul, (n_nested, n_mul)
    # an exception inside one operation surfaces in the caller, the others still complete
    import pytest

    with pytest.raises(ZeroDivisionError):
        isks.gather([lambda: 1 // 0, lambda: isks.smart_bitxor(ce.clone(), cf.clone())])
    # a launch that fails (here: the backend raises) unwinds every gathered operation and surfaces once
    real = backend.ks_pbs if not device else backend.ks_pbs_device

    def broken(*args, **kwargs):
        raise RuntimeError("launch failed")

    name = "ks_pbs_device" if device else "ks_pbs"
    setattr(backend, name, broken)
    try:
        with pytest.raises(RuntimeError, match="launch failed"):
            isks.gather([lambda: isks.smart_bitxor(ce.clone(), cf.clone()), lambda: isks.smart_bitand(ce.clone(), cf.clone())])
    finally:
        setattr(backend, name, real)
    t.eq(icks.decrypt(isks.smart_bitand(ce.clone(), cf.clone())), e & f)
    # a fresh integer needs no bootstrap to be "propagated"
    _, n_prop = launches_of(lambda: isks.full_propagate(ca.clone()))
    if n_prop is not None:
        assert n_prop == 0


def make_basis(message_modulus: int):
    """concrete-integer/src/server_key/crt/tests.rs:25-33."""
    if message_modulus in (2, 3):
        return [2]
    if message_modulus < 8:
        return [2, 3]
    if message_modulus < 16:
        return [2, 5, 7]
    return [3, 7, 13]


def check_crt_ops(cks, sks, count=6, device=False, seed=5, tally=None, rounds=6):
    """The CRT tests of the reference (crt/tests.rs:35-273) on a batch that holds every residue
    pair at least once; ``rounds`` stands for NB_TEST in the loops that feed a result back."""
    t = tally or Tally()
    icks = INT.ClientKey.from_shortint(cks)
    isks = INT.ServerKey(sks)
    basis = make_basis(sks.message_modulus)
    mod = int(np.prod(basis))
    rs = np.random.RandomState(seed)
    x = rs.randint(0, mod, size=count).astype(np.uint64)
    y = rs.randint(0, mod, size=count).astype(np.uint64)
    x[0], y[0] = mod - 1, mod - 1
    x[1], y[1] = 0, mod - 1
    dev = "cuda" if device else None

    def enc(v):
        return icks.encrypt_crt_many(v, basis, device=dev)

    def dec(ct):
        return icks.decrypt_crt(ct)

    t.eq(dec(enc(x)), x)
    # unchecked mul
    a = enc(x)
    isks.unchecked_crt_mul_assign(a, enc(y))
    t.eq(dec(a), (x * y) % mod)
    # smart add, fed back
    a, b, clear = enc(x), enc(y), x.copy()
    for _ in range(rounds):
        isks.smart_crt_add_assign(a, b)
        clear = (clear + y) % mod
        t.eq(dec(a), clear)
    # smart mul, fed back (operands pick up degrees that force reductions)
    a, b, clear = enc(x), enc(y), x.copy()
    for _ in range(max(2, rounds // 2)):
        isks.smart_crt_mul_assign(a, b)
        clear = (clear * y) % mod
        t.eq(dec(a), clear)
    # smart neg, fed back
    a, clear = enc(x), x.copy()
    for _ in range(rounds):
        isks.smart_crt_neg_assign(a)
        clear = (mod - clear) % mod
        t.eq(dec(a), clear)
    # scalar add / mul on fresh ciphertexts, scalar sub fed back
    for s in (0, 1, mod - 1, 1234567):
        a = enc(x)
        isks.smart_crt_scalar_add_assign(a, s)
        t.eq(dec(a), (x + s) % mod)
        a = enc(x)
        isks.smart_crt_scalar_mul_assign(a, s)
        t.eq(dec(a), (x * (s % mod)) % mod)
    a, clear, s = enc(x), x.copy(), mod - 2
    for _ in range(rounds):
        isks.smart_crt_scalar_sub_assign(a, s)
        clear = (clear + mod - s) % mod
        t.eq(dec(a), clear)
    # smart sub, fed back
    a, b, clear = enc(x), enc(y), x.copy()
    for _ in range(rounds):
        isks.smart_crt_sub_assign(a, b)
        clear = (clear + mod - y) % mod
        t.eq(dec(a), clear)
    # a function that commutes with the reductions (crt/mod.rs:44-65: x -> x*x)
    a = enc(x)
    isks.pbs_crt_compliant_function_assign(a, lambda v: v * v)
    t.eq(dec(a), (x * x) % mod)
    # checked forms and the value-returning forms leave their input alone
    import pytest

    a = enc(x)
    r = isks.unchecked_crt_scalar_add(a, 1)
    t.eq(dec(a), x)
    t.eq(dec(r), (x + 1) % mod)
    big = a
    while isks.is_crt_add_possible(big, big):
        big = isks.unchecked_crt_add(big, big)
    with pytest.raises(SI.CheckError):
        isks.checked_crt_scalar_mul(big, mod - 1)
    # single integers use the same code (reference signatures); native CRT is client-side only
    if not device:
        one, two = icks.encrypt_crt(5 % mod, basis), icks.encrypt_crt(4 % mod, basis)
        assert icks.decrypt_crt(isks.smart_crt_mul(one, two)) == (5 * 4) % mod
        crt_key = INT.CrtClientKey.from_key(icks, basis)
        assert crt_key.decrypt(crt_key.encrypt(mod - 1)) == mod - 1
        native = icks.encrypt_native_crt(13 % mod, basis)
        assert icks.decrypt_native_crt(native) == 13 % mod
