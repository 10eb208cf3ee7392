"""GPU tests of the host-side mirrors of concrete-shortint and concrete-boolean: the reference's
own test properties (concrete-shortint/src/server_key/tests.rs:219-368,
concrete-boolean/src/server_key/tests.rs:162-640) run against the GPU backend, on keys made by
the package's own client code."""
import numpy as np
import pytest

import concrete_amd.parameters as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def shortint_keys(gpu_ctx):
    from concrete_amd.shortint import ClientKey, ServerKey

    cks = ClientKey.new(P.PARAM_MESSAGE_2_CARRY_2, seed=11)
    sks = ServerKey.new(cks, ctx=gpu_ctx)
    return cks, sks


def test_shortint_keyswitch_bootstrap(shortint_keys):
    """tests.rs:219-250: keyswitch_bootstrap keeps the message."""
    cks, sks = shortint_keys
    for m in range(4):
        ct = cks.encrypt(m)
        out = sks.keyswitch_bootstrap(ct)
        assert cks.decrypt(out) == m
        assert out.degree == ct.degree  # the reference does not touch the degree here


def test_shortint_programmable_and_accumulator(shortint_keys):
    """tests.rs:252-277 (identity table) and :343-368 (table for 2x)."""
    cks, sks = shortint_keys
    acc_id = sks.generate_accumulator(lambda x: x)
    acc_2x = sks.generate_accumulator(lambda x: (2 * x) % 4)
    assert sks.generate_accumulator(lambda x: x) is acc_id  # cached by content
    for m in range(4):
        ct = cks.encrypt(m)
        assert cks.decrypt(sks.keyswitch_programmable_bootstrap(ct, acc_id)) == m
        assert cks.decrypt(sks.keyswitch_programmable_bootstrap(ct, acc_2x)) == (2 * m) % 4


def test_shortint_carry_and_message_extract(shortint_keys):
    """tests.rs:280-340."""
    cks, sks = shortint_keys
    for clear in (0, 5, 10, 15):
        ct = cks.unchecked_encrypt(clear)
        carry = sks.carry_extract(ct)
        msg = sks.message_extract(ct)
        assert cks.decrypt_message_and_carry(carry) == clear // 4 and carry.degree == 3
        assert cks.decrypt_message_and_carry(msg) == clear % 4 and msg.degree == 3


def test_shortint_bivariate_and_batch(shortint_keys):
    cks, sks = shortint_keys
    a, b = cks.encrypt(3), cks.encrypt(2)
    prod = sks.unchecked_functional_bivariate_pbs(a, b, lambda x: ((x // 4) * (x % 4)) % 4)
    assert cks.decrypt(prod) == (3 * 2) % 4
    msgs = [i % 4 for i in range(37)]
    cts = cks.encrypt_batch(msgs)
    accs = [sks.generate_accumulator(lambda x: (x + 1) % 4), sks.generate_accumulator(lambda x: x % 4)]
    out = sks.keyswitch_programmable_bootstrap_batch(cts, [accs[i % 2] for i in range(37)])
    dec = cks.decrypt_message_and_carry_batch(out)
    assert list(dec) == [((m + 1) % 4) if i % 2 == 0 else m for i, m in enumerate(msgs)]
    triv = sks.create_trivial(3)
    assert cks.decrypt(sks.keyswitch_bootstrap(triv)) == 3


def test_boolean_gates_and_mux(gpu_ctx):
    """Truth tables of every gate, ct/ct, ct/bool and trivial mixes, then MUX and a small circuit."""
    from concrete_amd.boolean import Ciphertext, ClientKey, ServerKey

    cks = ClientKey.new(P.DEFAULT_PARAMETERS, seed=5)
    sks = ServerKey.new(cks, ctx=gpu_ctx)
    clear = {"and_": lambda x, y: x and y, "nand": lambda x, y: not (x and y), "or_": lambda x, y: x or y,
             "nor": lambda x, y: not (x or y), "xor": lambda x, y: x != y, "xnor": lambda x, y: x == y}
    for name, f in clear.items():
        gate = getattr(sks, name)
        for x in (False, True):
            for y in (False, True):
                assert cks.decrypt(gate(cks.encrypt(x), cks.encrypt(y))) == f(x, y), (name, x, y)
                assert cks.decrypt(gate(cks.encrypt(x), y)) == f(x, y)
                assert cks.decrypt(gate(x, cks.encrypt(y))) == f(x, y)
                assert cks.decrypt(gate(Ciphertext(trivial=x), cks.encrypt(y))) == f(x, y)
                assert gate(sks.trivial_encrypt(x), Ciphertext(trivial=y)).trivial == f(x, y)
    for c in (False, True):
        for t in (False, True):
            for e in (False, True):
                out = sks.mux(cks.encrypt(c), cks.encrypt(t), cks.encrypt(e))
                assert cks.decrypt(out) == (t if c else e)
    assert cks.decrypt(sks.not_(cks.encrypt(True))) is False
    # batched gate == single gates
    xs = [bool(i & 1) for i in range(20)]
    ys = [bool(i & 2) for i in range(20)]
    out = sks.gate_batch("xor", cks.encrypt_batch(xs), cks.encrypt_batch(ys))
    assert list(cks.decrypt_batch(out)) == [x != y for x, y in zip(xs, ys)]
    # deep-ish circuit: a chain of 64 gates keeps decrypting correctly (tests.rs:611-640 in small)
    rs = np.random.RandomState(3)
    vals = [bool(v) for v in rs.randint(0, 2, 8)]
    cts = [cks.encrypt(v) for v in vals]
    names = list(clear)
    for step in range(64):
        i, j, g = rs.randint(0, 8), rs.randint(0, 8), names[rs.randint(0, 6)]
        cts[i] = getattr(sks, g)(cts[i], cts[j])
        vals[i] = clear[g](vals[i], vals[j])
    assert [cks.decrypt(c) for c in cts] == vals


def test_boolean_gate_batches_on_device(gpu_ctx):
    """Every gate on a device-resident batch (linear part by camd_lwe_axpby_device, 32-bit torus)
    equals the host path ciphertext for ciphertext, and decrypts to the truth table."""
    import torch

    import concrete_amd.parameters as P
    from concrete_amd import boolean as B

    cks = B.ClientKey(P.DEFAULT_PARAMETERS, seed=5)
    sks = B.ServerKey.new(cks, ctx=gpu_ctx)
    bits_a = np.array([0, 0, 1, 1] * 8, dtype=bool)
    bits_b = np.array([0, 1, 0, 1] * 8, dtype=bool)
    ca, cb = cks.encrypt_batch(bits_a), cks.encrypt_batch(bits_b)
    da = torch.from_numpy(ca.view(np.int32)).cuda()
    db = torch.from_numpy(cb.view(np.int32)).cuda()
    clear = {"and": bits_a & bits_b, "nand": ~(bits_a & bits_b), "or": bits_a | bits_b, "nor": ~(bits_a | bits_b),
             "xor": bits_a ^ bits_b, "xnor": ~(bits_a ^ bits_b)}
    for gate, want in clear.items():
        host = sks.gate_batch(gate, ca, cb)
        dev = sks.gate_batch(gate, da, db).cpu().numpy().view(np.uint32)
        assert np.array_equal(host, dev), gate
        assert np.array_equal(cks.decrypt_batch(dev), want), gate


def test_boolean_gates_on_the_wave_kernel(gpu_ctx):
    """Every gate's truth table, MUX and a device-resident gate batch with the wave-per-ciphertext
    bootstrap selected explicitly for the default parameters."""
    import torch

    from concrete_amd import boolean as B
    from concrete_amd.backend import CAMD_PBS_WAVE

    cks = B.ClientKey(P.DEFAULT_PARAMETERS, seed=17)
    sks = B.ServerKey.new(cks, ctx=gpu_ctx, variant=CAMD_PBS_WAVE)
    assert sks.bootstrapper.key.variant == CAMD_PBS_WAVE
    gates = {"and_": lambda x, y: x and y, "nand": lambda x, y: not (x and y), "or_": lambda x, y: x or y,
             "nor": lambda x, y: not (x or y), "xor": lambda x, y: x != y, "xnor": lambda x, y: x == y}
    for name, f in gates.items():
        for x in (False, True):
            for y in (False, True):
                assert cks.decrypt(getattr(sks, name)(cks.encrypt(x), cks.encrypt(y))) == f(x, y), (name, x, y)
    for c in (False, True):
        assert cks.decrypt(sks.mux(cks.encrypt(c), cks.encrypt(True), cks.encrypt(False))) == c
    rng = np.random.default_rng(3)
    a_bits, b_bits = rng.integers(0, 2, 2000).astype(bool), rng.integers(0, 2, 2000).astype(bool)
    a = torch.from_numpy(cks.encrypt_batch(a_bits).view(np.int32)).cuda()
    b = torch.from_numpy(cks.encrypt_batch(b_bits).view(np.int32)).cuda()
    out = sks.gate_batch("xor", a, b)
    got = cks.decrypt_batch(out.cpu().numpy().view(np.uint32))
    assert np.array_equal(got, a_bits ^ b_bits)
