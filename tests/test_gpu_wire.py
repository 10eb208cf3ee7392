"""Server keys through the wire formats and the key cache, on the GPU: a key read back from its
serialized form must produce bit-identical ciphertexts, because the device-resident key material
travels as it lies in memory (shortint) or is converted by the same kernel again (boolean)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("variant_name", ["auto", "generic", "block"])
def test_shortint_server_key_round_trip_is_bit_exact(gpu_ctx, variant_name):
    import concrete_amd.parameters as P
    from concrete_amd import backend, shortint, wire

    variant = {"auto": backend.CAMD_PBS_AUTO, "generic": backend.CAMD_PBS_GENERIC, "block": backend.CAMD_PBS_BLOCK}[variant_name]
    p = P.PARAM_MESSAGE_2_CARRY_2
    cks = shortint.ClientKey(p, seed=21)
    sks = shortint.ServerKey.new(cks, ctx=gpu_ctx, variant=variant)
    data = wire.serialize_shortint_server_key(sks)
    back = wire.deserialize_shortint_server_key(data, p, ctx=gpu_ctx)
    assert back.key.variant == sks.key.variant
    assert (back.message_modulus, back.carry_modulus, back.max_degree) == (4, 4, 15)
    cts = cks.encrypt_batch([i % 4 for i in range(96)])
    f = lambda x: (3 * x + 1) % 4   # noqa: E731
    want = sks.keyswitch_programmable_bootstrap_batch(cts, sks.generate_accumulator(f))
    got = back.keyswitch_programmable_bootstrap_batch(cts, back.generate_accumulator(f))
    np.testing.assert_array_equal(got, want)
    dec = cks.decrypt_message_and_carry_batch(got)
    assert dec.tolist() == [f(i % 4) for i in range(96)]
    # a second serialization of the reloaded key is byte-identical
    assert wire.serialize_shortint_server_key(back) == data
    # the wrong parameter set is refused before anything reaches the GPU
    with pytest.raises(wire.WireError):
        wire.deserialize_shortint_server_key(data, P.PARAM_MESSAGE_1_CARRY_1, ctx=gpu_ctx)
    # a server key with a restricted degree keeps it (server_key/mod.rs:463)
    sks.max_degree = 7
    assert wire.deserialize_shortint_server_key(wire.serialize_shortint_server_key(sks), p, ctx=gpu_ctx).max_degree == 7


def test_boolean_server_key_round_trip_is_bit_exact(gpu_ctx):
    import concrete_amd.parameters as P
    from concrete_amd import boolean, wire

    p = P.DEFAULT_PARAMETERS
    cks = boolean.ClientKey(p, seed=4)
    sks = boolean.ServerKey.new(cks, ctx=gpu_ctx)
    data = wire.serialize_boolean_server_key(sks.standard_bootstrap_key, sks.key_switching_key, p)
    back = wire.deserialize_boolean_server_key(data, p, ctx=gpu_ctx)
    rng = np.random.default_rng(0)
    a_bits, b_bits = rng.integers(0, 2, 64).astype(bool), rng.integers(0, 2, 64).astype(bool)
    a, b = cks.encrypt_batch(a_bits), cks.encrypt_batch(b_bits)
    for gate, f in (("nand", lambda x, y: ~(x & y)), ("xor", lambda x, y: x ^ y)):
        want, got = sks.gate_batch(gate, a, b), back.gate_batch(gate, a, b)
        np.testing.assert_array_equal(got, want)
        np.testing.assert_array_equal(cks.decrypt_batch(got), f(a_bits, b_bits))
    assert wire.serialize_boolean_server_key(back.standard_bootstrap_key, back.key_switching_key, p) == data


def test_key_cache_generates_once_then_loads(gpu_ctx, tmp_path, monkeypatch):
    import concrete_amd.parameters as P
    from concrete_amd import keycache, shortint

    p = P.PARAM_MESSAGE_1_CARRY_1
    storage = keycache.FileStorage(str(tmp_path / "keys"))
    first = keycache.KeyCache(storage).get_from_param(p, gpu_ctx, seed=7)
    path = storage.path(p)
    assert os.path.exists(path)
    cache = keycache.KeyCache(storage)
    # a second cache over the same directory must not generate anything
    monkeypatch.setattr(shortint.ServerKey, "new", classmethod(lambda *a, **k: pytest.fail("key generated again")))
    second = cache.get_from_param(p, gpu_ctx, seed=8)
    assert cache.get_from_param(p, gpu_ctx) is second           # memory hit
    np.testing.assert_array_equal(second.client_key().lwe_secret_key, first.client_key().lwe_secret_key)
    cts = first.client_key().encrypt_batch([0, 1, 0, 1])
    want = first.server_key().keyswitch_bootstrap_batch(cts)
    got = second.server_key().keyswitch_bootstrap_batch(cts)
    np.testing.assert_array_equal(got, want)
    assert second.client_key().decrypt_message_and_carry_batch(got).tolist() == [0, 1, 0, 1]
    # integer keys over the same cache (concrete-integer/src/keycache.rs:11-19)
    icks, isks = keycache.IntegerKeyCache(cache).get_from_params(p, gpu_ctx)
    assert isks.key.max_degree == (p.message_modulus - 1) * p.carry_modulus - 1
    assert icks.decrypt_radix(icks.encrypt_radix(5, 3)) == 5
    # a file for other parameters under this name, or a damaged file, is a miss (keycache.rs:59-61)
    monkeypatch.undo()
    with open(path, "r+b") as f:
        f.seek(0)
        f.write(b"\x01")
    assert storage.load(p, gpu_ctx) is None
    with open(path, "wb") as f:
        f.write(b"garbage")
    assert storage.load(p, gpu_ctx) is None
    third = keycache.KeyCache(storage).get_from_param(p, gpu_ctx, seed=7)
    assert storage.load(p, gpu_ctx) is not None
    np.testing.assert_array_equal(third.client_key().lwe_secret_key, first.client_key().lwe_secret_key)
