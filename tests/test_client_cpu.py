"""Client-side encodings of the shortint ClientKey that have no padding bit, written after the
reference's own tests (concrete-shortint/src/server_key/tests.rs:197-217) and doc examples
(client_key/mod.rs:186-319), for every published parameter set that the reference's macro covers."""
import numpy as np
import pytest

from concrete_amd import parameters as P
from concrete_amd import shortint

SETS = [P.PARAM_MESSAGE_1_CARRY_1, P.PARAM_MESSAGE_2_CARRY_2, P.PARAM_MESSAGE_3_CARRY_3, P.PARAM_MESSAGE_4_CARRY_4]


@pytest.mark.parametrize("p", SETS, ids=lambda p: f"{p.message_modulus}x{p.carry_modulus}")
def test_encrypt_decrypt_without_padding(p):
    cks = shortint.ClientKey(p, seed=31)
    modulus = p.message_modulus * p.carry_modulus   # the largest space without a padding bit
    for clear in range(modulus):
        ct = cks.encrypt_without_padding(clear)
        assert (ct.degree, ct.message_modulus, ct.carry_modulus) == (p.message_modulus - 1, p.message_modulus, p.carry_modulus)
        assert cks.decrypt_message_and_carry_without_padding(ct) == clear
        assert cks.decrypt_without_padding(ct) == clear % p.message_modulus
    # the encoding really has no padding bit: the top value sits in the upper half of the torus
    top = cks.encrypt_without_padding(modulus - 1)
    assert cks.decrypt_message_and_carry(top) == 2 * (modulus - 1)


@pytest.mark.parametrize("p", SETS[:3], ids=lambda p: f"{p.message_modulus}x{p.carry_modulus}")
def test_native_crt_encoding(p):
    cks = shortint.ClientKey(p, seed=32)
    for basis in (2, 3, 5, 7):
        if basis > p.message_modulus * p.carry_modulus:
            continue
        for msg in range(2 * basis):
            ct = cks.encrypt_native_crt(msg, basis)
            assert (ct.degree, ct.message_modulus, ct.carry_modulus) == (basis - 1, basis, 1)
            assert cks.decrypt_message_native_crt(ct, basis) == msg % basis


def test_batches_decode_elementwise():
    p = P.PARAM_MESSAGE_2_CARRY_2
    cks = shortint.ClientKey(p, seed=33)
    rows = np.stack([cks.encrypt_without_padding(m).ct for m in (0, 5, 15, 9)])
    batch = shortint.Ciphertext(rows, 3, 4, 4)
    assert cks.decrypt_message_and_carry_without_padding(batch).tolist() == [0, 5, 15, 9]
    assert cks.decrypt_without_padding(batch).tolist() == [0, 1, 3, 1]
    rows = np.stack([cks.encrypt_native_crt(m, 5).ct for m in (4, 0, 7)])
    assert cks.decrypt_message_native_crt(shortint.Ciphertext(rows, 4, 5, 1), 5).tolist() == [4, 0, 2]


# ---- randomness: keys that protect data come from a CSPRNG keyed by the operating system ----

def test_secure_generator_is_chacha20():
    """Known answer: ChaCha20 with an all-zero key and nonce, block 0, starts 76 b8 e0 ad a0 f1 3d 90
    (the classic test vector). Encrypting with a zero key and no noise exposes the mask words,
    which are outputs 0 and 3 of the stream (two words go to the noise sample in between)."""
    import ctypes as C

    from concrete_amd import _ffi, client as CL

    rng = CL.Rng(1)
    C.memset(C.byref(rng.state), 0, C.sizeof(rng.state))
    rng.state.secure = 1
    cts = CL.lwe_encrypt(rng, 64, np.zeros(1, dtype=np.uint64), np.zeros(2, dtype=np.uint64), 0.0)
    assert int(cts[0, 0]) == 0x903DF1A0ADE0B876
    assert int(cts[1, 0]) == 0xC70D778BCCEF36A8


def test_default_client_keys_are_fresh_and_seeded_ones_reproducible():
    p = P.PARAM_MESSAGE_2_CARRY_2
    a, b = shortint.ClientKey.new(p), shortint.ClientKey.new(p)
    assert not np.array_equal(a.glwe_secret_key, b.glwe_secret_key)
    assert not np.array_equal(a.lwe_secret_key_after_ks, b.lwe_secret_key_after_ks)
    assert set(np.unique(a.glwe_secret_key)) <= {0, 1}
    assert 0.4 < a.glwe_secret_key.mean() < 0.6
    s1, s2 = shortint.ClientKey.new(p, seed=7), shortint.ClientKey.new(p, seed=7)
    assert np.array_equal(s1.glwe_secret_key, s2.glwe_secret_key)
    # both kinds encrypt and decrypt
    for cks in (a, s1):
        for m in range(4):
            assert cks.decrypt(cks.encrypt(m)) == m


@pytest.mark.parametrize("seed", [None, 5])
def test_masks_after_reload_carry_no_key_bits(seed):
    """A key that is written out and read back must not replay the sequence that produced its
    secret bits: with the old behaviour the top bit of mask word i of the first ciphertext after a
    reload WAS bit i of the secret key."""
    from concrete_amd import wire

    p = P.PARAM_MESSAGE_2_CARRY_2
    cks = shortint.ClientKey.new(p, seed=seed)
    back = wire.deserialize_shortint_client_key(wire.serialize_shortint_client_key(cks), seed=seed)
    assert np.array_equal(back.glwe_secret_key, cks.glwe_secret_key)
    ct = back.encrypt(1)
    words = np.asarray(ct.ct if hasattr(ct, "ct") else ct, dtype=np.uint64).ravel()
    n = p.lwe_dimension
    top = (words[:n] >> np.uint64(63)).astype(np.int64)
    agree = int((top == cks.lwe_secret_key_after_ks.astype(np.int64)).sum())
    # independent bits agree about half the time (769 trials: 6 sigma is +-83)
    assert abs(agree - n / 2) < 90, agree
    top2 = (words[: 2 * 1024] >> np.uint64(63)).astype(np.int64)
    agree2 = int((top2 == cks.glwe_secret_key.astype(np.int64)).sum())
    assert abs(agree2 - 1024) < 140, agree2
