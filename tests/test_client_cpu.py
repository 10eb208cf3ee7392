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
