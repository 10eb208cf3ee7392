"""Wire formats on the CPU: frames against bytes written out by hand from the bincode rules and the
reference's struct definitions (the reference holds no serialized fixture, so byte-level parity
with files written by the reference itself is unpinned), round trips, and rejection of damaged
input. Server keys need a GPU and are covered by the GPU tests."""
import struct

import numpy as np
import pytest

from concrete_amd import boolean, integer, keycache, shortint, wire
from concrete_amd import parameters as P


def u64(*vals):
    return b"".join(struct.pack("<Q", v) for v in vals)


def lwe_payload_by_hand(words, fmt):
    return struct.pack("<I", 0) + struct.pack("<Q", len(words)) + b"".join(struct.pack(fmt, w) for w in words)


def test_shortint_ciphertext_frame_by_hand():
    # ciphertext/mod.rs:91-97: data: Vec<u8>, degree, message_modulus, carry_modulus
    words = [1, 0xFFFFFFFFFFFFFFFF, 0x0123456789ABCDEF]
    ct = shortint.Ciphertext(np.array(words, dtype=np.uint64), 3, 4, 4)
    payload = lwe_payload_by_hand(words, "<Q")
    expected = u64(len(payload)) + payload + u64(3, 4, 4)
    assert wire.serialize_shortint_ciphertext(ct) == expected
    assert len(expected) == 8 + (4 + 8 + 24) + 24
    back = wire.deserialize_shortint_ciphertext(expected)
    assert back.ct.dtype == np.uint64 and back.ct.tolist() == words
    assert (back.degree, back.message_modulus, back.carry_modulus) == (3, 4, 4)


def test_boolean_ciphertext_frames_by_hand():
    # ciphertext/mod.rs:17-21: enum, Encrypted(Vec<u8>) = 0, Trivial(bool) = 1
    words = [7, 0xFFFFFFFF, 5]
    payload = lwe_payload_by_hand(words, "<I")
    enc = struct.pack("<I", 0) + u64(len(payload)) + payload
    ct = boolean.Ciphertext(lwe=np.array(words, dtype=np.uint32))
    assert wire.serialize_boolean_ciphertext(ct) == enc
    assert wire.deserialize_boolean_ciphertext(enc).lwe.tolist() == words
    for b in (False, True):
        data = wire.serialize_boolean_ciphertext(boolean.Ciphertext(trivial=b))
        assert data == struct.pack("<I", 1) + bytes([int(b)])
        back = wire.deserialize_boolean_ciphertext(data)
        assert back.is_trivial and back.trivial is b
    with pytest.raises(wire.WireError):
        wire.deserialize_boolean_ciphertext(struct.pack("<I", 2))
    with pytest.raises(wire.WireError):
        wire.deserialize_boolean_ciphertext(struct.pack("<I", 1) + b"\x02")


def test_shortint_parameters_frame():
    # parameters/mod.rs:28-45: sixteen 8-byte fields in declaration order
    p = P.PARAM_MESSAGE_2_CARRY_2
    data = wire.serialize_shortint_parameters(p)
    expected = (u64(769, 2, 1024) + struct.pack("<dd", 4.3131554647504185e-06, 2.9403601535432533e-16)
                + u64(23, 1, 4, 3) + u64(1, 23) + struct.pack("<d", 2.9403601535432533e-16) + u64(0, 0) + u64(4, 4))
    assert data == expected and len(data) == 128
    for name, q in P.SHORTINT_BY_NAME.items():
        assert wire.deserialize_shortint_parameters(wire.serialize_shortint_parameters(q)) == q, name
    for q in (P.DEFAULT_PARAMETERS, P.TFHE_LIB_PARAMETERS):
        data = wire.serialize_boolean_parameters(q)
        assert len(data) == 72 and wire.deserialize_boolean_parameters(data) == q


def test_shortint_client_key_round_trip_decrypts():
    p = P.PARAM_MESSAGE_1_CARRY_1
    cks = shortint.ClientKey(p, seed=11)
    data = wire.serialize_shortint_client_key(cks)
    # three Vec<u8> payloads (4-byte tag, 8-byte count, words; the GLWE key adds its polynomial
    # size) and the 128-byte parameter frame
    big = p.glwe_dimension * p.polynomial_size
    assert len(data) == (8 + 12 + 8 * big) + (8 + 12 + 8 * big + 8) + (8 + 12 + 8 * p.lwe_dimension) + 128
    back = wire.deserialize_shortint_client_key(data, seed=5)
    assert back.parameters == p
    np.testing.assert_array_equal(back.lwe_secret_key, cks.lwe_secret_key)
    np.testing.assert_array_equal(back.glwe_secret_key, cks.glwe_secret_key)
    np.testing.assert_array_equal(back.lwe_secret_key_after_ks, cks.lwe_secret_key_after_ks)
    for m in range(p.message_modulus):
        assert back.decrypt(cks.encrypt(m)) == m
        assert cks.decrypt(wire.deserialize_shortint_ciphertext(wire.serialize_shortint_ciphertext(back.encrypt(m)))) == m


def test_boolean_client_key_round_trip_decrypts():
    cks = boolean.ClientKey(P.DEFAULT_PARAMETERS, seed=3)
    back = wire.deserialize_boolean_client_key(wire.serialize_boolean_client_key(cks), seed=9)
    assert back.parameters == P.DEFAULT_PARAMETERS
    for m in (False, True):
        ct = wire.deserialize_boolean_ciphertext(wire.serialize_boolean_ciphertext(cks.encrypt(m)))
        assert back.decrypt(ct) is m


def test_batches_and_radix_round_trip():
    p = P.PARAM_MESSAGE_2_CARRY_2
    cks = integer.RadixClientKey(p, 4, seed=2)
    one = cks.encrypt(0xB7)
    data = wire.serialize_radix_ciphertext(one)
    assert data[:8] == u64(4)
    back = wire.deserialize_radix_ciphertext(data)
    assert cks.decrypt(back) == 0xB7
    many = cks.encrypt_many([1, 200, 77])
    many.blocks[2].degree = 5
    back = wire.deserialize_radix_ciphertext_batch(wire.serialize_radix_ciphertext(many))
    assert cks.decrypt(back).tolist() == [1, 200, 77]
    assert [b.degree for b in back.blocks] == [3, 3, 5, 3]
    batch = cks.key.encrypt_many([0, 1, 2, 3, 2])
    back = wire.deserialize_shortint_ciphertext_batch(wire.serialize_shortint_ciphertext(batch))
    np.testing.assert_array_equal(back.ct, batch.ct)
    assert cks.key.decrypt_message_and_carry(back).tolist() == [0, 1, 2, 3, 2]


def test_damaged_input_is_rejected():
    ct = shortint.Ciphertext(np.arange(9, dtype=np.uint64), 3, 4, 4)
    data = wire.serialize_shortint_ciphertext(ct)
    for cut in range(len(data)):
        with pytest.raises(wire.WireError):
            wire.deserialize_shortint_ciphertext(data[:cut])
    with pytest.raises(wire.WireError):
        wire.deserialize_shortint_ciphertext(data + b"\x00")
    bad_version = bytearray(data)
    bad_version[8] = 1   # payload version tag
    with pytest.raises(wire.WireError, match="version"):
        wire.deserialize_shortint_ciphertext(bytes(bad_version))
    huge = bytearray(data)
    huge[12:20] = struct.pack("<Q", 1 << 60)   # word count far beyond the data
    with pytest.raises(wire.WireError):
        wire.deserialize_shortint_ciphertext(bytes(huge))
    zero_modulus = data[:-16] + u64(0, 4)
    with pytest.raises(wire.WireError, match="modulus"):
        wire.deserialize_shortint_ciphertext(zero_modulus)
    cks = shortint.ClientKey(P.PARAM_MESSAGE_1_CARRY_1, seed=1)
    key = bytearray(wire.serialize_shortint_client_key(cks))
    key[8 + 12] = 2   # first key word
    with pytest.raises(wire.WireError, match="binary"):
        wire.deserialize_shortint_client_key(bytes(key))


def test_fourier_payload_of_another_writer_is_refused():
    p = P.PARAM_MESSAGE_1_CARRY_1
    with pytest.raises(wire.WireError, match="not written by this library"):
        wire.parse_fourier_key_payload(struct.pack("<I", 0) + b"\x00" * 64, p, 64)


def test_fourier_payload_of_another_build_is_refused():
    """The Fourier key bytes are in the order of the kernels that wrote them. The payload carries
    the library's layout id for (parameter set, kernel family); one written under another id - a
    build whose kernels place the spectrum differently - is refused as a WireError, which the key
    cache treats as a miss. So is the format that had no id."""
    import dataclasses

    from concrete_amd.backend import fourier_layout_id

    p = dataclasses.replace(P.PARAM_MESSAGE_2_CARRY_2, lwe_dimension=1)
    size = p.lwe_dimension * p.pbs_level * (p.glwe_dimension + 1) ** 2 * (p.polynomial_size // 2) * 16

    def payload(fmt, variant, layout):
        w = wire.Writer().raw(wire.FOURIER_MAGIC).u32(fmt).u32(variant).u32(64)
        if fmt >= 2:
            w.u64(layout)
        w.u64(p.lwe_dimension).u64(p.glwe_dimension).u64(p.polynomial_size).u64(p.pbs_level).u64(p.pbs_base_log)
        w.byte_vec(bytes(size))
        return w.finish()

    ids = {v: fourier_layout_id(p, v, 64) for v in (1, 2, 3)}
    assert len(set(ids.values())) == 3                      # the three families lay the key out differently
    assert ids[2] == fourier_layout_id(p, 0, 64)            # AUTO resolves to the wave family for this set
    other = dataclasses.replace(P.PARAM_MESSAGE_4_CARRY_4, lwe_dimension=1)
    assert fourier_layout_id(other, 3, 64) != ids[3]
    for v in (1, 2, 3):
        variant, body = wire.parse_fourier_key_payload(payload(wire.FOURIER_FORMAT, v, ids[v]), p, 64)
        assert variant == v and len(body) == size
        with pytest.raises(wire.WireError, match="another build"):
            wire.parse_fourier_key_payload(payload(wire.FOURIER_FORMAT, v, ids[v] ^ 1), p, 64)
    with pytest.raises(wire.WireError, match="another build"):
        wire.parse_fourier_key_payload(payload(wire.FOURIER_FORMAT, 2, ids[3]), p, 64)   # another family's order
    with pytest.raises(wire.WireError, match="unknown Fourier key format"):
        wire.parse_fourier_key_payload(payload(1, 2, 0), p, 64)
    with pytest.raises(wire.WireError, match="unknown kernel family"):
        wire.parse_fourier_key_payload(payload(wire.FOURIER_FORMAT, 9, ids[2]), p, 64)


def test_kernel_family_that_auto_selects_for_every_parameter_set():
    """CAMD_PBS_AUTO resolves (on the host, no GPU) to the wave family for every shortint set of the
    reference except the three N = 256 sets and the three-level 1_7, and for both boolean sets; the
    layout id of AUTO then equals the wave family's and differs from the block family's.
    Parameter sets: concrete-shortint/src/parameters/mod.rs:179-808, concrete-boolean/src/parameters/mod.rs:30-72."""
    from concrete_amd.backend import CAMD_PBS_AUTO, CAMD_PBS_BLOCK, CAMD_PBS_WAVE, fourier_layout_id

    block_sets = {"PARAM_MESSAGE_1_CARRY_0", "PARAM_MESSAGE_1_CARRY_1", "PARAM_MESSAGE_2_CARRY_0", "PARAM_MESSAGE_1_CARRY_7"}
    seen = 0
    for name, p in sorted(P.SHORTINT_BY_NAME.items()):
        auto, block = fourier_layout_id(p, CAMD_PBS_AUTO, 64), fourier_layout_id(p, CAMD_PBS_BLOCK, 64)
        if name in block_sets:
            assert auto == block, name
        else:
            assert auto == fourier_layout_id(p, CAMD_PBS_WAVE, 64) != block, name
        seen += 1
    assert seen >= 30
    # the three-level set has a wave instance, on request only
    assert fourier_layout_id(P.PARAM_MESSAGE_1_CARRY_7, CAMD_PBS_WAVE, 64) != fourier_layout_id(P.PARAM_MESSAGE_1_CARRY_7, CAMD_PBS_BLOCK, 64)
    for p in (P.DEFAULT_PARAMETERS, P.TFHE_LIB_PARAMETERS):
        assert fourier_layout_id(p, CAMD_PBS_AUTO, 32) == fourier_layout_id(p, CAMD_PBS_WAVE, 32)


def test_param_names():
    assert keycache.param_name(P.PARAM_MESSAGE_2_CARRY_2) == "PARAM_MESSAGE_2_CARRY_2"
    assert keycache.FileStorage("some/dir").path(P.PARAM_MESSAGE_4_CARRY_4) == "some/dir/PARAM_MESSAGE_4_CARRY_4.bin"
    with pytest.raises(KeyError, match="Unnamed parameters"):
        keycache.param_name(P.Parameters(10, 1, 256, 1e-3, 1e-9, 10, 1, 2, 3, 2, 2))
