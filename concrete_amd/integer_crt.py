"""The CRT representation of concrete-integer: an integer modulo a product of pairwise coprime
moduli is one shortint ciphertext per modulus (``concrete-integer/src/ciphertext/mod.rs:57-61``),
and every operation acts on the residues independently
(``concrete-integer/src/server_key/crt/*.rs``; ``crt_parallel/*.rs`` hands the residues to rayon
workers).

Independence is what the GPU path wants: the bootstraps of one operation over all residues — and,
because every block may be a batch, over many integers — are collected and issued as ONE launch
with a look-up table id per ciphertext (``shortint.ServerKey.deferred``). The ``*_parallelized``
names are aliases of that single schedule.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Iterable, List

import numpy as np

from . import shortint as _si
from .shortint_ops import CheckError


@dataclass
class CrtCiphertext:
    """ciphertext/mod.rs:57-61: ``blocks[i]`` encrypts the value modulo ``moduli[i]``."""
    blocks: List[_si.Ciphertext]
    moduli: List[int]

    def clone(self) -> "CrtCiphertext":
        return CrtCiphertext([b.clone() for b in self.blocks], list(self.moduli))

    @property
    def count(self) -> int:
        return self.blocks[0].count

    def to_device(self, device="cuda") -> "CrtCiphertext":
        return CrtCiphertext([b.to_device(device) for b in self.blocks], list(self.moduli))

    def to_host(self) -> "CrtCiphertext":
        return CrtCiphertext([b.to_host() for b in self.blocks], list(self.moduli))


def extended_euclid_s(f: int, g: int) -> int:
    """The Bezout coefficient of ``f`` when the remainder sequence of (f, g) reaches the gcd
    (client_key/utils.rs:37-54: ``s[l]``)."""
    s0, s1, r0, r1 = 1, 0, int(f), int(g)
    while r1 != 0:
        q = abs(r0) // abs(r1) * (1 if (r0 >= 0) == (r1 >= 0) else -1)   # i64 division truncates
        r0, r1 = r1, r0 - q * r1
        s0, s1 = s1, s0 - q * s1
    return s0


def i_crt(moduli: Iterable[int], values: Iterable[int]) -> int:
    """client_key/utils.rs:56-76, u64 arithmetic."""
    moduli, values = [int(m) for m in moduli], [int(v) for v in values]
    mask = 0xFFFFFFFFFFFFFFFF
    big_mod = 1
    for m in moduli:
        big_mod = (big_mod * m) & mask
    out = 0
    for m, v in zip(moduli, values):
        tmp_mod = big_mod // m
        s = extended_euclid_s(tmp_mod, m)
        if s < 0:
            # Rust's % keeps the sign of the dividend
            s = -((-s) % m) + m
        c = ((v * s) & mask) % m
        out = (out + c * tmp_mod) & mask
    return out % big_mod


class CrtClientOps:
    """CRT entry points of the integer ClientKey (client_key/mod.rs:279-431); mixed into
    ``integer.ClientKey``, which supplies ``self.key`` (the shortint client key)."""

    def encrypt_crt(self, message: int, base_vec: Iterable[int]) -> CrtCiphertext:
        """client_key/mod.rs:299-316."""
        base = [int(b) for b in base_vec]
        return CrtCiphertext([self.key.encrypt_with_message_modulus(message, m) for m in base], base)

    def encrypt_crt_many(self, messages: Iterable[int], base_vec: Iterable[int], device=None) -> CrtCiphertext:
        """A batch of integers: block i holds the residue modulo ``base_vec[i]`` of every message."""
        base = [int(b) for b in base_vec]
        messages = [int(m) for m in messages]
        blocks = []
        for m in base:
            rows = np.stack([self.key.encrypt_with_message_modulus(x, m).ct for x in messages])
            p = self.key.parameters
            blocks.append(_si.Ciphertext(rows, m - 1, m, (p.message_modulus * p.carry_modulus) // m))
        ct = CrtCiphertext(blocks, base)
        return ct.to_device(device) if device is not None else ct

    def _recombine(self, moduli, residues):
        whole = 1
        for m in moduli:
            whole *= m
        if np.ndim(residues[0]) == 0:
            return i_crt(moduli, residues) % whole
        return np.array([i_crt(moduli, [int(r[j]) for r in residues]) % whole for j in range(len(residues[0]))],
                        dtype=np.uint64)

    def decrypt_crt(self, ctxt: CrtCiphertext):
        """client_key/mod.rs:339-355."""
        residues = [self.key.decrypt_message_and_carry(c) % m for c, m in zip(ctxt.blocks, ctxt.moduli)]
        return self._recombine(ctxt.moduli, residues)

    def encrypt_native_crt(self, message: int, base_vec: Iterable[int]) -> CrtCiphertext:
        """client_key/mod.rs:377-394."""
        base = [int(b) for b in base_vec]
        return CrtCiphertext([self.key.encrypt_native_crt(message, m) for m in base], base)

    def decrypt_native_crt(self, ct: CrtCiphertext):
        """client_key/mod.rs:416-431."""
        residues = [self.key.decrypt_message_native_crt(c, m) for c, m in zip(ct.blocks, ct.moduli)]
        return self._recombine(ct.moduli, residues)


class CrtClientKey:
    """client_key/crt.rs:6-43: an integer client key that remembers its CRT basis."""

    def __init__(self, parameters, moduli: Iterable[int], seed: int | None = None, key=None):
        from .integer import ClientKey

        self.key = key if key is not None else ClientKey(parameters, seed)
        self.moduli = [int(m) for m in moduli]

    @classmethod
    def new(cls, parameters, moduli: Iterable[int], seed: int | None = None) -> "CrtClientKey":
        return cls(parameters, moduli, seed)

    @classmethod
    def from_key(cls, key, moduli: Iterable[int]) -> "CrtClientKey":
        return cls(None, moduli, key=key)

    def as_ref(self):
        return self.key

    def encrypt(self, message: int) -> CrtCiphertext:
        return self.key.encrypt_crt(message, self.moduli)

    def encrypt_many(self, messages: Iterable[int], device=None) -> CrtCiphertext:
        return self.key.encrypt_crt_many(messages, self.moduli, device)

    def decrypt(self, ciphertext: CrtCiphertext):
        return self.key.decrypt_crt(ciphertext)

    def parameters(self):
        return self.key.parameters()


class CrtServerOps:
    """server_key/crt/*.rs; mixed into ``integer.ServerKey``, which supplies ``self.key`` (the
    shortint server key)."""

    # ------------------------------------------------------------------ crt/mod.rs --
    def _full_extract_many(self, ctxts) -> None:
        with self.key.deferred():
            for ctxt in ctxts:
                for b in ctxt.blocks:
                    self.key.message_extract_assign(b)

    def full_extract(self, ctxt: CrtCiphertext) -> None:
        """crt/mod.rs:38-42: every residue reduced to its message, all of them in one launch."""
        self._full_extract_many([ctxt])

    full_extract_parallelized = full_extract

    def pbs_crt_compliant_function_assign(self, ct1: CrtCiphertext, f: Callable[[int], int]) -> None:
        """crt/mod.rs:66-81: ``f`` must commute with the reductions modulo the basis."""
        with self.key.deferred():
            for block, b in zip(ct1.blocks, ct1.moduli):
                acc = self.key.generate_accumulator(lambda x, _b=b: f(x) % _b)
                self.key.keyswitch_programmable_bootstrap_assign(block, acc)

    def pbs_crt_compliant_function(self, ct1: CrtCiphertext, f: Callable[[int], int]) -> CrtCiphertext:
        res = ct1.clone()
        self.pbs_crt_compliant_function_assign(res, f)
        return res

    # ---------------------------------------------------------------- add_crt.rs --
    def is_crt_add_possible(self, lhs: CrtCiphertext, rhs: CrtCiphertext) -> bool:
        return all(self.key.is_add_possible(a, b) for a, b in zip(lhs.blocks, rhs.blocks))

    def unchecked_crt_add_assign(self, lhs: CrtCiphertext, rhs: CrtCiphertext) -> None:
        for a, b in zip(lhs.blocks, rhs.blocks):
            self.key.unchecked_add_assign(a, b)

    def unchecked_crt_add(self, lhs: CrtCiphertext, rhs: CrtCiphertext) -> CrtCiphertext:
        res = lhs.clone()
        self.unchecked_crt_add_assign(res, rhs)
        return res

    def smart_crt_add(self, lhs: CrtCiphertext, rhs: CrtCiphertext) -> CrtCiphertext:
        """add_crt.rs:32-43; both operands are extracted in one launch."""
        if not self.is_crt_add_possible(lhs, rhs):
            self._full_extract_many([lhs, rhs])
        return self.unchecked_crt_add(lhs, rhs)

    def smart_crt_add_assign(self, lhs: CrtCiphertext, rhs: CrtCiphertext) -> None:
        if not self.is_crt_add_possible(lhs, rhs):
            self._full_extract_many([lhs, rhs])
        self.unchecked_crt_add_assign(lhs, rhs)

    # ---------------------------------------------------------------- mul_crt.rs --
    def unchecked_crt_mul_assign(self, lhs: CrtCiphertext, rhs: CrtCiphertext) -> None:
        """mul_crt.rs:31-35: residue-wise low product, one launch."""
        with self.key.deferred():
            for a, b in zip(lhs.blocks, rhs.blocks):
                self.key.unchecked_mul_lsb_assign(a, b)

    def unchecked_crt_mul(self, lhs: CrtCiphertext, rhs: CrtCiphertext) -> CrtCiphertext:
        res = lhs.clone()
        self.unchecked_crt_mul_assign(res, rhs)
        return res

    def smart_crt_mul_assign(self, lhs: CrtCiphertext, rhs: CrtCiphertext) -> None:
        """mul_crt.rs:71-75 applies smart_mul_lsb_assign residue by residue; here the reductions
        that the degrees call for share one launch and the products another."""
        self.key.smart_mul_lsb_assign_many(zip(lhs.blocks, rhs.blocks))

    def smart_crt_mul(self, lhs: CrtCiphertext, rhs: CrtCiphertext) -> CrtCiphertext:
        """mul_crt.rs:77-85 (the left operand is cloned before the smart product)."""
        res = lhs.clone()
        self.smart_crt_mul_assign(res, rhs)
        return res

    # ---------------------------------------------------------------- neg_crt.rs --
    def is_crt_neg_possible(self, ctxt: CrtCiphertext) -> bool:
        return all(self.key.is_neg_possible(b) for b in ctxt.blocks)

    def unchecked_crt_neg_assign(self, ctxt: CrtCiphertext) -> None:
        for b in ctxt.blocks:
            self.key.unchecked_neg_assign(b)

    def unchecked_crt_neg(self, ctxt: CrtCiphertext) -> CrtCiphertext:
        res = ctxt.clone()
        self.unchecked_crt_neg_assign(res)
        return res

    def smart_crt_neg_assign(self, ctxt: CrtCiphertext) -> None:
        if not self.is_crt_neg_possible(ctxt):
            self.full_extract(ctxt)
        self.unchecked_crt_neg_assign(ctxt)

    def smart_crt_neg(self, ctxt: CrtCiphertext) -> CrtCiphertext:
        if not self.is_crt_neg_possible(ctxt):
            self.full_extract(ctxt)
        return self.unchecked_crt_neg(ctxt)

    # ---------------------------------------------------------------- sub_crt.rs --
    def is_crt_sub_possible(self, lhs: CrtCiphertext, rhs: CrtCiphertext) -> bool:
        return all(self.key.is_sub_possible(a, b) for a, b in zip(lhs.blocks, rhs.blocks))

    def unchecked_crt_sub_assign(self, lhs: CrtCiphertext, rhs: CrtCiphertext) -> None:
        """sub_crt.rs:60-67."""
        self.unchecked_crt_add_assign(lhs, self.unchecked_crt_neg(rhs))

    def unchecked_crt_sub(self, lhs: CrtCiphertext, rhs: CrtCiphertext) -> CrtCiphertext:
        res = lhs.clone()
        self.unchecked_crt_sub_assign(res, rhs)
        return res

    def smart_crt_sub_assign(self, lhs: CrtCiphertext, rhs: CrtCiphertext) -> None:
        if not self.is_crt_sub_possible(lhs, rhs):
            self._full_extract_many([lhs, rhs])
        self.unchecked_crt_sub_assign(lhs, rhs)

    def smart_crt_sub(self, lhs: CrtCiphertext, rhs: CrtCiphertext) -> CrtCiphertext:
        if not self.is_crt_sub_possible(lhs, rhs):
            self._full_extract_many([lhs, rhs])
        return self.unchecked_crt_sub(lhs, rhs)

    # ------------------------------------------- scalar_{add,sub,mul}_crt.rs --
    @staticmethod
    def _u64(scalar: int) -> int:
        return int(scalar) & 0xFFFFFFFFFFFFFFFF

    def _scalar_add_residues(self, ct: CrtCiphertext, scalar: int):
        return [(self._u64(scalar) % m) & 0xFF for m in ct.moduli]

    def _scalar_sub_residues(self, ct: CrtCiphertext, scalar: int):
        return [((m - self._u64(scalar) % m) % m) & 0xFF for m in ct.moduli]

    def unchecked_crt_scalar_add_assign(self, ct: CrtCiphertext, scalar: int) -> None:
        """scalar_add_crt.rs:59-65."""
        for b, s in zip(ct.blocks, self._scalar_add_residues(ct, scalar)):
            self.key.unchecked_scalar_add_assign(b, s)

    def is_crt_scalar_add_possible(self, ct: CrtCiphertext, scalar: int) -> bool:
        return all(self.key.is_scalar_add_possible(b, s) for b, s in zip(ct.blocks, self._scalar_add_residues(ct, scalar)))

    def unchecked_crt_scalar_sub_assign(self, ct: CrtCiphertext, scalar: int) -> None:
        """scalar_sub_crt.rs:57-64: adds the negated residue with the key's own scaling."""
        for b, s in zip(ct.blocks, self._scalar_sub_residues(ct, scalar)):
            self.key.unchecked_scalar_add_assign_crt(b, s)

    def is_crt_scalar_sub_possible(self, ct: CrtCiphertext, scalar: int) -> bool:
        return all(self.key.is_scalar_add_possible(b, s) for b, s in zip(ct.blocks, self._scalar_sub_residues(ct, scalar)))

    def unchecked_crt_scalar_mul_assign(self, ct: CrtCiphertext, scalar: int) -> None:
        """scalar_mul_crt.rs:59-64."""
        for b, m in zip(ct.blocks, ct.moduli):
            self.key.unchecked_scalar_mul_assign(b, (self._u64(scalar) % m) & 0xFF)

    def is_crt_scalar_mul_possible(self, ct: CrtCiphertext, scalar: int) -> bool:
        return all(self.key.is_scalar_mul_possible(b, (self._u64(scalar) % m) & 0xFF) for b, m in zip(ct.blocks, ct.moduli))

    def is_crt_small_scalar_mul_possible(self, ct: CrtCiphertext, scalar: int) -> bool:
        """scalar_mul_crt.rs:217-224: the scalar is NOT reduced modulo the basis here."""
        return all(self.key.is_scalar_mul_possible(b, self._u64(scalar) & 0xFF) for b in ct.blocks)

    def smart_crt_scalar_mul_assign(self, ct: CrtCiphertext, scalar: int) -> None:
        """scalar_mul_crt.rs:208-214 (tests the unreduced scalar, unlike smart_crt_scalar_mul)."""
        if not self.is_crt_small_scalar_mul_possible(ct, scalar):
            self.full_extract(ct)
        self.unchecked_crt_scalar_mul_assign(ct, scalar)


def _install_scalar_forms() -> None:
    """unchecked_/checked_/smart_ value and assign forms of the three scalar operations
    (scalar_add_crt.rs, scalar_sub_crt.rs, scalar_mul_crt.rs), and the ``*_parallelized`` aliases of
    every CRT operation (crt_parallel/*.rs)."""
    for name in ("add", "sub", "mul"):
        unchecked_assign = getattr(CrtServerOps, f"unchecked_crt_scalar_{name}_assign")
        possible = getattr(CrtServerOps, f"is_crt_scalar_{name}_possible")

        def unchecked(self, ct, scalar, _ua=unchecked_assign):
            res = ct.clone()
            _ua(self, res, scalar)
            return res

        def checked(self, ct, scalar, _ua=unchecked_assign, _p=possible):
            if not _p(self, ct, scalar):
                raise CheckError()
            res = ct.clone()
            _ua(self, res, scalar)
            return res

        def checked_assign(self, ct, scalar, _ua=unchecked_assign, _p=possible) -> None:
            if not _p(self, ct, scalar):
                raise CheckError()
            _ua(self, ct, scalar)

        def smart(self, ct, scalar, _ua=unchecked_assign, _p=possible):
            if not _p(self, ct, scalar):
                self.full_extract(ct)
            res = ct.clone()
            _ua(self, res, scalar)
            return res

        def smart_assign(self, ct, scalar, _ua=unchecked_assign, _p=possible) -> None:
            if not _p(self, ct, scalar):
                self.full_extract(ct)
            _ua(self, ct, scalar)

        setattr(CrtServerOps, f"unchecked_crt_scalar_{name}", unchecked)
        setattr(CrtServerOps, f"checked_crt_scalar_{name}", checked)
        setattr(CrtServerOps, f"checked_crt_scalar_{name}_assign", checked_assign)
        setattr(CrtServerOps, f"smart_crt_scalar_{name}", smart)
        if name != "mul":   # the assign form of mul has its own test (see above)
            setattr(CrtServerOps, f"smart_crt_scalar_{name}_assign", smart_assign)
    for attr in list(vars(CrtServerOps)):
        if "crt" in attr and not attr.startswith(("_", "is_")) and not attr.endswith("_parallelized"):
            setattr(CrtServerOps, attr + "_parallelized", getattr(CrtServerOps, attr))


_install_scalar_forms()
