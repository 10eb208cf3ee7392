"""The concrete-shortint operation layer on top of the GPU keyswitch + bootstrap path.

Mirrors, name for name, ``concrete-shortint/src/server_key/{add,sub,neg,scalar_add,scalar_sub,
scalar_mul,mul,bitwise_op,comp_op,div_mod,shift}.rs`` and their engine counterparts under
``engine/server_side/``: every operation is a short linear combination of LWE ciphertexts
followed (or not) by one keyswitch + programmable bootstrap with a look-up table that depends on
the operation and on the operands' degrees. Degree bookkeeping and the ``is_*_possible``
predicates are the reference's.

A ``Ciphertext`` here may hold one LWE ciphertext (1-D numpy array), a host batch (2-D numpy
array) or a device-resident batch (2-D torch CUDA int64 tensor). Degrees are bookkeeping that does
not depend on the encrypted data, so one degree describes a whole batch and every operation below
works unchanged on batches: the linear part runs as one strided kernel on the GPU for device
batches (``camd_lwe_axpby_device``), the bootstraps of a batch are one launch, and look-up tables
are cached on the device by content instead of being rebuilt per call.
"""
from __future__ import annotations

from typing import Callable, Sequence

import numpy as np

M64 = (1 << 64) - 1


class CheckError(Exception):
    """``CheckError::CarryFull`` (server_key/mod.rs:32-46)."""

    def __init__(self, kind: str = "CarryFull"):
        super().__init__("The carry buffer is full")
        self.kind = kind


def _is_host(a) -> bool:
    return isinstance(a, np.ndarray)


# Degree helpers, ciphertext/mod.rs:14-75
def degree_after_bitxor(a: int, b: int) -> int:
    hi, lo = max(a, b), min(a, b)
    return max([hi] + [hi ^ i for i in range(lo + 1)])


def degree_after_bitor(a: int, b: int) -> int:
    hi, lo = max(a, b), min(a, b)
    return max([hi] + [hi | i for i in range(lo + 1)])


def degree_after_bitand(a: int, b: int) -> int:
    return min(a, b)


def degree_after_left_shift(degree: int, shift: int, modulus: int) -> int:
    return max((i << shift) % modulus for i in range(degree + 1))


def _division(x: int, modulus: int) -> int:
    """engine/server_side/div_mod.rs:5-11: 0 when the divisor digit is 0."""
    return 0 if x % modulus == 0 else (x // modulus) // (x % modulus)


class ShortintOps:
    """Mixed into ``shortint.ServerKey``; relies on its ``key``, ``parameters``, moduli,
    ``max_degree``, ``generate_accumulator`` and ``keyswitch_programmable_bootstrap_assign``."""

    # ------------------------------------------------------------------ linear part --
    @property
    def _delta(self) -> int:
        return (1 << 63) // (self.message_modulus * self.carry_modulus)

    def _lin(self, x, alpha: int, y=None, beta: int = 0, body_add: int = 0):
        """alpha*x + beta*y with ``body_add`` added to the body, wrapping mod 2^64."""
        if _is_host(x):
            out = x * np.uint64(alpha & M64)
            if y is not None:
                out = out + y * np.uint64(beta & M64)
            if body_add & M64:
                out[..., -1] += np.uint64(body_add & M64)
            return out
        import torch

        out = torch.empty_like(x)
        self.key.ctx.lwe_axpby_device(out, x, alpha, y, beta, body_add, 64, torch.cuda.current_stream().cuda_stream)
        return out

    def apply_lookup_table_many(self, items: Sequence[tuple]) -> None:
        """One launch for several (ciphertext, look-up table) pairs: each ciphertext (single or
        batch) is bootstrapped in place with its own table. This is what the reference spreads
        over rayon workers (concrete-integer radix_parallel/mod.rs:17-31, mul.rs:157-181)."""
        items = [(ct, acc) for ct, acc in items]
        if not items:
            return
        if len(items) == 1:
            self.keyswitch_programmable_bootstrap_assign(items[0][0], items[0][1])
            return
        host = _is_host(items[0][0].ct)
        parts, ids, shapes = [], [], []
        for ct, acc in items:
            a = ct.ct if ct.ct.ndim == 2 else ct.ct[None, :]
            parts.append(a)
            shapes.append((ct.ct.ndim, a.shape[0]))
            ids.append(np.full(a.shape[0], acc.lut_id, dtype=np.uint32))
        ids = np.concatenate(ids)
        if host:
            out = self.key.ks_pbs(np.concatenate(parts), ids)
        else:
            import torch

            batch = torch.cat(parts)
            out = torch.empty_like(batch)
            d_ids = torch.from_numpy(ids.view(np.int32)).to(batch.device)
            self.key.ks_pbs_device(batch, d_ids, out, batch.shape[0], torch.cuda.current_stream().cuda_stream)
        pos = 0
        for (ct, _), (ndim, rows) in zip(items, shapes):
            piece = out[pos:pos + rows]
            ct.ct = piece[0] if ndim == 1 else piece
            pos += rows

    def _pbs(self, ct, f: Callable[[int], int]) -> None:
        self.keyswitch_programmable_bootstrap_assign(ct, self.generate_accumulator(f))

    # ------------------------------------------------------------------------ add --
    def is_add_possible(self, lhs, rhs) -> bool:
        """server_key/add.rs:101-104."""
        return lhs.degree + rhs.degree <= self.max_degree

    def unchecked_add(self, lhs, rhs):
        res = lhs.clone()
        self.unchecked_add_assign(res, rhs)
        return res

    def unchecked_add_assign(self, lhs, rhs) -> None:
        """engine/server_side/add.rs:17-26: wrapping add of the LWE words, degrees add."""
        lhs.ct = self._lin(lhs.ct, 1, rhs.ct, 1)
        lhs.degree += rhs.degree

    def checked_add(self, lhs, rhs):
        if not self.is_add_possible(lhs, rhs):
            raise CheckError()
        return self.unchecked_add(lhs, rhs)

    def checked_add_assign(self, lhs, rhs) -> None:
        if not self.is_add_possible(lhs, rhs):
            raise CheckError()
        self.unchecked_add_assign(lhs, rhs)

    def smart_add(self, lhs, rhs):
        res = lhs.clone()
        self.smart_add_assign(res, rhs)
        return res

    def smart_add_assign(self, lhs, rhs) -> None:
        """engine/server_side/add.rs:40-58."""
        if not self.is_add_possible(lhs, rhs):
            if lhs.message_modulus - 1 + rhs.degree <= self.max_degree:
                self.message_extract_assign(lhs)
            elif rhs.message_modulus - 1 + lhs.degree <= self.max_degree:
                self.message_extract_assign(rhs)
            else:
                self.apply_lookup_table_many([(lhs, self._msg_acc(lhs)), (rhs, self._msg_acc(rhs))])
                lhs.degree, rhs.degree = lhs.message_modulus - 1, rhs.message_modulus - 1
        self.unchecked_add_assign(lhs, rhs)

    def _msg_acc(self, ct):
        modulus = ct.message_modulus
        return self.generate_accumulator(lambda x: x % modulus)

    def _extract_both(self, lhs, rhs) -> None:
        """message_extract_assign on two operands, as one launch."""
        self.apply_lookup_table_many([(lhs, self._msg_acc(lhs)), (rhs, self._msg_acc(rhs))])
        lhs.degree, rhs.degree = lhs.message_modulus - 1, rhs.message_modulus - 1

    # ------------------------------------------------------------------- neg / sub --
    def _neg_z(self, ct) -> int:
        msg_mod = ct.message_modulus
        return ((ct.degree + msg_mod - 1) // msg_mod) * msg_mod

    def is_neg_possible(self, ct) -> bool:
        """server_key/neg.rs:100-110."""
        msg_mod = self.message_modulus
        z = ((ct.degree + msg_mod - 1) // msg_mod) * msg_mod
        return z // (self.message_modulus - 1) <= self.max_degree

    def unchecked_neg_assign_with_z(self, ct) -> int:
        """engine/server_side/neg.rs:42-68: -ct + z*delta with z the next multiple of the
        message modulus above the degree; the result encrypts z - m."""
        z = self._neg_z(ct)
        ct.ct = self._lin(ct.ct, -1, body_add=z * self._delta)
        ct.degree = z
        return z

    def unchecked_neg_with_z(self, ct):
        res = ct.clone()
        z = self.unchecked_neg_assign_with_z(res)
        return res, z

    def unchecked_neg(self, ct):
        return self.unchecked_neg_with_z(ct)[0]

    def unchecked_neg_assign(self, ct) -> None:
        self.unchecked_neg_assign_with_z(ct)

    def checked_neg(self, ct):
        if not self.is_neg_possible(ct):
            raise CheckError()
        return self.unchecked_neg(ct)

    def checked_neg_assign(self, ct) -> None:
        if not self.is_neg_possible(ct):
            raise CheckError()
        self.unchecked_neg_assign(ct)

    def smart_neg(self, ct):
        """engine/server_side/neg.rs:70-78."""
        if not self.is_neg_possible(ct):
            self.keyswitch_bootstrap_assign(ct)
        return self.unchecked_neg(ct)

    def smart_neg_assign(self, ct) -> None:
        if not self.is_neg_possible(ct):
            self.keyswitch_bootstrap_assign(ct)
        self.unchecked_neg_assign(ct)

    def is_sub_possible(self, lhs, rhs) -> bool:
        """server_key/sub.rs:98-107."""
        msg_mod = self.message_modulus
        z = ((rhs.degree + msg_mod - 1) // msg_mod) * msg_mod
        return lhs.degree + z <= self.max_degree

    def unchecked_sub_with_correcting_term_assign(self, lhs, rhs) -> int:
        """engine/server_side/sub.rs:40-52: lhs + (-rhs + z*delta); returns z."""
        z = self._neg_z(rhs)
        lhs.ct = self._lin(lhs.ct, 1, rhs.ct, -1, body_add=z * self._delta)
        lhs.degree += z
        return z

    def unchecked_sub_with_correcting_term(self, lhs, rhs):
        res = lhs.clone()
        z = self.unchecked_sub_with_correcting_term_assign(res, rhs)
        return res, z

    def unchecked_sub(self, lhs, rhs):
        return self.unchecked_sub_with_correcting_term(lhs, rhs)[0]

    def unchecked_sub_assign(self, lhs, rhs) -> None:
        self.unchecked_sub_with_correcting_term_assign(lhs, rhs)

    def checked_sub(self, lhs, rhs):
        if not self.is_sub_possible(lhs, rhs):
            raise CheckError()
        return self.unchecked_sub(lhs, rhs)

    def checked_sub_assign(self, lhs, rhs) -> None:
        if not self.is_sub_possible(lhs, rhs):
            raise CheckError()
        self.unchecked_sub_assign(lhs, rhs)

    def smart_sub(self, lhs, rhs):
        """engine/server_side/sub.rs:54-65."""
        if not self.is_sub_possible(lhs, rhs):
            self._extract_both(lhs, rhs)
        return self.unchecked_sub(lhs, rhs)

    def smart_sub_assign(self, lhs, rhs) -> None:
        if not self.is_sub_possible(lhs, rhs):
            self._extract_both(lhs, rhs)
        self.unchecked_sub_assign(lhs, rhs)

    def smart_sub_with_correcting_term(self, lhs, rhs):
        if not self.is_sub_possible(lhs, rhs):
            self._extract_both(lhs, rhs)
        return self.unchecked_sub_with_correcting_term(lhs, rhs)

    # ---------------------------------------------------------------- scalar add --
    def is_scalar_add_possible(self, ct, scalar: int) -> bool:
        return int(scalar) + ct.degree <= self.max_degree

    def unchecked_scalar_add(self, ct, scalar: int):
        res = ct.clone()
        self.unchecked_scalar_add_assign(res, scalar)
        return res

    def unchecked_scalar_add_assign(self, ct, scalar: int) -> None:
        """engine/server_side/scalar_add.rs:19-32."""
        scalar = int(scalar) & 0xFF
        delta = (1 << 63) // (ct.message_modulus * ct.carry_modulus)
        ct.ct = self._lin(ct.ct, 1, body_add=scalar * delta)
        ct.degree += scalar

    def unchecked_scalar_add_assign_crt(self, ct, scalar: int) -> None:
        """engine/server_side/scalar_add.rs:32-47: the scaling comes from the server key's moduli,
        not the ciphertext's (ciphertexts of the CRT representation carry their own modulus)."""
        scalar = int(scalar) & 0xFF
        delta = (1 << 63) // (self.message_modulus * self.carry_modulus)
        ct.ct = self._lin(ct.ct, 1, body_add=scalar * delta)
        ct.degree += scalar

    def checked_scalar_add(self, ct, scalar: int):
        if not self.is_scalar_add_possible(ct, scalar):
            raise CheckError()
        return self.unchecked_scalar_add(ct, scalar)

    def checked_scalar_add_assign(self, ct, scalar: int) -> None:
        if not self.is_scalar_add_possible(ct, scalar):
            raise CheckError()
        self.unchecked_scalar_add_assign(ct, scalar)

    def smart_scalar_add(self, ct, scalar: int):
        res = ct.clone()
        self.smart_scalar_add_assign(res, scalar)
        return res

    def smart_scalar_add_assign(self, ct, scalar: int) -> None:
        """engine/server_side/scalar_add.rs:62-78."""
        scalar = int(scalar) & 0xFF
        modulus = self.message_modulus
        if self.is_scalar_add_possible(ct, scalar):
            self.unchecked_scalar_add_assign(ct, scalar)
        else:
            self._pbs(ct, lambda x: (scalar + x) % modulus)
            ct.degree = self.message_modulus - 1

    # ---------------------------------------------------------------- scalar sub --
    def _neg_scalar(self, scalar: int) -> int:
        return ((-int(scalar)) & 0xFF) % self.message_modulus  # u8 wrapping_neg, then mod

    def is_scalar_sub_possible(self, ct, scalar: int) -> bool:
        return self._neg_scalar(scalar) + ct.degree <= self.max_degree

    def unchecked_scalar_sub(self, ct, scalar: int):
        res = ct.clone()
        self.unchecked_scalar_sub_assign(res, scalar)
        return res

    def unchecked_scalar_sub_assign(self, ct, scalar: int) -> None:
        """engine/server_side/scalar_sub.rs:19-33: adds (-scalar mod 256) mod message_modulus."""
        neg_scalar = ((-int(scalar)) & 0xFF) % ct.message_modulus
        delta = (1 << 63) // (ct.message_modulus * ct.carry_modulus)
        ct.ct = self._lin(ct.ct, 1, body_add=neg_scalar * delta)
        ct.degree += neg_scalar

    def checked_scalar_sub(self, ct, scalar: int):
        if not self.is_scalar_sub_possible(ct, scalar):
            raise CheckError()
        return self.unchecked_scalar_sub(ct, scalar)

    def checked_scalar_sub_assign(self, ct, scalar: int) -> None:
        if not self.is_scalar_sub_possible(ct, scalar):
            raise CheckError()
        self.unchecked_scalar_sub_assign(ct, scalar)

    def smart_scalar_sub(self, ct, scalar: int):
        res = ct.clone()
        self.smart_scalar_sub_assign(res, scalar)
        return res

    def smart_scalar_sub_assign(self, ct, scalar: int) -> None:
        """engine/server_side/scalar_sub.rs:46-63 (the table wraps like the reference's u64)."""
        modulus = self.message_modulus
        if self.is_scalar_sub_possible(ct, scalar):
            self.unchecked_scalar_sub_assign(ct, scalar)
        else:
            s = int(scalar) & 0xFF
            self._pbs(ct, lambda x: ((x - s) & M64) % modulus)
            ct.degree = self.message_modulus - 1

    # ---------------------------------------------------------------- scalar mul --
    def is_scalar_mul_possible(self, ct, scalar: int) -> bool:
        return int(scalar) * ct.degree <= self.max_degree

    def unchecked_scalar_mul(self, ct, scalar: int):
        res = ct.clone()
        self.unchecked_scalar_mul_assign(res, scalar)
        return res

    def unchecked_scalar_mul_assign(self, ct, scalar: int) -> None:
        """engine/server_side/scalar_mul.rs:19-30."""
        scalar = int(scalar) & 0xFF
        ct.ct = self._lin(ct.ct, scalar)
        ct.degree *= scalar

    def checked_scalar_mul(self, ct, scalar: int):
        if not self.is_scalar_mul_possible(ct, scalar):
            raise CheckError()
        return self.unchecked_scalar_mul(ct, scalar)

    def checked_scalar_mul_assign(self, ct, scalar: int) -> None:
        if not self.is_scalar_mul_possible(ct, scalar):
            raise CheckError()
        self.unchecked_scalar_mul_assign(ct, scalar)

    def smart_scalar_mul(self, ct, scalar: int):
        res = ct.clone()
        self.smart_scalar_mul_assign(res, scalar)
        return res

    def smart_scalar_mul_assign(self, ct, scalar: int) -> None:
        """engine/server_side/scalar_mul.rs:43-63. The reference multiplies the degree a second
        time on the leveled branch (:52); kept, since degrees steer later decisions."""
        scalar = int(scalar) & 0xFF
        modulus = self.message_modulus
        if self.is_scalar_mul_possible(ct, scalar):
            self.unchecked_scalar_mul_assign(ct, scalar)
            ct.degree *= scalar
        else:
            self._pbs(ct, lambda x: (scalar * x) % modulus)
            ct.degree = self.message_modulus - 1

    # -------------------------------------------------------------------- shifts --
    def unchecked_scalar_right_shift(self, ct, shift: int):
        res = ct.clone()
        self.unchecked_scalar_right_shift_assign(res, shift)
        return res

    def unchecked_scalar_right_shift_assign(self, ct, shift: int) -> None:
        """engine/server_side/shift.rs:16-26."""
        self._pbs(ct, lambda x: x >> shift)
        ct.degree >>= shift

    def is_scalar_left_shift_possible(self, ct, shift: int) -> bool:
        return (ct.degree << shift) <= self.max_degree

    def unchecked_scalar_left_shift(self, ct, shift: int):
        res = ct.clone()
        self.unchecked_scalar_left_shift_assign(res, shift)
        return res

    def unchecked_scalar_left_shift_assign(self, ct, shift: int) -> None:
        self.unchecked_scalar_mul_assign(ct, 1 << shift)

    def checked_scalar_left_shift(self, ct, shift: int):
        if not self.is_scalar_left_shift_possible(ct, shift):
            raise CheckError()
        return self.unchecked_scalar_left_shift(ct, shift)

    def checked_scalar_left_shift_assign(self, ct, shift: int) -> None:
        if not self.is_scalar_left_shift_possible(ct, shift):
            raise CheckError()
        self.unchecked_scalar_left_shift_assign(ct, shift)

    def smart_scalar_left_shift(self, ct, shift: int):
        res = ct.clone()
        self.smart_scalar_left_shift_assign(res, shift)
        return res

    def smart_scalar_left_shift_assign(self, ct, shift: int) -> None:
        """engine/server_side/shift.rs:59-76."""
        if self.is_scalar_left_shift_possible(ct, shift):
            self.unchecked_scalar_left_shift_assign(ct, shift)
        else:
            modulus = self.message_modulus
            self._pbs(ct, lambda x: (x << shift) % modulus)
            ct.degree = degree_after_left_shift(ct.degree, shift, modulus)

    # --------------------------------------------------------- bivariate bootstraps --
    def is_functional_bivariate_pbs_possible(self, ct1, ct2) -> bool:
        """server_key/mod.rs:266-270."""
        return ct1.degree * (ct2.degree + 1) + ct2.degree < ct1.carry_modulus * ct1.message_modulus

    def unchecked_functional_bivariate_pbs(self, lhs, rhs, f: Callable[[int], int]):
        res = lhs.clone()
        self.unchecked_functional_bivariate_pbs_assign(res, rhs, f)
        return res

    def unchecked_functional_bivariate_pbs_assign(self, lhs, rhs, f: Callable[[int], int]) -> None:
        """engine/server_side/mod.rs:172-196: lhs*(deg_r+1) + rhs, then one bootstrap with f.
        On a device-resident batch the shift and the add ride on the keyswitch's loads."""
        modulus = rhs.degree + 1
        self.bootstrap_of_combination_assign(lhs, modulus & 0xFF, rhs, 1, self.generate_accumulator(f))
        lhs.degree = lhs.degree * (modulus & 0xFF) + rhs.degree

    def _bivariate_smart_prepare(self, lhs, rhs) -> None:
        if not self.is_functional_bivariate_pbs_possible(lhs, rhs):
            self._extract_both(lhs, rhs)

    # ----------------------------------------------------------------------- mul --
    def is_mul_possible(self, ct1, ct2) -> bool:
        return self.is_functional_bivariate_pbs_possible(ct1, ct2)

    def is_mul_small_carry_possible(self, lhs, rhs) -> bool:
        return self.is_add_possible(lhs, rhs) and self.is_sub_possible(lhs, rhs)

    def unchecked_mul_lsb(self, lhs, rhs):
        res = lhs.clone()
        self.unchecked_mul_lsb_assign(res, rhs)
        return res

    def unchecked_mul_lsb_assign(self, lhs, rhs) -> None:
        """engine/server_side/mul.rs:17-42."""
        modulus = rhs.degree + 1
        res_modulus = lhs.message_modulus
        self.unchecked_functional_bivariate_pbs_assign(lhs, rhs, lambda x: ((x // modulus) * (x % modulus)) % res_modulus)
        lhs.degree = lhs.message_modulus - 1

    def unchecked_mul_msb(self, lhs, rhs):
        res = lhs.clone()
        self.unchecked_mul_msb_assign(res, rhs)
        return res

    def unchecked_mul_msb_assign(self, lhs, rhs) -> None:
        """engine/server_side/mul.rs:56-83."""
        modulus = rhs.degree + 1
        deg = (lhs.degree * rhs.degree) // rhs.message_modulus
        res_modulus = self.message_modulus
        self.unchecked_functional_bivariate_pbs_assign(lhs, rhs, lambda x: ((x // modulus) * (x % modulus)) // res_modulus)
        lhs.degree = deg

    def unchecked_mul_lsb_small_carry(self, ct1, ct2):
        """engine/server_side/mul.rs:85-109: ((a+b)^2/4 - (a-b)^2/4) with two bootstraps in one
        launch; the second table is offset by the correcting term z of the subtraction."""
        left = self.unchecked_add(ct1, ct2)
        right, z = self.unchecked_sub_with_correcting_term(ct1, ct2)
        modulus = ct1.message_modulus
        acc_add = self.generate_accumulator(lambda x: ((x * x) // 4) % modulus)
        acc_sub = self.generate_accumulator(lambda x: ((((x - z) & M64) * ((x - z) & M64) & M64) // 4) % modulus)
        self.apply_lookup_table_many([(left, acc_add), (right, acc_sub)])
        return self.unchecked_sub(left, right)

    def unchecked_mul_lsb_small_carry_assign(self, ct1, ct2) -> None:
        res = self.unchecked_mul_lsb_small_carry(ct1, ct2)
        ct1.ct, ct1.degree = res.ct, res.degree

    def checked_mul_lsb(self, lhs, rhs):
        if not self.is_mul_possible(lhs, rhs):
            raise CheckError()
        return self.unchecked_mul_lsb(lhs, rhs)

    def checked_mul_lsb_assign(self, lhs, rhs) -> None:
        if not self.is_mul_possible(lhs, rhs):
            raise CheckError()
        self.unchecked_mul_lsb_assign(lhs, rhs)

    def checked_mul_msb(self, lhs, rhs):
        if not self.is_mul_possible(lhs, rhs):
            raise CheckError()
        return self.unchecked_mul_msb(lhs, rhs)

    def checked_mul_lsb_with_small_carry(self, lhs, rhs):
        """server_key/mul.rs:438-450."""
        if not self.is_mul_small_carry_possible(lhs, rhs):
            raise CheckError()
        res = self.unchecked_mul_lsb_small_carry(lhs, rhs)
        res.degree = lhs.degree * 2
        return res

    def smart_mul_lsb(self, lhs, rhs):
        res = lhs.clone()
        self.smart_mul_lsb_assign(res, rhs)
        return res

    def _smart_mul_lsb_reductions(self, lhs, rhs) -> list:
        """The operands that engine/server_side/mul.rs:121-156 reduces to their messages before
        the product (the decision depends on degrees only, never on data)."""
        if lhs.message_modulus > lhs.carry_modulus:
            return [] if self.is_mul_small_carry_possible(lhs, rhs) else [lhs, rhs]
        if self.is_mul_possible(lhs, rhs):
            return []
        full = rhs.carry_modulus * rhs.message_modulus - 1
        if self.message_modulus * (rhs.degree + 1) < full:
            return [lhs]
        if (self.message_modulus + 1) + (lhs.degree + 1) < full:
            return [rhs]
        return [lhs, rhs]

    def smart_mul_lsb_assign_many(self, pairs) -> None:
        """smart_mul_lsb_assign on several independent (lhs, rhs) pairs: the reductions of all
        pairs in one launch, then the products in another."""
        pairs = list(pairs)
        seen: set = set()
        with self.deferred():
            for lhs, rhs in pairs:
                for c in self._smart_mul_lsb_reductions(lhs, rhs):
                    if id(c) not in seen:   # an operand shared by several pairs is reduced once
                        seen.add(id(c))
                        self.message_extract_assign(c)
        with self.deferred():
            for lhs, rhs in pairs:
                if lhs.message_modulus > lhs.carry_modulus:
                    self.unchecked_mul_lsb_small_carry_assign(lhs, rhs)
                else:
                    self.unchecked_mul_lsb_assign(lhs, rhs)

    def smart_mul_lsb_assign(self, lhs, rhs) -> None:
        """engine/server_side/mul.rs:121-156."""
        self.smart_mul_lsb_assign_many([(lhs, rhs)])

    def smart_mul_msb(self, lhs, rhs):
        res = lhs.clone()
        self.smart_mul_msb_assign(res, rhs)
        return res

    def smart_mul_msb_assign(self, lhs, rhs) -> None:
        if not self.is_mul_possible(lhs, rhs):
            self._extract_both(lhs, rhs)
        self.unchecked_mul_msb_assign(lhs, rhs)

    # ------------------------------------------------------------------- bitwise --
    def _bitwise(self, name: str, lhs, rhs) -> None:
        modulus = rhs.degree + 1
        dl, dr = lhs.degree, rhs.degree
        if name == "and":
            f, deg = (lambda x: (x // modulus) & (x % modulus)), degree_after_bitand(dl, dr)
        elif name == "or":
            f, deg = (lambda x: (x // modulus) | (x % modulus)), degree_after_bitor(dl, dr)
        else:
            f, deg = (lambda x: (x // modulus) ^ (x % modulus)), degree_after_bitxor(dl, dr)
        self.unchecked_functional_bivariate_pbs_assign(lhs, rhs, f)
        lhs.degree = deg

    # ---------------------------------------------------------------- comparisons --
    def _compare(self, name: str, lhs, rhs) -> None:
        """engine/server_side/comp_op.rs: the left digit is reduced mod message_modulus inside
        the table, the right one is read from the low part."""
        modulus = rhs.degree + 1
        modulus_msg = lhs.message_modulus
        large_mod = modulus * modulus_msg
        op = {"greater": int.__gt__, "greater_or_equal": int.__ge__, "less": int.__lt__, "less_or_equal": int.__le__,
              "equal": int.__eq__}[name]
        self.unchecked_functional_bivariate_pbs_assign(
            lhs, rhs, lambda x: int(op(((x % large_mod) // modulus) % modulus_msg, x % modulus_msg)))
        lhs.degree = 1

    def _scalar_compare(self, name: str, ct, scalar: int):
        """engine/server_side/comp_op.rs:269-385."""
        scalar = int(scalar) & 0xFF
        modulus = ct.message_modulus
        f = {"equal": lambda x: int(x % modulus == scalar), "greater_or_equal": lambda x: int(x >= scalar),
             "less_or_equal": lambda x: int(x <= scalar), "greater": lambda x: int(x > scalar),
             "less": lambda x: int(x < scalar)}[name]
        res = ct.clone()
        self._pbs(res, f)
        res.degree = 1
        return res

    # ----------------------------------------------------------------------- div --
    def unchecked_div(self, lhs, rhs):
        res = lhs.clone()
        self.unchecked_div_assign(res, rhs)
        return res

    def unchecked_div_assign(self, lhs, rhs) -> None:
        """engine/server_side/div_mod.rs:25-37 (the degree is left as the packed one, as there)."""
        modulus = rhs.degree + 1
        self.unchecked_functional_bivariate_pbs_assign(lhs, rhs, lambda x: _division(x, modulus))

    def smart_div(self, lhs, rhs):
        res = lhs.clone()
        self.smart_div_assign(res, rhs)
        return res

    def smart_div_assign(self, lhs, rhs) -> None:
        """engine/server_side/div_mod.rs:50-70."""
        if not self.is_functional_bivariate_pbs_possible(lhs, rhs):
            if lhs.message_modulus + rhs.degree <= self.max_degree:
                self.message_extract_assign(lhs)
            elif rhs.message_modulus + (lhs.degree + 1) <= self.max_degree:
                self.message_extract_assign(rhs)
            else:
                self._extract_both(lhs, rhs)
        self.unchecked_div_assign(lhs, rhs)

    def unchecked_scalar_div(self, ct, scalar: int):
        res = ct.clone()
        self.unchecked_scalar_div_assign(res, scalar)
        return res

    def unchecked_scalar_div_assign(self, ct, scalar: int) -> None:
        """engine/server_side/div_mod.rs:84-96."""
        scalar = int(scalar) & 0xFF
        if scalar == 0:
            raise ZeroDivisionError("scalar division by zero")
        self._pbs(ct, lambda x: x // scalar)
        ct.degree //= scalar

    def unchecked_scalar_mod(self, ct, modulus: int):
        res = ct.clone()
        self.unchecked_scalar_mod_assign(res, modulus)
        return res

    def unchecked_scalar_mod_assign(self, ct, modulus: int) -> None:
        """engine/server_side/div_mod.rs:118-129."""
        modulus = int(modulus) & 0xFF
        if modulus == 0:
            raise ZeroDivisionError("scalar modulus of zero")
        self._pbs(ct, lambda x: x % modulus)
        ct.degree = modulus - 1


def _install_binary_families() -> None:
    """unchecked_/checked_/smart_ forms of the bitwise operations and comparisons
    (server_key/bitwise_op.rs, server_key/comp_op.rs): they differ only in the table."""

    def make(kind: str, name: str, impl: str):
        def unchecked(self, lhs, rhs):
            res = lhs.clone()
            getattr(self, impl)(name, res, rhs)
            return res

        def unchecked_assign(self, lhs, rhs) -> None:
            getattr(self, impl)(name, lhs, rhs)

        def checked(self, lhs, rhs):
            if not self.is_functional_bivariate_pbs_possible(lhs, rhs):
                raise CheckError()
            return unchecked(self, lhs, rhs)

        def checked_assign(self, lhs, rhs) -> None:
            if not self.is_functional_bivariate_pbs_possible(lhs, rhs):
                raise CheckError()
            unchecked_assign(self, lhs, rhs)

        def smart(self, lhs, rhs):
            self._bivariate_smart_prepare(lhs, rhs)
            return unchecked(self, lhs, rhs)

        def smart_assign(self, lhs, rhs) -> None:
            self._bivariate_smart_prepare(lhs, rhs)
            unchecked_assign(self, lhs, rhs)

        public = f"bit{name}" if kind == "bit" else name
        for prefix, fn in (("unchecked_", unchecked), ("checked_", checked), ("smart_", smart)):
            setattr(ShortintOps, prefix + public, fn)
        for prefix, fn in (("unchecked_", unchecked_assign), ("checked_", checked_assign), ("smart_", smart_assign)):
            setattr(ShortintOps, prefix + public + "_assign", fn)

    for n in ("and", "or", "xor"):
        make("bit", n, "_bitwise")
    for n in ("greater", "greater_or_equal", "less", "less_or_equal", "equal"):
        make("cmp", n, "_compare")
        setattr(ShortintOps, "smart_scalar_" + n, (lambda nm: lambda self, ct, scalar: self._scalar_compare(nm, ct, scalar))(n))


_install_binary_families()
