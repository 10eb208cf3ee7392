"""Host-side mirror of concrete-integer's radix layer, driven through the GPU shortint backend.

* ``ClientKey`` / ``RadixClientKey`` — ``concrete-integer/src/client_key/{mod,radix}.rs``
* ``RadixCiphertext``               — ``concrete-integer/src/ciphertext/mod.rs:8-15``: blocks in
  base ``message_modulus``, least significant first
* ``ServerKey``                     — ``concrete-integer/src/server_key/{mod,radix/*,radix_parallel/*}.rs``

The algorithms (carry propagation, schoolbook block multiplication, tree reduction, degree-driven
"smart" operations) are the reference's. What changes is how their bootstraps reach the
hardware: every block of a ``RadixCiphertext`` here may be a *batch* (the same block of many
integers, host or device resident), and bootstraps that are independent of each other — the two
extractions of a propagation step, the lsb/msb products of every block pair of a multiplication,
the blocks of a bitwise operation — are collected and issued as one launch
(``shortint.ServerKey.deferred``). The ``*_parallelized`` names of the reference are therefore
aliases: there is one schedule, and it is the batched one.
"""
from __future__ import annotations

import os

from dataclasses import dataclass
from typing import Iterable, List

import numpy as np

from . import shortint as _si
from .backend import CAMD_PBS_AUTO, Context
from .integer_crt import CrtCiphertext, CrtClientKey, CrtClientOps, CrtServerOps  # noqa: F401
from .shortint_ops import CheckError


@dataclass
class RadixCiphertext:
    """ciphertext/mod.rs:8-15."""
    blocks: List[_si.Ciphertext]

    def clone(self) -> "RadixCiphertext":
        return RadixCiphertext([b.clone() for b in self.blocks])

    @property
    def count(self) -> int:
        return self.blocks[0].count

    def to_device(self, device="cuda") -> "RadixCiphertext":
        return RadixCiphertext([b.to_device(device) for b in self.blocks])

    def to_host(self) -> "RadixCiphertext":
        return RadixCiphertext([b.to_host() for b in self.blocks])


class ClientKey(CrtClientOps):
    """client_key/mod.rs:25-93 (the CRT entry points, :279-431, come from ``CrtClientOps``)."""

    def __init__(self, parameters, seed: int | None = None):
        self.key = _si.ClientKey(parameters, seed)

    @classmethod
    def new(cls, parameters, seed: int | None = None) -> "ClientKey":
        return cls(parameters, seed)

    @classmethod
    def from_shortint(cls, key: _si.ClientKey) -> "ClientKey":
        """client_key/mod.rs:31-35 (``From<ShortintClientKey>``)."""
        self = cls.__new__(cls)
        self.key = key
        return self

    def parameters(self):
        return self.key.parameters

    def _digits(self, message: int, num_blocks: int) -> list[int]:
        # client_key/mod.rs:98-111: digit i = (message & (mask * power)) / power, all in u64
        modulus = self.key.parameters.message_modulus
        mask = modulus - 1
        message = int(message) & 0xFFFFFFFFFFFFFFFF
        return [(message & (mask * modulus ** i)) // modulus ** i for i in range(num_blocks)]

    def encrypt_radix(self, message: int, num_blocks: int) -> RadixCiphertext:
        """client_key/mod.rs:95-114."""
        return RadixCiphertext([self.key.encrypt(d) for d in self._digits(message, num_blocks)])

    def encrypt_radix_many(self, messages: Iterable[int], num_blocks: int, device=None) -> RadixCiphertext:
        """A batch of integers: block i of the result holds digit i of every message."""
        digits = [self._digits(m, num_blocks) for m in messages]
        blocks = [self.key.encrypt_many([d[i] for d in digits]) for i in range(num_blocks)]
        ct = RadixCiphertext(blocks)
        return ct.to_device(device) if device is not None else ct

    def encrypt_radix_without_padding(self, message: int, num_blocks: int) -> RadixCiphertext:
        """client_key/mod.rs:136-160."""
        return RadixCiphertext([self.key.encrypt_without_padding(d) for d in self._digits(message, num_blocks)])

    def encrypt_one_block(self, message: int) -> _si.Ciphertext:
        """client_key/mod.rs:184-186."""
        return self.key.encrypt(message)

    def decrypt_one_block(self, ct: _si.Ciphertext):
        """client_key/mod.rs:191-193."""
        return self.key.decrypt(ct)

    def _recompose(self, vals, modulus: int, num_blocks: int):
        whole = modulus ** num_blocks
        if np.ndim(vals[0]) == 0:
            return sum(int(v) * modulus ** i for i, v in enumerate(vals)) % (1 << 64) % whole
        total = np.zeros(len(vals[0]), dtype=object)
        for i, v in enumerate(vals):
            total = total + v.astype(object) * (modulus ** i)
        return np.array([int(t) % (1 << 64) % whole for t in total], dtype=np.uint64)

    def decrypt_radix_without_padding(self, ctxt: RadixCiphertext):
        """client_key/mod.rs:256-277."""
        vals = [self.key.decrypt_message_and_carry_without_padding(b) for b in ctxt.blocks]
        return self._recompose(vals, self.key.parameters.message_modulus, len(ctxt.blocks))

    def decrypt_radix(self, ctxt: RadixCiphertext):
        """client_key/mod.rs:215-234: blocks are decrypted with their carries and recombined."""
        modulus = self.key.parameters.message_modulus
        whole = modulus ** len(ctxt.blocks)
        vals = [self.key.decrypt_message_and_carry(b) for b in ctxt.blocks]
        if np.ndim(vals[0]) == 0:
            return sum(int(v) * modulus ** i for i, v in enumerate(vals)) % whole
        total = np.zeros(len(vals[0]), dtype=object)
        for i, v in enumerate(vals):
            total = total + v.astype(object) * (modulus ** i)
        return np.array([int(t) % whole for t in total], dtype=np.uint64)


class RadixClientKey(ClientKey):
    """client_key/radix.rs: a client key that remembers the number of blocks."""

    def __init__(self, parameters, num_blocks: int, seed: int | None = None):
        super().__init__(parameters, seed)
        self.num_blocks = int(num_blocks)

    def encrypt(self, message: int) -> RadixCiphertext:
        return self.encrypt_radix(message, self.num_blocks)

    def encrypt_many(self, messages: Iterable[int], device=None) -> RadixCiphertext:
        return self.encrypt_radix_many(messages, self.num_blocks, device)

    def decrypt(self, ctxt: RadixCiphertext):
        return self.decrypt_radix(ctxt)


class ServerKey(CrtServerOps):
    """server_key/mod.rs:24-68: a shortint server key whose max_degree leaves room for one carry.
    The radix operations are below; the CRT ones come from ``CrtServerOps``."""

    def __init__(self, key: _si.ServerKey, reference_schedule: bool | None = None):
        p = key.parameters
        key.max_degree = (p.message_modulus - 1) * p.carry_modulus - 1
        self.key = key
        # True: a propagation step bootstraps exactly what the reference's does (two bootstraps
        # per block, every block, radix_parallel/mod.rs:44-56). False (default): the bootstraps
        # that cannot change any decryption are not launched (see _propagate_many). The
        # environment variable CAMD_REFERENCE_SCHEDULE=1 makes True the default.
        if reference_schedule is None:
            reference_schedule = os.environ.get("CAMD_REFERENCE_SCHEDULE", "0") not in ("", "0")
        self.reference_schedule = bool(reference_schedule)

    @classmethod
    def new(cls, cks: ClientKey, ctx: Context | None = None, variant: int = CAMD_PBS_AUTO,
            reference_schedule: bool | None = None) -> "ServerKey":
        return cls(_si.ServerKey.new(cks.key, ctx=ctx, variant=variant), reference_schedule=reference_schedule)

    @classmethod
    def from_shortint(cls, cks: ClientKey, key: _si.ServerKey) -> "ServerKey":
        return cls(key)

    def gather(self, thunks):
        """Independent operations side by side, their bootstraps launched together round after
        round (``shortint.ServerKey.gather``): ``sks.gather([lambda: sks.smart_mul(a, b), lambda:
        sks.smart_add(c, d)])`` issues what rayon workers calling the reference concurrently would."""
        return self.key.gather(thunks)

    # ------------------------------------------------------------- radix/mod.rs --
    def create_trivial_zero_radix(self, num_blocks: int, like: RadixCiphertext | None = None) -> RadixCiphertext:
        ref = like.blocks[0] if like is not None else None
        return RadixCiphertext([self.key.create_trivial(0, like=ref) for _ in range(num_blocks)])

    def _propagate_many(self, pairs) -> None:
        """One propagation step on several (ciphertext, index) pairs whose results do not depend
        on each other: all carry and message extractions in one launch, then the carry adds."""
        # A block whose degree is below the message modulus has no carry and is its own message:
        # the step changes nothing in it (the reference bootstraps it all the same, which only
        # refreshes its noise; every degree check that follows treats it as it treats a fresh
        # block). Skipping it is what makes an operation on fresh inputs start without a
        # propagation chain of launches.
        ref = self.reference_schedule
        if not ref:
            pairs = [(c, i) for c, i in pairs if c.blocks[i].degree >= self.key.message_modulus]
        carries = []
        with self.key.deferred():
            for ctxt, index in pairs:
                # radix/mod.rs:58-68 extracts the carry of the last block too and then drops it;
                # that bootstrap has no effect on any output, so it is only launched under the
                # reference's schedule.
                last = index >= len(ctxt.blocks) - 1
                carries.append(None if (last and not ref) else self.key.carry_extract(ctxt.blocks[index]))
                self.key.message_extract_assign(ctxt.blocks[index])
        for (ctxt, index), carry in zip(pairs, carries):
            if carry is not None and index < len(ctxt.blocks) - 1:
                self.key.unchecked_add_assign(ctxt.blocks[index + 1], carry)

    def propagate(self, ctxt: RadixCiphertext, index: int) -> None:
        """radix/mod.rs:58-68 (radix_parallel/mod.rs:17-31 runs the two extractions side by side;
        here they are one launch)."""
        self._propagate_many([(ctxt, index)])

    def full_propagate(self, ctxt: RadixCiphertext) -> None:
        """radix/mod.rs:95-101: the carry chain is inherently sequential over the blocks."""
        for i in range(len(ctxt.blocks)):
            self.propagate(ctxt, i)

    def _full_propagate_many(self, ctxts) -> None:
        """full_propagate of independent integers, block step by block step (rayon::join in
        radix_parallel/add.rs:14-19, mul.rs:163-166)."""
        ctxts = list(ctxts)
        if not ctxts:
            return
        for i in range(max(len(c.blocks) for c in ctxts)):
            self._propagate_many([(c, i) for c in ctxts if i < len(c.blocks)])

    propagate_parallelized = propagate
    full_propagate_parallelized = full_propagate

    # ------------------------------------------------------------------- add.rs --
    def unchecked_add(self, lhs: RadixCiphertext, rhs: RadixCiphertext) -> RadixCiphertext:
        res = lhs.clone()
        self.unchecked_add_assign(res, rhs)
        return res

    def unchecked_add_assign(self, lhs: RadixCiphertext, rhs: RadixCiphertext) -> None:
        for a, b in zip(lhs.blocks, rhs.blocks):
            self.key.unchecked_add_assign(a, b)

    def is_add_possible(self, lhs: RadixCiphertext, rhs: RadixCiphertext) -> bool:
        return all(self.key.is_add_possible(a, b) for a, b in zip(lhs.blocks, rhs.blocks))

    def checked_add(self, lhs, rhs) -> RadixCiphertext:
        if not self.is_add_possible(lhs, rhs):
            raise CheckError()
        return self.unchecked_add(lhs, rhs)

    def checked_add_assign(self, lhs, rhs) -> None:
        if not self.is_add_possible(lhs, rhs):
            raise CheckError()
        self.unchecked_add_assign(lhs, rhs)

    def smart_add(self, lhs: RadixCiphertext, rhs: RadixCiphertext) -> RadixCiphertext:
        """radix/add.rs:205-217; both operands are propagated together."""
        if not self.is_add_possible(lhs, rhs):
            self._full_propagate_many([lhs, rhs])
        return self.unchecked_add(lhs, rhs)

    def smart_add_assign(self, lhs: RadixCiphertext, rhs: RadixCiphertext) -> None:
        if not self.is_add_possible(lhs, rhs):
            self._full_propagate_many([lhs, rhs])
        self.unchecked_add_assign(lhs, rhs)

    smart_add_parallelized = smart_add
    smart_add_assign_parallelized = smart_add_assign

    def smart_binary_op_seq_parallelized(self, ct_seq, op):
        """radix_parallel/add.rs:55-140: pairwise tree reduction (an odd element stays in
        front). Each level's operations are independent; their propagations share launches when
        ``op`` is one of this class's smart operations."""
        seq = list(ct_seq)
        if not seq:
            return None
        while len(seq) > 1:
            prefix = len(seq) % 2
            pairs = [(seq[i], seq[i + 1]) for i in range(prefix, len(seq), 2)]
            if op in (ServerKey.smart_add, ServerKey.smart_add_parallelized):
                need = [c for a, b in pairs if not self.is_add_possible(a, b) for c in (a, b)]
                self._full_propagate_many(need)
                results = [self.unchecked_add(a, b) for a, b in pairs]
            else:
                # independent operations of one level: their bootstraps share launches
                results = self.gather([lambda a=a, b=b: op(self, a, b) for a, b in pairs])
            seq = seq[:prefix] + results
        return seq[0]

    # ------------------------------------------------------------ neg.rs, sub.rs --
    def unchecked_neg(self, ctxt: RadixCiphertext) -> RadixCiphertext:
        res = ctxt.clone()
        self.unchecked_neg_assign(res)
        return res

    def unchecked_neg_assign(self, ctxt: RadixCiphertext) -> None:
        """radix/neg.rs:56-72: block-wise negation; the correcting term of block i is paid back
        as a scalar added to block i+1."""
        n = len(ctxt.blocks)
        for i in range(n):
            z = self.key.unchecked_neg_assign_with_z(ctxt.blocks[i])
            z_b = z // self.key.message_modulus
            if i < n - 1:
                self.key.unchecked_scalar_add_assign(ctxt.blocks[i + 1], z_b)

    def is_neg_possible(self, ctxt: RadixCiphertext) -> bool:
        return all(self.key.is_neg_possible(b) for b in ctxt.blocks)

    def checked_neg(self, ctxt):
        if not self.is_neg_possible(ctxt):
            raise CheckError()
        return self.unchecked_neg(ctxt)

    def smart_neg(self, ctxt: RadixCiphertext) -> RadixCiphertext:
        if not self.is_neg_possible(ctxt):
            self.full_propagate(ctxt)
        return self.unchecked_neg(ctxt)

    smart_neg_parallelized = smart_neg

    def unchecked_sub(self, lhs: RadixCiphertext, rhs: RadixCiphertext) -> RadixCiphertext:
        res = lhs.clone()
        self.unchecked_sub_assign(res, rhs)
        return res

    def unchecked_sub_assign(self, lhs: RadixCiphertext, rhs: RadixCiphertext) -> None:
        """radix/sub.rs:57-64."""
        self.unchecked_add_assign(lhs, self.unchecked_neg(rhs))

    def is_sub_possible(self, lhs: RadixCiphertext, rhs: RadixCiphertext) -> bool:
        return all(self.key.is_sub_possible(a, b) for a, b in zip(lhs.blocks, rhs.blocks))

    def checked_sub(self, lhs, rhs):
        if not self.is_sub_possible(lhs, rhs):
            raise CheckError()
        return self.unchecked_sub(lhs, rhs)

    def checked_sub_assign(self, lhs, rhs) -> None:
        if not self.is_sub_possible(lhs, rhs):
            raise CheckError()
        self.unchecked_sub_assign(lhs, rhs)

    def smart_sub_assign(self, lhs: RadixCiphertext, rhs: RadixCiphertext) -> None:
        """radix/sub.rs:233-248."""
        if not self.is_neg_possible(rhs):
            self.full_propagate(rhs)
        if not self.is_sub_possible(lhs, rhs):
            self._full_propagate_many([lhs, rhs])
        self.unchecked_sub_assign(lhs, rhs)

    def smart_sub(self, lhs: RadixCiphertext, rhs: RadixCiphertext) -> RadixCiphertext:
        if not self.is_neg_possible(rhs):
            self.full_propagate(rhs)
        if not self.is_sub_possible(lhs, rhs):
            self._full_propagate_many([lhs, rhs])
        return self.unchecked_sub(lhs, rhs)

    smart_sub_parallelized = smart_sub
    smart_sub_assign_parallelized = smart_sub_assign

    # ------------------------------------------------ scalar_add.rs, scalar_sub.rs --
    def _scalar_digits(self, scalar: int, num_blocks: int) -> list[int]:
        modulus = self.key.message_modulus
        mask = modulus - 1
        scalar = int(scalar) & 0xFFFFFFFFFFFFFFFF
        return [((scalar & (mask * modulus ** i)) // modulus ** i) & 0xFF for i in range(num_blocks)]

    def unchecked_scalar_add(self, ct: RadixCiphertext, scalar: int) -> RadixCiphertext:
        res = ct.clone()
        self.unchecked_scalar_add_assign(res, scalar)
        return res

    def unchecked_scalar_add_assign(self, ct: RadixCiphertext, scalar: int) -> None:
        """radix/scalar_add.rs:52-66: the scalar is added digit by digit."""
        for b, d in zip(ct.blocks, self._scalar_digits(scalar, len(ct.blocks))):
            self.key.unchecked_scalar_add_assign(b, d)

    def is_scalar_add_possible(self, ct: RadixCiphertext, scalar: int) -> bool:
        return all(self.key.is_scalar_add_possible(b, d)
                   for b, d in zip(ct.blocks, self._scalar_digits(scalar, len(ct.blocks))))

    def checked_scalar_add(self, ct, scalar: int):
        if not self.is_scalar_add_possible(ct, scalar):
            raise CheckError()
        return self.unchecked_scalar_add(ct, scalar)

    def smart_scalar_add(self, ct: RadixCiphertext, scalar: int) -> RadixCiphertext:
        if not self.is_scalar_add_possible(ct, scalar):
            self.full_propagate(ct)
        return self.unchecked_scalar_add(ct, scalar)

    def smart_scalar_add_assign(self, ct: RadixCiphertext, scalar: int) -> None:
        if not self.is_scalar_add_possible(ct, scalar):
            self.full_propagate(ct)
        self.unchecked_scalar_add_assign(ct, scalar)

    smart_scalar_add_parallelized = smart_scalar_add
    smart_scalar_add_assign_parallelized = smart_scalar_add_assign

    def _neg_scalar(self, scalar: int, num_blocks: int) -> int:
        modulus = self.key.message_modulus ** num_blocks
        return ((-int(scalar)) & 0xFFFFFFFFFFFFFFFF) % modulus

    def unchecked_scalar_sub(self, ct: RadixCiphertext, scalar: int) -> RadixCiphertext:
        res = ct.clone()
        self.unchecked_scalar_sub_assign(res, scalar)
        return res

    def unchecked_scalar_sub_assign(self, ct: RadixCiphertext, scalar: int) -> None:
        """radix/scalar_sub.rs:49-66: adds the two's complement of the scalar."""
        self.unchecked_scalar_add_assign(ct, self._neg_scalar(scalar, len(ct.blocks)))

    def is_scalar_sub_possible(self, ct: RadixCiphertext, scalar: int) -> bool:
        return self.is_scalar_add_possible(ct, self._neg_scalar(scalar, len(ct.blocks)))

    def checked_scalar_sub(self, ct, scalar: int):
        if not self.is_scalar_sub_possible(ct, scalar):
            raise CheckError()
        return self.unchecked_scalar_sub(ct, scalar)

    def smart_scalar_sub(self, ct: RadixCiphertext, scalar: int) -> RadixCiphertext:
        if not self.is_scalar_sub_possible(ct, scalar):
            self.full_propagate(ct)
        return self.unchecked_scalar_sub(ct, scalar)

    def smart_scalar_sub_assign(self, ct: RadixCiphertext, scalar: int) -> None:
        if not self.is_scalar_sub_possible(ct, scalar):
            self.full_propagate(ct)
        self.unchecked_scalar_sub_assign(ct, scalar)

    smart_scalar_sub_parallelized = smart_scalar_sub
    smart_scalar_sub_assign_parallelized = smart_scalar_sub_assign

    # --------------------------------------------------------------- scalar_mul.rs --
    def unchecked_small_scalar_mul(self, ctxt: RadixCiphertext, scalar: int) -> RadixCiphertext:
        res = ctxt.clone()
        self.unchecked_small_scalar_mul_assign(res, scalar)
        return res

    def unchecked_small_scalar_mul_assign(self, ctxt: RadixCiphertext, scalar: int) -> None:
        for b in ctxt.blocks:
            self.key.unchecked_scalar_mul_assign(b, int(scalar) & 0xFF)

    def is_small_scalar_mul_possible(self, ctxt: RadixCiphertext, scalar: int) -> bool:
        return all(self.key.is_scalar_mul_possible(b, int(scalar) & 0xFF) for b in ctxt.blocks)

    def checked_small_scalar_mul(self, ctxt, scalar: int):
        if not self.is_small_scalar_mul_possible(ctxt, scalar):
            raise CheckError()
        return self.unchecked_small_scalar_mul(ctxt, scalar)

    def smart_small_scalar_mul(self, ctxt: RadixCiphertext, scalar: int) -> RadixCiphertext:
        if not self.is_small_scalar_mul_possible(ctxt, scalar):
            self.full_propagate(ctxt)
        return self.unchecked_small_scalar_mul(ctxt, scalar)

    def smart_small_scalar_mul_assign(self, ctxt: RadixCiphertext, scalar: int) -> None:
        if not self.is_small_scalar_mul_possible(ctxt, scalar):
            self.full_propagate(ctxt)
        self.unchecked_small_scalar_mul_assign(ctxt, scalar)

    def blockshift(self, ctxt: RadixCiphertext, shift: int) -> RadixCiphertext:
        """radix/scalar_mul.rs:280-292: multiply by message_modulus^shift (high blocks fall off)."""
        n = len(ctxt.blocks)
        if shift < 0:
            raise ValueError("blockshift: negative shift")
        shift = min(int(shift), n)   # shifting by the whole width or more leaves zeros only
        zero = self.key.create_trivial(0, like=ctxt.blocks[0])
        return RadixCiphertext([zero.clone() for _ in range(shift)] + [b.clone() for b in ctxt.blocks[:n - shift]])

    def blockshift_right(self, ctxt: RadixCiphertext, shift: int) -> RadixCiphertext:
        """radix/shift.rs:33-46."""
        n = len(ctxt.blocks)
        if shift < 0:
            raise ValueError("blockshift_right: negative shift")
        shift = min(int(shift), n)
        zero = self.key.create_trivial(0, like=ctxt.blocks[0])
        return RadixCiphertext([b.clone() for b in ctxt.blocks[shift:]] + [zero.clone() for _ in range(shift)])

    def smart_scalar_mul(self, ctxt: RadixCiphertext, scalar: int) -> RadixCiphertext:
        """radix/scalar_mul.rs:316-358: one small-scalar product per distinct non-trivial digit of
        the scalar, shifted and summed."""
        n = len(ctxt.blocks)
        self.full_propagate(ctxt)
        cache: dict[int, RadixCiphertext] = {}
        result = self.create_trivial_zero_radix(n, like=ctxt)
        for i, u_i in enumerate(self._scalar_digits(scalar, n)):
            if u_i == 0:
                continue
            if u_i == 1:
                tmp = self.blockshift(ctxt, i)
            else:
                if u_i not in cache:
                    cache[u_i] = self.smart_small_scalar_mul(ctxt, u_i)
                tmp = self.blockshift(cache[u_i], i)
            result = self.smart_add(result, tmp)
        return result

    def smart_scalar_mul_assign(self, ctxt: RadixCiphertext, scalar: int) -> None:
        ctxt.blocks = self.smart_scalar_mul(ctxt, scalar).blocks

    smart_scalar_mul_parallelized = smart_scalar_mul

    # ----------------------------------------------------------------- shift.rs --
    def unchecked_scalar_left_shift(self, ct: RadixCiphertext, shift: int) -> RadixCiphertext:
        res = ct.clone()
        self.unchecked_scalar_left_shift_assign(res, shift)
        return res

    def unchecked_scalar_left_shift_assign(self, ct: RadixCiphertext, shift: int) -> None:
        """radix/shift.rs:196-199: multiplication by 2^shift."""
        self.smart_scalar_mul_assign(ct, 1 << shift)

    def _partial_propagate(self, ctxt: RadixCiphertext) -> None:
        for i in range(len(ctxt.blocks) - 1):
            self.propagate(ctxt, i)

    def unchecked_scalar_right_shift(self, ct: RadixCiphertext, shift: int) -> RadixCiphertext:
        res = ct.clone()
        self.unchecked_scalar_right_shift_assign(res, shift)
        return res

    def unchecked_scalar_right_shift_assign(self, ct: RadixCiphertext, shift: int) -> None:
        """radix/shift.rs:106-136: a right shift by u = p*q + r bits (p bits per block) is a left
        shift by p - r bits followed by a shift of q + 1 whole blocks."""
        nb_bits = self.key.message_modulus.bit_length() - 1
        quotient, remainder = divmod(int(shift), nb_bits)
        if remainder == 0:
            self.full_propagate(ct)
            ct.blocks = self.blockshift_right(ct, quotient).blocks
        else:
            self.unchecked_scalar_left_shift_assign(ct, nb_bits - remainder)
            self._partial_propagate(ct)
            ct.blocks = self.blockshift_right(ct, 1).blocks
            self.propagate(ct, len(ct.blocks) - 2)
            ct.blocks = self.blockshift_right(ct, quotient).blocks

    unchecked_scalar_left_shift_parallelized = unchecked_scalar_left_shift
    unchecked_scalar_right_shift_parallelized = unchecked_scalar_right_shift

    # ------------------------------------------------------------------- mul.rs --
    def _block_mul_parts(self, ct1: RadixCiphertext, ct2_block: _si.Ciphertext, index: int):
        """radix/mul.rs:54-76: low and high halves of ct1 * (one block of ct2), the bootstraps
        left to the caller's ``deferred`` scope."""
        shifted = self.blockshift(ct1, index)
        lsb, msb = shifted.clone(), shifted
        n = len(lsb.blocks)
        for b in lsb.blocks[index:]:
            self.key.unchecked_mul_lsb_assign(b, ct2_block)
        for b in msb.blocks[index:n - 1]:
            self.key.unchecked_mul_msb_assign(b, ct2_block)
        return lsb, msb

    def unchecked_block_mul(self, ct1: RadixCiphertext, ct2_block: _si.Ciphertext, index: int) -> RadixCiphertext:
        with self.key.deferred():
            lsb, msb = self._block_mul_parts(ct1, ct2_block, index)
        return self.unchecked_add(lsb, self.blockshift(msb, 1))

    def unchecked_block_mul_assign(self, ct1: RadixCiphertext, ct2_block: _si.Ciphertext, index: int) -> None:
        ct1.blocks = self.unchecked_block_mul(ct1, ct2_block, index).blocks

    def smart_block_mul(self, ct1: RadixCiphertext, ct2_block: _si.Ciphertext, index: int) -> RadixCiphertext:
        """radix/mul.rs:113-139."""
        self.full_propagate(ct1)
        with self.key.deferred():
            lsb, msb = self._block_mul_parts(ct1, ct2_block, index)
        return self.smart_add(lsb, self.blockshift(msb, 1))

    def smart_block_mul_assign(self, ct1: RadixCiphertext, ct2_block: _si.Ciphertext, index: int) -> None:
        ct1.blocks = self.smart_block_mul(ct1, ct2_block, index).blocks

    def _mul_terms(self, ct1: RadixCiphertext, ct2: RadixCiphertext) -> list[RadixCiphertext]:
        """Every block product of the schoolbook multiplication, in ONE launch
        (radix_parallel/mul.rs:111-125 spreads them over rayon workers)."""
        with self.key.deferred():
            parts = [self._block_mul_parts(ct1, b, i) for i, b in enumerate(ct2.blocks)]
        return [self.unchecked_add(lsb, self.blockshift(msb, 1)) for lsb, msb in parts]

    def unchecked_mul(self, ct1: RadixCiphertext, ct2: RadixCiphertext) -> RadixCiphertext:
        """radix/mul.rs:186-200."""
        result = self.create_trivial_zero_radix(len(ct1.blocks), like=ct1)
        for term in self._mul_terms(ct1, ct2):
            self.unchecked_add_assign(result, term)
        return result

    def unchecked_mul_assign(self, ct1: RadixCiphertext, ct2: RadixCiphertext) -> None:
        ct1.blocks = self.unchecked_mul(ct1, ct2).blocks

    def smart_mul(self, ct1: RadixCiphertext, ct2: RadixCiphertext) -> RadixCiphertext:
        """radix_parallel/mul.rs:157-181 (radix/mul.rs:253-268 sums the terms one after the other;
        the tree of the parallel form is used for both names)."""
        self._full_propagate_many([ct1, ct2])
        terms = self._mul_terms(ct1, ct2)
        res = self.smart_binary_op_seq_parallelized(terms, ServerKey.smart_add)
        return res if res is not None else self.create_trivial_zero_radix(len(ct1.blocks), like=ct1)

    def smart_mul_assign(self, ct1: RadixCiphertext, ct2: RadixCiphertext) -> None:
        ct1.blocks = self.smart_mul(ct1, ct2).blocks

    unchecked_block_mul_parallelized = unchecked_block_mul
    smart_block_mul_parallelized = smart_block_mul
    unchecked_mul_parallelized = unchecked_mul
    smart_mul_parallelized = smart_mul
    smart_mul_assign_parallelized = smart_mul_assign

    # ------------------------------------------------------------- bitwise_op.rs --
    def is_functional_bivariate_pbs_possible(self, lhs: RadixCiphertext, rhs: RadixCiphertext) -> bool:
        return all(self.key.is_functional_bivariate_pbs_possible(a, b) for a, b in zip(lhs.blocks, rhs.blocks))

    def _blockwise(self, name: str, lhs: RadixCiphertext, rhs: RadixCiphertext) -> None:
        op = getattr(self.key, f"unchecked_bit{name}_assign")
        with self.key.deferred():
            for a, b in zip(lhs.blocks, rhs.blocks):
                op(a, b)

    def _smart_prepare(self, lhs: RadixCiphertext, rhs: RadixCiphertext) -> None:
        if not self.is_functional_bivariate_pbs_possible(lhs, rhs):
            self._full_propagate_many([lhs, rhs])


def _install_remaining_forms() -> None:
    """The assign / checked forms and ``*_parallelized`` names that differ from the methods above
    only in where the result goes (radix/{neg,scalar_add,scalar_sub,scalar_mul,shift,mul}.rs and
    their radix_parallel counterparts)."""
    def checked_neg_assign(self, ctxt) -> None:
        if not self.is_neg_possible(ctxt):
            raise CheckError()
        self.unchecked_neg_assign(ctxt)

    def checked_scalar_add_assign(self, ct, scalar) -> None:
        if not self.is_scalar_add_possible(ct, scalar):
            raise CheckError()
        self.unchecked_scalar_add_assign(ct, scalar)

    def checked_scalar_sub_assign(self, ct, scalar) -> None:
        if not self.is_scalar_sub_possible(ct, scalar):
            raise CheckError()
        self.unchecked_scalar_sub_assign(ct, scalar)

    def checked_small_scalar_mul_assign(self, ct, scalar) -> None:
        if not self.is_small_scalar_mul_possible(ct, scalar):
            raise CheckError()
        self.unchecked_small_scalar_mul_assign(ct, scalar)

    def blockshift_right_assign(self, ctxt, shift) -> None:
        """radix/shift.rs:48-50."""
        ctxt.blocks = self.blockshift_right(ctxt, shift).blocks

    for f in (checked_neg_assign, checked_scalar_add_assign, checked_scalar_sub_assign, checked_small_scalar_mul_assign,
              blockshift_right_assign):
        setattr(ServerKey, f.__name__, f)
    for name in ("checked_small_scalar_mul", "checked_small_scalar_mul_assign", "smart_small_scalar_mul",
                 "smart_small_scalar_mul_assign", "unchecked_small_scalar_mul", "unchecked_small_scalar_mul_assign",
                 "smart_block_mul_assign", "smart_scalar_mul_assign", "unchecked_block_mul_assign", "unchecked_mul_assign",
                 "unchecked_scalar_left_shift_assign", "unchecked_scalar_right_shift_assign"):
        setattr(ServerKey, name + "_parallelized", getattr(ServerKey, name))


def _install_bitwise() -> None:
    """unchecked_/checked_/smart_ bitand, bitor, bitxor (radix/bitwise_op.rs,
    radix_parallel/bitwise_op.rs): block-wise, all blocks in one launch."""
    for name in ("and", "or", "xor"):
        def unchecked_assign(self, lhs, rhs, _n=name) -> None:
            self._blockwise(_n, lhs, rhs)

        def unchecked(self, lhs, rhs, _n=name):
            res = lhs.clone()
            self._blockwise(_n, res, rhs)
            return res

        def checked(self, lhs, rhs, _n=name):
            if not self.is_functional_bivariate_pbs_possible(lhs, rhs):
                raise CheckError()
            res = lhs.clone()
            self._blockwise(_n, res, rhs)
            return res

        def smart(self, lhs, rhs, _n=name):
            self._smart_prepare(lhs, rhs)
            res = lhs.clone()
            self._blockwise(_n, res, rhs)
            return res

        def smart_assign(self, lhs, rhs, _n=name) -> None:
            self._smart_prepare(lhs, rhs)
            self._blockwise(_n, lhs, rhs)

        setattr(ServerKey, f"unchecked_bit{name}", unchecked)
        setattr(ServerKey, f"unchecked_bit{name}_assign", unchecked_assign)
        def checked_assign(self, lhs, rhs, _n=name) -> None:
            if not self.is_functional_bivariate_pbs_possible(lhs, rhs):
                raise CheckError()
            self._blockwise(_n, lhs, rhs)

        setattr(ServerKey, f"checked_bit{name}", checked)
        setattr(ServerKey, f"checked_bit{name}_assign", checked_assign)
        setattr(ServerKey, f"smart_bit{name}", smart)
        setattr(ServerKey, f"smart_bit{name}_assign", smart_assign)
        setattr(ServerKey, f"smart_bit{name}_parallelized", smart)
        setattr(ServerKey, f"smart_bit{name}_assign_parallelized", smart_assign)


_install_bitwise()
_install_remaining_forms()
