"""Host-side mirror of concrete-boolean for the gate bootstrap path (32-bit torus).

* ``ClientKey``      — ``concrete-boolean/src/engine/mod.rs:84-147`` (keys, encrypt, decrypt)
* ``ServerKey``      — ``concrete-boolean/src/server_key/mod.rs:22-160`` (gates) over a
  ``Bootstrapper`` whose three entry points (``bootstrap``, ``keyswitch``,
  ``bootstrap_keyswitch``; ``engine/bootstrapping/mod.rs:22-48``) run on the GPU
* ``Ciphertext``     — ``Encrypted(lwe)`` or ``Trivial(bool)`` as in ``ciphertext/mod.rs``

Gate linear combinations follow ``engine/mod.rs:217-581`` word for word (wrapping u32 arithmetic);
the order is bootstrap then keyswitch. Batched gates take arrays of ciphertexts.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Union

import numpy as np

from . import client as _client
from .backend import CAMD_PBS_AUTO, Context, DeviceServerKey

PLAINTEXT_LOG_SCALING_FACTOR = 3                       # lib.rs:76
PLAINTEXT_TRUE = np.uint32(1 << (32 - 3))              # 1/8,  lib.rs:79
PLAINTEXT_FALSE = np.uint32(7 << (32 - 3))             # -1/8, lib.rs:82


@dataclass
class Ciphertext:
    """Either an LWE ciphertext under the small key (n+1 uint32 words) or a trivial boolean."""
    lwe: Optional[np.ndarray] = None
    trivial: Optional[bool] = None

    @property
    def is_trivial(self) -> bool:
        return self.lwe is None

    def clone(self) -> "Ciphertext":
        return Ciphertext(None if self.lwe is None else self.lwe.copy(), self.trivial)


class ClientKey:
    def __init__(self, parameters, seed: int | None = None):
        self.parameters = parameters
        self._rng = _client.Rng(seed)
        self.lwe_secret_key = _client.gen_binary_key(self._rng, 32, parameters.lwe_dimension)
        self.glwe_secret_key = _client.gen_binary_key(self._rng, 32, parameters.glwe_dimension * parameters.polynomial_size)

    @classmethod
    def new(cls, parameters, seed: int | None = None) -> "ClientKey":
        return cls(parameters, seed)

    def encrypt(self, message: bool) -> Ciphertext:
        pt = PLAINTEXT_TRUE if message else PLAINTEXT_FALSE
        ct = _client.lwe_encrypt(self._rng, 32, self.lwe_secret_key, [pt], self.parameters.lwe_modular_std_dev)[0]
        return Ciphertext(lwe=ct)

    def encrypt_batch(self, messages) -> np.ndarray:
        pts = [PLAINTEXT_TRUE if m else PLAINTEXT_FALSE for m in messages]
        return _client.lwe_encrypt(self._rng, 32, self.lwe_secret_key, pts, self.parameters.lwe_modular_std_dev)

    def decrypt(self, ct: Ciphertext) -> bool:
        if ct.is_trivial:
            return bool(ct.trivial)
        return int(_client.lwe_phase(32, self.lwe_secret_key, ct.lwe)[0]) < (1 << 31)   # engine/mod.rs:144

    def decrypt_batch(self, cts: np.ndarray) -> np.ndarray:
        return _client.lwe_phase(32, self.lwe_secret_key, cts) < np.uint32(1 << 31)


class GpuBootstrapper:
    """The reference's ``Bootstrapper`` trait on the GPU (one look-up table: body = 1/8)."""

    def __init__(self, device_key: DeviceServerKey, parameters):
        self.key = device_key
        self.parameters = parameters
        self.key.upload_lut(_client.boolean_accumulator(parameters))  # cpu.rs:49-53

    def bootstrap(self, lwe: np.ndarray) -> np.ndarray:
        """n+1 words in, kN+1 words out (cpu.rs:162-194)."""
        return self.key.pbs(np.atleast_2d(lwe))

    def keyswitch(self, lwe: np.ndarray) -> np.ndarray:
        """kN+1 words in, n+1 words out (cpu.rs:196-213)."""
        return self.key.ks(np.atleast_2d(lwe))

    def bootstrap_keyswitch(self, lwe: np.ndarray) -> np.ndarray:
        """n+1 words in and out (cpu.rs:215-259)."""
        return self.key.pbs_ks(np.atleast_2d(lwe))


BoolOrCt = Union[Ciphertext, bool]


class ServerKey:
    def __init__(self, bootstrapper: GpuBootstrapper):
        self.bootstrapper = bootstrapper
        self.parameters = bootstrapper.parameters

    @classmethod
    def new(cls, cks: ClientKey, ctx: Context | None = None, variant: int = CAMD_PBS_AUTO) -> "ServerKey":
        """cpu.rs:100-142: bootstrap key under (lwe, glwe) keys, keyswitch key from the flattened
        GLWE key back to the LWE key."""
        p = cks.parameters
        bsk = _client.gen_bootstrap_key(cks._rng, p, 32, cks.lwe_secret_key, cks.glwe_secret_key)
        ksk = _client.gen_keyswitch_key(cks._rng, p, 32, cks.glwe_secret_key, cks.lwe_secret_key)
        ctx = ctx or Context(0)
        dk = DeviceServerKey.from_standard_keys(ctx, p, bsk, ksk, variant=variant, torus_bits=32)
        sks = cls(GpuBootstrapper(dk, p))
        # kept on the host like the reference's CpuBootstrapKey (cpu.rs:78-82): the serialized
        # form of a boolean server key is the standard-domain key (cpu.rs:262-290)
        sks.standard_bootstrap_key, sks.key_switching_key = bsk, ksk
        return sks

    def trivial_encrypt(self, message: bool) -> Ciphertext:
        """server_key/mod.rs:135-137."""
        return Ciphertext(trivial=bool(message))

    # ---- helpers ----
    def _lwe_of(self, ct: Ciphertext) -> np.ndarray:
        """convert_into_lwe_ciphertext_32 (engine/mod.rs:196-215)."""
        if not ct.is_trivial:
            return ct.lwe
        out = np.zeros(self.parameters.lwe_dimension + 1, dtype=np.uint32)
        out[-1] = PLAINTEXT_TRUE if ct.trivial else PLAINTEXT_FALSE
        return out

    @staticmethod
    def _add_plain(lwe: np.ndarray, pt) -> np.ndarray:
        out = lwe.copy()
        out[..., -1] += np.uint32(pt)
        return out

    # linear combinations ahead of the bootstrap, engine/mod.rs:347-353, :388-397, :431-440,
    # :472-478, :515-525, :558-572 (arrays broadcast over a leading batch dimension)
    @classmethod
    def _combine(cls, gate: str, a: np.ndarray, b: np.ndarray) -> np.ndarray:
        s = (a + b).astype(np.uint32)
        neg = lambda v: (np.uint32(0) - v).astype(np.uint32)  # noqa: E731
        if gate == "and":
            return cls._add_plain(s, PLAINTEXT_FALSE)
        if gate == "nand":
            return cls._add_plain(neg(s), PLAINTEXT_TRUE)
        if gate == "nor":
            return cls._add_plain(neg(s), PLAINTEXT_FALSE)
        if gate == "or":
            return cls._add_plain(s, PLAINTEXT_TRUE)
        if gate == "xor":
            return (cls._add_plain(s, PLAINTEXT_TRUE) * np.uint32(2)).astype(np.uint32)
        if gate == "xnor":
            return (neg(cls._add_plain(s, PLAINTEXT_TRUE)) * np.uint32(2)).astype(np.uint32)
        raise ValueError(gate)

    _CLEAR = {"and": lambda x, y: x and y, "nand": lambda x, y: not (x and y), "nor": lambda x, y: not (x or y),
              "or": lambda x, y: x or y, "xor": lambda x, y: x != y, "xnor": lambda x, y: x == y}

    def _with_bool(self, gate: str, ct: Ciphertext, b: bool) -> Ciphertext:
        """ciphertext-with-cleartext shortcuts, engine/mod.rs:583-700: no bootstrap."""
        table = {"and": (ct.clone, lambda: Ciphertext(trivial=False)),
                 "nand": (lambda: self.not_(ct), lambda: Ciphertext(trivial=True)),
                 "nor": (lambda: Ciphertext(trivial=False), lambda: self.not_(ct)),
                 "or": (lambda: Ciphertext(trivial=True), ct.clone),
                 "xor": (lambda: self.not_(ct), ct.clone),
                 "xnor": (ct.clone, lambda: self.not_(ct))}
        when_true, when_false = table[gate]
        return when_true() if b else when_false()

    def _gate(self, gate: str, left: BoolOrCt, right: BoolOrCt) -> Ciphertext:
        if isinstance(left, bool) and isinstance(right, bool):
            return Ciphertext(trivial=bool(self._CLEAR[gate](left, right)))
        if isinstance(left, bool):
            left, right = right, left  # every binary gate here is symmetric (engine/mod.rs:702-770)
        if isinstance(right, bool):
            return self._with_bool(gate, left, right)
        if left.is_trivial and right.is_trivial:
            return Ciphertext(trivial=bool(self._CLEAR[gate](left.trivial, right.trivial)))
        if right.is_trivial:
            return self._with_bool(gate, left, bool(right.trivial))
        if left.is_trivial:
            return self._with_bool(gate, right, bool(left.trivial))
        buf = self._combine(gate, left.lwe, right.lwe)
        return Ciphertext(lwe=self.bootstrapper.bootstrap_keyswitch(buf)[0])

    # ---- gates (server_key/mod.rs:79-160) ----
    def and_(self, a: BoolOrCt, b: BoolOrCt) -> Ciphertext:
        return self._gate("and", a, b)

    def nand(self, a: BoolOrCt, b: BoolOrCt) -> Ciphertext:
        return self._gate("nand", a, b)

    def nor(self, a: BoolOrCt, b: BoolOrCt) -> Ciphertext:
        return self._gate("nor", a, b)

    def or_(self, a: BoolOrCt, b: BoolOrCt) -> Ciphertext:
        return self._gate("or", a, b)

    def xor(self, a: BoolOrCt, b: BoolOrCt) -> Ciphertext:
        return self._gate("xor", a, b)

    def xnor(self, a: BoolOrCt, b: BoolOrCt) -> Ciphertext:
        return self._gate("xnor", a, b)

    def not_(self, ct: Ciphertext) -> Ciphertext:
        """engine/mod.rs:149-161: negation, no bootstrap."""
        if ct.is_trivial:
            return Ciphertext(trivial=not ct.trivial)
        return Ciphertext(lwe=(np.uint32(0) - ct.lwe).astype(np.uint32))

    def mux(self, cond: Ciphertext, then: Ciphertext, other: Ciphertext) -> Ciphertext:
        """engine/mod.rs:217-316: two bootstraps, one add, one keyswitch."""
        if cond.is_trivial:
            return then.clone() if cond.trivial else other.clone()
        if then.is_trivial:
            return self.or_(cond, other) if then.trivial else self.and_(self.not_(cond), other)
        if other.is_trivial:
            return self.or_(then, self.not_(cond)) if other.trivial else self.and_(cond, then)
        c, t, e = cond.lwe, self._lwe_of(then), self._lwe_of(other)
        first = self._add_plain((c + t).astype(np.uint32), PLAINTEXT_FALSE)
        second = self._add_plain(((np.uint32(0) - c) + e).astype(np.uint32), PLAINTEXT_FALSE)
        pbs = self.bootstrapper.bootstrap(np.stack([first, second]))
        summed = self._add_plain((pbs[0] + pbs[1]).astype(np.uint32), PLAINTEXT_TRUE)
        return Ciphertext(lwe=self.bootstrapper.keyswitch(summed)[0])

    # ---- batched gates: arrays [count][n+1] of encrypted inputs ----
    # linear part of each gate as (alpha, beta, plaintext) of alpha*a + beta*b + plaintext, then an
    # optional doubling: engine/mod.rs:347-353, :388-397, :431-440, :472-478, :515-525, :558-572
    _LINEAR = {"and": (1, 1, int(PLAINTEXT_FALSE)), "nand": (-1, -1, int(PLAINTEXT_TRUE)),
               "nor": (-1, -1, int(PLAINTEXT_FALSE)), "or": (1, 1, int(PLAINTEXT_TRUE)),
               "xor": (2, 2, 2 * int(PLAINTEXT_TRUE)), "xnor": (-2, -2, -2 * int(PLAINTEXT_TRUE))}

    def gate_batch(self, gate: str, a, b):
        """One gate on many ciphertext pairs. Host arrays ([count][n+1] uint32) go through the host
        entry point; torch CUDA int32 tensors stay on the GPU, where the gate is one call
        (``camd_axpby_pbs_ks_batch_device``: linear combination, bootstrap, keyswitch)."""
        if isinstance(a, np.ndarray):
            a = np.ascontiguousarray(a, dtype=np.uint32)
            b = np.ascontiguousarray(b, dtype=np.uint32)
            return self.bootstrapper.bootstrap_keyswitch(self._combine(gate, a, b))
        import torch

        alpha, beta, plain = self._LINEAR[gate]
        key = self.bootstrapper.key
        stream = torch.cuda.current_stream().cuda_stream
        out = torch.empty((a.shape[0], a.shape[1]), dtype=a.dtype, device=a.device)
        key.axpby_pbs_ks_device(a, alpha, b, beta, plain, None, out, stream)
        return out
