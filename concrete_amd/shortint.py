"""Host-side mirror of concrete-shortint's interface for the keyswitch + bootstrap path.

Same names, argument meaning and error behaviour as the reference, with the computation on the
GPU through the C ABI:

* ``ClientKey``  — ``concrete-shortint/src/client_key/mod.rs`` + ``engine/client_side.rs:9-144``
* ``ServerKey``  — ``concrete-shortint/src/server_key/mod.rs:54-229`` +
  ``engine/server_side/mod.rs:22-311``
* ``Ciphertext`` — ``concrete-shortint/src/ciphertext/mod.rs`` (LWE words, degree, moduli)

What the reference does one ciphertext at a time with a thread-local engine is also offered in
batched form (``*_batch``): many ciphertexts, one look-up table id each, one launch.
Look-up tables are cached on the device by content, where the reference rebuilds them per call
(``engine/server_side/mod.rs:205,231``).
"""
from __future__ import annotations

import contextlib
import threading
import hashlib
from dataclasses import dataclass
from typing import Callable, Iterable, Sequence

import numpy as np

from . import client as _client
from .backend import CAMD_PBS_AUTO, Context, DeviceServerKey
from .shortint_ops import CheckError, ShortintOps  # noqa: F401


@dataclass
class Ciphertext:
    """LWE ciphertext under the large key, with the reference's bookkeeping fields.

    ``ct`` is one ciphertext (kN+1 uint64 words, body last), a host batch ([count][kN+1] numpy)
    or a device-resident batch ([count][kN+1] torch CUDA int64 tensor); the bookkeeping fields
    do not depend on the encrypted data and describe every ciphertext of a batch."""
    ct: object
    degree: int
    message_modulus: int
    carry_modulus: int

    def clone(self) -> "Ciphertext":
        data = self.ct.copy() if isinstance(self.ct, np.ndarray) else self.ct.clone()
        return Ciphertext(data, self.degree, self.message_modulus, self.carry_modulus)

    @property
    def count(self) -> int:
        return 1 if self.ct.ndim == 1 else int(self.ct.shape[0])

    def to_device(self, device="cuda") -> "Ciphertext":
        """Host batch -> device-resident batch (the words are reinterpreted as int64)."""
        import torch

        a = np.atleast_2d(np.ascontiguousarray(self.ct, dtype=np.uint64))
        return Ciphertext(torch.from_numpy(a.view(np.int64)).to(device), self.degree, self.message_modulus,
                          self.carry_modulus)

    def to_host(self) -> "Ciphertext":
        if isinstance(self.ct, np.ndarray):
            return self.clone()
        return Ciphertext(self.ct.cpu().numpy().view(np.uint64), self.degree, self.message_modulus, self.carry_modulus)


class LookupTable:
    """A look-up table ("accumulator") resident on the GPU: trivial GLWE + its device id."""

    def __init__(self, glwe: np.ndarray, lut_id: int):
        self.glwe = glwe
        self.lut_id = lut_id


class ClientKey:
    """Secret keys of one parameter set (engine/client_side.rs:9-31)."""

    def __init__(self, parameters, seed: int | None = None):
        self.parameters = parameters
        self._rng = _client.Rng(seed)
        p = parameters
        self.lwe_secret_key_after_ks = _client.gen_binary_key(self._rng, 64, p.lwe_dimension)
        self.glwe_secret_key = _client.gen_binary_key(self._rng, 64, p.glwe_dimension * p.polynomial_size)
        # transform_glwe_secret_key_to_lwe_secret_key: the flattened GLWE key
        self.lwe_secret_key = self.glwe_secret_key

    @classmethod
    def new(cls, parameters, seed: int | None = None) -> "ClientKey":
        return cls(parameters, seed)

    @property
    def _delta(self) -> int:
        p = self.parameters
        return (1 << 63) // (p.message_modulus * p.carry_modulus)

    def encrypt(self, message: int) -> Ciphertext:
        return self.encrypt_with_message_modulus(message, self.parameters.message_modulus)

    def encrypt_with_message_modulus(self, message: int, message_modulus: int) -> Ciphertext:
        """engine/client_side.rs:45-83."""
        p = self.parameters
        carry_modulus = (p.message_modulus * p.carry_modulus) // message_modulus
        m = int(message) % message_modulus
        ct = _client.lwe_encrypt(self._rng, 64, self.lwe_secret_key, [m * self._delta], p.lwe_modular_std_dev)[0]
        return Ciphertext(ct, message_modulus - 1, message_modulus, carry_modulus)

    def unchecked_encrypt(self, message: int) -> Ciphertext:
        """engine/client_side.rs:85-112: no reduction of the message."""
        p = self.parameters
        pt = (int(message) * self._delta) & 0xFFFFFFFFFFFFFFFF
        ct = _client.lwe_encrypt(self._rng, 64, self.lwe_secret_key, [pt], p.lwe_modular_std_dev)[0]
        return Ciphertext(ct, p.message_modulus * p.carry_modulus - 1, p.message_modulus, p.carry_modulus)

    def encrypt_batch(self, messages: Iterable[int]) -> np.ndarray:
        """Fresh encryptions of ``messages`` (reduced mod message_modulus) as a [count][kN+1] array."""
        p = self.parameters
        pts = [(int(m) % p.message_modulus) * self._delta for m in messages]
        return _client.lwe_encrypt(self._rng, 64, self.lwe_secret_key, pts, p.lwe_modular_std_dev)

    def decode(self, phase: int) -> int:
        """Rounding of decrypt_message_and_carry (engine/client_side.rs:128-138)."""
        delta = self._delta
        rounding = (int(phase) & (delta >> 1)) << 1
        return ((int(phase) + rounding) & 0xFFFFFFFFFFFFFFFF) // delta

    def decrypt_message_and_carry(self, ct: Ciphertext):
        """One ciphertext -> int; a batch (host or device) -> int64 array."""
        data = ct.ct if isinstance(ct.ct, np.ndarray) else ct.ct.cpu().numpy().view(np.uint64)
        if data.ndim == 1:
            return self.decode(_client.lwe_phase(64, self.lwe_secret_key, data)[0])
        return self.decrypt_message_and_carry_batch(np.ascontiguousarray(data))

    def encrypt_many(self, messages: Iterable[int]) -> Ciphertext:
        """A host batch of fresh encryptions sharing one set of bookkeeping fields."""
        p = self.parameters
        return Ciphertext(self.encrypt_batch(messages), p.message_modulus - 1, p.message_modulus, p.carry_modulus)

    def decrypt(self, ct: Ciphertext) -> int:
        return self.decrypt_message_and_carry(ct) % ct.message_modulus

    # ---- encodings without the padding bit, and the native CRT encoding ----
    def _phases(self, ct: Ciphertext) -> np.ndarray:
        data = ct.ct if isinstance(ct.ct, np.ndarray) else ct.ct.cpu().numpy().view(np.uint64)
        return _client.lwe_phase(64, self.lwe_secret_key, np.ascontiguousarray(data))

    @staticmethod
    def _shape(ct: Ciphertext, values: list):
        return values[0] if ct.ct.ndim == 1 else np.array(values, dtype=np.int64)

    def encrypt_without_padding(self, message: int) -> Ciphertext:
        """engine/client_side.rs:146-175: the scaling is doubled, the message is not reduced."""
        p = self.parameters
        pt = (int(message) * self._delta * 2) & 0xFFFFFFFFFFFFFFFF
        ct = _client.lwe_encrypt(self._rng, 64, self.lwe_secret_key, [pt], p.lwe_modular_std_dev)[0]
        return Ciphertext(ct, p.message_modulus - 1, p.message_modulus, p.carry_modulus)

    def decrypt_message_and_carry_without_padding(self, ct: Ciphertext):
        """engine/client_side.rs:177-203."""
        delta = self._delta * 2
        out = []
        for ph in self._phases(ct):
            rounding = (int(ph) & (delta >> 1)) << 1
            out.append(((int(ph) + rounding) & 0xFFFFFFFFFFFFFFFF) // delta)
        return self._shape(ct, out)

    def decrypt_without_padding(self, ct: Ciphertext):
        """engine/client_side.rs:205-212."""
        return self.decrypt_message_and_carry_without_padding(ct) % ct.message_modulus

    def encrypt_native_crt(self, message: int, message_modulus: int) -> Ciphertext:
        """engine/client_side.rs:214-241: the whole torus is split into ``message_modulus`` parts
        (128-bit intermediate, floor division), carry modulus 1."""
        p = self.parameters
        message_modulus = int(message_modulus) & 0xFF
        m = int(message) % message_modulus
        pt = ((m << 64) // message_modulus) & 0xFFFFFFFFFFFFFFFF
        ct = _client.lwe_encrypt(self._rng, 64, self.lwe_secret_key, [pt], p.lwe_modular_std_dev)[0]
        return Ciphertext(ct, message_modulus - 1, message_modulus, 1)

    def decrypt_message_native_crt(self, ct: Ciphertext, message_modulus: int):
        """engine/client_side.rs:243-262 (u128 arithmetic: the product cannot wrap)."""
        basis = int(message_modulus) & 0xFF
        out = []
        for ph in self._phases(ct):
            result = int(ph) * basis
            result = (result + ((result & (1 << 63)) << 1)) >> 64
            out.append((result & 0xFFFFFFFFFFFFFFFF) % basis)
        return self._shape(ct, out)

    def decrypt_message_and_carry_batch(self, cts: np.ndarray) -> np.ndarray:
        phases = _client.lwe_phase(64, self.lwe_secret_key, cts)
        return np.array([self.decode(int(ph)) for ph in phases], dtype=np.int64)


class ServerKey(ShortintOps):
    """GPU-resident server key: keyswitch key + Fourier bootstrap key + moduli. The operation
    layer (add, mul, comparisons, ...) comes from ``ShortintOps``.

    ``ServerKey.new(cks)`` follows engine/server_side/mod.rs:22-72: the bootstrap key is generated
    in the standard domain, converted to the Fourier domain (here: on the GPU), and dropped.
    """

    def __init__(self, device_key: DeviceServerKey, parameters, max_degree: int | None = None):
        self.key = device_key
        self.parameters = parameters
        self.message_modulus = parameters.message_modulus
        self.carry_modulus = parameters.carry_modulus
        self.max_degree = (parameters.message_modulus * parameters.carry_modulus - 1) if max_degree is None else max_degree
        self._lut_cache: dict[bytes, LookupTable] = {}
        self._device_ids: dict = {}   # (lut id, count) -> device array of ids, reused across calls
        self._queue = None            # bootstraps collected by deferred()
        self._tls = threading.local()  # .gather: the Gather a worker thread of gather() belongs to
        # engine/mod.rs:195-199: the default accumulator is x -> x % message_modulus
        msg = self.message_modulus
        self._default_acc = self.generate_accumulator(lambda x: x % msg)

    @classmethod
    def new(cls, cks: ClientKey, ctx: Context | None = None, variant: int = CAMD_PBS_AUTO,
            threads: int | None = None) -> "ServerKey":
        p = cks.parameters
        bsk = _client.gen_bootstrap_key(cks._rng, p, 64, cks.lwe_secret_key_after_ks, cks.glwe_secret_key, threads=threads)
        ksk = _client.gen_keyswitch_key(cks._rng, p, 64, cks.lwe_secret_key, cks.lwe_secret_key_after_ks)
        return cls.from_standard_keys(p, bsk, ksk, ctx=ctx, variant=variant)

    @classmethod
    def new_with_max_degree(cls, cks: ClientKey, max_degree: int, ctx: Context | None = None,
                            variant: int = CAMD_PBS_AUTO) -> "ServerKey":
        """server_key/mod.rs:86-90 -> engine/server_side/mod.rs:31-72."""
        sks = cls.new(cks, ctx=ctx, variant=variant)
        sks.max_degree = int(max_degree)
        return sks

    @classmethod
    def from_standard_keys(cls, parameters, bsk_standard: np.ndarray, ksk: np.ndarray, ctx: Context | None = None,
                           variant: int = CAMD_PBS_AUTO) -> "ServerKey":
        ctx = ctx or Context(0)
        dk = DeviceServerKey.from_standard_keys(ctx, parameters, bsk_standard, ksk, variant=variant, torus_bits=64)
        return cls(dk, parameters)

    # ---- look-up tables ----
    def generate_accumulator(self, f: Callable[[int], int]) -> LookupTable:
        """server_key/mod.rs:119-126 -> engine/mod.rs:130-178, cached on the device by content."""
        glwe = _client.shortint_accumulator(self.parameters, f)
        digest = hashlib.blake2b(glwe.tobytes(), digest_size=16).digest()
        lut = self._lut_cache.get(digest)
        if lut is None:
            lut = LookupTable(glwe, self.key.upload_lut(glwe))
            self._lut_cache[digest] = lut
        return lut

    # ---- the path, one ciphertext (reference signatures) ----
    def keyswitch_programmable_bootstrap(self, ct: Ciphertext, acc: LookupTable) -> Ciphertext:
        """server_key/mod.rs:207-217 (clones, engine/server_side/mod.rs:120-129)."""
        res = ct.clone()
        self.keyswitch_programmable_bootstrap_assign(res, acc)
        return res

    @contextlib.contextmanager
    def deferred(self):
        """Collect the bootstraps issued inside the ``with`` block and run them as ONE launch when
        it ends. Valid whenever the operations inside are independent of each other's results
        (degrees and moduli are data-independent bookkeeping and are updated immediately): what
        the reference hands to rayon workers (concrete-integer radix_parallel/mul.rs:55-72,
        bitwise_op.rs) becomes a single batch here. Nested uses flush at the outermost level."""
        if self._queue is not None:
            yield
            return
        self._queue = []
        try:
            yield
        finally:
            items, self._queue = self._queue, None
        self._launch(items)

    def _launch(self, items) -> None:
        """One launch for ``items`` - or, inside ``gather``, this operation's share of the next
        combined launch."""
        if not items:
            return
        g = getattr(self._tls, "gather", None)
        if g is None:
            self.apply_lookup_table_many(items)
        else:
            g.park(items)

    def gather(self, thunks):
        """Run independent operations side by side and issue the bootstraps each of them needs
        next as ONE launch, round after round, whatever the operations are (a multiplication next
        to an addition next to a comparison, of different integers). The reference gets the same
        concurrency from rayon workers that call the server key at the same time
        (concrete-integer radix_parallel/mul.rs:319-328, add.rs:14-19); here every operation runs
        until it needs a bootstrap result, hands its requests over and waits, and when all have
        done so the requests go to the GPU together. Operations take turns (one runs at a time),
        so the result is that of calling them one after the other. Returns their results in order.
        """
        if self._queue is not None:
            # the workers' bootstraps would join the caller's deferred queue instead of parking,
            # and dependent steps inside a thunk would read ciphertexts that were never bootstrapped
            raise RuntimeError("gather() inside deferred(): the operations of a deferred block must not depend on bootstrap results")
        return _Gather(self, list(thunks)).run()

    def keyswitch_programmable_bootstrap_assign(self, ct: Ciphertext, acc: LookupTable) -> None:
        """server_key/mod.rs:219-229 -> engine/server_side/mod.rs:131-155. Degree is not touched."""
        if self._queue is not None:
            self._queue.append((ct, acc))
            return
        if getattr(self._tls, "gather", None) is not None:
            self._launch([(ct, acc)])
            return
        if isinstance(ct.ct, np.ndarray):
            if ct.ct.ndim == 1:
                ct.ct = self.key.ks_pbs(ct.ct[None, :], np.array([acc.lut_id], dtype=np.uint32))[0]
            else:
                ct.ct = self.key.ks_pbs(ct.ct, np.full(ct.ct.shape[0], acc.lut_id, dtype=np.uint32))
            return
        import torch

        count = ct.ct.shape[0]
        ids = self._device_ids.get((acc.lut_id, count))
        if ids is None:
            ids = torch.full((count,), acc.lut_id, dtype=torch.int32, device=ct.ct.device)
            self._device_ids[(acc.lut_id, count)] = ids
        out = torch.empty_like(ct.ct)
        self.key.ks_pbs_device(ct.ct, ids, out, count, torch.cuda.current_stream().cuda_stream)
        ct.ct = out

    def bootstrap_of_combination_assign(self, ct: Ciphertext, alpha: int, other: Ciphertext, beta: int,
                                        acc: LookupTable) -> None:
        """ct <- bootstrap(alpha*ct + beta*other) with ``acc``: the shift-and-add and the bootstrap
        of engine/server_side/mod.rs:182-195. A device-resident batch goes through one call
        (``camd_axpby_ks_pbs_batch_device``: the keyswitch forms the combination while it reads);
        everything else takes the linear pass and then the bootstrap. Degree is not touched."""
        deferred = self._queue is not None or getattr(self._tls, "gather", None) is not None
        if deferred or isinstance(ct.ct, np.ndarray) or ct.ct.ndim != 2:
            ct.ct = self._lin(ct.ct, alpha, other.ct, beta)
            self.keyswitch_programmable_bootstrap_assign(ct, acc)
            return
        import torch

        count = ct.ct.shape[0]
        ids = self._device_ids.get((acc.lut_id, count))
        if ids is None:
            ids = torch.full((count,), acc.lut_id, dtype=torch.int32, device=ct.ct.device)
            self._device_ids[(acc.lut_id, count)] = ids
        out = torch.empty((count, ct.ct.shape[1]), dtype=ct.ct.dtype, device=ct.ct.device)
        self.key.axpby_ks_pbs_device(ct.ct, alpha, other.ct, beta, 0, ids, out, torch.cuda.current_stream().cuda_stream)
        ct.ct = out

    def keyswitch_bootstrap(self, ct: Ciphertext) -> Ciphertext:
        """server_key/mod.rs:172-175 -> engine/server_side/mod.rs:85-118 (default accumulator)."""
        return self.keyswitch_programmable_bootstrap(ct, self._default_acc)

    def keyswitch_bootstrap_assign(self, ct: Ciphertext) -> None:
        self.keyswitch_programmable_bootstrap_assign(ct, self._default_acc)

    def carry_extract_assign(self, ct: Ciphertext) -> None:
        """engine/server_side/mod.rs:198-212."""
        modulus = ct.message_modulus
        acc = self.generate_accumulator(lambda x: x // modulus)
        self.keyswitch_programmable_bootstrap_assign(ct, acc)
        ct.degree = min(modulus - 1, ct.degree // modulus)

    def carry_extract(self, ct: Ciphertext) -> Ciphertext:
        res = ct.clone()
        self.carry_extract_assign(res)
        return res

    def message_extract_assign(self, ct: Ciphertext) -> None:
        """engine/server_side/mod.rs:224-237."""
        modulus = ct.message_modulus
        acc = self.generate_accumulator(lambda x: x % modulus)
        self.keyswitch_programmable_bootstrap_assign(ct, acc)
        ct.degree = ct.message_modulus - 1

    def message_extract(self, ct: Ciphertext) -> Ciphertext:
        res = ct.clone()
        self.message_extract_assign(res)
        return res

    def create_trivial(self, value: int, like: Ciphertext | None = None) -> Ciphertext:
        """engine/server_side/mod.rs:250-282: mask zero, sized with the bootstrap output dimension.
        With ``like`` the result is a batch of the same size and residency as ``like``."""
        p = self.parameters
        modular_value = int(value) % self.message_modulus
        delta = (1 << 63) // (self.message_modulus * self.carry_modulus)
        words = p.glwe_dimension * p.polynomial_size + 1
        if like is None or (isinstance(like.ct, np.ndarray) and like.ct.ndim == 1):
            ct = np.zeros(words, dtype=np.uint64)
            ct[-1] = np.uint64(modular_value * delta)
        elif isinstance(like.ct, np.ndarray):
            ct = np.zeros((like.ct.shape[0], words), dtype=np.uint64)
            ct[:, -1] = np.uint64(modular_value * delta)
        else:
            import torch

            ct = torch.zeros((like.ct.shape[0], words), dtype=torch.int64, device=like.ct.device)
            ct[:, -1] = modular_value * delta
        return Ciphertext(ct, modular_value, self.message_modulus, self.carry_modulus)

    def create_trivial_assign(self, ct: Ciphertext, value: int) -> None:
        """engine/server_side/mod.rs:284-310. As in the reference, the new ciphertext is sized with
        the bootstrap key's INPUT dimension (n+1 words; ``create_trivial`` uses the output
        dimension, :257-261) and the moduli of ``ct`` are left as they are."""
        modular_value = (int(value) & 0xFF) % self.message_modulus
        delta = (1 << 63) // (self.message_modulus * self.carry_modulus)
        words = np.zeros(self.parameters.lwe_dimension + 1, dtype=np.uint64)
        words[-1] = np.uint64(modular_value * delta)
        ct.ct = words
        ct.degree = modular_value

    # ---- batched forms ----
    def keyswitch_programmable_bootstrap_batch(self, cts: np.ndarray, accs: Sequence[LookupTable] | LookupTable) -> np.ndarray:
        """Many ciphertexts ([count][kN+1] uint64), one look-up table each (or one for all)."""
        cts = np.ascontiguousarray(cts, dtype=np.uint64)
        if isinstance(accs, LookupTable):
            ids = np.full(cts.shape[0], accs.lut_id, dtype=np.uint32)
        else:
            ids = np.array([a.lut_id for a in accs], dtype=np.uint32)
        return self.key.ks_pbs(cts, ids)

    def keyswitch_bootstrap_batch(self, cts: np.ndarray) -> np.ndarray:
        return self.keyswitch_programmable_bootstrap_batch(cts, self._default_acc)


class _Gather:
    """Cooperative scheduler behind ``ServerKey.gather``: one thread per operation, exactly one of
    them (or the caller) running at any time. A worker runs until it has bootstraps to launch
    (``park``) or returns; when every live worker has done either, the caller's thread launches
    the parked requests together and the round repeats."""

    def __init__(self, key: "ServerKey", thunks):
        self.key = key
        self.thunks = thunks
        self.resume = [threading.Semaphore(0) for _ in thunks]
        self.arrived = threading.Semaphore(0)
        self.pending = [None] * len(thunks)
        self.done = [False] * len(thunks)
        self.results = [None] * len(thunks)
        self.errors = [None] * len(thunks)
        self.failure = None               # exception of a combined launch, handed to every parked operation

    def _worker(self, w: int) -> None:
        self.resume[w].acquire()
        self.key._tls.gather = self
        self.key._tls.index = w
        try:
            if self.failure is not None:   # the caller gave up before this operation started
                raise self.failure
            self.results[w] = self.thunks[w]()
        except BaseException as e:  # re-raised by run() in the caller's thread
            self.errors[w] = e
        finally:
            self.key._tls.gather = None
            self.done[w] = True
            self.arrived.release()

    def park(self, items) -> None:
        w = self.key._tls.index
        if self.failure is not None:       # the caller has given up: do not wait for a launch that will not come
            raise self.failure
        self.pending[w] = list(items)
        self.arrived.release()
        self.resume[w].acquire()
        if self.failure is not None:       # the combined launch failed: unwind this operation too
            raise self.failure

    def run(self):
        threads = [threading.Thread(target=self._worker, args=(w,), daemon=True) for w in range(len(self.thunks))]
        for t in threads:
            t.start()
        live = list(range(len(self.thunks)))
        try:
            while live:
                for w in live:                 # one at a time: each runs to its next launch point
                    self.resume[w].release()
                    self.arrived.acquire()
                batch = []
                for w in live:
                    if self.pending[w]:
                        batch.extend(self.pending[w])
                    self.pending[w] = None
                live = [w for w in live if not self.done[w]]
                if batch:
                    # inside an outer gather this parks in turn; otherwise it is the launch
                    try:
                        self.key._launch(batch)
                    except BaseException as e:
                        self.failure = e
        except BaseException as e:
            # the caller was interrupted while the workers are parked (KeyboardInterrupt in
            # acquire): hand every parked operation the failure so that its thread unwinds
            # instead of waiting forever, then let the interruption through
            self.failure = self.failure or e
            for w in range(len(self.thunks)):
                self.resume[w].release()
            raise
        for t in threads:
            t.join()
        if self.failure is not None:
            raise self.failure
        for e in self.errors:
            if e is not None:
                raise e
        return self.results
