"""Key cache: keys are generated once per parameter set, kept in memory, and persisted to disk.

Mirrors ``concrete-utils/src/keycache.rs`` (``NamedParam``, ``FileStorage``, ``KeyCache``) and its
shortint and integer instances (``concrete-shortint/src/keycache.rs:168-226``,
``concrete-integer/src/keycache.rs:7-28``). A cache file is ``<prefix>/<PARAMETER_NAME>.bin`` and
holds the bincode tuple ``(parameters, (client key, server key))``; a file whose parameters differ
from the requested ones is ignored (keycache.rs:51-65).

What differs from the reference: the server key in the file carries the device-resident Fourier
key of this library (see ``wire``), so loading is one upload; and because a device key belongs to
one GPU context, the cache is asked with the context to load onto.
"""
from __future__ import annotations

import os
import threading
from typing import Optional

from . import parameters as _params
from . import wire

DEFAULT_PREFIX = os.environ.get("CAMD_KEY_DIR", os.path.join("keys", "shortint", "client_server"))


def param_name(param) -> str:
    """``NamedParam::name`` (concrete-shortint/src/keycache.rs:14-166): the name of the published
    constant equal to ``param``; unnamed parameters are an error, as in the reference."""
    for name, value in _params.SHORTINT_BY_NAME.items():
        if value == param:
            return name
    raise KeyError("Unnamed parameters")


class SharedKey:
    """keycache.rs:190-206."""

    def __init__(self, client_key, server_key):
        self._pair = (client_key, server_key)

    def client_key(self):
        return self._pair[0]

    def server_key(self):
        return self._pair[1]


class FileStorage:
    """keycache.rs:36-77, for (shortint ClientKey, shortint ServerKey) pairs."""

    def __init__(self, prefix: str = DEFAULT_PREFIX):
        self.prefix = prefix

    def path(self, param) -> str:
        return os.path.join(self.prefix, param_name(param) + ".bin")

    def load(self, param, ctx=None, seed: int | None = None) -> Optional[SharedKey]:
        path = self.path(param)
        if not os.path.exists(path):
            return None
        try:
            with open(path, "rb") as f:
                r = wire.Reader(f.read())
            stored = wire._read_shortint_parameters(r)
            if stored != param:
                return None
            cks = wire._read_shortint_client_key(r, seed)
            sks = wire._read_shortint_server_key(r, param, ctx)
            r.expect_end()
        except wire.WireError:
            return None   # keycache.rs:59-61: an unreadable file counts as a miss
        return SharedKey(cks, sks)

    def store(self, param, key: SharedKey) -> None:
        os.makedirs(self.prefix, exist_ok=True)
        path = self.path(param)
        tmp = f"{path}.{os.getpid()}.tmp"
        with open(tmp, "wb") as f:
            f.write(wire.serialize_shortint_parameters(param))
            f.write(wire.serialize_shortint_client_key(key.client_key()))
            f.write(wire.serialize_shortint_server_key(key.server_key()))
        os.replace(tmp, path)   # readers never see a half-written file


class KeyCache:
    """keycache.rs:99-182: memory first, then persistent storage, then key generation (stored
    afterwards). One entry per (parameter set, GPU)."""

    def __init__(self, storage: Optional[FileStorage] = None):
        self.persistent_storage = storage or FileStorage()
        self._memory: list = []
        self._lock = threading.Lock()

    def get_from_param(self, param, ctx=None, seed: int | None = None) -> SharedKey:
        from . import shortint
        from .backend import Context

        with self._lock:
            ctx = ctx or Context(0)
            for p, device, key in self._memory:
                if p == param and device == ctx.device_id:
                    return key
            key = self.persistent_storage.load(param, ctx, seed)
            if key is None:
                cks = shortint.ClientKey(param, seed)
                key = SharedKey(cks, shortint.ServerKey.new(cks, ctx=ctx))
                self.persistent_storage.store(param, key)
            self._memory.append((param, ctx.device_id, key))
            return key


class IntegerKeyCache:
    """concrete-integer/src/keycache.rs:7-28."""

    def __init__(self, cache: Optional[KeyCache] = None):
        self.cache = cache or KEY_CACHE

    def get_from_params(self, params, ctx=None):
        from . import integer

        keys = self.cache.get_from_param(params, ctx)
        cks = integer.ClientKey.from_shortint(keys.client_key())
        return cks, integer.ServerKey.from_shortint(cks, keys.server_key())

    def get_shortint_from_params(self, params, ctx=None):
        keys = self.cache.get_from_param(params, ctx)
        return keys.client_key(), keys.server_key()


KEY_CACHE = KeyCache()
