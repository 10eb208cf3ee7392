"""Host-side client operations (keys, encryption, phase, look-up tables) over the C ABI of
``include/concrete_amd_client.h``. numpy in, numpy out; nothing here touches the GPU.

``Rng()`` is ChaCha20 keyed from the operating system; ``Rng(seed)`` is a deterministic
xoshiro256** sequence for tests and benchmarks and must not generate keys that protect data.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _ffi
from ._ffi import check


def _ptr(a: np.ndarray):
    if not a.flags["C_CONTIGUOUS"]:
        raise ValueError("array must be C-contiguous")
    return a.ctypes.data_as(C.c_void_p)


def _dtype(bits: int):
    return np.uint64 if bits == 64 else np.uint32


class Rng:
    """``seed=None``: cryptographically secure (ChaCha20, key from getrandom). An integer seed gives
    the reproducible test sequence."""

    def __init__(self, seed: int | None = None):
        self.state = _ffi.CamdRng()
        self.secure = seed is None
        if seed is None:
            check(_ffi.lib().camd_rng_seed_secure(C.byref(self.state)))
        else:
            _ffi.lib().camd_rng_seed(C.byref(self.state), C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF))


def gen_binary_key(rng: Rng, bits: int, length: int) -> np.ndarray:
    key = np.zeros(length, dtype=_dtype(bits))
    check(_ffi.lib().camd_client_gen_binary_key(C.byref(rng.state), bits, length, _ptr(key)))
    return key


def gen_bootstrap_key(rng: Rng, params, bits: int, lwe_key: np.ndarray, glwe_key: np.ndarray,
                      glwe_std_dev: float | None = None, threads: int | None = None) -> np.ndarray:
    """Standard-domain bootstrap key [n][level][k+1][k+1][N]; n = len(lwe_key)."""
    import dataclasses

    p = dataclasses.replace(params, lwe_dimension=len(lwe_key))
    cp = _ffi.make_params(p, bits)
    k, N = int(p.glwe_dimension), int(p.polynomial_size)
    bsk = np.zeros((len(lwe_key), int(p.pbs_level), k + 1, k + 1, N), dtype=_dtype(bits))
    std = float(params.glwe_modular_std_dev if glwe_std_dev is None else glwe_std_dev)
    th = int(threads) if threads else min(16, os.cpu_count() or 1)
    check(_ffi.lib().camd_client_gen_bootstrap_key(C.byref(rng.state), C.byref(cp), C.c_double(std), _ptr(lwe_key),
                                                   _ptr(glwe_key), _ptr(bsk), th))
    return bsk


def gen_keyswitch_key(rng: Rng, params, bits: int, in_key: np.ndarray, out_key: np.ndarray,
                      lwe_std_dev: float | None = None) -> np.ndarray:
    lv = int(params.ks_level)
    ksk = np.zeros((len(in_key), lv, len(out_key) + 1), dtype=_dtype(bits))
    std = float(params.lwe_modular_std_dev if lwe_std_dev is None else lwe_std_dev)
    check(_ffi.lib().camd_client_gen_keyswitch_key(C.byref(rng.state), bits, _ptr(in_key), len(in_key), _ptr(out_key),
                                                   len(out_key), int(params.ks_base_log), lv, C.c_double(std), _ptr(ksk)))
    return ksk


def lwe_encrypt(rng: Rng, bits: int, key: np.ndarray, plaintexts, std_dev: float) -> np.ndarray:
    pts = np.ascontiguousarray(np.atleast_1d(plaintexts), dtype=_dtype(bits))
    out = np.zeros((pts.shape[0], len(key) + 1), dtype=_dtype(bits))
    check(_ffi.lib().camd_client_lwe_encrypt(C.byref(rng.state), bits, _ptr(key), len(key), _ptr(pts), pts.shape[0],
                                             C.c_double(std_dev), _ptr(out)))
    return out


def lwe_phase(bits: int, key: np.ndarray, cts) -> np.ndarray:
    a = np.ascontiguousarray(cts, dtype=_dtype(bits))
    if a.ndim == 1:
        a = a[None, :]
    if a.shape[1] != len(key) + 1:
        raise ValueError("ciphertext width does not match the key")
    out = np.zeros(a.shape[0], dtype=_dtype(bits))
    check(_ffi.lib().camd_client_lwe_phase(bits, _ptr(key), len(key), _ptr(a), a.shape[0], _ptr(out)))
    return out


def shortint_accumulator(params, f) -> np.ndarray:
    """Trivial GLWE holding the look-up table of ``f`` (engine/mod.rs:130-178 of concrete-shortint)."""
    sup = int(params.message_modulus) * int(params.carry_modulus)
    vals = np.array([int(f(i)) & 0xFFFFFFFFFFFFFFFF for i in range(sup)], dtype=np.uint64)
    k, N = int(params.glwe_dimension), int(params.polynomial_size)
    out = np.zeros((k + 1) * N, dtype=np.uint64)
    check(_ffi.lib().camd_client_shortint_accumulator(k, N, int(params.message_modulus), int(params.carry_modulus),
                                                      _ptr(vals), _ptr(out)))
    return out


def boolean_accumulator(params) -> np.ndarray:
    k, N = int(params.glwe_dimension), int(params.polynomial_size)
    out = np.zeros((k + 1) * N, dtype=np.uint32)
    check(_ffi.lib().camd_client_boolean_accumulator(k, N, _ptr(out)))
    return out
