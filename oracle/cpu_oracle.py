"""ctypes front end of the CPU oracle (``oracle/tfhe_oracle.c``).

TEST INFRASTRUCTURE ONLY: imported by ``tests/``, by ``__graft_entry__.smoke`` and by the
``cpu_baseline`` leg of ``bench.py``. The product package never imports it.

The oracle restates the keyswitch + programmable bootstrap the reference performs at
``concrete-shortint/src/engine/server_side/mod.rs:131-155`` (shortint, KS then PBS) and at
``concrete-boolean/src/engine/bootstrapping/cpu.rs:215-259`` (boolean, PBS then KS).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "build", "liboracle.so")


class OrcParams(C.Structure):
    _fields_ = [
        ("lwe_dimension", C.c_uint32),
        ("glwe_dimension", C.c_uint32),
        ("polynomial_size", C.c_uint32),
        ("pbs_base_log", C.c_uint32),
        ("pbs_level", C.c_uint32),
        ("ks_base_log", C.c_uint32),
        ("ks_level", C.c_uint32),
        ("message_modulus", C.c_uint32),
        ("carry_modulus", C.c_uint32),
        ("lwe_modular_std_dev", C.c_double),
        ("glwe_modular_std_dev", C.c_double),
    ]


class OrcRng(C.Structure):
    _fields_ = [("s", C.c_uint64 * 4)]


def build(force: bool = False) -> str:
    """Compile the oracle with the committed Makefile (gcc only)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _declare(_lib)
    return _lib


def _declare(l: C.CDLL) -> None:
    vp, u32, u64, sz, dbl, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_size_t, C.c_double, C.c_int
    PP = C.POINTER(OrcParams)
    RP = C.POINTER(OrcRng)
    l.orc_rng_seed.argtypes = [RP, u64]
    l.orc_rng_next.argtypes = [RP]
    l.orc_rng_next.restype = u64
    l.orc_rng_gauss.argtypes = [RP]
    l.orc_rng_gauss.restype = dbl
    l.orc_closest_representable.argtypes = [u64, u32, u32, u32]
    l.orc_closest_representable.restype = u64
    l.orc_decompose.argtypes = [u64, u32, u32, u32, vp]
    l.orc_modswitch.argtypes = [u64, u32, u32]
    l.orc_modswitch.restype = u32
    l.orc_generate_accumulator_u64.argtypes = [PP, vp, vp]
    l.orc_boolean_accumulator_u32.argtypes = [PP, vp]
    l.orc_fft_plan_create.argtypes = [u32]
    l.orc_fft_plan_create.restype = vp
    l.orc_fft_plan_destroy.argtypes = [vp]
    l.orc_negacyclic_schoolbook_u64.argtypes = [u32, vp, vp, vp]
    for s, word in (("u64", u64), ("u32", u32)):
        getattr(l, f"orc_fft_forward_as_integer_{s}").argtypes = [vp, vp, vp]
        getattr(l, f"orc_fft_forward_as_torus_{s}").argtypes = [vp, vp, vp]
        getattr(l, f"orc_fft_add_backward_as_torus_{s}").argtypes = [vp, vp, vp]
        getattr(l, f"orc_gen_binary_key_{s}").argtypes = [RP, sz, vp]
        getattr(l, f"orc_lwe_encrypt_{s}").argtypes = [RP, vp, sz, word, dbl, vp]
        f = getattr(l, f"orc_lwe_phase_{s}")
        f.argtypes = [vp, sz, vp]
        f.restype = word
        getattr(l, f"orc_gen_ksk_{s}").argtypes = [RP, vp, sz, vp, sz, u32, u32, dbl, vp]
        getattr(l, f"orc_gen_bsk_{s}").argtypes = [RP, PP, vp, vp, vp]
        getattr(l, f"orc_keyswitch_{s}").argtypes = [vp, sz, sz, u32, u32, vp, vp]
        getattr(l, f"orc_bsk_to_fourier_{s}").argtypes = [vp, PP, vp, vp]
        getattr(l, f"orc_bootstrap_{s}").argtypes = [vp, PP, vp, vp, vp, vp]
        getattr(l, f"orc_blind_rotate_{s}").argtypes = [vp, PP, vp, vp, vp, vp]
        getattr(l, f"orc_sample_extract_{s}").argtypes = [PP, vp, vp]
        getattr(l, f"orc_ks_pbs_batch_{s}").argtypes = [vp, PP, vp, vp, vp, vp, vp, vp, sz, i32]
        getattr(l, f"orc_pbs_ks_batch_{s}").argtypes = [vp, PP, vp, vp, vp, vp, vp, vp, sz, i32]


def _p(a: np.ndarray):
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def make_params(p) -> OrcParams:
    """Build the C struct from any object carrying the reference's parameter field names."""
    return OrcParams(
        int(p.lwe_dimension), int(p.glwe_dimension), int(p.polynomial_size),
        int(p.pbs_base_log), int(p.pbs_level), int(p.ks_base_log), int(p.ks_level),
        int(getattr(p, "message_modulus", 0) or 0), int(getattr(p, "carry_modulus", 0) or 0),
        float(p.lwe_modular_std_dev), float(p.glwe_modular_std_dev),
    )


class Rng:
    def __init__(self, seed: int):
        self.state = OrcRng()
        lib().orc_rng_seed(C.byref(self.state), C.c_uint64(seed))

    def next(self) -> int:
        return lib().orc_rng_next(C.byref(self.state))

    def gauss(self) -> float:
        return lib().orc_rng_gauss(C.byref(self.state))


class Oracle:
    """CPU restatement of the reference path for one parameter set and one torus width."""

    def __init__(self, params, bits: int = 64):
        assert bits in (32, 64)
        self.p = params
        self.cp = make_params(params)
        self.bits = bits
        self.dtype = np.uint64 if bits == 64 else np.uint32
        self.s = "u64" if bits == 64 else "u32"
        self.n = self.cp.lwe_dimension
        self.k = self.cp.glwe_dimension
        self.N = self.cp.polynomial_size
        self.big = self.k * self.N
        self.plan = lib().orc_fft_plan_create(self.N)

    def __del__(self):
        try:
            if self.plan:
                lib().orc_fft_plan_destroy(self.plan)
                self.plan = None
        except Exception:
            pass

    def _f(self, name):
        return getattr(lib(), f"{name}_{self.s}")

    # ---- keys ----
    def gen_binary_key(self, rng: Rng, length: int) -> np.ndarray:
        key = np.zeros(length, dtype=self.dtype)
        self._f("orc_gen_binary_key")(C.byref(rng.state), length, _p(key))
        return key

    def gen_ksk(self, rng: Rng, in_key, out_key, std=None) -> np.ndarray:
        lv = self.cp.ks_level
        ksk = np.zeros((len(in_key), lv, len(out_key) + 1), dtype=self.dtype)
        std = self.cp.lwe_modular_std_dev if std is None else std
        self._f("orc_gen_ksk")(C.byref(rng.state), _p(in_key), len(in_key), _p(out_key), len(out_key),
                               self.cp.ks_base_log, lv, C.c_double(std), _p(ksk))
        return ksk

    def gen_bsk(self, rng: Rng, lwe_key, glwe_key, std=None) -> np.ndarray:
        cp = make_params(self.p)
        cp.lwe_dimension = len(lwe_key)
        if std is not None:
            cp.glwe_modular_std_dev = std
        bsk = np.zeros((len(lwe_key), cp.pbs_level, self.k + 1, self.k + 1, self.N), dtype=self.dtype)
        self._f("orc_gen_bsk")(C.byref(rng.state), C.byref(cp), _p(lwe_key), _p(glwe_key), _p(bsk))
        return bsk

    def bsk_to_fourier(self, bsk: np.ndarray) -> np.ndarray:
        out = np.zeros(bsk.shape, dtype=np.float64)
        cp = make_params(self.p)
        cp.lwe_dimension = bsk.shape[0]
        self._f("orc_bsk_to_fourier")(self.plan, C.byref(cp), _p(bsk), _p(out))
        return out

    # ---- ciphertexts ----
    def lwe_encrypt(self, rng: Rng, key, plaintext: int, std: float) -> np.ndarray:
        ct = np.zeros(len(key) + 1, dtype=self.dtype)
        word = C.c_uint64 if self.bits == 64 else C.c_uint32
        self._f("orc_lwe_encrypt")(C.byref(rng.state), _p(key), len(key), word(plaintext & ((1 << self.bits) - 1)),
                                   C.c_double(std), _p(ct))
        return ct

    def lwe_phase(self, key, ct) -> int:
        ct = np.ascontiguousarray(ct, dtype=self.dtype)
        return int(self._f("orc_lwe_phase")(_p(key), len(key), _p(ct)))

    # ---- the path ----
    def keyswitch(self, ksk: np.ndarray, ct_in: np.ndarray) -> np.ndarray:
        in_dim, lv, out_size = ksk.shape
        ct_in = np.ascontiguousarray(ct_in, dtype=self.dtype)
        assert ct_in.shape[-1] == in_dim + 1
        out = np.zeros(out_size, dtype=self.dtype)
        self._f("orc_keyswitch")(_p(ksk), in_dim, out_size - 1, self.cp.ks_base_log, lv, _p(ct_in), _p(out))
        return out

    def _params_for(self, fbsk: np.ndarray) -> OrcParams:
        cp = make_params(self.p)
        cp.lwe_dimension = fbsk.shape[0]
        return cp

    def bootstrap(self, fbsk: np.ndarray, acc: np.ndarray, ct_in: np.ndarray) -> np.ndarray:
        cp = self._params_for(fbsk)
        ct_in = np.ascontiguousarray(ct_in, dtype=self.dtype)
        acc = np.ascontiguousarray(acc, dtype=self.dtype)
        assert ct_in.shape[-1] == cp.lwe_dimension + 1 and acc.size == (self.k + 1) * self.N
        out = np.zeros(self.big + 1, dtype=self.dtype)
        self._f("orc_bootstrap")(self.plan, C.byref(cp), _p(fbsk), _p(acc), _p(ct_in), _p(out))
        return out

    def blind_rotate(self, fbsk, acc, ct_in) -> np.ndarray:
        cp = self._params_for(fbsk)
        ct_in = np.ascontiguousarray(ct_in, dtype=self.dtype)
        acc = np.ascontiguousarray(acc, dtype=self.dtype)
        out = np.zeros((self.k + 1) * self.N, dtype=self.dtype)
        self._f("orc_blind_rotate")(self.plan, C.byref(cp), _p(fbsk), _p(acc), _p(ct_in), _p(out))
        return out

    def sample_extract(self, glwe) -> np.ndarray:
        glwe = np.ascontiguousarray(glwe, dtype=self.dtype)
        out = np.zeros(self.big + 1, dtype=self.dtype)
        self._f("orc_sample_extract")(C.byref(self.cp), _p(glwe), _p(out))
        return out

    def _batch(self, fn, ksk, fbsk, accs, lut_index, ct_in, out_width, threads):
        cp = self._params_for(fbsk)
        ct_in = np.ascontiguousarray(ct_in, dtype=self.dtype)
        accs = np.ascontiguousarray(accs, dtype=self.dtype)
        count = ct_in.shape[0]
        out = np.zeros((count, out_width), dtype=self.dtype)
        li = None
        if lut_index is not None:
            li = np.ascontiguousarray(lut_index, dtype=np.uint32)
            assert li.shape[0] == count
        self._f(fn)(self.plan, C.byref(cp), _p(ksk), _p(fbsk), _p(accs), _p(li) if li is not None else None,
                    _p(ct_in), _p(out), count, int(threads))
        return out

    def ks_pbs_batch(self, ksk, fbsk, accs, lut_index, ct_in, threads=1) -> np.ndarray:
        """shortint order: keyswitch then bootstrap; in/out are [count][kN+1]."""
        return self._batch("orc_ks_pbs_batch", ksk, fbsk, accs, lut_index, ct_in, self.big + 1, threads)

    def pbs_ks_batch(self, ksk, fbsk, accs, lut_index, ct_in, threads=1) -> np.ndarray:
        """boolean order: bootstrap then keyswitch; in/out are [count][n+1]."""
        return self._batch("orc_pbs_ks_batch", ksk, fbsk, accs, lut_index, ct_in, fbsk.shape[0] + 1, threads)

    # ---- look-up tables ----
    def generate_accumulator(self, f) -> np.ndarray:
        """engine/mod.rs:130-178 with f given as a Python callable on [0, msg*carry)."""
        assert self.bits == 64
        sup = self.cp.message_modulus * self.cp.carry_modulus
        vals = np.array([int(f(i)) & 0xFFFFFFFFFFFFFFFF for i in range(sup)], dtype=np.uint64)
        out = np.zeros((self.k + 1) * self.N, dtype=np.uint64)
        lib().orc_generate_accumulator_u64(C.byref(self.cp), _p(vals), _p(out))
        return out

    def boolean_accumulator(self) -> np.ndarray:
        assert self.bits == 32
        out = np.zeros((self.k + 1) * self.N, dtype=np.uint32)
        lib().orc_boolean_accumulator_u32(C.byref(self.cp), _p(out))
        return out

    # ---- transforms (for the FFT known-answer test) ----
    def fft_forward_integer(self, poly) -> np.ndarray:
        poly = np.ascontiguousarray(poly, dtype=self.dtype)
        out = np.zeros(self.N, dtype=np.float64)
        self._f("orc_fft_forward_as_integer")(self.plan, _p(poly), _p(out))
        return out

    def fft_forward_torus(self, poly) -> np.ndarray:
        poly = np.ascontiguousarray(poly, dtype=self.dtype)
        out = np.zeros(self.N, dtype=np.float64)
        self._f("orc_fft_forward_as_torus")(self.plan, _p(poly), _p(out))
        return out

    def fft_add_backward_torus(self, fourier, poly) -> np.ndarray:
        fourier = np.ascontiguousarray(fourier, dtype=np.float64)
        poly = np.array(poly, dtype=self.dtype, copy=True)
        self._f("orc_fft_add_backward_as_torus")(self.plan, _p(fourier), _p(poly))
        return poly


# ---- shortint / boolean client side (client_side.rs:45-143, boolean engine/mod.rs:107-147) ----
def shortint_delta(params) -> int:
    return (1 << 63) // (params.message_modulus * params.carry_modulus)


def shortint_decode(params, phase: int) -> int:
    """decrypt_message_and_carry rounding (client_side.rs:128-138)."""
    delta = shortint_delta(params)
    rounding = (phase & (delta >> 1)) << 1
    return ((phase + rounding) & 0xFFFFFFFFFFFFFFFF) // delta


def closest_representable(x, base_log, level, bits=64) -> int:
    return int(lib().orc_closest_representable(C.c_uint64(x), base_log, level, bits))


def decompose(x, base_log, level, bits=64):
    d = np.zeros(level, dtype=np.int64)
    lib().orc_decompose(C.c_uint64(x), base_log, level, bits, _p(d))
    return [int(v) for v in d]


def modswitch(x, log2_n, bits=64) -> int:
    return int(lib().orc_modswitch(C.c_uint64(x), log2_n, bits))


def negacyclic_schoolbook(a, b) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.int64)
    out = np.zeros(len(a), dtype=np.uint64)
    lib().orc_negacyclic_schoolbook_u64(len(a), _p(a), _p(b), _p(out))
    return out
