"""ctypes binding of ``libconcrete_amd.so`` (the C ABI declared in ``include/concrete_amd.h``).

The library is built in-tree by ``__graft_entry__.build()``. There is no CPU fallback: if the
shared object is missing, importing this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CONCRETE_AMD_LIB") or os.path.join(_HERE, "_lib", "libconcrete_amd.so")

CAMD_PBS_AUTO, CAMD_PBS_GENERIC, CAMD_PBS_WAVE, CAMD_PBS_BLOCK = 0, 1, 2, 3


class CamdParams(C.Structure):
    _fields_ = [
        ("lwe_dimension", C.c_uint32),
        ("glwe_dimension", C.c_uint32),
        ("polynomial_size", C.c_uint32),
        ("pbs_base_log", C.c_uint32),
        ("pbs_level", C.c_uint32),
        ("ks_base_log", C.c_uint32),
        ("ks_level", C.c_uint32),
        ("torus_bits", C.c_uint32),
    ]


class CamdError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"concrete_amd error {code}: {message}")
        self.code = code


# every symbol include/concrete_amd.h declares: name -> (restype, argtypes)
_vp, _u32p, _sz, _i = C.c_void_p, C.POINTER(C.c_uint32), C.c_size_t, C.c_int
_PP = C.POINTER(CamdParams)
SYMBOLS = {
    "camd_last_error": (C.c_char_p, []),
    "camd_version": (C.c_char_p, []),
    "camd_context_create": (_i, [_i, C.POINTER(_vp)]),
    "camd_context_destroy": (_i, [_vp]),
    "camd_context_device": (_i, [_vp, C.POINTER(_i)]),
    "camd_context_synchronize": (_i, [_vp]),
    "camd_server_key_create": (_i, [_vp, _PP, _vp, _vp, _i, C.POINTER(_vp)]),
    "camd_server_key_destroy": (_i, [_vp]),
    "camd_server_key_params": (_i, [_vp, _PP]),
    "camd_server_key_variant": (_i, [_vp, C.POINTER(_i)]),
    "camd_server_key_blob_size": (_i, [_vp, C.POINTER(_sz)]),
    "camd_fourier_layout_id": (_i, [_PP, _i, C.POINTER(C.c_uint64)]),
    "camd_server_key_export_blob": (_i, [_vp, _vp, _vp]),
    "camd_server_key_import_blob": (_i, [_vp, _PP, _i, _vp, _sz, _vp, C.POINTER(_vp)]),
    "camd_lut_upload": (_i, [_vp, _vp, _u32p]),
    "camd_lut_count": (_i, [_vp, _u32p]),
    "camd_ks_pbs_batch": (_i, [_vp, _vp, _vp, _vp, _sz]),
    "camd_pbs_ks_batch": (_i, [_vp, _vp, _vp, _vp, _sz]),
    "camd_pbs_batch": (_i, [_vp, _vp, _vp, _vp, _sz]),
    "camd_ks_batch": (_i, [_vp, _vp, _vp, _sz]),
    "camd_reserve": (_i, [_vp, _sz]),
    "camd_server_key_check": (_i, [_vp, _vp]),
    "camd_ks_pbs_batch_device": (_i, [_vp, _vp, _vp, _vp, _sz, _vp]),
    "camd_ks_pbs_batch_device_noalloc": (_i, [_vp, _vp, _vp, _vp, _sz, _vp]),
    "camd_pbs_ks_batch_device": (_i, [_vp, _vp, _vp, _vp, _sz, _vp]),
    "camd_pbs_batch_device": (_i, [_vp, _vp, _vp, _vp, _sz, _vp]),
    "camd_ks_batch_device": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "camd_lwe_axpby_device": (_i, [_vp, C.c_uint32, _vp, _sz, _vp, _sz, C.c_uint64, _vp, _sz, C.c_uint64, C.c_uint64, _sz,
                                   _sz, _vp]),
    "camd_axpby_ks_pbs_batch_device": (_i, [_vp, _vp, _sz, C.c_uint64, _vp, _sz, C.c_uint64, C.c_uint64, _vp, _vp, _sz, _vp]),
    "camd_axpby_pbs_ks_batch_device": (_i, [_vp, _vp, _sz, C.c_uint64, _vp, _sz, C.c_uint64, C.c_uint64, _vp, _vp, _sz, _vp]),
    "camd_measure_fp64_fma_rate": (_i, [_vp, C.POINTER(C.c_double)]),
    "camd_timing_enable": (_i, [_vp, _i]),
    "camd_timing_read": (_i, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_double),
                              C.POINTER(C.c_uint64)]),
}



class CamdRng(C.Structure):
    _fields_ = [("s", C.c_uint64 * 4), ("counter", C.c_uint64), ("secure", C.c_uint32), ("have", C.c_uint32),
                ("block", C.c_uint64 * 8)]


_RP = C.POINTER(CamdRng)
_dbl, _u32, _u64 = C.c_double, C.c_uint32, C.c_uint64
# every symbol include/concrete_amd_client.h declares
SYMBOLS.update({
    "camd_rng_seed": (None, [_RP, _u64]),
    "camd_rng_seed_secure": (_i, [_RP]),
    "camd_client_gen_binary_key": (_i, [_RP, _u32, _sz, _vp]),
    "camd_client_gen_bootstrap_key": (_i, [_RP, _PP, _dbl, _vp, _vp, _vp, _i]),
    "camd_client_gen_keyswitch_key": (_i, [_RP, _u32, _vp, _sz, _vp, _sz, _u32, _u32, _dbl, _vp]),
    "camd_client_lwe_encrypt": (_i, [_RP, _u32, _vp, _sz, _vp, _sz, _dbl, _vp]),
    "camd_client_lwe_phase": (_i, [_u32, _vp, _sz, _vp, _sz, _vp]),
    "camd_client_shortint_accumulator": (_i, [_u32, _u32, _u32, _u32, _vp, _vp]),
    "camd_client_boolean_accumulator": (_i, [_u32, _u32, _vp]),
})

_lib = None


def _preload_torch_hip_runtime() -> None:
    """One process, one HIP runtime. PyTorch-ROCm ships its own libamdhip64; if this library were
    loaded first it would bind to the system copy and torch would later bring a second, mismatched
    runtime (its device enumeration then fails). Loading torch's copy first makes both share it.
    A caller that never uses torch (a C or Rust host) simply gets the system runtime."""
    import importlib.util

    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return
    for name in ("libamdhip64.so",):
        path = os.path.join(list(spec.submodule_search_locations)[0], "lib", name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                pass


def lib() -> C.CDLL:
    """Load the shared library (once). Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "to compile the HIP kernels for gfx950. There is no CPU fallback.")
        _preload_torch_hip_runtime()
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        msg = lib().camd_last_error()
        raise CamdError(rc, msg.decode() if msg else "")


def make_params(p, torus_bits: int | None = None) -> CamdParams:
    bits = torus_bits if torus_bits is not None else int(getattr(p, "torus_bits", 64))
    return CamdParams(int(p.lwe_dimension), int(p.glwe_dimension), int(p.polynomial_size), int(p.pbs_base_log),
                      int(p.pbs_level), int(p.ks_base_log), int(p.ks_level), bits)
