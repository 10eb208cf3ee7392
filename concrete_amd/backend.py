"""Thin object layer over the C ABI: one ``Context`` per GPU, ``DeviceServerKey`` per key.

Host entry points take numpy arrays; device entry points take torch CUDA tensors (torch is
only the owner of device memory and streams here). Everything computes in the HIP library.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import CAMD_PBS_AUTO, CAMD_PBS_BLOCK, CAMD_PBS_GENERIC, CAMD_PBS_WAVE, check  # noqa: F401


def _np_ptr(a: np.ndarray):
    if not a.flags["C_CONTIGUOUS"]:
        raise ValueError("array must be C-contiguous")
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """One per (process, GPU). Replaces the reference's thread-local engines."""

    def __init__(self, device_id: int = 0):
        self._h = C.c_void_p()
        check(_ffi.lib().camd_context_create(int(device_id), C.byref(self._h)))
        self.device_id = int(device_id)

    def synchronize(self) -> None:
        check(_ffi.lib().camd_context_synchronize(self._h))

    def lwe_axpby_device(self, out, x, alpha: int, y=None, beta: int = 0, body_add: int = 0, torus_bits: int = 64,
                         stream: int = 0) -> None:
        """out = alpha*x + beta*y (+ body_add on the last word of each row), wrapping, on the GPU.
        ``out``, ``x``, ``y`` are 2-D torch CUDA tensors [count][words] whose rows may be strided
        (a block of a radix batch is such a view); the last dimension must be contiguous."""
        for t in (out, x) + ((y,) if y is not None else ()):
            if t.dim() != 2 or t.stride(1) != 1 or t.shape != out.shape or t.element_size() * 8 != torus_bits:
                raise ValueError("lwe_axpby_device: tensors must be [count][words] of the torus width, rows contiguous")
        mask = (1 << 64) - 1
        check(_ffi.lib().camd_lwe_axpby_device(
            self._h, int(torus_bits), C.c_void_p(out.data_ptr()), out.stride(0), C.c_void_p(x.data_ptr()), x.stride(0),
            int(alpha) & mask, C.c_void_p(y.data_ptr()) if y is not None else None, y.stride(0) if y is not None else 0,
            int(beta) & mask, int(body_add) & mask, out.shape[1], out.shape[0], C.c_void_p(stream)))

    def measure_fp64_fma_rate(self) -> float:
        """Sustained v_fma_f64 rate of this GPU in TFLOP/s (a 5 ms probe kernel)."""
        t = C.c_double()
        check(_ffi.lib().camd_measure_fp64_fma_rate(self._h, C.byref(t)))
        return t.value

    def close(self) -> None:
        if self._h:
            _ffi.lib().camd_context_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def fourier_layout_id(params, variant: int, torus_bits: int | None = None) -> int:
    """The library's id for the layout of the Fourier bootstrap key of ``params`` under kernel
    family ``variant``; a stored blob is only valid while this value is the one it was stored with."""
    cp = _ffi.make_params(params, torus_bits)
    out = C.c_uint64()
    check(_ffi.lib().camd_fourier_layout_id(C.byref(cp), int(variant), C.byref(out)))
    return out.value


class DeviceServerKey:
    """Fourier bootstrap key + keyswitch key + look-up tables resident on one GPU."""

    def __init__(self, ctx: Context, handle: C.c_void_p, params, torus_bits: int):
        self.ctx = ctx
        self._h = handle
        self.params = params
        self.torus_bits = torus_bits
        self.dtype = np.uint64 if torus_bits == 64 else np.uint32
        self.n = int(params.lwe_dimension)
        self.big = int(params.glwe_dimension) * int(params.polynomial_size)
        self.glwe_words = (int(params.glwe_dimension) + 1) * int(params.polynomial_size)
        self.launches = 0      # batch calls made through this object (host or device entry points)
        self.bootstraps = 0    # ciphertexts in them

    # ---- construction ----
    @classmethod
    def from_standard_keys(cls, ctx: Context, params, bsk_standard: np.ndarray, ksk: np.ndarray,
                           variant: int = CAMD_PBS_AUTO, torus_bits: int | None = None) -> "DeviceServerKey":
        bits = torus_bits if torus_bits is not None else int(getattr(params, "torus_bits", 64))
        dtype = np.uint64 if bits == 64 else np.uint32
        n, k, N = int(params.lwe_dimension), int(params.glwe_dimension), int(params.polynomial_size)
        bsk = np.ascontiguousarray(bsk_standard, dtype=dtype)
        ks = np.ascontiguousarray(ksk, dtype=dtype)
        if bsk.size != n * int(params.pbs_level) * (k + 1) * (k + 1) * N:
            raise ValueError("bootstrap key has the wrong number of words for the parameter set")
        if ks.size != k * N * int(params.ks_level) * (n + 1):
            raise ValueError("keyswitch key has the wrong number of words for the parameter set")
        cp = _ffi.make_params(params, bits)
        h = C.c_void_p()
        check(_ffi.lib().camd_server_key_create(ctx._h, C.byref(cp), _np_ptr(bsk), _np_ptr(ks), int(variant), C.byref(h)))
        return cls(ctx, h, params, bits)

    @classmethod
    def from_device_blob(cls, ctx: Context, params, blob, variant: int, torus_bits: int | None = None,
                         stream: int = 0) -> "DeviceServerKey":
        """Adopt a copy of a key blob that is already on this GPU (a torch uint8 CUDA tensor)."""
        bits = torus_bits if torus_bits is not None else int(getattr(params, "torus_bits", 64))
        cp = _ffi.make_params(params, bits)
        h = C.c_void_p()
        check(_ffi.lib().camd_server_key_import_blob(ctx._h, C.byref(cp), int(variant), C.c_void_p(blob.data_ptr()),
                                                     blob.numel() * blob.element_size(), C.c_void_p(stream), C.byref(h)))
        return cls(ctx, h, params, bits)

    @property
    def variant(self) -> int:
        v = C.c_int()
        check(_ffi.lib().camd_server_key_variant(self._h, C.byref(v)))
        return v.value

    def layout_id(self) -> int:
        """Identifies the order of the Fourier key inside this key's blob (see ``fourier_layout_id``)."""
        return fourier_layout_id(self.params, self.variant, self.torus_bits)

    def blob_size(self) -> int:
        s = C.c_size_t()
        check(_ffi.lib().camd_server_key_blob_size(self._h, C.byref(s)))
        return s.value

    def export_blob(self, dst, stream: int = 0) -> None:
        """Copy the device-resident key material into ``dst`` (torch uint8 CUDA tensor)."""
        if dst.numel() * dst.element_size() < self.blob_size():
            raise ValueError("destination tensor is smaller than the key blob")
        check(_ffi.lib().camd_server_key_export_blob(self._h, C.c_void_p(dst.data_ptr()), C.c_void_p(stream)))

    def export_blob_host(self) -> np.ndarray:
        """The key material as host bytes (uint8 array): what a key file stores."""
        out = np.empty(self.blob_size(), dtype=np.uint8)
        check(_ffi.lib().camd_server_key_export_blob(self._h, _np_ptr(out), C.c_void_p(0)))
        self.ctx.synchronize()
        return out

    @classmethod
    def from_host_blob(cls, ctx: Context, params, blob: np.ndarray, variant: int,
                       torus_bits: int | None = None) -> "DeviceServerKey":
        """Rebuild a key from ``export_blob_host`` bytes; ``variant`` must be the exporting key's."""
        bits = torus_bits if torus_bits is not None else int(getattr(params, "torus_bits", 64))
        data = np.ascontiguousarray(blob, dtype=np.uint8)
        cp = _ffi.make_params(params, bits)
        h = C.c_void_p()
        check(_ffi.lib().camd_server_key_import_blob(ctx._h, C.byref(cp), int(variant), _np_ptr(data), data.size,
                                                     C.c_void_p(0), C.byref(h)))
        return cls(ctx, h, params, bits)

    def close(self) -> None:
        if self._h:
            _ffi.lib().camd_server_key_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- look-up tables ----
    def upload_lut(self, glwe: np.ndarray) -> int:
        g = np.ascontiguousarray(glwe, dtype=self.dtype)
        if g.size != self.glwe_words:
            raise ValueError("look-up table must hold (k+1)*N words")
        lut_id = C.c_uint32()
        check(_ffi.lib().camd_lut_upload(self._h, _np_ptr(g), C.byref(lut_id)))
        return lut_id.value

    @property
    def lut_count(self) -> int:
        c = C.c_uint32()
        check(_ffi.lib().camd_lut_count(self._h, C.byref(c)))
        return c.value

    # ---- host batches ----
    def _host(self, fn, ct_in, lut_ids, in_words, out_words, with_ids=True):
        a = np.ascontiguousarray(ct_in, dtype=self.dtype)
        if a.ndim == 1:
            a = a[None, :]
        if a.ndim != 2 or a.shape[1] != in_words:
            raise ValueError(f"expected ciphertexts of {in_words} words, got shape {a.shape}")
        count = a.shape[0]
        self.launches += 1
        self.bootstraps += count
        out = np.empty((count, out_words), dtype=self.dtype)
        if with_ids:
            ids = None
            if lut_ids is not None:
                ids = np.ascontiguousarray(lut_ids, dtype=np.uint32)
                if ids.shape != (count,):
                    raise ValueError("lut_ids must have one entry per ciphertext")
            check(fn(self._h, _np_ptr(a), _np_ptr(ids) if ids is not None else None, _np_ptr(out), count))
        else:
            check(fn(self._h, _np_ptr(a), _np_ptr(out), count))
        return out

    def ks_pbs(self, ct_in, lut_ids=None) -> np.ndarray:
        return self._host(_ffi.lib().camd_ks_pbs_batch, ct_in, lut_ids, self.big + 1, self.big + 1)

    def pbs_ks(self, ct_in, lut_ids=None) -> np.ndarray:
        return self._host(_ffi.lib().camd_pbs_ks_batch, ct_in, lut_ids, self.n + 1, self.n + 1)

    def pbs(self, ct_in, lut_ids=None) -> np.ndarray:
        return self._host(_ffi.lib().camd_pbs_batch, ct_in, lut_ids, self.n + 1, self.big + 1)

    def ks(self, ct_in) -> np.ndarray:
        return self._host(_ffi.lib().camd_ks_batch, ct_in, None, self.big + 1, self.n + 1, with_ids=False)

    # ---- device batches (torch tensors own the memory) ----
    def reserve(self, count: int) -> None:
        check(_ffi.lib().camd_reserve(self._h, int(count)))

    def check(self, stream: int = 0) -> None:
        """Synchronise ``stream`` and raise if a device batch since the last check used a look-up
        table id that is out of range (such ciphertexts were bootstrapped with table 0)."""
        check(_ffi.lib().camd_server_key_check(self._h, C.c_void_p(stream)))

    def _dev(self, fn, ct_in, lut_ids, ct_out, count, stream, with_ids=True):
        ids_ptr = C.c_void_p(lut_ids.data_ptr()) if lut_ids is not None else None
        self.launches += 1
        self.bootstraps += int(count)
        if with_ids:
            check(fn(self._h, C.c_void_p(ct_in.data_ptr()), ids_ptr, C.c_void_p(ct_out.data_ptr()), int(count),
                     C.c_void_p(stream)))
        else:
            check(fn(self._h, C.c_void_p(ct_in.data_ptr()), C.c_void_p(ct_out.data_ptr()), int(count), C.c_void_p(stream)))

    def ks_pbs_device(self, ct_in, lut_ids, ct_out, count: int, stream: int = 0) -> None:
        self._dev(_ffi.lib().camd_ks_pbs_batch_device, ct_in, lut_ids, ct_out, count, stream)

    def ks_pbs_device_noalloc(self, ct_in, lut_ids, ct_out, count: int, stream: int = 0) -> None:
        """ks_pbs_device through the entry point that allocates nothing (fixed scratch, bounded batch)."""
        self._dev(_ffi.lib().camd_ks_pbs_batch_device_noalloc, ct_in, lut_ids, ct_out, count, stream)

    def pbs_ks_device(self, ct_in, lut_ids, ct_out, count: int, stream: int = 0) -> None:
        self._dev(_ffi.lib().camd_pbs_ks_batch_device, ct_in, lut_ids, ct_out, count, stream)

    def pbs_device(self, ct_in, lut_ids, ct_out, count: int, stream: int = 0) -> None:
        self._dev(_ffi.lib().camd_pbs_batch_device, ct_in, lut_ids, ct_out, count, stream)

    def _combined(self, fn, words, x, alpha, y, beta, body_add, lut_ids, ct_out, stream):
        for t in (x,) + ((y,) if y is not None else ()):
            if t.dim() != 2 or t.stride(1) != 1 or t.shape[1] != words or t.shape[0] != x.shape[0]:
                raise ValueError("combined batch: operands must be [count][words] tensors with contiguous rows")
        if ct_out.dim() != 2 or not ct_out.is_contiguous() or ct_out.shape != (x.shape[0], words):
            raise ValueError("combined batch: the output must be a contiguous [count][words] tensor")
        mask = (1 << 64) - 1
        count = x.shape[0]
        self.launches += 1
        self.bootstraps += int(count)
        check(fn(self._h, C.c_void_p(x.data_ptr()), x.stride(0), int(alpha) & mask,
                 C.c_void_p(y.data_ptr()) if y is not None else None, y.stride(0) if y is not None else 0, int(beta) & mask,
                 int(body_add) & mask, C.c_void_p(lut_ids.data_ptr()) if lut_ids is not None else None,
                 C.c_void_p(ct_out.data_ptr()), int(count), C.c_void_p(stream)))

    def axpby_ks_pbs_device(self, x, alpha: int, y, beta: int, body_add: int, lut_ids, ct_out, stream: int = 0) -> None:
        """ks_pbs_device on the rows alpha*x + beta*y (+ body_add on the body), formed by the keyswitch
        while it reads; ``x`` and ``y`` (or None) may be row-strided views."""
        self._combined(_ffi.lib().camd_axpby_ks_pbs_batch_device, self.big + 1, x, alpha, y, beta, body_add, lut_ids, ct_out,
                       stream)

    def axpby_pbs_ks_device(self, x, alpha: int, y, beta: int, body_add: int, lut_ids, ct_out, stream: int = 0) -> None:
        """pbs_ks_device on the rows alpha*x + beta*y (+ body_add on the body)."""
        self._combined(_ffi.lib().camd_axpby_pbs_ks_batch_device, self.n + 1, x, alpha, y, beta, body_add, lut_ids, ct_out,
                       stream)

    def ks_device(self, ct_in, ct_out, count: int, stream: int = 0) -> None:
        self._dev(_ffi.lib().camd_ks_batch_device, ct_in, None, ct_out, count, stream, with_ids=False)

    # ---- measurement ----
    def timing_enable(self, on: bool = True) -> None:
        check(_ffi.lib().camd_timing_enable(self._h, 1 if on else 0))

    def timing_read(self):
        pm, km = C.c_double(), C.c_double()
        pn, kn = C.c_uint64(), C.c_uint64()
        check(_ffi.lib().camd_timing_read(self._h, C.byref(pm), C.byref(pn), C.byref(km), C.byref(kn)))
        return {"pbs_ms": pm.value, "pbs_launches": pn.value, "ks_ms": km.value, "ks_launches": kn.value}
