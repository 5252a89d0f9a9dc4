"""``UMat``: a device-resident (HBM) array handle -- the stand-in for ``cv.UMat``.

The reference wraps arrays in ``cv.UMat`` so that OpenCV keeps them on the OpenCL device
(stitching_detailed_enhanced.py:1539-1541, :1886) and calls ``.get()`` to read them back (:1599).
Here a ``UMat`` owns an ``ssp_image`` handle; passing a ``UMat`` into warp / apply / feed keeps the whole
chain on the GPU, passing an ``ndarray`` uploads it for the duration of the call (cv2 semantics).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _lib

U8, S16, F32 = 0, 3, 5
_DEPTH_OF = {np.dtype(np.uint8): U8, np.dtype(np.int16): S16, np.dtype(np.float32): F32}
_DTYPE_OF = {U8: np.uint8, S16: np.int16, F32: np.float32}


class UMat:
    _ver = 0      # bumped by in-place writers (compensator.apply, SeamFinder.find): deferred results remember their operands' versions (deferred.py)

    def __init__(self, array: Optional[np.ndarray] = None, _handle: Optional[int] = None):
        self._h = None
        if _handle is not None:
            self._h = C.c_void_p(_handle)
        elif array is not None:
            if isinstance(array, UMat):
                _lib.check(_lib.lib().ssp_image_retain(array._h))
                self._h = C.c_void_p(array._h.value)
                return
            a = np.ascontiguousarray(array)
            if a.dtype not in _DEPTH_OF or a.ndim not in (2, 3):
                raise _lib.error(f"UMat: unsupported array {a.dtype} with shape {a.shape}")
            h, w = a.shape[:2]
            cn = 1 if a.ndim == 2 else a.shape[2]
            out = C.c_void_p()
            _lib.check(_lib.lib().ssp_image_upload(a.ctypes.data, w, h, cn, _DEPTH_OF[a.dtype], C.byref(out)))
            self._h = out
        else:
            raise _lib.error("UMat() needs an array")

    @classmethod
    def from_handle(cls, handle) -> "UMat":
        return cls(_handle=handle.value if isinstance(handle, C.c_void_p) else int(handle))

    @classmethod
    def empty(cls, width: int, height: int, channels: int, dtype) -> "UMat":
        out = C.c_void_p()
        _lib.check(_lib.lib().ssp_image_create(width, height, channels, _DEPTH_OF[np.dtype(dtype)], C.byref(out)))
        return cls.from_handle(out)

    @classmethod
    def wrap_device(cls, dev_ptr: int, width: int, height: int, channels: int, dtype, pitch: int = 0) -> "UMat":
        """Borrow an existing device allocation (e.g. ``torch.Tensor.data_ptr()``); not freed on release."""
        out = C.c_void_p()
        _lib.check(_lib.lib().ssp_image_wrap(C.c_void_p(dev_ptr), pitch, width, height, channels, _DEPTH_OF[np.dtype(dtype)], C.byref(out)))
        return cls.from_handle(out)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                _lib.lib().ssp_image_release(h)
            except Exception:  # interpreter shutdown
                pass
            self._h = None

    def info(self) -> Tuple[int, int, int, int, int, int]:
        """(width, height, channels, depth code, row pitch in bytes, device pointer): fixed for the life of the handle, so asked once."""
        cached = self.__dict__.get("_info")
        if cached is None:
            cached = self.__dict__["_info"] = self._query_info()
        return cached

    def _query_info(self) -> Tuple[int, int, int, int, int, int]:
        w, h, cn, d = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        pitch = C.c_size_t()
        ptr = C.c_void_p()
        _lib.check(_lib.lib().ssp_image_info(self._h, C.byref(w), C.byref(h), C.byref(cn), C.byref(d), C.byref(pitch), C.byref(ptr)))
        return w.value, h.value, cn.value, d.value, pitch.value, ptr.value

    @property
    def shape(self):
        w, h, cn, _, _, _ = self.info()
        return (h, w) if cn == 1 else (h, w, cn)

    @property
    def dtype(self):
        return np.dtype(_DTYPE_OF[self.info()[3]])

    def get(self) -> np.ndarray:
        w, h, cn, d, _, _ = self.info()
        out = np.empty((h, w) if cn == 1 else (h, w, cn), _DTYPE_OF[d])
        _lib.check(_lib.lib().ssp_image_download(self._h, out.ctypes.data))
        return out

    def astype(self, dtype) -> "UMat":
        """``ndarray.astype`` for the conversions on the path (uint8 -> int16 at sde.py:1755)."""
        out = C.c_void_p()
        _lib.check(_lib.lib().ssp_image_convert(self._h, _DEPTH_OF[np.dtype(dtype)], C.byref(out)))
        return UMat.from_handle(out)


def as_umat(x) -> Tuple[UMat, bool]:
    """Return (UMat, was_device): ndarrays (and anything exposing ``.get()``) are uploaded."""
    if isinstance(x, UMat):
        return x, True
    if not isinstance(x, np.ndarray) and hasattr(x, "get"):
        x = x.get()
    return UMat(np.asarray(x)), False
