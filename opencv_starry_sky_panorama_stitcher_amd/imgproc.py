"""The three mask-preparation calls between warp and feed (stitching_detailed_enhanced.py:1760-1772),
with cv2 names: ``dilate(mask, None)``, ``resize(mask, dsize, 0, 0, INTER_LINEAR_EXACT)``, ``bitwise_and``.
ndarray in -> ndarray out, UMat in -> UMat out."""
from __future__ import annotations

import ctypes as C

from . import _lib
from .umat import UMat, as_umat

INTER_NEAREST, INTER_LINEAR, INTER_CUBIC, INTER_AREA, INTER_LANCZOS4, INTER_LINEAR_EXACT = 0, 1, 2, 3, 4, 5
BORDER_CONSTANT, BORDER_REPLICATE, BORDER_REFLECT, BORDER_WRAP, BORDER_REFLECT_101 = 0, 1, 2, 3, 4


def dilate(src, kernel=None):
    if kernel is not None:
        raise _lib.error("dilate: only the default 3x3 rectangular kernel (kernel=None, sde.py:1760-1764) is implemented")
    s, dev = as_umat(src)
    out = C.c_void_p()
    _lib.check(_lib.lib().ssp_dilate3x3(s._h, C.byref(out)))
    d = UMat.from_handle(out)
    return d if dev else d.get()


def resize(src, dsize, fx=0, fy=0, interpolation=INTER_LINEAR_EXACT):
    if interpolation != INTER_LINEAR_EXACT:
        raise _lib.error("resize: only INTER_LINEAR_EXACT on 8UC1 masks (sde.py:1767-1768) is implemented on this path")
    s, dev = as_umat(src)
    out = C.c_void_p()
    _lib.check(_lib.lib().ssp_resize_linear_exact(s._h, int(dsize[0]), int(dsize[1]), C.byref(out)))
    d = UMat.from_handle(out)
    return d if dev else d.get()


def bitwise_and(a, b):
    ua, da = as_umat(a)
    ub, db = as_umat(b)
    out = C.c_void_p()
    _lib.check(_lib.lib().ssp_bitwise_and(ua._h, ub._h, C.byref(out)))
    d = UMat.from_handle(out)
    return d if (da or db) else d.get()
