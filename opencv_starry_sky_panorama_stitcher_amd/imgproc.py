"""The image operators around warp and feed, with cv2 names: the mask preparation between warp and feed
(stitching_detailed_enhanced.py:1760-1772: ``dilate(mask, None)``, ``resize(mask, dsize, 0, 0, INTER_LINEAR_EXACT)``,
``bitwise_and``) and the frame prologue before the warp (sde.py:1699-1711: ``resize(img, None, fx, fy, INTER_AREA)``,
``adjust_black_and_white_point``).
ndarray in -> ndarray out, UMat in -> UMat out."""
from __future__ import annotations

import ctypes as C

from . import _lib
from . import deferred as _deferred
from .umat import UMat, as_umat

INTER_NEAREST, INTER_LINEAR, INTER_CUBIC, INTER_AREA, INTER_LANCZOS4, INTER_LINEAR_EXACT = 0, 1, 2, 3, 4, 5
BORDER_CONSTANT, BORDER_REPLICATE, BORDER_REFLECT, BORDER_WRAP, BORDER_REFLECT_101 = 0, 1, 2, 3, 4


def dilate(src, kernel=None):
    if kernel is not None:
        raise _lib.error("dilate: only the default 3x3 rectangular kernel (kernel=None, sde.py:1760-1764) is implemented")
    if isinstance(src, UMat) and _deferred.enabled():
        w, h, cn, depth = src.info()[:4]
        if cn == 1 and depth == 0:
            return _deferred.DeferredUMat("dilate", (src,), w, h, 1, "uint8")
    s, dev = as_umat(src)
    out = C.c_void_p()
    _lib.check(_lib.lib().ssp_dilate3x3(s._h, C.byref(out)))
    d = UMat.from_handle(out)
    return d if dev else d.get()


def resize(src, dsize, fx=0, fy=0, interpolation=INTER_LINEAR_EXACT):
    """The two cv.resize calls on the path: INTER_LINEAR_EXACT of an 8UC1 mask to ``dsize`` (sde.py:1767-1768) and the
    INTER_AREA decimation of the full frame by ``fx, fy`` (sde.py:1701-1707; ``dsize`` None or (0, 0))."""
    if interpolation == INTER_AREA:
        if dsize is not None and tuple(dsize) != (0, 0):
            raise _lib.error("resize(INTER_AREA): pass dsize=None with fx/fy (sde.py:1701-1707); explicit dsize is not implemented")
        return _resize_area(src, fx, fy, None)
    if interpolation != INTER_LINEAR_EXACT:
        raise _lib.error("resize: only INTER_LINEAR_EXACT on 8UC1 masks (sde.py:1767-1768) and INTER_AREA decimation (sde.py:1701) are implemented")
    if isinstance(src, UMat) and _deferred.enabled() and int(dsize[0]) > 0 and int(dsize[1]) > 0:
        _, _, cn, depth = src.info()[:4]
        if cn == 1 and depth == 0:
            return _deferred.DeferredUMat("resize_exact", (src, (int(dsize[0]), int(dsize[1]))), int(dsize[0]), int(dsize[1]), 1, "uint8")
    s, dev = as_umat(src)
    out = C.c_void_p()
    _lib.check(_lib.lib().ssp_resize_linear_exact(s._h, int(dsize[0]), int(dsize[1]), C.byref(out)))
    d = UMat.from_handle(out)
    return d if dev else d.get()


def _bw_lut(black_and_white_point_tpl):
    lut = (C.c_uint8 * 256)()
    black, white = black_and_white_point_tpl
    _lib.check(_lib.lib().ssp_bw_point_lut(int(black), int(white), lut))
    return lut


def _resize_area(src, fx, fy, lut):
    s, dev = as_umat(src)
    out = C.c_void_p()
    _lib.check(_lib.lib().ssp_resize_area(s._h, float(fx), float(fy), lut, C.byref(out)))
    d = UMat.from_handle(out)
    return d if dev else d.get()


def adjust_black_and_white_point(img, black_and_white_point_tpl):
    """image_processors.py:32-41 -- clip to [black, white] and stretch to 0..255 (truncating); a falsy tuple returns ``img``."""
    if not black_and_white_point_tpl:
        return img
    s, dev = as_umat(img)
    out = C.c_void_p()
    _lib.check(_lib.lib().ssp_apply_lut(s._h, _bw_lut(black_and_white_point_tpl), C.byref(out)))
    d = UMat.from_handle(out)
    return d if dev else d.get()


def prepare_frame(full_img, compose_scale, black_and_white_point_tpl=None):
    """The frame prologue of the compose loop in one pass (sde.py:1699-1711): INTER_AREA decimation by ``compose_scale`` when
    ``abs(compose_scale - 1) > 1e-1``, then the black / white point stretch."""
    lut = _bw_lut(black_and_white_point_tpl) if black_and_white_point_tpl else None
    if abs(compose_scale - 1) > 1e-1:
        return _resize_area(full_img, compose_scale, compose_scale, lut)
    return adjust_black_and_white_point(full_img, black_and_white_point_tpl)


def bitwise_and(a, b, dst=None, mask=None):
    """cv.bitwise_and(a, b) (sde.py:1772) and cv.bitwise_and(a, b, mask=mask) (sde.py:1842: zero where the mask is zero)."""
    if mask is None and isinstance(a, UMat) and isinstance(b, UMat) and _deferred.enabled() and (_deferred.is_pending(a) or _deferred.is_pending(b)):
        ia, ib = a.info()[:4], b.info()[:4]
        if ia == ib and ia[2] == 1 and ia[3] == 0:
            return _deferred.DeferredUMat("and", (a, b), ia[0], ia[1], 1, "uint8")
    ua, da = as_umat(a)
    ub, db = as_umat(b)
    out = C.c_void_p()
    if mask is None:
        _lib.check(_lib.lib().ssp_bitwise_and(ua._h, ub._h, C.byref(out)))
        dm = False
    else:
        um, dm = as_umat(mask)
        _lib.check(_lib.lib().ssp_bitwise_and_masked(ua._h, ub._h, um._h, C.byref(out)))
    d = UMat.from_handle(out)
    return d if (da or db or dm) else d.get()
