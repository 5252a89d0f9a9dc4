"""``PyRotationWarper``: host-side mirror of ``cv.PyRotationWarper`` over the HIP library.

Reference usage: stitching_detailed_enhanced.py:1545-1546 / :1684-1688 (construction), :1696 (warpRoi),
:1557 / :1731 (warp image: INTER_AREA|INTER_LINEAR + BORDER_REFLECT), :1591 / :1740 (warp mask:
INTER_NEAREST + BORDER_CONSTANT).  Same names, argument meaning and error behaviour as cv2: K and R must be
3x3 float32, an unknown type string raises, ndarray in -> ndarray out, UMat in -> UMat out.
"""
from __future__ import annotations

import ctypes as C
from typing import Tuple

import numpy as np

from . import _lib
from .umat import UMat, as_umat

WARP_TYPES = (
    "plane", "affine", "cylindrical", "spherical", "fisheye", "stereographic",
    "compressedPlaneA2B1", "compressedPlaneA1.5B1", "compressedPlanePortraitA2B1", "compressedPlanePortraitA1.5B1",
    "paniniA2B1", "paniniA1.5B1", "paniniPortraitA2B1", "paniniPortraitA1.5B1", "mercator", "transverseMercator",
)  # sde.py:218-237


_NP_OF_DEPTH = {0: np.dtype(np.uint8), 3: np.dtype(np.int16), 5: np.dtype(np.float32)}


def _mat3(a, name: str):
    a = np.asarray(a)
    if a.shape != (3, 3) or a.dtype != np.float32:
        # cv2: "K.size() == Size(3, 3) && K.type() == CV_32F" assertion (hence the astype at sde.py:1550, :1695)
        raise _lib.error(f"{name} must be a 3x3 float32 array (got shape {a.shape}, dtype {a.dtype})")
    a = np.ascontiguousarray(a)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


class PyRotationWarper:
    def __init__(self, type: str, scale: float):
        self._h = C.c_void_p()
        _lib.check(_lib.lib().ssp_warper_create(str(type).encode(), float(scale), C.byref(self._h)))
        self.type = type
        self._scale = float(np.float32(scale))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                _lib.lib().ssp_warper_destroy(h)
            except Exception:
                pass
            self._h = None

    # -- cv2 API ------------------------------------------------------------------------------------------
    def getScale(self) -> float:
        return self._scale

    def setScale(self, scale: float) -> None:
        _lib.check(_lib.lib().ssp_warper_set_scale(self._h, float(scale)))
        self._scale = float(np.float32(scale))

    def warpRoi(self, src_size: Tuple[int, int], K, R) -> Tuple[int, int, int, int]:
        _, kp = _mat3(K, "K")
        _, rp = _mat3(R, "R")
        roi = (C.c_int * 4)()
        _lib.check(_lib.lib().ssp_warper_roi(self._h, int(src_size[0]), int(src_size[1]), kp, rp, roi))
        return tuple(roi)

    def liveParts(self, src_size: Tuple[int, int], K, R, num_bands: int):
        """(no cv2 counterpart) the rectangles (x, y, w, h) of ``warpRoi`` that a ``num_bands`` multiband blender has to see: the roi itself,
        or -- a frame that straddles u = +-pi*scale, whose roi spans the full circle -- its two live column ranges grown by 4 * 2^num_bands."""
        _, kp = _mat3(K, "K")
        _, rp = _mat3(R, "R")
        parts, n = (C.c_int * 8)(), C.c_int()
        _lib.check(_lib.lib().ssp_warper_live_parts(self._h, int(src_size[0]), int(src_size[1]), kp, rp, int(num_bands), parts, 2, C.byref(n)))
        return [tuple(parts[4 * k:4 * k + 4]) for k in range(n.value)]

    def warp(self, src, K, R, interp_mode: int, border_mode: int, dst=None):
        """ndarray in -> ndarray out.  UMat in -> a deferred UMat (deferred.py): its size and corner are known now (warpRoi), its pixels are
        computed by this very call's kernel when somebody reads them -- or, when it reaches ``blender.feed`` the way the reference's loop
        feeds it (sde.py:1731-1886), inside the fused warp of ``blender.blend``."""
        from . import deferred
        if isinstance(src, UMat) and deferred.enabled() and int(interp_mode) in (0, 1, 3) and 0 <= int(border_mode) <= 4:
            k, kp = _mat3(K, "K")
            r, rp = _mat3(R, "R")
            w, h, cn, depth = src.info()[:4]
            roi = (C.c_int * 4)()
            _lib.check(_lib.lib().ssp_warper_roi(self._h, w, h, kp, rp, roi))      # (a process-wide cache in the library: one scan per camera, not per call)
            out = deferred.DeferredUMat("warp", (self, src, k.copy(), r.copy(), int(interp_mode), int(border_mode)), roi[2], roi[3], cn, _NP_OF_DEPTH[depth])
            out.corner = (roi[0], roi[1])
            return out.corner, out
        return self._warp_now(src, K, R, interp_mode, border_mode)

    def _warp_now(self, src, K, R, interp_mode: int, border_mode: int):
        k, kp = _mat3(K, "K")
        r, rp = _mat3(R, "R")
        s, on_device = as_umat(src)
        out = C.c_void_p()
        corner = (C.c_int * 2)()
        _lib.check(_lib.lib().ssp_warper_warp_image(self._h, s._h, kp, rp, int(interp_mode), int(border_mode), C.byref(out), corner))
        d = UMat.from_handle(out)
        return (corner[0], corner[1]), (d if on_device else d.get())

    def warpBackward(self, src, K, R, interp_mode: int, border_mode: int, dst_size, dst=None):
        """cv.PyRotationWarper.warpBackward: ``src`` is a warped image of size ``warpRoi(dst_size, K, R)``; returns the frame."""
        _, kp = _mat3(K, "K")
        _, rp = _mat3(R, "R")
        s, on_device = as_umat(src)
        out = C.c_void_p()
        _lib.check(_lib.lib().ssp_warper_warp_backward(self._h, s._h, kp, rp, int(interp_mode), int(border_mode), int(dst_size[0]), int(dst_size[1]), C.byref(out)))
        d = UMat.from_handle(out)
        return d if on_device else d.get()

    def buildMaps(self, src_size: Tuple[int, int], K, R, xmap=None, ymap=None):
        _, kp = _mat3(K, "K")
        _, rp = _mat3(R, "R")
        x, y, w, h = self.warpRoi(src_size, K, R)
        xm = np.empty((h, w), np.float32)
        ym = np.empty((h, w), np.float32)
        roi = (C.c_int * 4)()
        fp = C.POINTER(C.c_float)
        _lib.check(_lib.lib().ssp_warper_build_maps(self._h, int(src_size[0]), int(src_size[1]), kp, rp, xm.ctypes.data_as(fp), ym.ctypes.data_as(fp), w, h, roi))
        # cv2 returns Rect(tl, br) whose width is br.x - tl.x, i.e. one less than the map size
        return (x, y, w - 1, h - 1), xm, ym

    def warpPoint(self, pt, K, R):
        _, kp = _mat3(K, "K")
        _, rp = _mat3(R, "R")
        uv = (C.c_float * 2)()
        _lib.check(_lib.lib().ssp_warper_warp_point(self._h, float(pt[0]), float(pt[1]), kp, rp, uv))
        return (uv[0], uv[1])

    def warpPointBackward(self, pt, K, R):
        _, kp = _mat3(K, "K")
        _, rp = _mat3(R, "R")
        xy = (C.c_float * 2)()
        _lib.check(_lib.lib().ssp_warper_warp_point_backward(self._h, float(pt[0]), float(pt[1]), kp, rp, xy))
        return (xy[0], xy[1])

    # -- fused device-resident form of the two warp calls at sde.py:1731 + :1740 ---------------------------------
    def warpWithMask(self, src, K, R, border_mode: int = 2):
        """image (INTER_LINEAR, ``border_mode``) and the all-255 mask (INTER_NEAREST, BORDER_CONSTANT) in one pass."""
        _, kp = _mat3(K, "K")
        _, rp = _mat3(R, "R")
        s, on_device = as_umat(src)
        d, m = C.c_void_p(), C.c_void_p()
        corner = (C.c_int * 2)()
        _lib.check(_lib.lib().ssp_warper_warp_with_mask(self._h, s._h, kp, rp, int(border_mode), C.byref(d), C.byref(m), corner))
        du, mu = UMat.from_handle(d), UMat.from_handle(m)
        if on_device:
            return (corner[0], corner[1]), du, mu
        return (corner[0], corner[1]), du.get(), mu.get()
