"""ctypes front-end of the CPU ORACLE (test infrastructure, not product code).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this.
The classes mirror the cv2 objects used at stitching_detailed_enhanced.py:1545-1930 so that
parity tests read like the reference's own call sequence.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

INTER_NEAREST, INTER_LINEAR, INTER_AREA = 0, 1, 3
BORDER_CONSTANT, BORDER_REPLICATE, BORDER_REFLECT, BORDER_WRAP, BORDER_REFLECT_101 = 0, 1, 2, 3, 4
U8, S16, F32 = 0, 3, 5
BLEND_NO, BLEND_FEATHER, BLEND_MULTIBAND = 0, 1, 2
COMP_NO, COMP_GAIN, COMP_GAIN_BLOCKS, COMP_CHANNELS, COMP_CHANNELS_BLOCKS = 0, 1, 2, 3, 4

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int)


def build(force: bool = False) -> None:
    """Compile oracle/*.c with gcc (building the checker is not using it)."""
    if force or not all(os.path.exists(os.path.join(_HERE, n)) for n in ("liborc.so", "liborc_libm.so", "liborc_omp.so")):
        subprocess.run(["make", "-C", _HERE, "-s"] + (["-B"] if force else []), check=True)


def _load(name: str) -> C.CDLL:
    path = os.path.join(_HERE, name)
    if not os.path.exists(path):
        build()
    lib = C.CDLL(path)
    lib.orc_last_error.restype = C.c_char_p
    lib.orc_warper_create.restype = C.c_void_p
    lib.orc_warper_create.argtypes = [C.c_char_p, C.c_float]
    lib.orc_warper_destroy.argtypes = [C.c_void_p]
    lib.orc_warper_set_camera.argtypes = [C.c_void_p, _f32p, _f32p]
    lib.orc_warper_get_projector.argtypes = [C.c_void_p, _f32p, _f32p, _f32p, _f32p, _f32p]
    lib.orc_warper_map_forward.argtypes = [C.c_void_p, C.c_float, C.c_float, _f32p, _f32p]
    lib.orc_warper_map_backward.argtypes = [C.c_void_p, C.c_float, C.c_float, _f32p, _f32p]
    lib.orc_warper_roi.argtypes = [C.c_void_p, C.c_int, C.c_int, _f32p, _f32p, _i32p]
    lib.orc_warper_build_maps.argtypes = [C.c_void_p, C.c_int, C.c_int, _f32p, _f32p, _f32p, _f32p, _i32p]
    lib.orc_warper_warp.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p, _f32p, C.c_int, C.c_int, C.c_void_p, _i32p]
    lib.orc_warper_warp_backward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.orc_remap.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    for fn in (lib.orc_pyr_down_s16, lib.orc_pyr_down_f32):
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    for fn in (lib.orc_pyr_up_s16, lib.orc_pyr_up_f32):
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
    lib.orc_dilate3x3_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.orc_resize_linear_exact_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
    lib.orc_resize_linear_f32.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
    lib.orc_resize_area_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
    lib.orc_resize_area_u8_scale.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    lib.orc_bw_point_lut.argtypes = [C.c_int, C.c_int, C.c_void_p]
    lib.orc_distance_l1.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.orc_seam_voronoi.argtypes = [C.c_int, _i32p, _i32p, C.POINTER(C.c_void_p)]
    lib.orc_result_roi.argtypes = [C.c_int, _i32p, _i32p, _i32p]
    lib.orc_blender_create.restype = C.c_void_p
    lib.orc_blender_create.argtypes = [C.c_int]
    lib.orc_blender_destroy.argtypes = [C.c_void_p]
    lib.orc_blender_set_num_bands.argtypes = [C.c_void_p, C.c_int]
    lib.orc_blender_num_bands.argtypes = [C.c_void_p]
    lib.orc_blender_set_sharpness.argtypes = [C.c_void_p, C.c_float]
    lib.orc_blender_set_float_mode.argtypes = [C.c_void_p, C.c_int]
    lib.orc_blender_prepare.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.orc_blender_feed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.orc_blender_blend.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.orc_blender_level_size.argtypes = [C.c_void_p, C.c_int, _i32p, _i32p]
    lib.orc_blender_level_lap.restype = C.c_void_p
    lib.orc_blender_level_lap.argtypes = [C.c_void_p, C.c_int]
    lib.orc_blender_level_weight.restype = C.c_void_p
    lib.orc_blender_level_weight.argtypes = [C.c_void_p, C.c_int]
    lib.orc_blender_add_partial.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.orc_comp_create.restype = C.c_void_p
    lib.orc_comp_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.orc_comp_destroy.argtypes = [C.c_void_p]
    lib.orc_comp_feed.argtypes = [C.c_void_p, C.c_int, _i32p, _i32p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    lib.orc_comp_apply.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]
    lib.orc_comp_num_images.argtypes = [C.c_void_p]
    lib.orc_comp_gains.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    lib.orc_comp_gain_map_size.argtypes = [C.c_void_p, C.c_int, _i32p, _i32p, _i32p]
    lib.orc_comp_gain_map.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    return lib


_libs = {}


_use_omp = False


def use_openmp(on: bool, threads: int = 0) -> int:
    """Objects created / functions called from now on run in liborc_omp.so (row loops in parallel; bit-identical results).
    bench.py's multi-core CPU baseline switches this on around its timed passes.  Returns the thread count in effect."""
    global _use_omp
    _use_omp = bool(on)
    if not on:
        return 1
    l = lib()
    l.orc_set_threads.argtypes = [C.c_int]
    l.orc_set_threads.restype = C.c_int
    return int(l.orc_set_threads(int(threads)))


def lib(libm: bool = False) -> C.CDLL:
    key = "liborc_libm.so" if libm else ("liborc_omp.so" if _use_omp else "liborc.so")
    if key not in _libs:
        _libs[key] = _load(key)
    return _libs[key]


class OracleError(RuntimeError):
    pass


def _f9(a) -> np.ndarray:
    a = np.ascontiguousarray(a)
    if a.dtype != np.float32 or a.shape != (3, 3):
        raise OracleError("K and R must be 3x3 float32 (CV_32F)")
    return a


def _fp(a: np.ndarray):
    return a.ctypes.data_as(_f32p)


def _depth(a: np.ndarray) -> int:
    return {np.dtype(np.uint8): U8, np.dtype(np.int16): S16, np.dtype(np.float32): F32}[a.dtype]


class PyRotationWarper:
    """cv.PyRotationWarper restated on the CPU."""

    def __init__(self, type: str, scale: float, libm: bool = False):
        self._l = lib(libm)
        self._h = self._l.orc_warper_create(type.encode(), float(scale))
        if not self._h:
            raise OracleError(self._l.orc_last_error().decode())
        self.scale = float(np.float32(scale))

    def __del__(self):
        if getattr(self, "_h", None):
            self._l.orc_warper_destroy(self._h)
            self._h = None

    def warpRoi(self, src_size: Tuple[int, int], K, R) -> Tuple[int, int, int, int]:
        K, R = _f9(K), _f9(R)
        roi = (C.c_int * 4)()
        self._l.orc_warper_roi(self._h, int(src_size[0]), int(src_size[1]), _fp(K), _fp(R), roi)
        return tuple(roi)

    def buildMaps(self, src_size, K, R):
        K, R = _f9(K), _f9(R)
        x, y, w, h = self.warpRoi(src_size, K, R)
        xm = np.empty((h, w), np.float32)
        ym = np.empty((h, w), np.float32)
        roi = (C.c_int * 4)()
        self._l.orc_warper_build_maps(self._h, int(src_size[0]), int(src_size[1]), _fp(K), _fp(R), _fp(xm), _fp(ym), roi)
        return (x, y, w, h), xm, ym

    def warpBackward(self, src: np.ndarray, K, R, interp_mode: int, border_mode: int, dst_size):
        K, R = _f9(K), _f9(R)
        src = np.ascontiguousarray(src)
        h, w = src.shape[:2]
        cn = 1 if src.ndim == 2 else src.shape[2]
        dw, dh = int(dst_size[0]), int(dst_size[1])
        dst = np.empty((dh, dw) if src.ndim == 2 else (dh, dw, cn), src.dtype)
        rc = self._l.orc_warper_warp_backward(self._h, src.ctypes.data, w, h, cn, _depth(src), _fp(K), _fp(R), int(interp_mode), int(border_mode), dw, dh, dst.ctypes.data)
        if rc:
            raise OracleError(self._l.orc_last_error().decode())
        return dst

    def warp(self, src: np.ndarray, K, R, interp_mode: int, border_mode: int):
        K, R = _f9(K), _f9(R)
        src = np.ascontiguousarray(src)
        h, w = src.shape[:2]
        cn = 1 if src.ndim == 2 else src.shape[2]
        x, y, dw, dh = self.warpRoi((w, h), K, R)
        dst = np.empty((dh, dw) if src.ndim == 2 else (dh, dw, cn), src.dtype)
        roi = (C.c_int * 4)()
        rc = self._l.orc_warper_warp(self._h, src.ctypes.data, w, h, cn, _depth(src), _fp(K), _fp(R), int(interp_mode), int(border_mode), dst.ctypes.data, roi)
        if rc:
            raise OracleError(self._l.orc_last_error().decode())
        return (roi[0], roi[1]), dst

    def setCameraParams(self, K, R):
        K, R = _f9(K), _f9(R)
        self._l.orc_warper_set_camera(self._h, _fp(K), _fp(R))

    def projector(self):
        arrs = [np.empty(9, np.float32) for _ in range(4)] + [np.empty(3, np.float32)]
        self._l.orc_warper_get_projector(self._h, *[_fp(a) for a in arrs])
        return dict(zip(("k", "rinv", "r_kinv", "k_rinv", "t"), arrs))

    def mapForward(self, x: float, y: float):
        u, v = C.c_float(), C.c_float()
        self._l.orc_warper_map_forward(self._h, float(x), float(y), C.byref(u), C.byref(v))
        return u.value, v.value

    def mapBackward(self, u: float, v: float):
        x, y = C.c_float(), C.c_float()
        self._l.orc_warper_map_backward(self._h, float(u), float(v), C.byref(x), C.byref(y))
        return x.value, y.value


def remap(src: np.ndarray, xmap: np.ndarray, ymap: np.ndarray, interp: int, border: int, libm: bool = False) -> np.ndarray:
    src = np.ascontiguousarray(src)
    xmap = np.ascontiguousarray(xmap, np.float32)
    ymap = np.ascontiguousarray(ymap, np.float32)
    h, w = src.shape[:2]
    cn = 1 if src.ndim == 2 else src.shape[2]
    dh, dw = xmap.shape
    dst = np.empty((dh, dw) if src.ndim == 2 else (dh, dw, cn), src.dtype)
    rc = lib(libm).orc_remap(src.ctypes.data, w, h, cn, _depth(src), _fp(xmap), _fp(ymap), dw, dh, interp, border, dst.ctypes.data)
    if rc:
        raise OracleError(lib(libm).orc_last_error().decode())
    return dst


def pyrDown(src: np.ndarray) -> np.ndarray:
    src = np.ascontiguousarray(src)
    h, w = src.shape[:2]
    cn = 1 if src.ndim == 2 else src.shape[2]
    shape = ((h + 1) // 2, (w + 1) // 2) + (() if src.ndim == 2 else (cn,))
    dst = np.empty(shape, src.dtype)
    fn = lib().orc_pyr_down_s16 if src.dtype == np.int16 else lib().orc_pyr_down_f32
    fn(src.ctypes.data, w, h, cn, dst.ctypes.data)
    return dst


def pyrUp(src: np.ndarray, dsize: Tuple[int, int] = None) -> np.ndarray:
    src = np.ascontiguousarray(src)
    h, w = src.shape[:2]
    cn = 1 if src.ndim == 2 else src.shape[2]
    dw, dh = dsize if dsize is not None else (w * 2, h * 2)
    dst = np.empty((dh, dw) + (() if src.ndim == 2 else (cn,)), src.dtype)
    fn = lib().orc_pyr_up_s16 if src.dtype == np.int16 else lib().orc_pyr_up_f32
    fn(src.ctypes.data, w, h, cn, dst.ctypes.data, dw, dh)
    return dst


def dilate(mask: np.ndarray) -> np.ndarray:
    mask = np.ascontiguousarray(mask, np.uint8)
    out = np.empty_like(mask)
    lib().orc_dilate3x3_u8(mask.ctypes.data, mask.shape[1], mask.shape[0], out.ctypes.data)
    return out


def resize_linear_exact(src: np.ndarray, dsize: Tuple[int, int]) -> np.ndarray:
    src = np.ascontiguousarray(src, np.uint8)
    out = np.empty((dsize[1], dsize[0]), np.uint8)
    lib().orc_resize_linear_exact_u8(src.ctypes.data, src.shape[1], src.shape[0], out.ctypes.data, dsize[0], dsize[1])
    return out


def resize_linear_f32(src: np.ndarray, dsize: Tuple[int, int]) -> np.ndarray:
    src = np.ascontiguousarray(src, np.float32)
    cn = 1 if src.ndim == 2 else src.shape[2]
    out = np.empty((dsize[1], dsize[0]) + (() if src.ndim == 2 else (cn,)), np.float32)
    lib().orc_resize_linear_f32(src.ctypes.data, src.shape[1], src.shape[0], cn, out.ctypes.data, dsize[0], dsize[1])
    return out


def resize_area(src: np.ndarray, fx: float, fy: float, bw_point=None) -> np.ndarray:
    """cv.resize(src, None, fx=, fy=, interpolation=INTER_AREA) [+ adjust_black_and_white_point(bw_point)] (sde.py:1701-1711)."""
    src = np.ascontiguousarray(src, np.uint8)
    h, w = src.shape[:2]
    cn = 1 if src.ndim == 2 else src.shape[2]
    dw, dh = int(np.rint(w * fx)), int(np.rint(h * fy))
    out = np.empty((dh, dw) + (() if src.ndim == 2 else (cn,)), np.uint8)
    lut = None
    if bw_point:
        lut = np.empty(256, np.uint8)
        lib().orc_bw_point_lut(int(bw_point[0]), int(bw_point[1]), lut.ctypes.data)
    lib().orc_resize_area_u8_scale(src.ctypes.data, w, h, cn, float(fx), float(fy), lut.ctypes.data if lut is not None else None, out.ctypes.data, dw, dh)
    return out


def bw_point_lut(black: int, white: int) -> np.ndarray:
    lut = np.empty(256, np.uint8)
    lib().orc_bw_point_lut(int(black), int(white), lut.ctypes.data)
    return lut


def distance_l1(mask: np.ndarray) -> np.ndarray:
    mask = np.ascontiguousarray(mask, np.uint8)
    out = np.empty(mask.shape, np.float32)
    lib().orc_distance_l1(mask.ctypes.data, mask.shape[1], mask.shape[0], out.ctypes.data)
    return out


def resultRoi(corners: Sequence[Tuple[int, int]], sizes: Sequence[Tuple[int, int]]) -> Tuple[int, int, int, int]:
    n = len(corners)
    c = (C.c_int * (2 * n))(*[int(v) for p in corners for v in p])
    s = (C.c_int * (2 * n))(*[int(v) for p in sizes for v in p])
    roi = (C.c_int * 4)()
    lib().orc_result_roi(n, c, s, roi)
    return tuple(roi)


class Blender:
    """cv.detail.Blender / FeatherBlender / MultiBandBlender restated on the CPU."""

    def __init__(self, type: int = BLEND_NO, float_mode: bool = False):
        self._l = lib()
        self._h = self._l.orc_blender_create(type)
        self.type = type
        self.float_mode = float_mode
        if float_mode:
            self._l.orc_blender_set_float_mode(self._h, 1)
        self._final = None

    def __del__(self):
        if getattr(self, "_h", None):
            self._l.orc_blender_destroy(self._h)
            self._h = None

    def setNumBands(self, n: int):
        self._l.orc_blender_set_num_bands(self._h, int(n))

    def numBands(self) -> int:
        return self._l.orc_blender_num_bands(self._h)

    def setSharpness(self, s: float):
        self._l.orc_blender_set_sharpness(self._h, float(s))

    def prepare(self, roi):
        self._final = tuple(int(v) for v in roi)
        self._l.orc_blender_prepare(self._h, *self._final)

    def feed(self, img: np.ndarray, mask: np.ndarray, tl):
        img = np.ascontiguousarray(img, np.float32 if self.float_mode else np.int16)
        mask = np.ascontiguousarray(mask, np.uint8)
        h, w = mask.shape
        assert img.shape == (h, w, 3)
        rc = self._l.orc_blender_feed(self._h, img.ctypes.data, mask.ctypes.data, w, h, int(tl[0]), int(tl[1]))
        if rc:
            raise OracleError(self._l.orc_last_error().decode())

    def level(self, i: int):
        w, h = C.c_int(), C.c_int()
        if self._l.orc_blender_level_size(self._h, i, C.byref(w), C.byref(h)):
            raise OracleError("no such level")
        lap = np.ctypeslib.as_array(C.cast(self._l.orc_blender_level_lap(self._h, i), C.POINTER(C.c_int16)), (h.value, w.value, 3)).copy()
        wgt = np.ctypeslib.as_array(C.cast(self._l.orc_blender_level_weight(self._h, i), C.POINTER(C.c_float)), (h.value, w.value)).copy()
        return lap, wgt

    def addPartial(self, i: int, lap32: np.ndarray, wgt: np.ndarray):
        lap32 = np.ascontiguousarray(lap32, np.int32)
        wgt = np.ascontiguousarray(wgt, np.float32)
        if self._l.orc_blender_add_partial(self._h, i, lap32.ctypes.data, wgt.ctypes.data):
            raise OracleError("addPartial failed")

    def blend(self):
        x, y, w, h = self._final
        dst = np.empty((h, w, 3), np.float32 if self.float_mode else np.int16)
        msk = np.empty((h, w), np.uint8)
        rc = self._l.orc_blender_blend(self._h, dst.ctypes.data, msk.ctypes.data)
        if rc:
            raise OracleError(self._l.orc_last_error().decode())
        return dst, msk


class ExposureCompensator:
    """cv.detail.ExposureCompensator family restated on the CPU."""

    def __init__(self, type: int, bl_w: int = 32, bl_h: int = 32, nr_feeds: int = 1, nr_filter: int = 2):
        self._l = lib()
        self.type = type
        self._h = self._l.orc_comp_create(type, bl_w, bl_h, nr_feeds, nr_filter)

    def __del__(self):
        if getattr(self, "_h", None):
            self._l.orc_comp_destroy(self._h)
            self._h = None

    def feed(self, corners, images: List[np.ndarray], masks: List[np.ndarray]):
        n = len(images)
        imgs = [np.ascontiguousarray(i, np.uint8) for i in images]
        msks = [np.ascontiguousarray(m, np.uint8) for m in masks]
        c = (C.c_int * (2 * n))(*[int(v) for p in corners for v in p])
        s = (C.c_int * (2 * n))(*[int(v) for i in imgs for v in (i.shape[1], i.shape[0])])
        ip = (C.c_void_p * n)(*[i.ctypes.data for i in imgs])
        mp = (C.c_void_p * n)(*[m.ctypes.data for m in msks])
        self._l.orc_comp_feed(self._h, n, c, s, ip, mp)

    def apply(self, index: int, corner, image: np.ndarray, mask: np.ndarray = None):
        assert image.dtype == np.uint8 and image.flags["C_CONTIGUOUS"] and image.shape[2] == 3
        rc = self._l.orc_comp_apply(self._h, int(index), image.ctypes.data, image.shape[1], image.shape[0])
        if rc:
            raise OracleError(self._l.orc_last_error().decode())

    def gains(self) -> np.ndarray:
        n = self._l.orc_comp_num_images(self._h)
        out = (C.c_double * (3 * n))()
        m = self._l.orc_comp_gains(self._h, out)
        return np.array(out[: max(m, 0)])

    def gainMap(self, index: int) -> np.ndarray:
        w, h, cn = C.c_int(), C.c_int(), C.c_int()
        if self._l.orc_comp_gain_map_size(self._h, index, C.byref(w), C.byref(h), C.byref(cn)):
            raise OracleError("no gain map")
        out = np.empty((h.value, w.value, cn.value), np.float32)
        self._l.orc_comp_gain_map(self._h, index, out.ctypes.data)
        return out


# ---- seam finder and timelapser ------------------------------------------------------------------
SEAM_NO, SEAM_VORONOI, SEAM_DP = 0, 1, 2
TIMELAPSER_AS_IS, TIMELAPSER_CROP = 0, 1


class SeamFinder:
    """cv.detail.SeamFinder_createDefault(type) / cv.detail_DpSeamFinder(costFunc) (sde.py:243-249); find() returns the (cut)
    masks like cv2 does.  ``SeamFinder_createDefault(SeamFinder_DP_SEAM)`` is DpSeamFinder with its default cost COLOR."""

    def __init__(self, type: int, cost_func: str = "COLOR"):
        if type not in (SEAM_NO, SEAM_VORONOI, SEAM_DP):
            raise OracleError(f"unknown seam finder type {type}")
        if cost_func not in ("COLOR", "COLOR_GRAD"):
            raise OracleError(f"DpSeamFinder: unknown cost function {cost_func!r}")
        self.type, self.cost_func = type, cost_func
        self.pair_order = None

    def find(self, images, corners, masks):
        masks = [np.ascontiguousarray(m, np.uint8).copy() for m in masks]
        if self.type == SEAM_NO or not masks:
            return tuple(masks)
        n = len(masks)
        cs = (C.c_int * (2 * n))(*[int(v) for c in corners for v in c])
        ss = (C.c_int * (2 * n))(*[int(v) for m in masks for v in (m.shape[1], m.shape[0])])
        ptrs = (C.c_void_p * n)(*[m.ctypes.data for m in masks])
        if self.type == SEAM_VORONOI:
            lib().orc_seam_voronoi(n, cs, ss, ptrs)
            return tuple(masks)
        imgs = [np.ascontiguousarray(im, np.float32) for im in images]
        for im, m in zip(imgs, masks):
            if im.ndim != 3 or im.shape[2] != 3 or im.shape[:2] != m.shape:
                raise OracleError("DpSeamFinder.find: images must be float32 HxWx3 of their masks' sizes")
        iptrs = (C.c_void_p * n)(*[im.ctypes.data for im in imgs])
        order = (C.c_int * max(n * (n - 1), 1))()
        f = lib().orc_seam_dp
        f.restype = C.c_int
        f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        if f(n, cs, ss, iptrs, ptrs, 1 if self.cost_func == "COLOR_GRAD" else 0, order) != 0:
            raise OracleError("orc_seam_dp failed")
        self.pair_order = [(order[2 * k], order[2 * k + 1]) for k in range(n * (n - 1) // 2)]
        return tuple(masks)


def dp_gradients(img: np.ndarray):
    """DpSeamFinder::computeGradients of one float32 BGR image -> (gradx, grady)."""
    im = np.ascontiguousarray(img, np.float32)
    h, w = im.shape[:2]
    gx, gy = np.empty((h, w), np.float32), np.empty((h, w), np.float32)
    f = lib().orc_seam_dp_gradients
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    f(im.ctypes.data, w, h, gx.ctypes.data, gy.ctypes.data)
    return gx, gy


class Timelapser:
    """cv.detail.Timelapser_createDefault(type) (sde.py:1822-1851; stitching/src/timelapsers.cpp): every process() clears the
    pano-sized int16 canvas and pastes the frame at its corner (pixels outside the canvas are dropped); the mask is ignored."""

    def __init__(self, type: int):
        if type not in (TIMELAPSER_AS_IS, TIMELAPSER_CROP):
            raise OracleError("unknown timelapser type")
        self.type = type
        self.roi = None
        self.dst = None

    def initialize(self, corners, sizes):
        if self.type == TIMELAPSER_AS_IS:
            x0 = min(c[0] for c in corners); y0 = min(c[1] for c in corners)
            x1 = max(c[0] + s[0] for c, s in zip(corners, sizes)); y1 = max(c[1] + s[1] for c, s in zip(corners, sizes))
        else:  # resultRoiIntersection
            x0 = max(c[0] for c in corners); y0 = max(c[1] for c in corners)
            x1 = min(c[0] + s[0] for c, s in zip(corners, sizes)); y1 = min(c[1] + s[1] for c, s in zip(corners, sizes))
        self.roi = (x0, y0, x1 - x0, y1 - y0)
        self.dst = np.zeros((max(y1 - y0, 0), max(x1 - x0, 0), 3), np.int16)

    def process(self, img, mask, tl):
        if img.dtype != np.int16 or img.ndim != 3 or img.shape[2] != 3:
            raise OracleError("Timelapser.process: image must be CV_16SC3")
        self.dst[:] = 0
        x0, y0, w, h = self.roi
        dx, dy = tl[0] - x0, tl[1] - y0
        ys, xs = max(0, -dy), max(0, -dx)
        ye, xe = min(img.shape[0], h - dy), min(img.shape[1], w - dx)
        if ye > ys and xe > xs:
            self.dst[dy + ys:dy + ye, dx + xs:dx + xe] = img[ys:ye, xs:xe]

    def getDst(self):
        return self.dst
