"""``cv.detail``-shaped objects over the HIP library: blenders, exposure compensators, resultRoi.

Reference usage (stitching_detailed_enhanced.py):
* :649-665   ``ExposureCompensator_createDefault(type)``, ``detail_ChannelsCompensator(nr_feeds)``,
             ``detail_BlocksChannelsCompensator(bw, bh, nr_feeds)``
* :1613      ``compensator.feed(corners=, images=, masks=)``
* :1754      ``compensator.apply(idx, corner, image, mask)`` -- relies on in-place mutation of ``image``
* :1806-1820 ``Blender_createDefault(Blender_NO)``, ``detail_MultiBandBlender()`` + ``setNumBands``,
             ``detail_FeatherBlender()`` + ``setSharpness``, ``prepare(resultRoi(...))``
* :1886      ``blender.feed(cv.UMat(image_warped_s), mask_warped, corner)``
* :1930      ``blender.blend(None, None) -> (result int16, result_mask uint8)``
Same names, argument meaning and error behaviour as cv2.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from . import deferred as _deferred
from .camera import CameraParams, wave_correct  # noqa: F401  (cv.detail.CameraParams / waveCorrect)
from .umat import UMat, as_umat

# ---- constants (cv.detail.*) ------------------------------------------------------------------------------------
Blender_NO, Blender_FEATHER, Blender_MULTI_BAND = 0, 1, 2
ExposureCompensator_NO = 0
ExposureCompensator_GAIN = 1
ExposureCompensator_GAIN_BLOCKS = 2
ExposureCompensator_CHANNELS = 3
ExposureCompensator_CHANNELS_BLOCKS = 4
WAVE_CORRECT_HORIZ, WAVE_CORRECT_VERT, WAVE_CORRECT_AUTO = 0, 1, 2


def waveCorrect(rmats, kind):
    return wave_correct(rmats, kind)


def resultRoi(corners: Sequence[Tuple[int, int]], sizes: Sequence[Tuple[int, int]]) -> Tuple[int, int, int, int]:
    """cv.detail.resultRoi(corners=, sizes=) (sde.py:1807)."""
    n = len(corners)
    if n == 0 or n != len(sizes):
        raise _lib.error("resultRoi: corners and sizes must be non-empty and of equal length")
    c = (C.c_int * (2 * n))(*[int(v) for p in corners for v in p])
    s = (C.c_int * (2 * n))(*[int(v) for p in sizes for v in p])
    roi = (C.c_int * 4)()
    _lib.check(_lib.lib().ssp_result_roi(n, c, s, roi))
    return tuple(roi)


# ---- blenders -----------------------------------------------------------------------------------------------------
class Blender:
    """cv.detail.Blender (type NO: last writer wins under the mask)."""

    _TYPE = Blender_NO

    def __init__(self, _type: Optional[int] = None):
        self._h = C.c_void_p()
        self._type = self._TYPE if _type is None else int(_type)
        _lib.check(_lib.lib().ssp_blender_create(self._type, C.byref(self._h)))
        self._keep = []
        self._prepared = None        # the rectangle of prepare() until blend() consumes the state
        self._feeds = []             # feeds kept for blend() (deferred.py): from the first deferred operand on, in order
        self._want_bands, self._sharpness, self._float = 5, 0.02, False

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                _lib.lib().ssp_blender_destroy(h)
            except Exception:
                pass
            self._h = None

    def prepare(self, *args):
        """``prepare(rect)`` or ``prepare(corners, sizes)``."""
        if len(args) == 1:
            x, y, w, h = [int(v) for v in args[0]]
        elif len(args) == 2:
            x, y, w, h = resultRoi(args[0], args[1])
        else:
            raise _lib.error("prepare(dst_roi) or prepare(corners, sizes)")
        _lib.check(_lib.lib().ssp_blender_prepare(self._h, x, y, w, h))
        self._prepared = (x, y, w, h)
        self._feeds = []

    def feed(self, img, mask, tl):
        """cv2's element types: ``Blender`` / ``FeatherBlender`` take CV_16SC3 only (cv2 asserts it; a uint8 image raises here as
        it does there).  ``MultiBandBlender`` also takes CV_8UC3, as cv2 does: cv2 then builds the pyramid with its 8-bit
        pyrDown / pyrUp and an 8U - 8U -> 16S subtract, which are the same rounded integer formulas on values that never leave
        [0, 255], so the library runs the image as int16 holding the same values (tests: ``test_multiband_u8_feed_equals_int16_feed``).
        That equality is derived from OpenCV's source, not measured against a cv2 binary ([CV-U] in DESIGN.md); the reference
        itself always feeds int16 (sde.py:1755)."""
        im, _ = as_umat(img)
        mk, _ = as_umat(mask)
        if self._type != Blender_MULTI_BAND and im.info()[3] != 3:
            raise _lib.error(f"feed: {type(self).__name__} takes CV_16SC3 images (cv2 asserts img.type() == CV_16SC3); convert with astype(np.int16) as sde.py:1755 does")
        if self._prepared is None:
            raise _lib.error("feed called before prepare (or after blend)")
        # deferred operands (UMat results of warp / astype / dilate / resize / bitwise_and): nothing of a fed image is observable before
        # blend() (sde.py:1886 -> :1930), so the feed is kept -- and every later one behind it, in order -- for blend() to run as one plan
        if self._feeds or _deferred.is_pending(im) or _deferred.is_pending(mk):
            mi, ii = mk.info(), im.info()
            if mi[2] != 1 or mi[3] != 0 or mi[:2] != ii[:2] or ii[2] != 3:
                raise _lib.error(f"feed: mask {mi[0]}x{mi[1]} ({mi[2]} channels) does not fit image {ii[0]}x{ii[1]} ({ii[2]} channels)")
            self._feeds.append(_deferred._Feed(im, mk, tl))
            return
        _lib.check(_lib.lib().ssp_blender_feed(self._h, im._h, mk._h, int(tl[0]), int(tl[1])))

    def blend(self, dst=None, dst_mask=None, device: bool = False, mosaic: bool = False):
        """-> (result int16 HxWx3, result_mask uint8).  ``device=True`` returns UMats; ``mosaic=True`` appends
        the saturated 8-bit panorama that ``cv.imwrite`` would store (sde.py:1938)."""
        if self._feeds:
            feeds, self._feeds = self._feeds, []
            plan = _deferred.plan_for(self, feeds)
            if plan is not None:
                # the reference's sequence: one fused warp for all frames, pyramids, collapse (compose.Composer) -- same kernels, same bits
                composer, frames = plan
                _deferred.stats["planned"] += 1
                composer.run(frames)
                mo_u, mk_u, rs_u = composer.result()
                self._prepared = None            # consumed, as cv2's blend() releases its state: a second blend() raises
                out = [rs_u, mk_u] + ([mo_u] if mosaic else [])
                return tuple(out if device else [o.get() for o in out])
            _deferred.stats["call_by_call"] += 1
            for f in feeds:                      # anything else: call by call, by the eager kernels
                _lib.check(_lib.lib().ssp_blender_feed(self._h, f.img._h, f.mask._h, f.tl[0], f.tl[1]))
        if self._prepared is None:
            raise _lib.error("blend called before prepare, or twice (the blender state is consumed by blend)")
        r, m, mo = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _lib.check(_lib.lib().ssp_blender_blend(self._h, C.byref(r), C.byref(m), C.byref(mo) if mosaic else None))
        self._prepared = None
        out = [UMat.from_handle(r), UMat.from_handle(m)]
        if mosaic:
            out.append(UMat.from_handle(mo))
        if not device:
            out = [o.get() for o in out]
        return tuple(out)


class FeatherBlender(Blender):
    _TYPE = Blender_FEATHER

    def __init__(self, sharpness: float = 0.02):
        super().__init__()
        self.setSharpness(sharpness)

    def setSharpness(self, val: float) -> None:
        self._sharpness = float(val)
        _lib.check(_lib.lib().ssp_blender_set_sharpness(self._h, float(val)))

    def sharpness(self) -> float:
        return self._sharpness


class MultiBandBlender(Blender):
    _TYPE = Blender_MULTI_BAND

    def __init__(self, try_gpu: int = 0, num_bands: int = 5, weight_type: int = 5, float_pyramids: bool = False):
        super().__init__()
        if weight_type != 5:
            raise _lib.error("MultiBandBlender: only CV_32F weights are implemented (the reference uses the default)")
        self.setNumBands(num_bands)
        if float_pyramids:
            _lib.check(_lib.lib().ssp_blender_set_float_mode(self._h, 1))
            self._float = True

    def setNumBands(self, val: int) -> None:
        _lib.check(_lib.lib().ssp_blender_set_num_bands(self._h, int(val)))
        self._want_bands = int(val)

    def numBands(self) -> int:
        n = C.c_int()
        _lib.check(_lib.lib().ssp_blender_get_num_bands(self._h, C.byref(n)))
        return n.value


def Blender_createDefault(type: int, try_gpu: bool = False) -> Blender:
    if type == Blender_NO:
        return Blender()
    if type == Blender_FEATHER:
        return FeatherBlender()
    if type == Blender_MULTI_BAND:
        return MultiBandBlender()
    raise _lib.error(f"Blender_createDefault: unknown blender type {type}")


# ---- exposure compensators ------------------------------------------------------------------------------------------
class ExposureCompensator:
    def __init__(self, type: int):
        self._h = C.c_void_p()
        self.type = type
        self._gen = 0        # feeds / setMatGains so far: a deferred apply belongs to one generation of gains
        _lib.check(_lib.lib().ssp_comp_create(int(type), C.byref(self._h)))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                _lib.lib().ssp_comp_destroy(h)
            except Exception:
                pass
            self._h = None

    def feed(self, corners, images, masks):
        n = len(images)
        if not (len(corners) == n == len(masks)):
            raise _lib.error("feed: corners, images and masks must have the same length")
        ims = [as_umat(i)[0] for i in images]
        mks = [as_umat(m)[0] for m in masks]
        c = (C.c_int * (2 * max(n, 1)))(*[int(v) for p in corners for v in p])
        ip = (C.c_void_p * max(n, 1))(*[i._h.value for i in ims])
        mp = (C.c_void_p * max(n, 1))(*[m._h.value for m in mks])
        _lib.check(_lib.lib().ssp_comp_feed(self._h, n, c, ip, mp))
        self._gen += 1

    def apply(self, index: int, corner, image, mask=None):
        """In place, as cv2: an ndarray argument is updated through a device round trip, a UMat on the GPU (a deferred warp: when it is
        evaluated -- inside the fused warp of blender.blend, or by the apply kernel behind its own warp)."""
        if isinstance(image, UMat):
            if self.type == ExposureCompensator_NO:
                return image
            if _deferred.is_pending(image) and image.op == "warp" and image.gain is None and image.info()[2:4] == (3, 0):
                image.gain = (self, int(index), self._gen)
                return image
            _lib.check(_lib.lib().ssp_comp_apply(self._h, int(index), image._h))
            image._ver += 1
            return image
        if not (isinstance(image, np.ndarray) and image.dtype == np.uint8 and image.ndim == 3 and image.shape[2] == 3):
            raise _lib.error("apply: image must be an 8UC3 ndarray or UMat")
        if self.type == ExposureCompensator_NO:
            return image
        u = UMat(image)
        _lib.check(_lib.lib().ssp_comp_apply(self._h, int(index), u._h))
        image[...] = u.get()
        return image

    def getMatGains(self) -> List[np.ndarray]:
        if self.type in (ExposureCompensator_GAIN_BLOCKS, ExposureCompensator_CHANNELS_BLOCKS):
            n = C.c_int()
            _lib.check(_lib.lib().ssp_comp_num_images(self._h, C.byref(n)))
            maps = [self.gainMap(i) for i in range(n.value)]
            return [m[:, :, 0] if m.shape[2] == 1 else m for m in maps]
        cnt = C.c_int()
        _lib.check(_lib.lib().ssp_comp_get_gains(self._h, None, 0, C.byref(cnt)))
        buf = (C.c_double * max(cnt.value, 1))()
        _lib.check(_lib.lib().ssp_comp_get_gains(self._h, buf, cnt.value, C.byref(cnt)))
        g = np.array(buf[: cnt.value], dtype=np.float64)
        if self.type in (ExposureCompensator_CHANNELS,):
            return [row.reshape(3, 1) for row in g.reshape(-1, 3)]
        return [np.array([[v]]) for v in g]

    def setMatGains(self, umv) -> None:
        """cv2: one 1x1 (GAIN) / 3x1 (CHANNELS) float64 matrix, or one float32 gain map (block kinds), per image."""
        mats = [m.get() if hasattr(m, "get") else np.asarray(m) for m in umv]
        if not mats:
            raise _lib.error("setMatGains: empty list")
        self._gen += 1
        if self.type in (ExposureCompensator_GAIN, ExposureCompensator_CHANNELS):
            flat = np.concatenate([np.asarray(m, np.float64).reshape(-1) for m in mats])
            buf = (C.c_double * flat.size)(*flat.tolist())
            _lib.check(_lib.lib().ssp_comp_set_gains(self._h, buf, flat.size))
        elif self.type in (ExposureCompensator_GAIN_BLOCKS, ExposureCompensator_CHANNELS_BLOCKS):
            for i, m in enumerate(mats):
                m = np.ascontiguousarray(m, np.float32)
                m3 = m.reshape(m.shape[0], m.shape[1], -1)
                _lib.check(_lib.lib().ssp_comp_set_gain_map(self._h, i, m3.ctypes.data_as(C.POINTER(C.c_float)), m3.shape[1], m3.shape[0], m3.shape[2]))
        else:
            raise _lib.error("setMatGains: the NO compensator has no gains")

    def gains(self) -> np.ndarray:
        cnt = C.c_int()
        _lib.check(_lib.lib().ssp_comp_get_gains(self._h, None, 0, C.byref(cnt)))
        buf = (C.c_double * max(cnt.value, 1))()
        _lib.check(_lib.lib().ssp_comp_get_gains(self._h, buf, cnt.value, C.byref(cnt)))
        return np.array(buf[: cnt.value], dtype=np.float64)

    def gainMap(self, index: int) -> np.ndarray:
        w, h, cn = C.c_int(), C.c_int(), C.c_int()
        _lib.check(_lib.lib().ssp_comp_get_gain_map(self._h, int(index), None, 0, C.byref(w), C.byref(h), C.byref(cn)))
        out = np.empty((h.value, w.value, cn.value), np.float32)
        _lib.check(_lib.lib().ssp_comp_get_gain_map(self._h, int(index), out.ctypes.data_as(C.POINTER(C.c_float)), out.size, C.byref(w), C.byref(h), C.byref(cn)))
        return out

    def setNrFeeds(self, n: int) -> None:
        _lib.check(_lib.lib().ssp_comp_set_nr_feeds(self._h, int(n)))

    def setBlockSize(self, w: int, h: int) -> None:
        _lib.check(_lib.lib().ssp_comp_set_block_size(self._h, int(w), int(h)))

    def setNrGainsFilteringIterations(self, n: int) -> None:
        _lib.check(_lib.lib().ssp_comp_set_nr_filtering(self._h, int(n)))


def ExposureCompensator_createDefault(type: int) -> ExposureCompensator:
    if type not in (0, 1, 2, 3, 4):
        raise _lib.error(f"ExposureCompensator_createDefault: unknown type {type}")
    return ExposureCompensator(type)


def ChannelsCompensator(nr_feeds: int = 1) -> ExposureCompensator:
    c = ExposureCompensator(ExposureCompensator_CHANNELS)
    c.setNrFeeds(nr_feeds)
    return c


def BlocksChannelsCompensator(bl_width: int = 32, bl_height: int = 32, nr_feeds: int = 1) -> ExposureCompensator:
    c = ExposureCompensator(ExposureCompensator_CHANNELS_BLOCKS)
    c.setBlockSize(bl_width, bl_height)
    c.setNrFeeds(nr_feeds)
    return c


def BlocksGainCompensator(bl_width: int = 32, bl_height: int = 32, nr_feeds: int = 1) -> ExposureCompensator:
    c = ExposureCompensator(ExposureCompensator_GAIN_BLOCKS)
    c.setBlockSize(bl_width, bl_height)
    c.setNrFeeds(nr_feeds)
    return c


def GainCompensator(nr_feeds: int = 1) -> ExposureCompensator:
    c = ExposureCompensator(ExposureCompensator_GAIN)
    c.setNrFeeds(nr_feeds)
    return c


# ---- seam finders (sde.py:243-249, :1618) ----------------------------------------------------------------------------
SeamFinder_NO, SeamFinder_VORONOI_SEAM, SeamFinder_DP_SEAM = 0, 1, 2


class SeamFinder:
    """cv.detail.SeamFinder: ``find(images, corners, masks) -> masks`` (UMats in -> UMats out, ndarrays in -> ndarrays out).
    NO returns the masks as they are; VORONOI_SEAM cuts them on the device (``ssp_seam_voronoi``); DP_SEAM is
    ``DpSeamFinder`` with cv2's default cost function 'COLOR' (``ssp_seam_dp``)."""

    def __init__(self, type: int, costFunc: str = "COLOR"):
        if type not in (SeamFinder_NO, SeamFinder_VORONOI_SEAM, SeamFinder_DP_SEAM):
            raise _lib.error(f"SeamFinder_createDefault: unknown type {type}")
        if costFunc not in ("COLOR", "COLOR_GRAD"):
            raise _lib.error(f"DpSeamFinder: unknown cost function {costFunc!r} (cv2 takes 'COLOR' or 'COLOR_GRAD')")
        self._type, self._cost = type, costFunc
        self.pair_order = None   # DP_SEAM: the image pairs in the order find() processed them

    def find(self, src, corners, masks):
        if self._type == SeamFinder_NO or len(masks) == 0:
            return tuple(masks)
        if len(corners) != len(masks):
            raise _lib.error("SeamFinder.find: corners and masks differ in length")
        ums, devs = [], []
        for m in masks:
            u, dev = as_umat(m)
            if dev:  # cv2 cuts UMats in place and returns them
                ums.append(u)
            else:
                ums.append(UMat(np.ascontiguousarray(m)))
            devs.append(dev)
        n = len(ums)
        cs = (C.c_int * (2 * n))(*[int(v) for c in corners for v in (c[0], c[1])])
        hs = (C.c_void_p * n)(*[u._h for u in ums])
        if self._type == SeamFinder_VORONOI_SEAM:
            _lib.check(_lib.lib().ssp_seam_voronoi(n, cs, hs))
        else:
            if len(src) != n:
                raise _lib.error("DpSeamFinder.find: images and masks differ in length")
            ims = [as_umat(im)[0] for im in src]          # float32 (sde.py:1601-1604) or 8-bit, 3 channels
            ih = (C.c_void_p * n)(*[u._h for u in ims])
            order = (C.c_int * max(n * (n - 1), 1))()
            _lib.check(_lib.lib().ssp_seam_dp(n, cs, ih, hs, 1 if self._cost == "COLOR_GRAD" else 0, order))
            self.pair_order = [(order[2 * k], order[2 * k + 1]) for k in range(n * (n - 1) // 2)]
        for u, d in zip(ums, devs):
            if d:
                u._ver += 1            # cut in place
        return tuple(u if d else u.get() for u, d in zip(ums, devs))


def SeamFinder_createDefault(type: int) -> SeamFinder:
    return SeamFinder(type)


def DpSeamFinder(costFunc: str = "COLOR") -> SeamFinder:
    """cv.detail_DpSeamFinder(costFunc) (sde.py:243-249: 'COLOR' for "dp_color", 'COLOR_GRAD' for the default "dp_colorgrad")."""
    return SeamFinder(SeamFinder_DP_SEAM, costFunc)


# ---- timelapser (sde.py:1822-1871) -----------------------------------------------------------------------------------
Timelapser_AS_IS, Timelapser_CROP = 0, 1


class Timelapser:
    """cv.detail.Timelapser: a pano-sized int16 canvas that every ``process`` clears and pastes one warped frame into."""

    def __init__(self, type: int):
        self._h = C.c_void_p()
        _lib.check(_lib.lib().ssp_timelapser_create(int(type), C.byref(self._h)))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                _lib.lib().ssp_timelapser_destroy(h)
            except Exception:
                pass
            self._h = None

    def initialize(self, corners, sizes):
        n = len(corners)
        if n == 0 or n != len(sizes):
            raise _lib.error("Timelapser.initialize: corners and sizes must be non-empty and of equal length")
        cs = (C.c_int * (2 * n))(*[int(v) for c in corners for v in (c[0], c[1])])
        ss = (C.c_int * (2 * n))(*[int(v) for s in sizes for v in (s[0], s[1])])
        _lib.check(_lib.lib().ssp_timelapser_initialize(self._h, n, cs, ss))

    def process(self, img, mask, tl):
        im, _ = as_umat(img)   # the mask is accepted and ignored, as in OpenCV
        _lib.check(_lib.lib().ssp_timelapser_process(self._h, im._h, int(tl[0]), int(tl[1])))

    def getDst(self) -> UMat:
        out = C.c_void_p()
        _lib.check(_lib.lib().ssp_timelapser_get_dst(self._h, C.byref(out)))
        return UMat.from_handle(out)

    def getDstRoi(self):
        roi = (C.c_int * 4)()
        _lib.check(_lib.lib().ssp_timelapser_dst_roi(self._h, roi))
        return tuple(roi)


def Timelapser_createDefault(type: int) -> Timelapser:
    return Timelapser(type)
