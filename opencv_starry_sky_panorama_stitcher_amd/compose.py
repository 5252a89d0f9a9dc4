"""The composition half of the reference pipeline, restated against a cv2-shaped namespace.

``compose_panorama`` follows ``StitchingDetailedPipeline.compose_imgs_to_panorama``
(stitching_detailed_enhanced.py:1537-1944) call for call -- seam-scale warps (:1543-1599), compensator feed
(:1612-1613), compose-scale warpRoi (:1689-1698), per image warp / warp mask / apply / astype / dilate / resize / and /
feed (:1731-1889), blend (:1930) and the 8-bit saturation that imwrite performs (:1938); optionally the frame prologue
(:1699-1711), the seam finder (:1615-1624; "no" or "voronoi") and the timelapser (:1822-1851) -- with registration and
file output left out.  ``cv`` is any namespace with the cv2 names
used there: this package (HIP), or the oracle adapter in tests/.  ``Composer`` is the batched device-resident
form of the same loop (one C call per panorama; bench.py's step).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .umat import UMat


@dataclass
class ComposeResult:
    result: np.ndarray          # int16 HxWx3 (float32 in float mode), as blender.blend returns
    result_mask: np.ndarray     # uint8 HxW
    mosaic: np.ndarray          # uint8 HxWx3, what cv.imwrite stores (saturate_cast)
    corners: List[Tuple[int, int]]
    sizes: List[Tuple[int, int]]
    pano_roi: Tuple[int, int, int, int]
    num_bands: int = 0
    timelapse: Optional[List[np.ndarray]] = None   # per frame: the int16 canvas after timelapser.process (sde.py:1857-1882)


def num_bands_for(blend_width: float) -> int:
    """sde.py:1814-1815: (log(blend_width)/log(2) - 1).astype(int32)."""
    return int((np.log(blend_width) / np.log(2.0) - 1.0).astype(np.int32))


def make_blender(cv, blend: str, dst_sz, blend_strength: Optional[float] = None, num_bands: Optional[int] = None, float_pyramids: bool = False):
    """sde.py:1805-1820."""
    blender = cv.detail.Blender_createDefault(cv.detail.Blender_NO)
    blend_width = np.sqrt(dst_sz[2] * dst_sz[3]) * (blend_strength if blend_strength is not None else 5.0) / 100
    if blend == "no" or (blend_strength is not None and blend_width < 1):
        blender = cv.detail.Blender_createDefault(cv.detail.Blender_NO)
    elif blend == "multiband":
        blender = cv.detail_MultiBandBlender(float_pyramids=True) if float_pyramids else cv.detail_MultiBandBlender()
        blender.setNumBands(num_bands if num_bands is not None else num_bands_for(blend_width))
    elif blend == "feather":
        blender = cv.detail_FeatherBlender()
        blender.setSharpness(1.0 / blend_width)
    blender.prepare(dst_sz)
    return blender


_ones_cache = {}


def _device_ones(cv, h: int, w: int):
    """The all-255 mask of sde.py:1739 as a device-resident constant (the reference builds it on the host for every image; a UMat caller
    keeps one per frame size)."""
    key = (id(cv), h, w)
    if key not in _ones_cache:
        if len(_ones_cache) > 8:
            _ones_cache.clear()
        _ones_cache[key] = cv.UMat(np.full((h, w), 255, np.uint8))
    return _ones_cache[key]


def seam_stage(cv, seam_frames, Ks, Rs, warp: str, warper_scale: float, seam_aspect: float, expos_comp: int = 0, seam: str = "no"):
    """sde.py:1543-1624: seam-scale warps of the frames and of their all-255 masks, ``compensator.feed``, the seam finder.
    -> (compensator, seam-scale masks).  The reference runs this once per panorama, in front of the compose loop."""
    n = len(seam_frames)
    on_device = n > 0 and not isinstance(seam_frames[0], np.ndarray) and hasattr(seam_frames[0], "get") and hasattr(cv, "UMat")

    def ones_mask(h, w):
        if not on_device:
            return 255 * np.ones((h, w), np.uint8)
        return _device_ones(cv, h, w)
    compensator = cv.detail.ExposureCompensator_createDefault(expos_comp)
    warper_s = cv.PyRotationWarper(warp, warper_scale * seam_aspect)
    corners_s, images_s, masks_seam = [], [], []
    for idx in range(n):
        K = np.array(Ks[idx], dtype=np.float32)
        K[0, 0] *= seam_aspect
        K[0, 2] *= seam_aspect
        K[1, 1] *= seam_aspect
        K[1, 2] *= seam_aspect
        corner, image_wp = warper_s.warp(seam_frames[idx], K, Rs[idx], cv.INTER_AREA, cv.BORDER_REFLECT)
        um = ones_mask(seam_frames[idx].shape[0], seam_frames[idx].shape[1])
        _, mask_wp = warper_s.warp(um, K, Rs[idx], cv.INTER_NEAREST, cv.BORDER_CONSTANT)
        corners_s.append(corner)
        images_s.append(image_wp)
        masks_seam.append(mask_wp)
    # ---- C: exposure compensation (sde.py:1612-1613) ----------------------------------------------------------
    compensator.feed(corners=corners_s, images=images_s, masks=masks_seam)
    # ---- D: seam estimation (sde.py:1615-1624) -----------------------------------------------------------------------
    if seam != "no":
        if seam in ("dp_color", "dp_colorgrad"):                                                              # sde.py:243-249
            finder = cv.detail_DpSeamFinder("COLOR" if seam == "dp_color" else "COLOR_GRAD")
        else:
            finder = cv.detail.SeamFinder_createDefault({"voronoi": cv.detail.SeamFinder_VORONOI_SEAM}[seam])
        masks_seam = list(finder.find([np.asarray(im.get() if hasattr(im, "get") else im).astype(np.float32) for im in images_s], corners_s, masks_seam))
    return compensator, masks_seam


def compose_panorama(cv, frames: Sequence[np.ndarray], Ks: Sequence[np.ndarray], Rs: Sequence[np.ndarray], warp: str, warper_scale: float,
                     blend: str = "multiband", num_bands: Optional[int] = 5, blend_strength: Optional[float] = None, expos_comp: int = 0,
                     seam_frames: Optional[Sequence[np.ndarray]] = None, seam_aspect: float = 1.0, mask_prep: bool = True,
                     float_pyramids: bool = False, seam: str = "no", timelapse_type: Optional[int] = None, compose_scale: float = 1.0,
                     black_and_white_point: Optional[Tuple[int, int]] = None, blend_masks: Optional[Sequence[np.ndarray]] = None,
                     seam_state=None) -> ComposeResult:
    """``frames`` are the full-resolution frames; with ``compose_scale`` / ``black_and_white_point`` they go through the prologue of
    sde.py:1699-1711 first (``Ks`` must already be the compose-scale cameras, sde.py:1689-1695).  ``blend_masks``: compose-scale
    seamed masks from elsewhere (a recorded run's ``masks_warped_and_seamed``), AND-ed with the warped validity mask in place of
    the seam-scale mask preparation.  ``seam_state``: the result of ``seam_stage`` on these cameras (then ``seam_frames`` is not needed):
    the compose loop alone (sde.py:1673-1930), what bench.py times as ``dropin_umat``."""
    n = len(frames)
    # device-resident form of the same calls: hand the frames in as UMat (cv.UMat(ndarray)); masks are then created as UMats too and
    # the result comes back as UMats (cv2's T-API convention: UMat in -> UMat out)
    on_device = n > 0 and not isinstance(frames[0], np.ndarray) and hasattr(frames[0], "get") and hasattr(cv, "UMat")

    def ones_mask(h, w):
        if not on_device:
            return 255 * np.ones((h, w), np.uint8)
        return _device_ones(cv, h, w)
    if abs(compose_scale - 1) > 1e-1 or black_and_white_point:
        frames = [cv.prepare_frame(f, compose_scale, black_and_white_point) for f in frames]
    # ---- B: seam-scale warps (sde.py:1543-1599) -------------------------------------------------------------------
    masks_seam = None
    if seam_state is not None:
        compensator, masks_seam = seam_state
    elif seam_frames is not None:
        compensator, masks_seam = seam_stage(cv, seam_frames, Ks, Rs, warp, warper_scale, seam_aspect, expos_comp, seam)
    else:
        compensator = cv.detail.ExposureCompensator_createDefault(expos_comp)
    # ---- E: compose scale (sde.py:1684-1698) ------------------------------------------------------------------------
    warper = cv.PyRotationWarper(warp, warper_scale)
    corners, sizes = [], []
    for i in range(n):
        sz = (frames[i].shape[1], frames[i].shape[0])
        roi = warper.warpRoi(sz, Ks[i], Rs[i])
        corners.append(roi[0:2])
        sizes.append(roi[2:4])
    blender = None
    dst_sz = None
    timelapser, tl_frames = None, None
    for idx in range(n):
        img = frames[idx]
        corner, image_warped = warper.warp(img, Ks[idx], Rs[idx], cv.INTER_LINEAR, cv.BORDER_REFLECT)           # :1731
        mask = ones_mask(img.shape[0], img.shape[1])
        _, mask_warped = warper.warp(mask, Ks[idx], Rs[idx], cv.INTER_NEAREST, cv.BORDER_CONSTANT)             # :1740
        if image_warped.dtype == np.uint8:
            compensator.apply(idx, corners[idx], image_warped, mask_warped)                                     # :1754
            image_warped_s = image_warped.astype(np.int16)                                                       # :1755
        else:
            image_warped_s = image_warped
        if mask_prep and masks_seam is not None:
            dilated_mask = cv.dilate(masks_seam[idx], None)                                                      # :1760
            seam_mask = cv.resize(dilated_mask, (mask_warped.shape[1], mask_warped.shape[0]), 0, 0, cv.INTER_LINEAR_EXACT)  # :1767
            mask_warped = cv.bitwise_and(seam_mask, mask_warped)                                                 # :1772
        if blend_masks is not None:
            mask_warped = cv.bitwise_and(blend_masks[idx], mask_warped)
        if blender is None:
            dst_sz = cv.detail.resultRoi(corners=corners, sizes=sizes)                                           # :1807
            blender = make_blender(cv, blend, dst_sz, blend_strength, num_bands, float_pyramids)
        if timelapse_type is not None:                                                                           # :1822-1851
            if timelapser is None:
                timelapser = cv.detail.Timelapser_createDefault(timelapse_type)
                timelapser.initialize(corners, sizes)
                tl_frames = []
            _, untouched = warper.warp(mask, Ks[idx], Rs[idx], cv.INTER_NEAREST, cv.BORDER_CONSTANT)
            ma_tones = np.ones((image_warped_s.shape[0], image_warped_s.shape[1]), np.uint8)
            timelapser.process(cv.bitwise_and(image_warped_s, image_warped_s, mask=untouched), ma_tones, corners[idx])
            dst = timelapser.getDst()
            tl_frames.append(np.array(dst.get() if hasattr(dst, "get") else dst))
        blender.feed(image_warped_s, mask_warped, corners[idx])                                                  # :1886
    nb = blender.numBands() if hasattr(blender, "numBands") else 0
    if on_device:
        result, result_mask, mosaic = blender.blend(None, None, device=True, mosaic=True)                        # :1930 (+ :1938 on the device)
        return ComposeResult(result, result_mask, mosaic, corners, sizes, tuple(dst_sz), nb, tl_frames)
    result, result_mask = blender.blend(None, None)                                                              # :1930
    if result.dtype == np.int16:
        mosaic = np.clip(result, 0, 255).astype(np.uint8)                                                        # imwrite's convertTo(CV_8U)
    else:
        mosaic = np.clip(np.rint(result), 0, 255).astype(np.uint8)
    return ComposeResult(result, result_mask, mosaic, corners, sizes, tuple(dst_sz), nb, tl_frames)


# ---- batched device-resident form ------------------------------------------------------------------------------------
class _Cfg(C.Structure):
    _fields_ = [
        ("warp_type", C.c_char_p), ("warper_scale", C.c_float), ("n_images", C.c_int), ("src_w", C.c_int), ("src_h", C.c_int),
        ("src_depth", C.c_int), ("K", C.POINTER(C.c_float)), ("R", C.POINTER(C.c_float)), ("blend_type", C.c_int), ("num_bands", C.c_int),
        ("sharpness", C.c_float), ("mask_prep", C.c_int), ("seam_w", C.c_int), ("seam_h", C.c_int), ("seam_aspect", C.c_float),
        ("want_result_s16", C.c_int), ("use_graph", C.c_int), ("external_seam_masks", C.c_int), ("coordinate_planes", C.c_int),
    ]


_BLEND_CODE = {"no": 0, "feather": 1, "multiband": 2}


class Composer:
    """One panorama = one ``run``: every frame warp+mask (+apply) -> pyramids -> blend, all on the GPU."""

    def __init__(self, warp: str, warper_scale: float, Ks, Rs, frame_size: Tuple[int, int], blend: str = "multiband", num_bands: int = 5,
                 sharpness: float = 0.02, float_frames: bool = False, mask_prep: bool = False, seam_size: Tuple[int, int] = (0, 0),
                 seam_aspect: float = 1.0, want_result_s16: bool = False, own_stream: bool = False, external_seam_masks: bool = False,
                 coordinate_planes: bool = False):
        """``own_stream=True`` gives the composer a HIP stream of its own: several composers then keep one panorama each in flight
        (bench.py --pipeline); ``result()`` waits for this composer's stream."""
        n = len(Ks)
        self.float_frames = bool(float_frames)
        self._stream = C.c_void_p()
        if own_stream:
            _lib.check(_lib.lib().ssp_stream_create(C.byref(self._stream)))
        self._K = np.ascontiguousarray(np.stack([np.asarray(k, np.float32).reshape(9) for k in Ks]))
        self._R = np.ascontiguousarray(np.stack([np.asarray(r, np.float32).reshape(9) for r in Rs]))
        self._warp = warp.encode()
        cfg = _Cfg(self._warp, float(warper_scale), n, int(frame_size[0]), int(frame_size[1]), 5 if float_frames else 0,
                   self._K.ctypes.data_as(C.POINTER(C.c_float)), self._R.ctypes.data_as(C.POINTER(C.c_float)), _BLEND_CODE[blend], int(num_bands),
                   float(sharpness), int(mask_prep), int(seam_size[0]), int(seam_size[1]), float(seam_aspect), int(want_result_s16), 0, int(external_seam_masks), int(coordinate_planes))
        self._h = C.c_void_p()
        self._use()   # the composer's persistent buffers belong to its own stream
        _lib.check(_lib.lib().ssp_composer_create(C.byref(cfg), C.byref(self._h)))
        self.n = n
        self._comp = None

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                self._use()
                _lib.lib().ssp_composer_destroy(h)
            except Exception:
                pass
            self._h = None
        st = getattr(self, "_stream", None)
        if st is not None and st.value:
            try:
                _lib.lib().ssp_use_stream(None)
                _lib.lib().ssp_stream_destroy(st)
            except Exception:
                pass
            self._stream = C.c_void_p()

    def _use(self) -> None:
        """Every following library call runs on this composer's stream (the library's own one without ``own_stream``)."""
        _lib.check(_lib.lib().ssp_use_stream(self._stream if self._stream.value else None))

    def sync(self) -> None:
        _lib.check(_lib.lib().ssp_stream_sync(self._stream if self._stream.value else None))

    def set_compensator(self, comp) -> None:
        self._comp = comp
        _lib.check(_lib.lib().ssp_composer_set_compensator(self._h, comp._h if comp is not None else None))

    def set_seam_masks(self, masks: Sequence[UMat]) -> None:
        """Seam-scale masks from the caller (what a seam finder returned, sde.py:1618) in place of the warped all-255 masks the composer
        makes itself with ``mask_prep``; call again when their contents change."""
        arr = (C.c_void_p * self.n)(*[m._h.value for m in masks])
        self._seam_keep = list(masks)
        _lib.check(_lib.lib().ssp_composer_set_seam_masks(self._h, self.n, arr))

    def pano_roi(self) -> Tuple[int, int, int, int]:
        roi = (C.c_int * 4)()
        _lib.check(_lib.lib().ssp_composer_pano_roi(self._h, roi))
        return tuple(roi)

    def image_roi(self, i: int) -> Tuple[int, int, int, int]:
        roi = (C.c_int * 4)()
        _lib.check(_lib.lib().ssp_composer_image_roi(self._h, int(i), roi))
        return tuple(roi)

    def parts(self) -> List[Tuple[int, Tuple[int, int, int, int]]]:
        """[(image index, (x, y, w, h))]: what the composer warps and feeds -- a frame's whole roi, or the two live column ranges of a frame
        that straddles u = +-pi*scale (its roi from warpRoi spans the full circle; ``image_roi`` / ``pano_roi`` stay OpenCV's)."""
        n = C.c_int()
        _lib.check(_lib.lib().ssp_composer_num_parts(self._h, C.byref(n)))
        out = []
        for k in range(n.value):
            img, roi = C.c_int(), (C.c_int * 4)()
            _lib.check(_lib.lib().ssp_composer_part(self._h, k, C.byref(img), roi))
            out.append((img.value, tuple(roi)))
        return out

    def run(self, frames: Sequence[UMat]) -> None:
        self._use()
        arr = (C.c_void_p * self.n)(*[f._h.value for f in frames])
        _lib.check(_lib.lib().ssp_composer_run(self._h, arr))

    # -- multi-GPU form of a step (see parallel.py) ------------------------------------------------------------------
    def set_pano_roi(self, roi) -> None:
        _lib.check(_lib.lib().ssp_composer_set_pano_roi(self._h, (C.c_int * 4)(*[int(v) for v in roi])))

    def feed(self, frames: Sequence[UMat]) -> None:
        self._use()
        arr = (C.c_void_p * self.n)(*[f._h.value for f in frames])
        _lib.check(_lib.lib().ssp_composer_feed(self._h, arr))

    def feed_planes(self, frames: Sequence[UMat]) -> None:
        """First half of ``feed`` (multi-GPU): warp, apply, level-0 borders -- the planes strips are exported from."""
        self._use()
        arr = (C.c_void_p * self.n)(*[f._h.value for f in frames])
        _lib.check(_lib.lib().ssp_composer_feed_planes(self._h, arr))

    def feed_pyramids(self) -> None:
        """Second half of ``feed``: the Gaussian pyramids of this GPU's own frames."""
        self._use()
        _lib.check(_lib.lib().ssp_composer_feed_pyramids(self._h))

    def blender_handle(self) -> C.c_void_p:
        h = C.c_void_p()
        _lib.check(_lib.lib().ssp_composer_blender(self._h, C.byref(h)))
        return h

    def finish_region(self, rect) -> None:
        self._use()
        _lib.check(_lib.lib().ssp_composer_finish_region(self._h, *[int(v) for v in rect]))

    def warp_rest_tiles(self) -> Tuple[int, int]:
        """(state, count): what the composer learnt from its first panorama about the tiles the LDS-staged warp cannot stage
        (state 0: nothing yet, 2: known); raises if the device-side list overflowed its capacity."""
        state, count = C.c_int(), C.c_int()
        _lib.check(_lib.lib().ssp_composer_warp_rest_tiles(self._h, C.byref(state), C.byref(count)))
        return state.value, count.value

    def forget_geometry(self) -> None:
        """The next ``run`` rebuilds the projection / resize tables and the rest list, as cv2 rebuilds its maps in every ``warp`` call."""
        _lib.check(_lib.lib().ssp_composer_forget_geometry(self._h))

    def result(self):
        """(mosaic u8, mask u8, result int16|None) as UMats borrowed from the composer (valid until the next run)."""
        if self._stream.value:   # the images were produced on this composer's stream; readers use whatever stream is current
            self.sync()
        mo, mk, rs = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _lib.check(_lib.lib().ssp_composer_result(self._h, C.byref(mo), C.byref(mk), C.byref(rs)))
        out = []
        for h in (mo, mk, rs):
            if h.value:
                _lib.check(_lib.lib().ssp_image_retain(h))
                out.append(UMat.from_handle(h))
            else:
                out.append(None)
        return tuple(out)

    def algorithmic_bytes(self) -> Tuple[float, float, float]:
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        _lib.check(_lib.lib().ssp_composer_algorithmic_bytes(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value
