"""Deferred evaluation of the reference's per-image call sequence on device-resident arrays.

The reference's compose loop (stitching_detailed_enhanced.py:1731-1889) is, per image::

    corner, image_warped = warper.warp(img, K, R, cv.INTER_LINEAR, cv.BORDER_REFLECT)        # :1731
    p, mask_warped = warper.warp(mask, K, R, cv.INTER_NEAREST, cv.BORDER_CONSTANT)             # :1740
    compensator.apply(idx, corners[idx], image_warped, mask_warped)                              # :1754
    image_warped_s = image_warped.astype(np.int16)                                               # :1755
    dilated_mask = cv.dilate(masks_warped[idx], None)                                            # :1760
    seam_mask = cv.resize(dilated_mask, (w, h), 0, 0, cv.INTER_LINEAR_EXACT)                     # :1767
    mask_warped = cv.bitwise_and(seam_mask, mask_warped)                                         # :1772
    blender.feed(cv.UMat(image_warped_s), mask_warped, corners[idx])                             # :1886

and nothing of it is observable before ``blender.blend`` (:1930).  Executed call by call on the GPU every line is a pass over a
warped frame (or several launches); executed as ONE plan it is the batched composer: one fused warp launch for all frames (maps never
materialised, mask / gains / mask preparation in its epilogue, written straight into the blender's planes), one pyramid launch per level.

So with ``UMat`` operands these calls return ``DeferredUMat`` objects -- a ``UMat`` whose shape and dtype are known and whose pixels are
computed when somebody needs them -- and ``blender.feed`` keeps them.  ``blender.blend`` recognises the sequence above (any subset of
apply / astype / mask preparation; all frames through one warper type and scale) and runs it through ``compose.Composer``; anything else
-- a deferred array handed to another function, ``.get()``, a sequence that does not match -- is evaluated call by call by the very
kernels the eager API uses.  Both ways give the same bits (tests/test_dropin.py).

In-place writers (``compensator.apply`` on a plain UMat, ``SeamFinder.find``) bump ``UMat._ver``; a deferred array remembers the version
of its operands and refuses to evaluate after they were overwritten (cv2 would have read the old contents at call time).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Tuple

import numpy as np

from . import _lib
from .umat import S16, U8, UMat

_DEPTH_OF = {np.dtype(np.uint8): 0, np.dtype(np.int16): 3, np.dtype(np.float32): 5}


stats = {"planned": 0, "call_by_call": 0}      # blends that ran as one Composer plan / whose kept feeds were evaluated call by call


def enabled() -> bool:
    import os
    return not os.environ.get("SSP_EAGER")


class DeferredUMat(UMat):
    """A UMat that is computed on first use.  ``op`` in {"warp", "astype", "dilate", "resize_exact", "and"}."""

    def __init__(self, op: str, args: tuple, width: int, height: int, channels: int, dtype):
        self._real: Optional[UMat] = None
        self.op, self.args = op, args
        self._shape = (height, width) if channels == 1 else (height, width, channels)
        self._dtype = np.dtype(dtype)
        self._whc = (int(width), int(height), int(channels))
        self.gain = None          # (compensator, index, generation): compensator.apply recorded on a deferred warp
        self._ver = 0
        self._srcs = [(a, a._ver) for a in args if isinstance(a, UMat)]

    # -- the UMat protocol ---------------------------------------------------------------------------------------
    @property
    def _h(self):
        return self.materialize()._h

    @_h.setter
    def _h(self, v):      # UMat.__del__ clears the handle of plain UMats; nothing to clear here
        pass

    def __del__(self):
        self._real = None

    def info(self):
        if self._real is not None:
            return self._real.info()
        w, h, cn = self._whc
        return w, h, cn, _DEPTH_OF[self._dtype], 0, 0

    @property
    def shape(self):
        return self._shape

    @property
    def dtype(self):
        return self._dtype

    @property
    def pending(self) -> bool:
        return self._real is None

    def get(self) -> np.ndarray:
        return self.materialize().get()

    def astype(self, dtype) -> UMat:
        if self._real is not None:
            return self._real.astype(dtype)
        w, h, cn = self._whc
        return DeferredUMat("astype", (self, np.dtype(dtype)), w, h, cn, dtype)

    # -- evaluation by the eager kernels ---------------------------------------------------------------------------------
    def _check_sources(self) -> None:
        for a, ver in self._srcs:
            if a._ver != ver:
                raise _lib.error(f"deferred {self.op}: an operand was overwritten in place (compensator.apply / SeamFinder.find) after the call that "
                                 "read it; evaluate the result first (.get()) or set SSP_EAGER=1")

    def materialize(self) -> UMat:
        if self._real is not None:
            return self._real
        self._check_sources()
        op, a = self.op, self.args
        if op == "warp":
            warper, src, K, R, interp, border = a
            _, out = warper._warp_now(src, K, R, interp, border)
            if self.gain is not None:
                comp, idx, gen = self.gain
                if comp._gen != gen:
                    raise _lib.error("deferred compensator.apply: the compensator was fed again before the image was evaluated")
                _lib.check(_lib.lib().ssp_comp_apply(comp._h, int(idx), out._h))
        elif op == "astype":
            out = a[0].materialize().astype(a[1]) if isinstance(a[0], DeferredUMat) else a[0].astype(a[1])
        elif op == "dilate":
            o = C.c_void_p()
            _lib.check(_lib.lib().ssp_dilate3x3(a[0]._h, C.byref(o)))
            out = UMat.from_handle(o)
        elif op == "resize_exact":
            o = C.c_void_p()
            _lib.check(_lib.lib().ssp_resize_linear_exact(a[0]._h, int(a[1][0]), int(a[1][1]), C.byref(o)))
            out = UMat.from_handle(o)
        elif op == "and":
            o = C.c_void_p()
            _lib.check(_lib.lib().ssp_bitwise_and(a[0]._h, a[1]._h, C.byref(o)))
            out = UMat.from_handle(o)
        else:  # pragma: no cover
            raise _lib.error(f"deferred: unknown op {op}")
        self._real = out
        return out


def is_pending(x) -> bool:
    return isinstance(x, DeferredUMat) and x._real is None


def all_255(u: UMat) -> bool:
    """Is this 8UC1 array 255 everywhere (the mask of sde.py:1739)?  One device reduction per array and version, cached."""
    cached = getattr(u, "_all255", None)
    if cached is not None and cached[0] == u._ver:
        return cached[1]
    flag = C.c_int()
    _lib.check(_lib.lib().ssp_image_all_equal(u._h, 255, C.byref(flag)))
    u._all255 = (u._ver, bool(flag.value))
    return bool(flag.value)


# ---- the plan behind blender.blend -------------------------------------------------------------------------------------------------------
class _Feed:
    __slots__ = ("img", "mask", "tl")

    def __init__(self, img, mask, tl):
        self.img, self.mask, self.tl = img, mask, (int(tl[0]), int(tl[1]))


def _match_image(node):
    """-> (warp node, wants_int16) or None.  image := astype(int16)?(warp(src, LINEAR|AREA, REFLECT))"""
    to16 = False
    if is_pending(node) and node.op == "astype" and node.args[1] == np.dtype(np.int16):
        to16, node = True, node.args[0]
    if not (is_pending(node) and node.op == "warp"):
        return None
    _, src, _, _, interp, border = node.args
    if interp not in (1, 3) or border != 2 or not isinstance(src, UMat) or is_pending(src):
        return None
    return node, to16


def _match_mask(node):
    """-> (mask warp node, seam mask or None) or None.  mask := warp(ones, NEAREST, CONSTANT) | and(resize_exact(dilate(seam), size), warp(...)) (either order)"""
    seam = None
    if is_pending(node) and node.op == "and":
        a, b = node.args
        for x, y in ((a, b), (b, a)):
            if is_pending(x) and x.op == "resize_exact" and is_pending(x.args[0]) and x.args[0].op == "dilate":
                s = x.args[0].args[0]
                if isinstance(s, UMat) and not is_pending(s) and is_pending(y) and y.op == "warp" and tuple(x.args[1]) == (y._whc[0], y._whc[1]):
                    seam, node = s, y
                    break
        else:
            return None
    if not (is_pending(node) and node.op == "warp"):
        return None
    _, src, _, _, interp, border = node.args
    if interp != 0 or border != 0 or not isinstance(src, UMat) or is_pending(src) or src.info()[2] != 1 or src.info()[3] != U8:
        return None
    return node, seam


_composers: List[tuple] = []      # (key, Composer), most recently used first


def plan_for(blender, feeds: List[_Feed]):
    """The feeds of ``blender`` as one Composer run, or None when they are not the reference's sequence."""
    from . import compose as cmp
    if not feeds or blender._prepared is None:
        return None
    n = len(feeds)
    imgs, masks = [], []
    for f in feeds:
        mi, mm = _match_image(f.img), _match_mask(f.mask)
        if mi is None or mm is None:
            return None
        imgs.append(mi); masks.append(mm)
    w0 = imgs[0][0].args[0]
    size0 = imgs[0][0].args[1].info()[:4]
    float_frames = size0[3] == 5
    if size0[2] != 3 or size0[3] not in (0, 5):
        return None
    seam_given = masks[0][1] is not None
    comp0 = imgs[0][0].gain
    Ks, Rs, srcs, seams = [], [], [], []
    for k, ((wn, to16), (mn, seam), f) in enumerate(zip(imgs, masks, feeds)):
        warper, src, K, R, _, _ = wn.args
        mw, msrc, mK, mR, _, _ = mn.args
        if warper.type != w0.type or warper.getScale() != w0.getScale() or mw.type != w0.type or mw.getScale() != w0.getScale():
            return None
        if src.info()[:4] != size0 or msrc.info()[:2] != size0[:2] or K.tobytes() != mK.tobytes() or R.tobytes() != mR.tobytes():
            return None
        if (seam is not None) != seam_given or to16 != (not float_frames) or f.tl != tuple(wn.corner) or tuple(mn.corner) != tuple(wn.corner):
            return None
        g = wn.gain
        if (g is None) != (comp0 is None) or (g is not None and (g[0] is not comp0[0] or g[1] != k or g[2] != g[0]._gen)):
            return None
        if not all_255(msrc):
            return None
        wn._check_sources(); mn._check_sources()
        Ks.append(K); Rs.append(R); srcs.append(src); seams.append(seam)
    btype = blender._type
    if float_frames and not (btype == 2 and blender._float):
        return None
    if btype == 2 and not float_frames and blender._float:
        return None
    key = (w0.type, float(w0.getScale()), n, size0, b"".join(k.tobytes() for k in Ks), b"".join(r.tobytes() for r in Rs), btype,
           blender._want_bands if btype == 2 else 0, blender._sharpness if btype == 1 else 0.0, seam_given,
           tuple((s.info()[0], s.info()[1]) for s in seams) if seam_given else None, tuple(blender._prepared))
    comp = None
    for i, (k, c) in enumerate(_composers):
        if k == key:
            comp = c
            if i:
                _composers.insert(0, _composers.pop(i))
            break
    if comp is None:
        blend = {0: "no", 1: "feather", 2: "multiband"}[btype]
        comp = cmp.Composer(w0.type, w0.getScale(), Ks, Rs, (size0[0], size0[1]), blend=blend, num_bands=blender._want_bands, sharpness=blender._sharpness,
                            float_frames=float_frames, mask_prep=seam_given, external_seam_masks=seam_given, want_result_s16=True)
        if tuple(comp.pano_roi()) != tuple(blender._prepared):
            try:
                comp.set_pano_roi(blender._prepared)       # prepare() got another rectangle than resultRoi of these images: it must contain them
            except _lib.error:
                return None
        _composers.insert(0, (key, comp))
        del _composers[4:]
    if seam_given:
        stamp = tuple((s._h.value, s._ver) for s in seams)
        if getattr(comp, "_seam_stamp", None) != stamp:
            comp.set_seam_masks(seams)
            comp._seam_stamp = stamp
    want = comp0[0] if comp0 is not None else None
    if getattr(comp, "_comp_obj", None) is not want:
        comp.set_compensator(want)
        comp._comp_obj = want
    return comp, srcs
