"""cv2-shaped namespace over the CPU oracle, so that the same call sequence (compose.compose_panorama, or a
test body) can run against the oracle and against the HIP library.  Test infrastructure only."""
import os
import sys
import types

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import oracle as orc  # noqa: E402

INTER_NEAREST, INTER_LINEAR, INTER_AREA, INTER_LINEAR_EXACT = 0, 1, 3, 5
BORDER_CONSTANT, BORDER_REPLICATE, BORDER_REFLECT, BORDER_WRAP, BORDER_REFLECT_101 = 0, 1, 2, 3, 4

PyRotationWarper = orc.PyRotationWarper


def dilate(src, kernel=None):
    return orc.dilate(src)


def resize(src, dsize, fx=0, fy=0, interpolation=INTER_LINEAR_EXACT):
    if interpolation == INTER_AREA:
        return orc.resize_area(src, fx, fy)
    assert interpolation == INTER_LINEAR_EXACT
    return orc.resize_linear_exact(src, dsize)


def adjust_black_and_white_point(img, tpl):
    return orc.bw_point_lut(*tpl)[img] if tpl else img


def prepare_frame(full_img, compose_scale, tpl=None):
    if abs(compose_scale - 1) > 1e-1:
        return orc.resize_area(full_img, compose_scale, compose_scale, tpl)
    return adjust_black_and_white_point(full_img, tpl)


def bitwise_and(a, b, dst=None, mask=None):
    r = np.bitwise_and(a, b)
    if mask is not None:
        r = np.where((mask != 0)[(...,) + (None,) * (r.ndim - 2)], r, 0).astype(r.dtype)
    return r


class _Blender(orc.Blender):
    def blend(self, dst=None, dst_mask=None):
        return super().blend()


def detail_MultiBandBlender(try_gpu=0, num_bands=5, weight_type=5, float_pyramids=False):
    b = _Blender(orc.BLEND_MULTIBAND, float_mode=float_pyramids)
    b.setNumBands(num_bands)
    return b


def detail_FeatherBlender(sharpness=0.02):
    b = _Blender(orc.BLEND_FEATHER)
    b.setSharpness(sharpness)
    return b


def _blender_default(type, try_gpu=False):
    return {0: lambda: _Blender(orc.BLEND_NO), 1: detail_FeatherBlender, 2: detail_MultiBandBlender}[type]()


class _Comp(orc.ExposureCompensator):
    def feed(self, corners, images, masks):
        return super().feed(corners, images, masks)


detail = types.SimpleNamespace(
    Blender_NO=0, Blender_FEATHER=1, Blender_MULTI_BAND=2,
    Blender_createDefault=_blender_default,
    ExposureCompensator_createDefault=lambda t: _Comp(t),
    resultRoi=lambda corners, sizes: orc.resultRoi(corners, sizes),
    SeamFinder_NO=0, SeamFinder_VORONOI_SEAM=1, SeamFinder_DP_SEAM=2,
    SeamFinder_createDefault=lambda t: orc.SeamFinder(t),
    Timelapser_AS_IS=0, Timelapser_CROP=1,
    Timelapser_createDefault=lambda t: orc.Timelapser(t),
)


def detail_ChannelsCompensator(nr_feeds=1):
    return _Comp(orc.COMP_CHANNELS, nr_feeds=nr_feeds)


def detail_BlocksChannelsCompensator(bw=32, bh=32, nr_feeds=1):
    return _Comp(orc.COMP_CHANNELS_BLOCKS, bw, bh, nr_feeds)


def detail_DpSeamFinder(costFunc="COLOR"):
    return orc.SeamFinder(orc.SEAM_DP, costFunc)
