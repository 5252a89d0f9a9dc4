"""MI355X-native warp / exposure-compensate / blend hot path behind the cv2 object protocol used by
joachim-broser/opencv-starry-sky-panorama-stitcher's ``compose_imgs_to_panorama``
(stitching_detailed_enhanced.py:1355-1954).

    import opencv_starry_sky_panorama_stitcher_amd as cv
    warper = cv.PyRotationWarper("spherical", scale)            # sde.py:1684
    corner, warped = warper.warp(img, K, R, cv.INTER_LINEAR, cv.BORDER_REFLECT)
    blender = cv.detail_MultiBandBlender(); blender.setNumBands(5)
    blender.prepare(cv.detail.resultRoi(corners=corners, sizes=sizes))
    blender.feed(cv.UMat(warped.astype("int16")), mask, corner); result, result_mask = blender.blend(None, None)

All arithmetic runs in hand-written HIP kernels for gfx950 (libssp_hip.so, C ABI in include/ssp.h); there is
no CPU fallback -- without the library or a GPU every operation raises ``error``.
"""
from . import _lib, camera, detail, imgproc  # noqa: F401
from ._lib import error  # noqa: F401
from .detail import (  # noqa: F401
    Blender_createDefault as detail_Blender_createDefault,
    ExposureCompensator_createDefault as detail_ExposureCompensator_createDefault,
)
from .detail import BlocksChannelsCompensator as detail_BlocksChannelsCompensator  # noqa: F401
from .detail import BlocksGainCompensator as detail_BlocksGainCompensator  # noqa: F401
from .detail import ChannelsCompensator as detail_ChannelsCompensator  # noqa: F401
from .detail import FeatherBlender as detail_FeatherBlender  # noqa: F401
from .detail import GainCompensator as detail_GainCompensator  # noqa: F401
from .detail import MultiBandBlender as detail_MultiBandBlender  # noqa: F401
from .detail import DpSeamFinder as detail_DpSeamFinder  # noqa: F401
from .imgproc import (  # noqa: F401
    BORDER_CONSTANT, BORDER_REFLECT, BORDER_REFLECT_101, BORDER_REPLICATE, BORDER_WRAP,
    INTER_AREA, INTER_LINEAR, INTER_LINEAR_EXACT, INTER_NEAREST,
    adjust_black_and_white_point, bitwise_and, dilate, prepare_frame, resize,
)
from .umat import UMat  # noqa: F401
from .warpers import WARP_TYPES, PyRotationWarper  # noqa: F401

__all__ = [
    "PyRotationWarper", "UMat", "detail", "error", "dilate", "resize", "bitwise_and", "adjust_black_and_white_point", "prepare_frame",
    "detail_MultiBandBlender", "detail_FeatherBlender", "detail_ChannelsCompensator", "detail_BlocksChannelsCompensator",
    "INTER_NEAREST", "INTER_LINEAR", "INTER_AREA", "INTER_LINEAR_EXACT", "BORDER_CONSTANT", "BORDER_REFLECT",
]


def device_available() -> bool:
    """True when libssp_hip.so loads and a gfx950 device answers."""
    try:
        import ctypes as C

        n = C.c_int()
        _lib.lib().ssp_device_count(C.byref(n))
        if n.value < 1:
            return False
        return _lib.lib().ssp_init(0) == 0
    except Exception:
        return False
