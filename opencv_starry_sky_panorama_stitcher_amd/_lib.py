"""ctypes binding of libssp_hip.so (the C ABI declared in include/ssp.h).

There is no CPU fallback: if the shared library is missing, or no gfx950 device is present, every
operation raises ``error`` (the stand-in for ``cv2.error``).
"""
from __future__ import annotations

import ctypes as C
import os
import re

_PKG = os.path.dirname(os.path.abspath(__file__))
# SSP_LIB: another build of the SAME library (A/B measurements of kernel variants, tools/ab_bench.sh) -- never a different backend
LIB_PATH = os.environ.get("SSP_LIB") or os.path.join(_PKG, "libssp_hip.so")
LIB_IS_VARIANT = bool(os.environ.get("SSP_LIB"))      # bench.py records the path and a hash of whatever was loaded
HEADER_PATH = os.path.join(os.path.dirname(_PKG), "include", "ssp.h")


class error(RuntimeError):
    """Raised where cv2 would raise ``cv2.error`` (sde.py:1567 catches it around warper.warp)."""

    def __init__(self, msg: str, code: int = -1):
        super().__init__(msg)
        self.code = code


_lib = None


def declared_symbols() -> list:
    """Every function name declared in include/ssp.h."""
    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ssp_[a-z0-9_]+)\s*\(", text)))


_vp = C.c_void_p
_vpp = C.POINTER(C.c_void_p)
_ip = C.POINTER(C.c_int)
_fp = C.POINTER(C.c_float)


def _declare(lib: C.CDLL) -> None:
    lib.ssp_last_error.restype = C.c_char_p
    sigs = {
        "ssp_init": [C.c_int],
        "ssp_device_count": [_ip],
        "ssp_device_name": [C.c_char_p, C.c_int],
        "ssp_sync": [],
        "ssp_device_copy": [_vp, _vp, C.c_size_t],
        "ssp_device_copy_kernel": [_vp, _vp, C.c_size_t],
        "ssp_stream_create": [_vpp],
        "ssp_stream_destroy": [_vp],
        "ssp_stream_sync": [_vp],
        "ssp_use_stream": [_vp],
        "ssp_set_stream": [_vp],
        "ssp_current_stream": [C.POINTER(C.c_void_p)],
        "ssp_pool_stats": [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)],
        "ssp_pool_trim": [],
        "ssp_timer_create": [_vpp],
        "ssp_timer_start": [_vp],
        "ssp_timer_stop": [_vp],
        "ssp_timer_elapsed_ms": [_vp, _fp],
        "ssp_timer_destroy": [_vp],
        "ssp_profile_enable": [C.c_int],
        "ssp_profile_reset": [],
        "ssp_profile_count": [_ip],
        "ssp_profile_get": [C.c_int, C.c_char_p, C.c_int, _ip, _fp, C.POINTER(C.c_double)],
        "ssp_calibrate_stream": [C.c_int, C.c_size_t, C.c_int],
        "ssp_image_create": [C.c_int, C.c_int, C.c_int, C.c_int, _vpp],
        "ssp_image_upload": [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vpp],
        "ssp_image_wrap": [_vp, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, _vpp],
        "ssp_image_download": [_vp, _vp],
        "ssp_image_info": [_vp, _ip, _ip, _ip, _ip, C.POINTER(C.c_size_t), _vpp],
        "ssp_image_retain": [_vp],
        "ssp_image_release": [_vp],
        "ssp_image_fill": [_vp, C.c_double],
        "ssp_image_convert": [_vp, C.c_int, _vpp],
        "ssp_image_all_equal": [_vp, C.c_int, _ip],
        "ssp_warper_create": [C.c_char_p, C.c_float, _vpp],
        "ssp_warper_destroy": [_vp],
        "ssp_warper_get_scale": [_vp, _fp],
        "ssp_warper_set_scale": [_vp, C.c_float],
        "ssp_warper_roi": [_vp, C.c_int, C.c_int, _fp, _fp, _ip],
        "ssp_warper_live_parts": [_vp, C.c_int, C.c_int, _fp, _fp, C.c_int, _ip, C.c_int, _ip],
        "ssp_warper_warp": [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp, C.c_int, C.c_int, _vp, C.c_int, C.c_int, _ip],
        "ssp_warper_warp_image": [_vp, _vp, _fp, _fp, C.c_int, C.c_int, _vpp, _ip],
        "ssp_warper_warp_with_mask": [_vp, _vp, _fp, _fp, C.c_int, _vpp, _vpp, _ip],
        "ssp_warper_warp_backward": [_vp, _vp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _vpp],
        "ssp_warper_build_maps": [_vp, C.c_int, C.c_int, _fp, _fp, _fp, _fp, C.c_int, C.c_int, _ip],
        "ssp_warper_warp_point": [_vp, C.c_float, C.c_float, _fp, _fp, _fp],
        "ssp_warper_warp_point_backward": [_vp, C.c_float, C.c_float, _fp, _fp, _fp],
        "ssp_result_roi": [C.c_int, _ip, _ip, _ip],
        "ssp_dilate3x3": [_vp, _vpp],
        "ssp_resize_linear_exact": [_vp, C.c_int, C.c_int, _vpp],
        "ssp_bitwise_and": [_vp, _vp, _vpp],
        "ssp_resize_area": [_vp, C.c_double, C.c_double, _vp, _vpp],
        "ssp_bw_point_lut": [C.c_int, C.c_int, _vp],
        "ssp_apply_lut": [_vp, _vp, _vpp],
        "ssp_seam_voronoi": [C.c_int, _ip, C.POINTER(C.c_void_p)],
        "ssp_seam_dp": [C.c_int, _ip, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, _ip],
        "ssp_timelapser_create": [C.c_int, _vpp],
        "ssp_timelapser_destroy": [_vp],
        "ssp_timelapser_initialize": [_vp, C.c_int, _ip, _ip],
        "ssp_timelapser_process": [_vp, _vp, C.c_int, C.c_int],
        "ssp_timelapser_get_dst": [_vp, _vpp],
        "ssp_timelapser_dst_roi": [_vp, _ip],
        "ssp_bitwise_and_masked": [_vp, _vp, _vp, _vpp],
        "ssp_comp_create": [C.c_int, _vpp],
        "ssp_comp_destroy": [_vp],
        "ssp_comp_set_nr_feeds": [_vp, C.c_int],
        "ssp_comp_set_block_size": [_vp, C.c_int, C.c_int],
        "ssp_comp_set_nr_filtering": [_vp, C.c_int],
        "ssp_comp_feed": [_vp, C.c_int, _ip, _vpp, _vpp],
        "ssp_comp_apply": [_vp, C.c_int, _vp],
        "ssp_comp_num_images": [_vp, _ip],
        "ssp_comp_get_gains": [_vp, C.POINTER(C.c_double), C.c_int, _ip],
        "ssp_comp_get_gain_map": [_vp, C.c_int, _fp, C.c_int, _ip, _ip, _ip],
        "ssp_comp_set_gains": [_vp, C.POINTER(C.c_double), C.c_int],
        "ssp_comp_set_gain_map": [_vp, C.c_int, _fp, C.c_int, C.c_int, C.c_int],
        "ssp_blender_create": [C.c_int, _vpp],
        "ssp_blender_destroy": [_vp],
        "ssp_blender_set_num_bands": [_vp, C.c_int],
        "ssp_blender_get_num_bands": [_vp, _ip],
        "ssp_blender_set_sharpness": [_vp, C.c_float],
        "ssp_blender_set_float_mode": [_vp, C.c_int],
        "ssp_blender_prepare": [_vp, C.c_int, C.c_int, C.c_int, C.c_int],
        "ssp_blender_feed": [_vp, _vp, _vp, C.c_int, C.c_int],
        "ssp_blender_feed_batch": [_vp, C.c_int, _vpp, _vpp, _ip],
        "ssp_blender_blend": [_vp, _vpp, _vpp, _vpp],
        "ssp_blender_level_info": [_vp, C.c_int, _ip, _ip],
        "ssp_blender_export_partial": [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp],
        "ssp_blender_import_partial": [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp],
        "ssp_blender_export_strips": [_vp, C.c_int, _ip, _ip, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)],
        "ssp_blender_feed_strips": [_vp, C.c_int, _ip, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)],
        "ssp_blender_feed_strips_begin": [_vp, C.c_int, _ip, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)],
        "ssp_blender_feed_end_pair": [_vp, _vp],
        "ssp_blender_export_level_strips": [_vp, C.c_int, _ip, _ip, C.POINTER(C.c_void_p), C.c_int],
        "ssp_blender_feed_level_strips": [_vp, C.c_int, _ip, _ip, C.POINTER(C.c_void_p), C.c_int],
        "ssp_level_strip_buffer_bytes": [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)],
        "ssp_blender_set_strip_layout": [_vp, C.c_int],
        "ssp_strip_buffer_bytes": [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)],
        "ssp_blender_order_feeds": [_vp, _ip, C.c_int],
        "ssp_blender_blend_region": [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vpp, _vpp, _vpp],
        "ssp_composer_create": [_vp, _vpp],
        "ssp_composer_destroy": [_vp],
        "ssp_composer_set_compensator": [_vp, _vp],
        "ssp_composer_set_seam_masks": [_vp, C.c_int, _vpp],
        "ssp_composer_warp_rest_tiles": [_vp, _ip, _ip],
        "ssp_composer_warp_rest_tiles_nowait": [_vp, _ip, _ip],
        "ssp_composer_forget_geometry": [_vp],
        "ssp_composer_pano_roi": [_vp, _ip],
        "ssp_composer_image_roi": [_vp, C.c_int, _ip],
        "ssp_composer_num_parts": [_vp, _ip],
        "ssp_composer_part": [_vp, C.c_int, _ip, _ip],
        "ssp_composer_run": [_vp, _vpp],
        "ssp_composer_result": [_vp, _vpp, _vpp, _vpp],
        "ssp_composer_set_pano_roi": [_vp, _ip],
        "ssp_composer_feed": [_vp, _vpp],
        "ssp_composer_feed_planes": [_vp, _vpp],
        "ssp_composer_feed_pyramids": [_vp],
        "ssp_composer_blender": [_vp, _vpp],
        "ssp_composer_finish_region": [_vp, C.c_int, C.c_int, C.c_int, C.c_int],
        "ssp_composer_algorithmic_bytes": [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)],
    }
    for name, argtypes in sigs.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.ssp_version.restype = C.c_int


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise error(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback."
            )
        if LIB_IS_VARIANT:
            import sys
            print(f"opencv_starry_sky_panorama_stitcher_amd: SSP_LIB is set -- loading the variant build {LIB_PATH} instead of the shipped library", file=sys.stderr)
        try:
            loaded = C.CDLL(LIB_PATH)
        except OSError as exc:  # pragma: no cover
            raise error(f"cannot load {LIB_PATH}: {exc}") from exc
        _declare(loaded)
        _lib = loaded
    return _lib


def lib_identity() -> dict:
    """Which binary is loaded: path, SHA-256 (first 16 hex digits), whether SSP_LIB redirected the load."""
    import hashlib
    h = hashlib.sha256()
    with open(LIB_PATH, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return {"path": LIB_PATH, "sha256_16": h.hexdigest()[:16], "variant_via_SSP_LIB": LIB_IS_VARIANT}


def check(rc: int) -> None:
    if rc != 0:
        raise error(lib().ssp_last_error().decode(errors="replace"), rc)
