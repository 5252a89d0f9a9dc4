"""GPU parity tests: the HIP path (through the C ABI, via the cv2-shaped Python mirror) against the CPU oracle on
the same seeded inputs.  Integer stages must match bit for bit; the float stages' tolerance is stated per test.
Run on the GPU box:  python -m pytest tests -m gpu -x -q
"""
import os

import numpy as np
import pytest

import opencv_starry_sky_panorama_stitcher_amd as cv
from opencv_starry_sky_panorama_stitcher_amd import compose as cmp
from opencv_starry_sky_panorama_stitcher_amd import starfield

import oracle_cv as ocv
from util import big_frame, camera, star_patch

pytestmark = pytest.mark.gpu

WARPS = list(cv.WARP_TYPES)


def _affine_R(yaw_px=30.0):
    # the affine warper takes a 3x3 homography-like matrix: rotation+translation in pixels
    a = np.deg2rad(3.0)
    return np.array([[np.cos(a), -np.sin(a), yaw_px], [np.sin(a), np.cos(a), -12.0], [0, 0, 1]], dtype=np.float32)


def _cam_for(warp, w, h, yaw=12.0, pitch=-7.0, roll=4.0):
    K, R, f = camera(w, h, 60.0, yaw, pitch, roll)
    if warp == "affine":
        R = _affine_R()
    return K, R, f


@pytest.mark.parametrize("warp", WARPS)
def test_warp_roi_and_maps_bit_exact(warp):
    """warpRoi and buildMaps for all 16 projections: bit-identical to the oracle (same binary32 op order, same
    deterministic transcendentals)."""
    w, h = 161, 97
    K, R, f = _cam_for(warp, w, h)
    g = cv.PyRotationWarper(warp, f)
    o = ocv.PyRotationWarper(warp, f)
    roi_g = g.warpRoi((w, h), K, R)
    roi_o = o.warpRoi((w, h), K, R)
    assert roi_g == roi_o
    _, xg, yg = g.buildMaps((w, h), K, R)
    _, xo, yo = o.buildMaps((w, h), K, R)
    assert np.array_equal(xg.view(np.uint32), xo.view(np.uint32))
    assert np.array_equal(yg.view(np.uint32), yo.view(np.uint32))


@pytest.mark.parametrize("warp", WARPS)
def test_warp_u8c3_linear_reflect_bit_exact(warp):
    w, h = 203, 131
    img = star_patch(w, h, seed=11)
    K, R, f = _cam_for(warp, w, h)
    cg, dg = cv.PyRotationWarper(warp, f).warp(img, K, R, cv.INTER_LINEAR, cv.BORDER_REFLECT)
    co, do = ocv.PyRotationWarper(warp, f).warp(img, K, R, ocv.INTER_LINEAR, ocv.BORDER_REFLECT)
    assert tuple(cg) == tuple(co)
    assert dg.shape == do.shape and np.array_equal(dg, do)


@pytest.mark.parametrize("warp", ["spherical", "cylindrical", "fisheye", "plane"])
def test_warp_mask_nearest_constant_bit_exact(warp):
    w, h = 150, 90
    mask = 255 * np.ones((h, w), np.uint8)
    K, R, f = _cam_for(warp, w, h, yaw=-20, pitch=10, roll=-3)
    cg, dg = cv.PyRotationWarper(warp, f).warp(mask, K, R, cv.INTER_NEAREST, cv.BORDER_CONSTANT)
    co, do = ocv.PyRotationWarper(warp, f).warp(mask, K, R, ocv.INTER_NEAREST, ocv.BORDER_CONSTANT)
    assert tuple(cg) == tuple(co) and np.array_equal(dg, do)
    assert set(np.unique(dg)) <= {0, 255}


@pytest.mark.parametrize("warp", ["spherical", "cylindrical", "mercator", "stereographic"])
def test_fused_warp_with_mask_equals_two_calls(warp):
    """The one-pass image+mask kernel equals the reference's two warp calls (sde.py:1731 + :1740)."""
    w, h = 260, 150
    img = star_patch(w, h, seed=5)
    K, R, f = _cam_for(warp, w, h, yaw=33, pitch=5, roll=1)
    o = ocv.PyRotationWarper(warp, f)
    co, do = o.warp(img, K, R, ocv.INTER_LINEAR, ocv.BORDER_REFLECT)
    _, mo = o.warp(255 * np.ones((h, w), np.uint8), K, R, ocv.INTER_NEAREST, ocv.BORDER_CONSTANT)
    cg, dg, mg = cv.PyRotationWarper(warp, f).warpWithMask(img, K, R, cv.BORDER_REFLECT)
    assert tuple(cg) == tuple(co)
    assert np.array_equal(dg, do)
    assert np.array_equal(mg, mo)


@pytest.mark.parametrize("border", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("interp", [0, 1])
def test_warp_all_border_modes(border, interp):
    w, h = 90, 60
    img = star_patch(w, h, seed=3)
    K, R, f = _cam_for("spherical", w, h, yaw=50, pitch=-25, roll=9)
    _, dg = cv.PyRotationWarper("spherical", f).warp(img, K, R, interp, border)
    _, do = ocv.PyRotationWarper("spherical", f).warp(img, K, R, interp, border)
    assert np.array_equal(dg, do)


@pytest.mark.parametrize("border", [1, 2, 4])
@pytest.mark.parametrize("size", [(700, 300), (260, 9), (3, 40), (4, 5), (2, 7)])
def test_warp_mirror_borders_outline_and_tiny_frames(border, size):
    """Waves that straddle the frame outline take the mirrored two-read path (one reflection away), far pixels and frames
    narrower than 3 columns the per-tap path; both must equal cv::remap's borderInterpolate."""
    w, h = size
    img = star_patch(w, h, seed=w + h, n_stars=40)
    K, R, f = camera(w, h, 65.0, yaw=8.0, pitch=-6.0, roll=21.0)
    for warp in ("spherical", "cylindrical"):
        cg, dg, mg = cv.PyRotationWarper(warp, f).warpWithMask(img, K, R, border)
        co, do = ocv.PyRotationWarper(warp, f).warp(img, K, R, ocv.INTER_LINEAR, border)
        assert cg == co and np.array_equal(dg, do), (warp, border, size)


@pytest.mark.parametrize("cn", [1, 3])
@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
def test_warp_types_generic_kernel(cn, dtype):
    """8U/32F, 1/3 channels through the generic kernel.  f32 interpolation uses float weights in OpenCV's order;
    the HIP kernel keeps that order without FMA contraction, so equality is exact."""
    w, h = 120, 80
    img = star_patch(w, h, seed=8, cn=cn, dtype=dtype)
    K, R, f = _cam_for("fisheye", w, h)
    _, dg = cv.PyRotationWarper("fisheye", f).warp(img, K, R, cv.INTER_LINEAR, cv.BORDER_REFLECT)
    _, do = ocv.PyRotationWarper("fisheye", f).warp(img, K, R, ocv.INTER_LINEAR, ocv.BORDER_REFLECT)
    assert dg.dtype == do.dtype and np.array_equal(dg, do)


def test_warp_behind_camera_and_wide_roi():
    """A frame looking backwards across u = +-pi*scale: OpenCV's by-border roi spans the whole sphere; every pixel
    (including z <= 0 -> (-1,-1) and far reflected taps) must still match."""
    w, h = 96, 64
    img = star_patch(w, h, seed=21)
    K, R, f = camera(w, h, 70.0, yaw=179.0, pitch=3.0)
    g = cv.PyRotationWarper("spherical", f)
    o = ocv.PyRotationWarper("spherical", f)
    assert g.warpRoi((w, h), K, R) == o.warpRoi((w, h), K, R)
    cg, dg, mg = g.warpWithMask(img, K, R, cv.BORDER_REFLECT)
    co, do = o.warp(img, K, R, ocv.INTER_LINEAR, ocv.BORDER_REFLECT)
    _, mo = o.warp(255 * np.ones((h, w), np.uint8), K, R, ocv.INTER_NEAREST, ocv.BORDER_CONSTANT)
    assert np.array_equal(dg, do) and np.array_equal(mg, mo)
    assert dg.shape[1] > 3 * w  # the roi really is the wide one


def test_spherical_pole_inside_frame():
    w, h = 120, 90
    K, R, f = camera(w, h, 80.0, yaw=10.0, pitch=-88.0)
    assert cv.PyRotationWarper("spherical", f).warpRoi((w, h), K, R) == ocv.PyRotationWarper("spherical", f).warpRoi((w, h), K, R)


def test_warp_errors_like_cv2():
    K, R, f = camera(32, 32)
    with pytest.raises(cv.error):
        cv.PyRotationWarper("no-such-warper", 10.0)
    w = cv.PyRotationWarper("plane", f)
    with pytest.raises(cv.error):
        w.warpRoi((32, 32), K.astype(np.float64), R)  # K must be CV_32F (sde.py:1550/:1695 cast it)
    with pytest.raises(cv.error):
        w.warp(np.zeros((8, 8, 3), np.float64), K, R, 1, 2)


# ---- mask preparation ---------------------------------------------------------------------------------------------------
def test_dilate_resize_and_bit_exact():
    rng = np.random.default_rng(2)
    m = (rng.uniform(size=(37, 53)) > 0.6).astype(np.uint8) * 255
    assert np.array_equal(cv.dilate(m, None), ocv.dilate(m, None))
    for dsize in [(211, 140), (53, 37), (54, 38), (400, 39), (30, 20)]:
        assert np.array_equal(cv.resize(m, dsize, 0, 0, cv.INTER_LINEAR_EXACT), ocv.resize(m, dsize, 0, 0, ocv.INTER_LINEAR_EXACT)), dsize
    # destinations of a megapixel and more take four rows per lane: odd sizes, a ragged last row group, down- and up-scaling
    big = rng.integers(0, 256, (131, 173), dtype=np.uint8)
    for dsize in [(1531, 1207), (2051, 514), (1024, 1030)]:
        assert np.array_equal(cv.resize(big, dsize, 0, 0, cv.INTER_LINEAR_EXACT), ocv.resize(big, dsize, 0, 0, ocv.INTER_LINEAR_EXACT)), dsize
    huge = rng.integers(0, 256, (1500, 1300), dtype=np.uint8)
    assert np.array_equal(cv.resize(huge, (1201, 1003), 0, 0, cv.INTER_LINEAR_EXACT), ocv.resize(huge, (1201, 1003), 0, 0, ocv.INTER_LINEAR_EXACT))
    a = rng.integers(0, 256, (20, 31), dtype=np.uint8)
    b = rng.integers(0, 256, (20, 31), dtype=np.uint8)
    assert np.array_equal(cv.bitwise_and(a, b), a & b)


# ---- frame prologue (sde.py:1699-1711) -------------------------------------------------------------------------------------
# fractional factors cover 1..4 twelve-byte loads per row and the generic kernel (> 14 source pixels per destination pixel)
@pytest.mark.parametrize("f", [0.9, 0.37, 0.1829, 0.1203, 0.085, 0.06, 0.5, 1.0 / 3.0, 0.25, 0.125])
def test_resize_area_bit_exact(f):
    rng = np.random.default_rng(int(f * 1e4))
    img = rng.integers(0, 256, size=(203, 331, 3), dtype=np.uint8)
    assert np.array_equal(cv.resize(img, None, fx=f, fy=f, interpolation=cv.INTER_AREA), ocv.resize(img, None, fx=f, fy=f, interpolation=ocv.INTER_AREA)), f
    g = img[:, :, 1].copy()
    assert np.array_equal(cv.resize(g, None, fx=f, fy=0.7 * f, interpolation=cv.INTER_AREA), ocv.resize(g, None, fx=f, fy=0.7 * f, interpolation=ocv.INTER_AREA)), f


@pytest.mark.parametrize("tpl", [(0, 150), (12, 201), None])
def test_black_and_white_point_and_fused_prologue_bit_exact(tpl):
    img = star_patch(517, 389, seed=8)
    assert np.array_equal(cv.adjust_black_and_white_point(img, tpl), ocv.adjust_black_and_white_point(img, tpl))
    for scale in (0.3405, 0.95):   # sde.py:1700: the resize is skipped when abs(compose_scale - 1) <= 1e-1
        got = cv.prepare_frame(img, scale, tpl)
        assert np.array_equal(got, ocv.prepare_frame(img, scale, tpl)), scale
        two_steps = cv.adjust_black_and_white_point(cv.resize(img, None, fx=scale, fy=scale, interpolation=cv.INTER_AREA) if abs(scale - 1) > 1e-1 else img, tpl)
        assert np.array_equal(got, two_steps)


def test_prologue_at_camera_resolution_and_errors():
    # the reference's frames are 5184 x 3456 decimated to compose_megapix 0.6: compose_scale = sqrt(0.6e6 / (w h))
    w, h = 5184, 3456
    scale = min(1.0, float(np.sqrt(0.6e6 / (w * h))))
    img = big_frame(w, h, seed=4)
    got = cv.prepare_frame(cv.UMat(img), scale, (0, 150)).get()
    assert got.shape == (int(np.rint(h * scale)), int(np.rint(w * scale)), 3)
    assert np.array_equal(got, ocv.prepare_frame(img, scale, (0, 150)))
    with pytest.raises(cv.error):
        cv.resize(img[:64, :64], None, fx=1.5, fy=1.5, interpolation=cv.INTER_AREA)      # only decimation is on the path
    with pytest.raises(cv.error):
        cv.resize(img[:64, :64].astype(np.int16), None, fx=0.5, fy=0.5, interpolation=cv.INTER_AREA)
    with pytest.raises(cv.error):
        cv.adjust_black_and_white_point(img[:64, :64], (200, 100))


# ---- seam finder and timelapser -------------------------------------------------------------------
@pytest.mark.parametrize("seed", [0, 1, 2, 5])
def test_voronoi_seam_finder_bit_exact(seed):
    from test_oracle_pixels import _seam_case
    corners, masks = _seam_case(seed, n=5)
    want = ocv.detail.SeamFinder_createDefault(ocv.detail.SeamFinder_VORONOI_SEAM).find(None, corners, masks)
    got = cv.detail.SeamFinder_createDefault(cv.detail.SeamFinder_VORONOI_SEAM).find(None, corners, masks)
    assert all(np.array_equal(a, b) for a, b in zip(got, want))
    # UMats are cut in place and returned, like cv2
    ums = [cv.UMat(m) for m in masks]
    out = cv.detail.SeamFinder_createDefault(cv.detail.SeamFinder_VORONOI_SEAM).find(None, corners, ums)
    assert all(o is u for o, u in zip(out, ums)) and all(np.array_equal(u.get(), b) for u, b in zip(ums, want))
    assert cv.detail.SeamFinder_createDefault(cv.detail.SeamFinder_NO).find(None, corners, masks)[1] is masks[1]
    with pytest.raises(cv.error):
        cv.detail_DpSeamFinder("GRADIENT")


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5, 6, 7])
@pytest.mark.parametrize("cost", ["COLOR", "COLOR_GRAD"])
def test_dp_seam_finder_bit_exact(seed, cost):
    """cv.detail_DpSeamFinder (sde.py:243-249, :1618): device gradients / edge costs / dynamic programme + host component graph
    against the oracle's restatement, which shares no code with it."""
    from test_seam_dp import blob_case
    corners, images, masks = blob_case(seed, n=3 + seed % 4)
    fo, fg = ocv.detail_DpSeamFinder(cost), cv.detail_DpSeamFinder(cost)
    want = fo.find(images, corners, masks)
    got = fg.find(images, corners, masks)
    assert fg.pair_order == fo.pair_order
    assert all(np.array_equal(a, b) for a, b in zip(got, want))
    if seed == 0 and cost == "COLOR":   # 8-bit images give the same colour costs as their float32 copies; UMats are cut in place
        ums = [cv.UMat(m) for m in masks]
        out = cv.detail_DpSeamFinder(cost).find([cv.UMat(im.astype(np.uint8)) for im in images], corners, ums)
        want8 = ocv.detail_DpSeamFinder(cost).find([im.astype(np.uint8).astype(np.float32) for im in images], corners, masks)
        assert all(o is u for o, u in zip(out, ums)) and all(np.array_equal(u.get(), b) for u, b in zip(ums, want8))
    if seed == 0 and cost == "COLOR_GRAD":
        # cv2's cvtColor(BGR2GRAY) on 8-bit images is a fixed-point grey rounded to uint8 (not restated): float32 only, as the reference passes
        with pytest.raises(cv.error, match="CV_32FC3"):
            cv.detail_DpSeamFinder(cost).find([cv.UMat(im.astype(np.uint8)) for im in images], corners, [cv.UMat(m) for m in masks])


@pytest.mark.parametrize("seed", range(4))
def test_fuzz_dp_seams(seed):
    """more layouts (3-7 images of 32-128 px), both cost functions; tools/fuzz_sweep.py runs further seeds"""
    from test_seam_dp import blob_case
    corners, images, masks = blob_case(100 + seed, n=3 + seed % 5, size=64 + (seed * 37) % 65)
    for cost in ("COLOR", "COLOR_GRAD"):
        fo, fg = ocv.detail_DpSeamFinder(cost), cv.detail_DpSeamFinder(cost)
        want, got = fo.find(images, corners, masks), fg.find(images, corners, masks)
        assert fg.pair_order == fo.pair_order and all(np.array_equal(a, b) for a, b in zip(got, want)), (seed, cost)


@pytest.mark.parametrize("size,seed", [(300, 0), (420, 1), (520, 2), (700, 3), (1100, 4), (2300, 5)])
def test_dp_seams_on_large_overlaps(size, seed):
    """Overlap boxes on both sides of k_dp_seam_lds's limits (131072 cells, lines of 1024): blocks of 256 / 512 / 1024 lanes in the
    LDS form, and the global-memory sweep k_dp_seam beyond it (the only form before round 3) -- both bit for bit against the oracle."""
    from test_seam_dp import blob_case
    corners, images, masks = blob_case(500 + seed, n=3, size=size)
    for cost in ("COLOR", "COLOR_GRAD"):
        fo, fg = ocv.detail_DpSeamFinder(cost), cv.detail_DpSeamFinder(cost)
        want, got = fo.find(images, corners, masks), fg.find(images, corners, masks)
        assert fg.pair_order == fo.pair_order and all(np.array_equal(a, b) for a, b in zip(got, want)), (size, cost)


def test_dp_seams_read_nothing_outside_the_overlap_window():
    """The host part of DpSeamFinder keeps its label image and the masks' outlines only inside the two images' overlap rectangle grown by 2 (the rest
    of the union canvas is never read).  With SSP_SEAM_DP_POISON set everything outside that window holds values that would change the outcome:
    the fuzz layouts and the recorded 21-frame run must still equal the oracle's masks (own process: the switch is read once)."""
    import subprocess
    import sys
    code = (
        "import sys, os, numpy as np\n"
        f"sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r}); sys.path.insert(0, {os.path.dirname(os.path.abspath(__file__))!r})\n"
        "import opencv_starry_sky_panorama_stitcher_amd as cv, oracle_cv as ocv, test_seam_dp as t\n"
        "for seed in range(6):\n"
        "    corners, images, masks = t.blob_case(100 + seed, n=3 + seed % 5, size=64 + (seed * 37) % 65)\n"
        "    for cost in ('COLOR', 'COLOR_GRAD'):\n"
        "        want, got = ocv.detail_DpSeamFinder(cost).find(images, corners, masks), cv.detail_DpSeamFinder(cost).find(images, corners, masks)\n"
        "        assert all(np.array_equal(a, b) for a, b in zip(got, want)), (seed, cost)\n"
        "corners, images, masks = t.recorded_seam_inputs(ocv)\n"
        "imf = [im.astype(np.float32) for im in images]\n"
        "want, got = ocv.detail_DpSeamFinder('COLOR_GRAD').find(imf, corners, masks), cv.detail_DpSeamFinder('COLOR_GRAD').find(imf, corners, masks)\n"
        "assert all(np.array_equal(a, b) for a, b in zip(got, want))\n"
        "print('WINDOW OK')\n")
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SSP_SEAM_DP_POISON="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "WINDOW OK" in r.stdout, r.stderr[-2000:]


def test_dp_seam_across_more_than_4096_pixels():
    """Two 4400 x 4400 frames overlapping in 4250 x 4280 pixels: the seam's sweep lines are longer than the 4096 cells the LDS lines of k_dp_seam hold
    (full-size 8K frames would do that; seam-scale frames never), so the two cost / reach lines live in global memory (k_dp_seam<true>).  Until round 4
    this was SSP_ERR_ARG."""
    from PIL import Image
    rng = np.random.default_rng(77)
    W = H = 4400
    images = []
    for k in range(2):
        base = rng.uniform(0, 255, (H // 64 + 2, W // 64 + 2, 3)).astype(np.uint8)
        img = np.asarray(Image.fromarray(base).resize((W, H), Image.BICUBIC), np.int16) + rng.integers(-6, 7, (H, W, 3), dtype=np.int16)
        images.append(np.clip(img, 0, 255).astype(np.uint8))
    corners = [(0, 0), (150, 120)]
    masks = [np.full((H, W), 255, np.uint8) for _ in range(2)]
    want = ocv.detail_DpSeamFinder("COLOR").find([im.astype(np.float32) for im in images], corners, [m.copy() for m in masks])
    ums = [cv.UMat(m) for m in masks]
    cv.detail_DpSeamFinder("COLOR").find([cv.UMat(im) for im in images], corners, ums)
    got = [u.get() for u in ums]
    assert all(np.array_equal(a, b) for a, b in zip(got, want))
    assert 0 < int((got[0] == 0).sum()) and 0 < int((got[1] == 0).sum())       # both masks were cut


def test_dp_seam_finder_default_type_is_color():
    from test_seam_dp import blob_case
    corners, images, masks = blob_case(3, n=4)
    a = cv.detail.SeamFinder_createDefault(cv.detail.SeamFinder_DP_SEAM).find(images, corners, masks)
    b = cv.detail_DpSeamFinder("COLOR").find(images, corners, masks)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_dp_seams_on_the_recorded_run():
    """The reference's own run (21 autumn-forest frames, KAT 26 cameras, 'dp_colorgrad'): HIP == oracle bit for bit on the real
    seam-scale warps, and both agree with the recorded seamed masks as tests/test_seam_dp.py states."""
    import test_seam_dp as t
    corners, images, masks = t.recorded_seam_inputs(cv)
    corners_o, images_o, masks_o = t.recorded_seam_inputs(ocv)
    assert corners == corners_o and all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(images, images_o))
    fg, fo = cv.detail_DpSeamFinder("COLOR_GRAD"), ocv.detail_DpSeamFinder("COLOR_GRAD")
    got = fg.find([np.asarray(im).astype(np.float32) for im in images], corners, [np.asarray(m) for m in masks])
    want = fo.find([im.astype(np.float32) for im in images_o], corners_o, masks_o)
    assert fg.pair_order == fo.pair_order
    assert all(np.array_equal(a, b) for a, b in zip(got, want))
    t.check_against_recorded(ocv, got, masks_o)
    # the second recorded run (other cameras, another compose scale): all 21 recorded masks reproduced
    corners2, images2, masks2 = t.recorded_seam_inputs(cv, run=2)
    got2 = cv.detail_DpSeamFinder("COLOR_GRAD").find([np.asarray(im).astype(np.float32) for im in images2], corners2, [np.asarray(m) for m in masks2])
    frac2 = t.second_run_agreement(ocv, got2, [np.asarray(m) for m in masks2])
    assert all(v < 0.015 for v in frac2), frac2


def test_voronoi_seams_on_seam_scale_warps_of_a_rig():
    rig = starfield.make_rig(3, scale_div=8, n_override=4)
    corners, masks_g, masks_o = [], [], []
    for i in range(rig.n):
        K = rig.Ks[i].astype(np.float32); R = rig.Rs[i].astype(np.float32)
        ones = 255 * np.ones((rig.height, rig.width), np.uint8)
        c, mg = cv.PyRotationWarper(rig.warp, rig.focal).warp(ones, K, R, cv.INTER_NEAREST, cv.BORDER_CONSTANT)
        _, mo = ocv.PyRotationWarper(rig.warp, rig.focal).warp(ones, K, R, ocv.INTER_NEAREST, ocv.BORDER_CONSTANT)
        corners.append(c); masks_g.append(mg); masks_o.append(mo)
    got = cv.detail.SeamFinder_createDefault(1).find(None, corners, masks_g)
    want = ocv.detail.SeamFinder_createDefault(1).find(None, corners, masks_o)
    assert all(np.array_equal(a, b) for a, b in zip(got, want))
    assert sum(int((m != 0).sum()) for m in got) < sum(int((m != 0).sum()) for m in masks_g)   # overlaps were cut


@pytest.mark.parametrize("kind", [0, 1])
def test_timelapser_bit_exact(kind):
    corners, sizes = [(-5, 3), (20, -4), (8, 10)], [(40, 30), (35, 32), (50, 28)]
    tg = cv.detail.Timelapser_createDefault(kind)
    to = ocv.detail.Timelapser_createDefault(kind)
    tg.initialize(corners, sizes); to.initialize(corners, sizes)
    assert tg.getDstRoi() == tuple(to.roi)
    rng = np.random.default_rng(kind + 3)
    for (cx, cy), (sw, sh) in zip(corners, sizes):
        img = rng.integers(-300, 300, size=(sh, sw, 3)).astype(np.int16)
        mask = (rng.uniform(size=(sh, sw)) > 0.3).astype(np.uint8) * 255
        # sde.py:1841-1845: the frame is masked first, then pasted
        mg = cv.bitwise_and(img, img, mask=mask)
        mo = ocv.bitwise_and(img, img, mask=mask)
        assert np.array_equal(mg, mo)
        tg.process(mg, np.ones((sh, sw), np.uint8), (cx, cy)); to.process(mo, None, (cx, cy))
        assert np.array_equal(tg.getDst().get(), to.getDst())
    with pytest.raises(cv.error):
        tg.process(np.zeros((4, 4, 3), np.uint8), None, (0, 0))            # OpenCV asserts CV_16SC3
    with pytest.raises(cv.error):
        cv.detail.Timelapser_createDefault(7)


# ---- blenders -----------------------------------------------------------------------------------------------------------
def _three_images(seed=0, w=150, h=100, dtype=np.int16):
    rng = np.random.default_rng(seed)
    imgs, masks, tls = [], [], []
    for i, (tx, ty) in enumerate([(-40, 7), (55, -3), (140, 12)]):
        im = star_patch(w, h, seed=seed * 10 + i, n_stars=None if w * h < 200000 else 300).astype(dtype)      # (star_patch is O(stars w h))
        if dtype == np.int16:
            im = (im.astype(np.int32) + rng.integers(-20, 20, im.shape)).astype(np.int16)  # int16 values outside [0,255] too
        mk = np.zeros((h, w), np.uint8)
        mk[3 + i: h - 5, 6: w - 4 - 2 * i] = 255
        mk[rng.integers(0, h, 40), rng.integers(0, w, 40)] = rng.integers(0, 256, 40)  # grey values as after the G1 mask prep
        imgs.append(im)
        masks.append(mk)
        tls.append((tx, ty))
    return imgs, masks, tls


def _blend_both(make_g, make_o, imgs, masks, tls):
    sizes = [(m.shape[1], m.shape[0]) for m in masks]
    roi = ocv.detail.resultRoi(tls, sizes)
    assert tuple(cv.detail.resultRoi(corners=tls, sizes=sizes)) == tuple(roi)
    bg, bo = make_g(), make_o()
    bg.prepare(roi)
    bo.prepare(roi)
    for im, mk, tl in zip(imgs, masks, tls):
        bg.feed(im, mk, tl)
        bo.feed(im, mk, tl)
    return bg.blend(None, None), bo.blend(None, None)


@pytest.mark.parametrize("bands", [1, 2, 3, 5, 7])
def test_multiband_bit_exact(bands):
    """MultiBandBlender (int16 Laplacian pyramids, f32 weights, truncating casts): result and mask identical."""
    imgs, masks, tls = _three_images(seed=bands)
    def mg():
        b = cv.detail_MultiBandBlender(); b.setNumBands(bands); return b
    def mo():
        return ocv.detail_MultiBandBlender(num_bands=bands)
    (rg, kg), (ro, ko) = _blend_both(mg, mo, imgs, masks, tls)
    assert np.array_equal(kg, ko)
    assert rg.dtype == np.int16 and np.array_equal(rg, ro)


def test_multiband_single_pixel_weights_and_odd_sizes():
    imgs, masks, tls = _three_images(seed=42, w=97, h=61)
    (rg, kg), (ro, ko) = _blend_both(lambda: cv.detail_MultiBandBlender(num_bands=4), lambda: ocv.detail_MultiBandBlender(num_bands=4), imgs, masks, tls)
    assert np.array_equal(kg, ko) and np.array_equal(rg, ro)


def test_multiband_u8_feed_equals_int16_feed():
    """Feeding the warped uint8 frame directly (device fast path) equals feeding its int16 copy (sde.py:1755)."""
    imgs, masks, tls = _three_images(seed=9, dtype=np.uint8)
    def mg():
        b = cv.detail_MultiBandBlender(); b.setNumBands(3); return b
    (rg, kg), (ro, ko) = _blend_both(mg, lambda: ocv.detail_MultiBandBlender(num_bands=3), imgs, masks, tls)
    assert np.array_equal(rg, ro) and np.array_equal(kg, ko)


@pytest.mark.parametrize("w,h,bands,order", [(150, 100, 3, "u8,s16,u8"), (97, 61, 4, "s16,u8,u8"), (640, 420, 4, "u8,u8,s16"), (1100, 1300, 5, "s16,u8,s16")])
def test_multiband_mixed_u8_and_int16_feeds(w, h, bands, order):
    """One blender, frames of both depths (cv2 takes CV_8UC3 and CV_16SC3 in any order).  An 8-bit fed pyramid keeps 8-bit Gaussian levels, an
    int16 fed one int16 levels: the blend kernels take each image's levels as they are stored (per-image flag) -- small, 4x2 and, at the last
    size, strip-kernel levels."""
    i16, masks, tls = _three_images(seed=31 + bands, w=w, h=h)
    u8, _, _ = _three_images(seed=31 + bands, w=w, h=h, dtype=np.uint8)
    tls = [(tx * w // 150, ty) for tx, ty in tls]
    imgs = [u8[i] if d == "u8" else i16[i] for i, d in enumerate(order.split(","))]
    (rg, kg), (ro, ko) = _blend_both(lambda: cv.detail_MultiBandBlender(num_bands=bands), lambda: ocv.detail_MultiBandBlender(num_bands=bands), imgs, masks, tls)
    assert np.array_equal(kg, ko) and np.array_equal(rg, ro)


@pytest.mark.parametrize("w,h,bands,kind", [(640, 420, 4, "noise"), (640, 420, 4, "spikes"), (1100, 1300, 5, "spikes"), (150, 100, 3, "noise")])
def test_multiband_int16_beyond_the_packed_range(w, h, bands, kind):
    """int16 frames far outside [0, 255]: the collapsed parent's pyrUp of the 4x2 form runs packed only while a lane's samples lie in
    [-512, 511]; every other lane takes the 32-bit form.  'noise': all lanes outside; 'spikes': a few pixels, so waves mix both forms."""
    imgs, masks, tls = _three_images(seed=77 + bands, w=w, h=h)
    tls = [(tx * w // 150, ty) for tx, ty in tls]
    rng = np.random.default_rng(5)
    for im in imgs:
        if kind == "noise":
            im[...] = rng.integers(-12000, 12000, im.shape, dtype=np.int16)
        else:
            ys, xs = rng.integers(0, h, 60), rng.integers(0, w, 60)
            im[ys, xs] = rng.integers(-30000, 30000, (60, 3), dtype=np.int16)
    (rg, kg), (ro, ko) = _blend_both(lambda: cv.detail_MultiBandBlender(num_bands=bands), lambda: ocv.detail_MultiBandBlender(num_bands=bands), imgs, masks, tls)
    assert np.array_equal(kg, ko) and np.array_equal(rg, ro)
    assert int(np.abs(ro.astype(np.int32)).max()) > 2000      # the case is what it says


def test_feather_and_no_blender_bit_exact():
    imgs, masks, tls = _three_images(seed=4)
    (rg, kg), (ro, ko) = _blend_both(lambda: cv.detail_FeatherBlender(0.05), lambda: ocv.detail_FeatherBlender(0.05), imgs, masks, tls)
    assert np.array_equal(kg, ko) and np.array_equal(rg, ro)
    (rg, kg), (ro, ko) = _blend_both(lambda: cv.detail.Blender_createDefault(cv.detail.Blender_NO), lambda: ocv.detail.Blender_createDefault(0), imgs, masks, tls)
    assert np.array_equal(kg, ko) and np.array_equal(rg, ro)


def test_multiband_float_pyramids_close():
    """f32 pyramid variant (BASELINE config 5, no OpenCV counterpart): same operation order as the oracle, compared
    with a tolerance of 1e-3 grey levels (float sums over images are taken in feed order on both sides)."""
    imgs, masks, tls = _three_images(seed=6, dtype=np.float32)
    (rg, kg), (ro, ko) = _blend_both(lambda: cv.detail_MultiBandBlender(num_bands=4, float_pyramids=True),
                                     lambda: ocv.detail_MultiBandBlender(num_bands=4, float_pyramids=True), imgs, masks, tls)
    assert np.array_equal(kg, ko)
    assert rg.dtype == np.float32 and np.max(np.abs(rg - ro)) <= 1e-3


def test_blender_state_errors():
    b = cv.detail_MultiBandBlender()
    with pytest.raises(cv.error):
        b.feed(np.zeros((4, 4, 3), np.int16), np.zeros((4, 4), np.uint8), (0, 0))  # feed before prepare
    b.prepare((0, 0, 16, 16))
    b.feed(np.zeros((4, 4, 3), np.int16), 255 * np.ones((4, 4), np.uint8), (2, 2))
    b.blend(None, None)
    with pytest.raises(cv.error):
        b.blend(None, None)  # state is consumed by blend, as in OpenCV


# ---- exposure compensation ------------------------------------------------------------------------------------------------
def _comp_inputs(n=4, w=90, h=60, seed=0):
    rng = np.random.default_rng(seed)
    base = star_patch(w + 40 * n, h + 20, seed=seed + 100)
    corners, images, masks = [], [], []
    for i in range(n):
        x0, y0 = 35 * i, (5 * i) % 17
        g = rng.uniform(0.7, 1.3)
        im = np.clip(np.rint(base[y0:y0 + h, x0:x0 + w].astype(np.float32) * g), 0, 255).astype(np.uint8)
        mk = 255 * np.ones((h, w), np.uint8)
        mk[:, : 3 + i] = 0
        mk[rng.integers(0, h, 30), rng.integers(0, w, 30)] = 128
        corners.append((x0 - 7, y0 + 3))
        images.append(np.ascontiguousarray(im))
        masks.append(mk)
    return corners, images, masks


@pytest.mark.parametrize("ctype", [1, 2, 3, 4])
def test_compensator_gains_and_apply(ctype):
    """Gains: the overlap sums are double precision with a different summation order on the GPU, so gains are compared
    at 1e-9 relative; the applied 8-bit images must then be identical."""
    corners, images, masks = _comp_inputs()
    cg = cv.detail.ExposureCompensator_createDefault(ctype)
    co = ocv.detail.ExposureCompensator_createDefault(ctype)
    if ctype in (2, 4):
        cg.setBlockSize(16, 16)
        co = ocv._Comp(ctype, 16, 16, 1, 2)
    cg.feed(corners=corners, images=images, masks=masks)
    co.feed(corners, images, masks)
    if ctype in (1, 3):
        assert np.allclose(cg.gains(), co.gains(), rtol=1e-9, atol=0)
    else:
        for i in range(len(images)):
            assert np.allclose(cg.gainMap(i), co.gainMap(i), rtol=1e-6, atol=0)
    changed = 0
    for i in range(len(images)):
        big = star_patch(images[i].shape[1] * 3 + 5, images[i].shape[0] * 3 + 2, seed=50 + i)
        a, b = big.copy(), big.copy()
        cg.apply(i, corners[i], a, None)
        co.apply(i, corners[i], b, None)
        changed += int(not np.array_equal(a, big))
        assert np.array_equal(a, b)
    assert changed > 0  # in-place mutation happened (sde.py:1754 relies on it)


def test_compensator_nr_feeds_two():
    corners, images, masks = _comp_inputs(seed=3)
    keep = [im.copy() for im in images]
    cg = cv.detail_ChannelsCompensator(2)
    co = ocv.detail_ChannelsCompensator(2)
    cg.feed(corners=corners, images=images, masks=masks)
    co.feed(corners, images, masks)
    assert all(np.array_equal(a, b) for a, b in zip(images, keep))  # feed does not modify the caller's images
    assert np.allclose(cg.gains(), co.gains(), rtol=1e-9, atol=0)


# ---- whole compose loop ---------------------------------------------------------------------------------------------------
def _rig_small(config, div, n=None):
    rig = starfield.make_rig(config, scale_div=div, n_override=n)
    frames, seams = starfield.make_frames(rig, want_seam=True)
    return rig, frames, seams


@pytest.mark.parametrize("config,div,n", [(1, 8, None), (2, 8, 3), (3, 8, 4)])
def test_compose_loop_matches_oracle(config, div, n):
    """BASELINE configs 1-3 at 1/8 frame size through the full sde.py:1537-1944 call sequence, HIP vs oracle.
    Without exposure compensation the mosaic is bit-identical; with GAIN_BLOCKS (config 3) the gains carry a 1e-9
    relative difference (double summation order), which may move a rounded 8-bit value: north_star tolerance is
    +-1 LSB, asserted here, and we also assert it affects < 0.01 % of the samples."""
    rig, frames, seams = _rig_small(config, div, n)
    kw = dict(warp=rig.warp, warper_scale=rig.focal, blend=rig.blend, num_bands=rig.num_bands if rig.blend == "multiband" else None,
              blend_strength=rig.blend_strength if rig.blend == "feather" else None, expos_comp=rig.expos_comp, seam_frames=seams,
              seam_aspect=rig.seam_scale)
    g = cmp.compose_panorama(cv, frames, rig.Ks, rig.Rs, **kw)
    o = cmp.compose_panorama(ocv, frames, rig.Ks, rig.Rs, **kw)
    assert g.pano_roi == o.pano_roi and g.corners == o.corners and g.sizes == o.sizes
    assert np.array_equal(g.result_mask, o.result_mask)
    diff = np.abs(g.mosaic.astype(np.int16) - o.mosaic.astype(np.int16))
    if rig.expos_comp == 0:
        assert diff.max() == 0 and np.array_equal(g.result, o.result)
    else:
        assert diff.max() <= 1 and (diff > 0).mean() < 1e-4


def test_compose_loop_with_prologue_voronoi_seams_and_timelapse():
    """The widened loop: full-resolution frames through resize(INTER_AREA) + black/white point, Voronoi
    seams on the seam-scale masks, timelapser canvases -- every stage bit-exact, so the mosaic is too."""
    rig, frames, seams = _rig_small(2, 8, 3)
    scale = 0.5
    full = [np.repeat(np.repeat(f, 2, axis=0), 2, axis=1) for f in frames]      # frames at twice the compose resolution
    kw = dict(blend=rig.blend, num_bands=rig.num_bands, seam_frames=seams, seam_aspect=rig.seam_scale, mask_prep=True, seam="voronoi",
              timelapse_type=0, compose_scale=scale, black_and_white_point=(0, 150))
    g = cmp.compose_panorama(cv, full, rig.Ks, rig.Rs, rig.warp, rig.focal, **kw)
    o = cmp.compose_panorama(ocv, full, rig.Ks, rig.Rs, rig.warp, rig.focal, **kw)
    assert g.pano_roi == o.pano_roi and len(g.timelapse) == len(frames)
    assert all(np.array_equal(a, b) for a, b in zip(g.timelapse, o.timelapse))
    assert np.array_equal(g.result, o.result) and np.array_equal(g.result_mask, o.result_mask) and np.array_equal(g.mosaic, o.mosaic)


@pytest.mark.parametrize("config,div,n,prep", [(2, 8, 3, False), (2, 8, 4, True), (3, 8, 3, True)])
def test_composer_equals_object_api(config, div, n, prep):
    """The batched device-resident Composer produces exactly what the object-by-object API produces."""
    rig, frames, seams = _rig_small(config, div, n)
    comp = cv.detail.ExposureCompensator_createDefault(rig.expos_comp)
    kw = dict(warp=rig.warp, warper_scale=rig.focal, blend=rig.blend, num_bands=rig.num_bands, expos_comp=rig.expos_comp,
              seam_frames=seams if (prep or rig.expos_comp) else None, seam_aspect=rig.seam_scale, mask_prep=prep)
    ref = cmp.compose_panorama(cv, frames, rig.Ks, rig.Rs, **kw)
    c = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), blend=rig.blend, num_bands=rig.num_bands, mask_prep=prep,
                     seam_size=rig.seam_size, seam_aspect=rig.seam_scale, want_result_s16=True)
    if rig.expos_comp:
        # same seam-scale feed as compose_panorama performs
        ws = cv.PyRotationWarper(rig.warp, rig.focal * rig.seam_scale)
        cs, ims, mks = [], [], []
        for i in range(rig.n):
            K = rig.Ks[i].copy(); K[0, 0] *= rig.seam_scale; K[0, 2] *= rig.seam_scale; K[1, 1] *= rig.seam_scale; K[1, 2] *= rig.seam_scale
            cnr, im = ws.warp(seams[i], K, rig.Rs[i], cv.INTER_AREA, cv.BORDER_REFLECT)
            _, mk = ws.warp(255 * np.ones(seams[i].shape[:2], np.uint8), K, rig.Rs[i], cv.INTER_NEAREST, cv.BORDER_CONSTANT)
            cs.append(cnr); ims.append(im); mks.append(mk)
        comp.feed(corners=cs, images=ims, masks=mks)
        c.set_compensator(comp)
    assert c.pano_roi() == ref.pano_roi
    dev_frames = [cv.UMat(f) for f in frames]
    for _ in range(2):  # a second run reuses pooled buffers
        c.run(dev_frames)
        mo, mk, rs = c.result()
        assert np.array_equal(mk.get(), ref.result_mask)
        assert np.array_equal(rs.get(), ref.result)
        assert np.array_equal(mo.get(), ref.mosaic)


@pytest.mark.parametrize("ctype", [1, 2, 3, 4])
def test_composer_fused_gain_equals_separate_apply(ctype):
    """Exposure compensation inside the warp epilogue (Composer) against compensator.apply as its own pass (object API): scalar gain,
    gain blocks, channel gains, channel blocks -- identical images, hence identical panoramas."""
    rig, frames, seams = _rig_small(3, 8, 4)
    rig.expos_comp = ctype
    kw = dict(warp=rig.warp, warper_scale=rig.focal, blend=rig.blend, num_bands=rig.num_bands, expos_comp=ctype, seam_frames=seams,
              seam_aspect=rig.seam_scale, mask_prep=True)
    ref = cmp.compose_panorama(cv, frames, rig.Ks, rig.Rs, **kw)
    comp = cv.detail.ExposureCompensator_createDefault(ctype)
    ws = cv.PyRotationWarper(rig.warp, rig.focal * rig.seam_scale)
    cs, ims, mks = [], [], []
    for i in range(rig.n):
        K = rig.Ks[i].copy(); K[0, 0] *= rig.seam_scale; K[0, 2] *= rig.seam_scale; K[1, 1] *= rig.seam_scale; K[1, 2] *= rig.seam_scale
        cnr, im = ws.warp(seams[i], K, rig.Rs[i], cv.INTER_AREA, cv.BORDER_REFLECT)
        _, mk = ws.warp(255 * np.ones(seams[i].shape[:2], np.uint8), K, rig.Rs[i], cv.INTER_NEAREST, cv.BORDER_CONSTANT)
        cs.append(cnr); ims.append(im); mks.append(mk)
    comp.feed(corners=cs, images=ims, masks=mks)
    c = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), blend=rig.blend, num_bands=rig.num_bands, mask_prep=True,
                     seam_size=rig.seam_size, seam_aspect=rig.seam_scale, want_result_s16=True)
    c.set_compensator(comp)
    c.run([cv.UMat(f) for f in frames])
    mo, mk, rs = c.result()
    assert np.array_equal(mk.get(), ref.result_mask) and np.array_equal(rs.get(), ref.result) and np.array_equal(mo.get(), ref.mosaic)
    # the gains did something
    plain = cmp.compose_panorama(cv, frames, rig.Ks, rig.Rs, **dict(kw, expos_comp=0))
    assert not np.array_equal(plain.mosaic, ref.mosaic)


def test_composer_float_frames_config5():
    """BASELINE config 5 at 1/32 size: float32 frames, 7-band float pyramids.  The Composer (frames warped straight into the planes,
    batched pyramids) equals the object-by-object API bit for bit, and the oracle within 1e-3 grey levels (float sums in feed order
    on both sides; the warp itself is bit-exact)."""
    rig = starfield.make_rig(5, scale_div=32, n_override=3)
    rig.yaws_deg, rig.pitches_deg, rig.Ks, rig.Rs = rig.yaws_deg[:3], rig.pitches_deg[:3], rig.Ks[:3], rig.Rs[:3]
    frames, seams = starfield.make_frames(rig, want_seam=True)
    assert frames[0].dtype == np.float32
    kw = dict(warp=rig.warp, warper_scale=rig.focal, blend="multiband", num_bands=rig.num_bands, seam_frames=seams, seam_aspect=rig.seam_scale,
              mask_prep=True, float_pyramids=True)
    g = cmp.compose_panorama(cv, frames, rig.Ks, rig.Rs, **kw)
    o = cmp.compose_panorama(ocv, frames, rig.Ks, rig.Rs, **kw)
    assert g.pano_roi == o.pano_roi and np.array_equal(g.result_mask, o.result_mask)
    assert g.result.dtype == np.float32 and np.max(np.abs(g.result - o.result)) <= 1e-3
    c = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), blend="multiband", num_bands=rig.num_bands, float_frames=True,
                     mask_prep=True, seam_size=rig.seam_size, seam_aspect=rig.seam_scale, want_result_s16=True)
    c.run([cv.UMat(f) for f in frames])
    mo, mk, rs = [u.get() for u in c.result()]
    assert np.array_equal(mk, g.result_mask) and np.array_equal(rs, g.result) and np.array_equal(mo, g.mosaic)


def test_composer_float_frames_config5_full_size():
    """BASELINE config 5 at its own frame size: two 7680x4320 float32 frames, 7-band float pyramids, mask preparation -- the Composer against
    the oracle's call sequence: identical roi and mask, result within 1e-3 grey levels (the float kernels and tolerance of the 1/32-size test,
    at the sizes where the strip / LDS paths and 32-bit offsets are exercised: a 400 MB plane per frame)."""
    rig = starfield.make_rig(5, scale_div=1, n_override=2)
    rig.yaws_deg, rig.pitches_deg, rig.Ks, rig.Rs = rig.yaws_deg[:2], rig.pitches_deg[:2], rig.Ks[:2], rig.Rs[:2]
    frames, seams = starfield.make_frames(rig, want_seam=True)
    assert frames[0].dtype == np.float32 and frames[0].shape[:2] == (4320, 7680)
    o = cmp.compose_panorama(ocv, frames, rig.Ks, rig.Rs, warp=rig.warp, warper_scale=rig.focal, blend="multiband", num_bands=rig.num_bands, seam_frames=seams,
                             seam_aspect=rig.seam_scale, mask_prep=True, float_pyramids=True)
    c = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), blend="multiband", num_bands=rig.num_bands, float_frames=True,
                     mask_prep=True, seam_size=rig.seam_size, seam_aspect=rig.seam_scale, want_result_s16=True)
    c.run([cv.UMat(f) for f in frames])
    mo, mk, rs = [u.get() for u in c.result()]
    assert c.pano_roi() == o.pano_roi and np.array_equal(mk, o.result_mask)
    assert rs.dtype == np.float32 and np.max(np.abs(rs - o.result)) <= 1e-3
    d = np.abs(mo.astype(np.int16) - o.mosaic.astype(np.int16))
    assert d.max() <= 1 and (d > 0).mean() < 1e-4          # the 8-bit mosaic rounds the float result: a 1e-3 difference can move a value at x.5


def test_partial_export_import_roundtrip():
    """Multi-GPU hooks: exporting the raw level sums of a blender holding images {0,1} and importing them into a blender
    holding image {2} gives the single-blender result (integer sums exact; weight sums differ in association only, and
    the masks/values here make them exact)."""
    imgs, masks, tls = _three_images(seed=12, dtype=np.uint8)
    for m in masks:
        m[(m != 0) & (m != 255)] = 255
    sizes = [(m.shape[1], m.shape[0]) for m in masks]
    roi = cv.detail.resultRoi(corners=tls, sizes=sizes)
    full = cv.detail_MultiBandBlender(num_bands=3); full.prepare(roi)
    a = cv.detail_MultiBandBlender(num_bands=3); a.prepare(roi)
    b = cv.detail_MultiBandBlender(num_bands=3); b.prepare(roi)
    for i in range(3):
        full.feed(imgs[i], masks[i], tls[i])
        (a if i < 2 else b).feed(imgs[i], masks[i], tls[i])
    import ctypes as C
    from opencv_starry_sky_panorama_stitcher_amd import _lib
    L = _lib.lib()
    W, H = C.c_int(), C.c_int()
    _lib.check(L.ssp_blender_level_info(a._h, 0, C.byref(W), C.byref(H)))  # padded pano size
    for lvl in range(4):
        w, h = W.value >> lvl, H.value >> lvl
        # export buffers are tightly packed: allocate them as single-row images (no row pitch padding)
        lap1 = cv.UMat.empty(w * h * 3, 1, 1, np.int16)
        wgt1 = cv.UMat.empty(w * h, 1, 1, np.float32)
        _lib.check(L.ssp_blender_export_partial(a._h, lvl, 0, 0, W.value, H.value, C.c_void_p(lap1.info()[5]), C.c_void_p(wgt1.info()[5])))
        _lib.check(L.ssp_blender_import_partial(b._h, lvl, 0, 0, W.value, H.value, C.c_void_p(lap1.info()[5]), C.c_void_p(wgt1.info()[5])))
    rf, kf = full.blend(None, None)
    rb, kb = b.blend(None, None)
    assert np.array_equal(kf, kb)
    assert np.max(np.abs(rf.astype(np.int32) - rb.astype(np.int32))) <= 1


def test_two_rank_emulation_on_one_gpu():
    """The multi-GPU step without RCCL: two Composers (one per emulated rank) feed their frames into blenders prepared with
    the global pano roi, exchange the partial sums of bbox[s] & bbox[d] through export/import_partial, and each finishes
    its own bbox.  Every pixel covered by a rank's frames must equal the single-composer panorama (integer sums exact;
    f32 weight sums differ in association only: tolerance 1, almost always 0)."""
    import ctypes as C
    from opencv_starry_sky_panorama_stitcher_amd import _lib, parallel
    L = _lib.lib()
    rig, frames, _ = _rig_small(3, 8, 5)
    nb = 4
    owner = [0, 0, 0, 1, 1]
    w = cv.PyRotationWarper(rig.warp, rig.focal)
    rois = [w.warpRoi((rig.width, rig.height), rig.Ks[i], rig.Rs[i]) for i in range(rig.n)]
    plan = parallel.plan_exchange([r[:2] for r in rois], [r[2:] for r in rois], owner, 2, nb)
    dev = [cv.UMat(f) for f in frames]
    full = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), num_bands=nb, want_result_s16=True)
    assert full.pano_roi() == plan.pano_roi
    full.run(dev)
    _, ref_mask, ref_res = [u.get() for u in full.result()]
    comps = []
    for r in range(2):
        idx = [i for i in range(rig.n) if owner[i] == r]
        c = cmp.Composer(rig.warp, rig.focal, [rig.Ks[i] for i in idx], [rig.Rs[i] for i in idx], (rig.width, rig.height), num_bands=nb, want_result_s16=True)
        c.set_pano_roi(plan.pano_roi)
        c.feed([dev[i] for i in idx])
        comps.append(c)
    for s, d, rect in plan.pairs:
        for lvl in range(plan.nb + 1):
            n = (rect[2] >> lvl) * (rect[3] >> lvl)
            lap = cv.UMat.empty(n * 3, 1, 1, np.int16)
            wgt = cv.UMat.empty(n, 1, 1, np.float32)
            _lib.check(L.ssp_blender_export_partial(comps[s].blender_handle(), lvl, *rect, C.c_void_p(lap.info()[5]), C.c_void_p(wgt.info()[5])))
            _lib.check(L.ssp_blender_import_partial(comps[d].blender_handle(), lvl, *rect, C.c_void_p(lap.info()[5]), C.c_void_p(wgt.info()[5])))
    own = parallel.owner_map(plan)
    covered = 0
    for r in range(2):
        comps[r].finish_region(plan.bbox[r])
        _, mk, rs = [u.get() for u in comps[r].result()]
        x0, y0 = plan.bbox[r][0], plan.bbox[r][1]
        hh, ww = mk.shape
        sel = own[y0:y0 + hh, x0:x0 + ww] == r
        covered += int(sel.sum())
        assert np.array_equal(mk[sel], ref_mask[y0:y0 + hh, x0:x0 + ww][sel])
        d = np.abs(rs.astype(np.int32) - ref_res[y0:y0 + hh, x0:x0 + ww].astype(np.int32))[sel]
        assert d.max() <= 1 and (d > 0).mean() < 1e-3
    assert covered == int((own >= 0).sum())


@pytest.mark.parametrize("levels", [True, False])
@pytest.mark.parametrize("world,owner,nb", [(2, [0, 0, 0, 1, 1], 4), (3, [0, 0, 1, 1, 2, 2], 3), (2, [0, 1, 0, 1], 3)])
def test_strip_exchange_emulation_is_bit_exact(world, owner, nb, levels):
    """The strip protocol (parallel.plan_strips) with every rank emulated on one GPU: each rank feeds its frames, receives the
    strips of foreign frames it needs (every pyramid level of them, or -- levels=False -- level 0 only, whose pyramids it rebuilds),
    orders all fed images globally and collapses its region.  Every pixel a rank owns must equal the single-composer panorama bit
    for bit -- result, mask AND weights' effect (same summation order)."""
    from opencv_starry_sky_panorama_stitcher_amd import parallel
    rig, frames, _ = _rig_small(3, 8, len(owner))
    w = cv.PyRotationWarper(rig.warp, rig.focal)
    rois = [w.warpRoi((rig.width, rig.height), rig.Ks[i], rig.Rs[i]) for i in range(rig.n)]
    plan = parallel.plan_strips([r[:2] for r in rois], [r[2:] for r in rois], owner, world, nb, levels=levels)
    dev = [cv.UMat(f) for f in frames]
    full = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), num_bands=nb, want_result_s16=True)
    assert full.pano_roi() == plan.pano_roi
    full.run(dev)
    ref_mos, ref_mask, ref_res = [u.get() for u in full.result()]
    exs, per_rank = [], []
    for r in range(world):
        idx = [i for i in range(rig.n) if owner[i] == r]
        c = cmp.Composer(rig.warp, rig.focal, [rig.Ks[i] for i in idx], [rig.Rs[i] for i in idx], (rig.width, rig.height), num_bands=nb, want_result_s16=True)
        exs.append(parallel.StripExchangeBase(c, plan, r, parallel._umat_alloc))
        per_rank.append([dev[i] for i in idx])
    parallel.emulate_strip_exchange(exs, per_rank)
    own = parallel.strip_owner_map(plan)
    assert np.all((own >= 0) | (ref_mask == 0))            # every blended pixel has an owner
    covered = 0
    for r in range(world):
        mos, mk, rs = [u.get() for u in exs[r].c.result()]
        x0, y0 = plan.region[r][0], plan.region[r][1]
        hh, ww = mk.shape
        sel = own[y0:y0 + hh, x0:x0 + ww] == r
        assert int(sel.sum()) == int((own == r).sum())     # the region holds everything the rank owns
        covered += int(sel.sum())
        assert np.array_equal(mk[sel], ref_mask[y0:y0 + hh, x0:x0 + ww][sel])
        assert np.array_equal(rs[sel], ref_res[y0:y0 + hh, x0:x0 + ww][sel])
        assert np.array_equal(mos[sel], ref_mos[y0:y0 + hh, x0:x0 + ww][sel])
    assert covered == int((own >= 0).sum())
    assert sum(plan.bytes_sent(r) for r in range(world)) > 0


def test_gpu_reproduces_committed_golden_vectors():
    """HIP path vs tests/golden/pixels.npz (no oracle involved at run time): warps of six projections, the three
    blenders, the mask helpers and the four compensators' applied images."""
    import os
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pixels.npz"))
    img, K, R, f = G["img"], G["cam_K"], G["cam_R"], float(G["cam_f"])
    for warp in ("spherical", "cylindrical", "fisheye", "plane", "paniniA2B1", "transverseMercator"):
        c, d, m = cv.PyRotationWarper(warp, f).warpWithMask(img, K, R, cv.BORDER_REFLECT)
        assert tuple(c) == tuple(G[f"warp_{warp}_corner"])
        assert np.array_equal(d, G[f"warp_{warp}_img"]) and np.array_equal(m, G[f"warp_{warp}_mask"])
    roi = tuple(int(v) for v in G["blend_roi"])
    for name, make in (("mb3", lambda: cv.detail_MultiBandBlender(num_bands=3)), ("feather", lambda: cv.detail_FeatherBlender(0.08)),
                       ("no", lambda: cv.detail.Blender_createDefault(0))):
        b = make()
        b.prepare(roi)
        for im, mk, tl in zip(G["blend_imgs"], G["blend_masks"], G["blend_tls"]):
            b.feed(im, mk, tuple(int(v) for v in tl))
        r, k = b.blend(None, None)
        assert np.array_equal(r, G[f"blend_{name}_result"]) and np.array_equal(k, G[f"blend_{name}_mask"])
    assert np.array_equal(cv.dilate(G["mask_small"], None), G["mask_dilated"])
    assert np.array_equal(cv.resize(G["mask_small"], (77, 52), 0, 0, cv.INTER_LINEAR_EXACT), G["mask_resized_77x52"])
    corners = [tuple(int(v) for v in c) for c in G["comp_corners"]]
    cimgs = [np.ascontiguousarray(i) for i in G["comp_imgs"]]
    cmasks = [255 * np.ones(cimgs[0].shape[:2], np.uint8) for _ in cimgs]
    for t in (1, 2, 3, 4):
        c = cv.detail.ExposureCompensator_createDefault(t)
        if t in (2, 4):
            c.setBlockSize(16, 16)
        c.feed(corners=corners, images=cimgs, masks=cmasks)
        big = star_patch(100, 72, seed=60)
        c.apply(1, corners[1], big, None)
        assert np.array_equal(big, G[f"comp_{t}_applied"])


def test_full_size_4k_compose_matches_oracle():
    """BASELINE size: two 3840x2160 frames of the bench rig, spherical warp + mask prep + 5-band multiband, through the
    batched Composer (the bench path: fused warp kernel, bordered planes, 2x2 pyrDown, quad blend) against the CPU oracle
    running the reference's call sequence.  Bit-identical mosaic, mask and int16 result."""
    import bench
    rig, _ = bench.block_rig(starfield, 1, 0, 1)
    idx = [0, 1]
    frames = starfield.make_frames(rig, indices=idx)
    cat = starfield.StarCatalogue(rig.config_id, density_per_sr=2500.0 / starfield._frame_solid_angle(rig))
    sw, sh = rig.seam_size
    seams = [np.rint(starfield.render_frame(rig, cat, i, res_scale=rig.seam_scale)[:sh, :sw]).astype(np.uint8) for i in idx]
    Ks, Rs = [rig.Ks[i] for i in idx], [rig.Rs[i] for i in idx]
    c = cmp.Composer(rig.warp, rig.focal, Ks, Rs, (rig.width, rig.height), blend="multiband", num_bands=5, mask_prep=True, seam_size=rig.seam_size,
                     seam_aspect=rig.seam_scale, want_result_s16=True)
    c.run([cv.UMat(f) for f in frames])
    mo, mk, rs = [u.get() for u in c.result()]
    ref = cmp.compose_panorama(ocv, frames, Ks, Rs, warp=rig.warp, warper_scale=rig.focal, blend="multiband", num_bands=5, seam_frames=seams,
                               seam_aspect=rig.seam_scale)
    assert c.pano_roi() == ref.pano_roi
    assert np.array_equal(mk, ref.result_mask)
    assert np.array_equal(rs, ref.result)
    assert np.array_equal(mo, ref.mosaic)
    assert mo.shape[1] > 5000 and (mk > 0).mean() > 0.8


# ---- size-independent properties at BASELINE frame size (no oracle in the loop) --------------------------------------------------
def test_full_size_properties_constant_frames():
    """3840x2160, the whole bench pipeline: (1) warping a constant frame gives that constant everywhere (taps of equal value with
    weights summing to 2^15; BORDER_REFLECT outside) and a binary mask; (2) blending constant frames of value c gives exactly
    c - 1 deep inside the union: every Laplacian level is 0 and the top level is (short)(n c / (n + 1e-5f)) = c - 1 for
    n = 1, 2, 3 overlapping frames -- the truncating cast of MultiBandBlender::blend; (3) the batched Composer and the object API
    agree bit for bit."""
    import bench
    rig, _ = bench.block_rig(starfield, 1, 0, 1)
    c0 = 137
    K, R = rig.Ks[0], rig.Rs[0]
    frame = np.full((rig.height, rig.width, 3), c0, np.uint8)
    w = cv.PyRotationWarper(rig.warp, rig.focal)
    corner, img, mask = w.warpWithMask(cv.UMat(frame), K, R, cv.BORDER_REFLECT)
    img, mask = img.get(), mask.get()
    assert np.all(img == c0) and set(np.unique(mask)) <= {0, 255} and 0.5 < (mask == 255).mean() < 1.0
    idx = [0, 1, 3, 4]
    Ks, Rs = [rig.Ks[i] for i in idx], [rig.Rs[i] for i in idx]
    comp = cmp.Composer(rig.warp, rig.focal, Ks, Rs, (rig.width, rig.height), blend="multiband", num_bands=5, want_result_s16=True)
    dev = [cv.UMat(frame)] * len(idx)
    comp.run(dev)
    mo, mk, rs = [u.get() for u in comp.result()]
    assert (mk > 0).mean() > 0.7
    # away from the outline of the union (where the blurred weights of the coarse levels are fractional and (short)(c w) loses more)
    from scipy.ndimage import distance_transform_cdt
    deep = distance_transform_cdt(mk > 0, metric="chessboard") > 6 * 32
    assert deep.mean() > 0.5
    # (where a frame ends inside another one the sum of two truncated products can lose one more: isolated c - 2 samples)
    assert rs[deep].max() == c0 - 1 and rs[deep].min() >= c0 - 2 and (rs[deep] != c0 - 1).mean() < 1e-4     # (4.4e-5 on the 30 degree block rig)
    assert np.array_equal(mo[deep], rs[deep].astype(np.uint8))
    assert np.all(rs[mk == 0] == 0) and rs.max() <= c0 and rs[mk > 0].min() > 0
    # object API on the same inputs
    blender = cv.detail_MultiBandBlender()
    blender.setNumBands(5)
    rois = [w.warpRoi((rig.width, rig.height), k, r) for k, r in zip(Ks, Rs)]
    blender.prepare(cv.detail.resultRoi([r[:2] for r in rois], [r[2:] for r in rois]))
    for k, r in zip(Ks, Rs):
        cnr, wi, wm = w.warpWithMask(dev[0], k, r, cv.BORDER_REFLECT)
        blender.feed(wi, wm, cnr)
    res, rmask = blender.blend(None, None)
    assert np.array_equal(res, rs) and np.array_equal(rmask, mk)


def test_pool_is_steady_and_released():
    """No allocation growth across composer steps (every temporary returns to the pool), and destroying the objects returns the
    bytes in use to the starting level."""
    import ctypes as C
    import gc
    from opencv_starry_sky_panorama_stitcher_amd import _lib
    L = _lib.lib()

    def in_use():
        a, b = C.c_size_t(), C.c_size_t()
        _lib.check(L.ssp_pool_stats(C.byref(a), C.byref(b)))
        return a.value, b.value

    gc.collect()
    base = in_use()[0]
    rig, frames, seams = _rig_small(2, 8, 3)
    dev = [cv.UMat(f) for f in frames]
    c = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), num_bands=4, mask_prep=True, seam_size=rig.seam_size,
                     seam_aspect=rig.seam_scale)
    for _ in range(8):        # the descriptor ring allocates its slots on first use
        c.run(dev)
    _lib.check(L.ssp_sync())
    first = in_use()
    for _ in range(8):
        c.run(dev)
    _lib.check(L.ssp_sync())
    later = in_use()
    assert later[0] == first[0] and later[1] == first[1]          # in use and cached bytes both steady
    del c, dev
    gc.collect()
    assert in_use()[0] == base


# ---- the multi-GPU step as real processes: two ranks share this box's one GPU, gloo carries the device tensors -----------------------
def _strip_rank_process(rank, world, port, owner, nb, outdir, pipelined, levels=True):
    import os
    import sys
    import torch                       # torch first: see INTEGRATION.md (its HIP runtime must be the one that initialises)
    import torch.distributed as dist
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ctypes as C
        import numpy as np
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, root)
        import opencv_starry_sky_panorama_stitcher_amd as cv
        from opencv_starry_sky_panorama_stitcher_amd import compose as cmp, parallel, starfield
        L = cv._lib.lib()
        cv._lib.check(L.ssp_init(0))
        cv._lib.check(L.ssp_set_stream(C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        rig = starfield.make_rig(3, scale_div=8, n_override=len(owner))
        frames = starfield.make_frames(rig)
        w = cv.PyRotationWarper(rig.warp, rig.focal)
        rois = [w.warpRoi((rig.width, rig.height), rig.Ks[i], rig.Rs[i]) for i in range(rig.n)]
        corners, sizes = [r[:2] for r in rois], [r[2:] for r in rois]
        idx = [i for i in range(rig.n) if owner[i] == rank]
        comp = cmp.Composer(rig.warp, rig.focal, [rig.Ks[i] for i in idx], [rig.Rs[i] for i in idx], (rig.width, rig.height), num_bands=nb, want_result_s16=True)
        mine = [cv.UMat(frames[i]) for i in idx]
        if pipelined:
            # double buffered: three steps, the composer that finished last holds a complete panorama, drain completes the other
            def make():
                return cmp.Composer(rig.warp, rig.focal, [rig.Ks[i] for i in idx], [rig.Rs[i] for i in idx], (rig.width, rig.height), num_bands=nb, want_result_s16=True)
            pipe = parallel.HipStripPipeline(make, dist, torch, corners, sizes, owner, nb, levels=levels)
            done = None
            for _ in range(3):
                done = pipe.step(mine)
            ex = done
            first = [u.get() for u in done.c.result()]
            pipe.drain()
            other = pipe.ex[0] if done is pipe.ex[1] else pipe.ex[1]
            second = [u.get() for u in other.c.result()]
            assert all(np.array_equal(a, b) for a, b in zip(first, second))
            mo, mk, rs = first
        else:
            ex = parallel.HipStripExchange(comp, dist, torch, corners, sizes, owner, nb, levels=levels)
            for _ in range(2):              # a second step reuses every buffer
                ex.run(mine)
            mo, mk, rs = [u.get() for u in comp.result()]
        # single-GPU reference of the whole panorama, computed by this rank too (the rig is small)
        full = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), num_bands=nb, want_result_s16=True)
        full.run([cv.UMat(f) for f in frames])
        ref_mo, ref_mk, ref_rs = [u.get() for u in full.result()]
        plan = ex.plan
        own = parallel.strip_owner_map(plan)
        x0, y0 = plan.region[rank][0], plan.region[rank][1]
        hh, ww = mk.shape
        sel = own[y0:y0 + hh, x0:x0 + ww] == rank
        ok = (int(sel.sum()) == int((own == rank).sum()) and np.array_equal(mk[sel], ref_mk[y0:y0 + hh, x0:x0 + ww][sel])
              and np.array_equal(rs[sel], ref_rs[y0:y0 + hh, x0:x0 + ww][sel]) and np.array_equal(mo[sel], ref_mo[y0:y0 + hh, x0:x0 + ww][sel]))
        np.save(os.path.join(outdir, f"rank_{rank}.npy"), np.array([int(ok), int(sel.sum()), plan.bytes_sent(rank)]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("pipelined,levels", [(False, True), (True, True), (True, False)])
@pytest.mark.parametrize("owner,nb", [([0, 0, 0, 1, 1], 4), ([0, 1, 0, 1, 0, 1], 3)])
def test_strip_exchange_two_processes_one_gpu(tmp_path, owner, nb, pipelined, levels):
    """parallel.HipStripExchange end to end: one process per rank (both on this GPU), torch.distributed point-to-point messages of
    strips (gloo here, RCCL on a multi-GPU node), two steps; and parallel.HipStripPipeline, the double-buffered flavour bench.py runs
    (three steps + drain, both buffer sets checked; all-level strips and the level-0 protocol it replaced).  Every owned pixel equals the
    single-process panorama bit for bit."""
    import torch.multiprocessing as mp
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_strip_rank_process, args=(2, port, owner, nb, str(tmp_path), pipelined, levels), nprocs=2, join=True)
    stats = [np.load(tmp_path / f"rank_{r}.npy") for r in range(2)]
    assert all(int(s[0]) == 1 for s in stats), stats
    assert all(int(s[1]) > 0 and int(s[2]) > 0 for s in stats)


# ---- randomised sweeps: sizes, positions and cameras that hit the kernels' special paths ---------------------------------------------
@pytest.mark.parametrize("seed", list(range(64)))
def test_fuzz_multiband_layouts(seed):
    """Random image counts, sizes (including widths/heights that are not multiples of anything), positions, masks (binary with holes,
    grey ramps) and band counts: bordered planes with every alignment shift, strip / 2x2 pyrDown with in-kernel aprons, 4x2 / 2x2 /
    per-pixel blend kernels, packed and general accumulation, the three normalisation paths -- against the oracle, bit for bit."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(1, 6))
    bands = int(rng.integers(1, 7))
    big = seed % 4 == 0
    imgs, masks, tls = [], [], []
    for i in range(n):
        w = int(rng.integers(5, 700 if big else 160))
        h = int(rng.integers(5, 500 if big else 120))
        dt = np.uint8 if seed % 3 else np.int16
        if dt == np.uint8:
            img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        else:
            img = rng.integers(-400, 700, size=(h, w, 3)).astype(np.int16)
        kind = int(rng.integers(0, 3))
        m = np.full((h, w), 255, np.uint8)
        if kind == 1:
            m[rng.integers(0, h, 8), rng.integers(0, w, 8)] = 0
            m[:, :int(rng.integers(0, max(1, w // 3)))] = 0
        elif kind == 2:
            m = np.clip(np.add.outer(np.arange(h), np.arange(w)) * (255.0 / max(1, h + w - 2)), 0, 255).astype(np.uint8)
        imgs.append(img); masks.append(m)
        tls.append((int(rng.integers(-300, 300)), int(rng.integers(-200, 200))))
    if seed % 3 == 0:   # the blender takes one depth per panorama
        imgs = [im.astype(np.int16) for im in imgs]
    (rg, kg), (ro, ko) = _blend_both(lambda: cv.detail_MultiBandBlender(num_bands=bands), lambda: ocv.detail_MultiBandBlender(num_bands=bands), imgs, masks, tls)
    assert np.array_equal(kg, ko) and np.array_equal(rg, ro), (seed, n, bands)


@pytest.mark.parametrize("seed", list(range(48)))
def test_fuzz_warp_cameras(seed):
    """Random frame sizes and cameras (yaw, pitch, roll, field of view) through the fused separable kernel (interior, mirrored-outline
    and per-tap paths, every column-group shift) and the generic kernel, image and mask, against the oracle."""
    rng = np.random.default_rng(2000 + seed)
    w, h = int(rng.integers(3, 420)), int(rng.integers(3, 300))
    img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    K, R, f = camera(w, h, float(rng.uniform(25, 100)), yaw=float(rng.uniform(-170, 170)), pitch=float(rng.uniform(-60, 60)), roll=float(rng.uniform(-40, 40)))
    warp = ["spherical", "cylindrical", "mercator", "fisheye", "plane", "stereographic"][seed % 6]
    if warp in ("plane", "stereographic", "fisheye"):
        K, R, f = camera(w, h, float(rng.uniform(25, 80)), yaw=float(rng.uniform(-25, 25)), pitch=float(rng.uniform(-20, 20)), roll=float(rng.uniform(-40, 40)))
    border = [cv.BORDER_REFLECT, cv.BORDER_REFLECT_101, cv.BORDER_REPLICATE, cv.BORDER_CONSTANT][int(rng.integers(0, 4))]
    g, o = cv.PyRotationWarper(warp, f), ocv.PyRotationWarper(warp, f)
    roi = o.warpRoi((w, h), K, R)
    if roi[2] <= 0 or roi[3] <= 0:
        # a projection that runs off to infinity inside the frame (mercator at a pole: tools/fuzz_sweep.py seed 20168): OpenCV's int(inf) roi has a
        # negative size and cv2 fails allocating the maps (sde.py:1576-1586 catches it); the library refuses the roi with its own cv.error
        with pytest.raises(cv.error, match="degenerate or absurd roi"):
            g.warpWithMask(img, K, R, border)
        return
    if roi[2] * roi[3] > 4_000_000:
        pytest.skip("roi too large for a quick oracle run")
    cg, dg, mg = g.warpWithMask(img, K, R, border)
    co, do = o.warp(img, K, R, ocv.INTER_LINEAR, border)
    _, mo = o.warp(255 * np.ones((h, w), np.uint8), K, R, ocv.INTER_NEAREST, ocv.BORDER_CONSTANT)
    assert cg == co and np.array_equal(dg, do) and np.array_equal(mg, mo), (seed, warp, w, h, border)


@pytest.mark.parametrize("seed", list(range(12)))
def test_fuzz_composer_rigs(seed):
    """Random small rigs (frame size, count, yaw / pitch / roll jitter, projection, bands, mask preparation) through the batched
    Composer against the oracle running the reference's call sequence."""
    from opencv_starry_sky_panorama_stitcher_amd.starfield import Rig, _finish
    rng = np.random.default_rng(3000 + seed)
    w, h = int(rng.integers(60, 330)), int(rng.integers(40, 220))
    n = int(rng.integers(2, 6))
    step = float(rng.uniform(12, 35))
    yaws = [float((i - (n - 1) / 2) * step + rng.uniform(-3, 3)) for i in range(n)]
    pitches = [float(rng.uniform(-5, 5)) for _ in range(n)]
    warp = ["spherical", "cylindrical", "mercator"][seed % 3]
    bands = int(rng.integers(2, 6))
    rig = _finish(Rig(f"fuzz {seed}", 9, w, h, 60.0, yaws, pitches, warp, "multiband", bands))
    frames, seams = starfield.make_frames(rig, want_seam=True)
    prep = bool(seed % 2)
    c = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), blend="multiband", num_bands=bands, mask_prep=prep, seam_size=rig.seam_size,
                     seam_aspect=rig.seam_scale, want_result_s16=True)
    c.run([cv.UMat(f) for f in frames])
    mo, mk, rs = [u.get() for u in c.result()]
    ref = cmp.compose_panorama(ocv, frames, rig.Ks, rig.Rs, warp=rig.warp, warper_scale=rig.focal, blend="multiband", num_bands=bands,
                               seam_frames=seams if prep else None, seam_aspect=rig.seam_scale, mask_prep=prep)
    assert c.pano_roi() == ref.pano_roi
    assert np.array_equal(mk, ref.result_mask) and np.array_equal(rs, ref.result) and np.array_equal(mo, ref.mosaic), (seed, w, h, n, warp, bands, prep)


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_composer_with_gains(seed):
    """Random small rigs with every compensator type: the Composer applies the gains in the warp epilogue (scalar gains, 1- and
    3-channel gain maps: k_warp_strip_batch<1..3>), the object API in a pass of its own (k_apply_*).  Same library, same gains:
    the panoramas must be bit-identical."""
    from opencv_starry_sky_panorama_stitcher_amd.starfield import Rig, _finish
    rng = np.random.default_rng(7000 + seed)
    w, h = int(rng.integers(80, 420)), int(rng.integers(60, 260))
    n = int(rng.integers(2, 6))
    step = float(rng.uniform(12, 30))
    yaws = [float((i - (n - 1) / 2) * step + rng.uniform(-3, 3)) for i in range(n)]
    pitches = [float(rng.uniform(-15, 15)) for _ in range(n)]
    warp = ["spherical", "cylindrical", "mercator"][seed % 3]
    bands = int(rng.integers(2, 5))
    kind = 1 + seed % 4                                   # GAIN, GAIN_BLOCKS, CHANNELS, CHANNELS_BLOCKS
    rig = _finish(Rig(f"gain fuzz {seed}", 9, w, h, 60.0, yaws, pitches, warp, "multiband", bands))
    frames, seams = starfield.make_frames(rig, want_seam=True)
    frames = [np.clip(f.astype(np.float32) * rng.uniform(0.7, 1.3) + 20, 0, 255).astype(np.uint8) for f in frames]   # exposures differ
    seams = [np.clip(sf.astype(np.float32) * 1.0 + 20, 0, 255).astype(np.uint8) for sf in seams]
    prep = bool(seed % 2)
    ref = cmp.compose_panorama(cv, frames, rig.Ks, rig.Rs, warp=rig.warp, warper_scale=rig.focal, blend="multiband", num_bands=bands, expos_comp=kind,
                               seam_frames=seams, seam_aspect=rig.seam_scale, mask_prep=prep)
    comp = cv.detail.ExposureCompensator_createDefault(kind)
    ws = cv.PyRotationWarper(rig.warp, rig.focal * rig.seam_scale)
    cs, ims, mks = [], [], []
    for i in range(rig.n):
        K = rig.Ks[i].copy(); K[0, 0] *= rig.seam_scale; K[0, 2] *= rig.seam_scale; K[1, 1] *= rig.seam_scale; K[1, 2] *= rig.seam_scale   # noqa: E702
        cnr, im = ws.warp(seams[i], K, rig.Rs[i], cv.INTER_AREA, cv.BORDER_REFLECT)
        _, mk = ws.warp(255 * np.ones(seams[i].shape[:2], np.uint8), K, rig.Rs[i], cv.INTER_NEAREST, cv.BORDER_CONSTANT)
        cs.append(cnr); ims.append(im); mks.append(mk)                                                                                     # noqa: E702
    comp.feed(corners=cs, images=ims, masks=mks)
    c = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), blend="multiband", num_bands=bands, mask_prep=prep, seam_size=rig.seam_size,
                     seam_aspect=rig.seam_scale, want_result_s16=True)
    c.set_compensator(comp)
    c.run([cv.UMat(f) for f in frames])
    mo, mk, rs = [u.get() for u in c.result()]
    assert c.pano_roi() == ref.pano_roi
    assert np.array_equal(mk, ref.result_mask) and np.array_equal(rs, ref.result) and np.array_equal(mo, ref.mosaic), (seed, w, h, n, warp, bands, kind, prep)


def test_pipelined_composers_on_two_streams():
    """bench.py --pipeline: two composers with a HIP stream each keep two panoramas in flight (the pool keeps per-stream free lists).
    Interleaved runs on DIFFERENT inputs must give exactly what each composer gives alone."""
    rig, frames, seams = _rig_small(2, 8, 4)
    alt = [np.ascontiguousarray(f[::-1, ::-1]) for f in frames]          # a second, different set of frames
    kw = dict(blend=rig.blend, num_bands=4, mask_prep=True, seam_size=rig.seam_size, seam_aspect=rig.seam_scale, want_result_s16=True)
    ref = []
    for fs in (frames, alt):
        c = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), **kw)
        c.run([cv.UMat(f) for f in fs])
        ref.append([u.get() for u in c.result()])
        del c
    a = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), own_stream=True, **kw)
    b = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), own_stream=True, **kw)
    da, db = [cv.UMat(f) for f in frames], [cv.UMat(f) for f in alt]
    for _ in range(6):               # no synchronisation between the launches of the two
        a.run(da)
        b.run(db)
    for comp, want in ((a, ref[0]), (b, ref[1])):
        got = [u.get() for u in comp.result()]
        assert all(np.array_equal(g, w) for g, w in zip(got, want))
    from opencv_starry_sky_panorama_stitcher_amd import _lib
    _lib.check(_lib.lib().ssp_use_stream(None))


@pytest.mark.parametrize("warp", ["spherical", "cylindrical", "mercator"])
@pytest.mark.parametrize("border", [0, 1, 2, 4])
def test_warp_f32c3_separable_kernel_bit_exact(warp, border):
    """Float frames through the table-based separable kernel (config 5's warp): same map, float weights in OpenCV's order -- bit for
    bit, including the outline (border rules) and frames smaller than a wave."""
    for (w, h, seed) in ((333, 207, 1), (40, 9, 2)):
        img = star_patch(w, h, seed=seed, dtype=np.float32, n_stars=30)
        K, R, f = camera(w, h, 70.0, yaw=14.0, pitch=-9.0, roll=12.0)
        cg, dg = cv.PyRotationWarper(warp, f).warp(img, K, R, cv.INTER_LINEAR, border)
        co, do = ocv.PyRotationWarper(warp, f).warp(img, K, R, ocv.INTER_LINEAR, border)
        assert cg == co and dg.dtype == np.float32 and np.array_equal(dg.view(np.uint32), do.view(np.uint32)), (warp, border, w, h)


@pytest.mark.parametrize("warp", ["spherical", "plane", "fisheye", "cylindrical"])
def test_warp_backward_bit_exact_and_round_trip(warp):
    """PyRotationWarper.warpBackward (not on the reference's path): HIP == oracle, and warp followed by warpBackward gives the frame
    back up to the two interpolations."""
    w, h = 140, 96
    img = star_patch(w, h, seed=31, n_stars=25)
    K, R, f = camera(w, h, 60.0, yaw=6.0, pitch=-4.0, roll=3.0)
    g, o = cv.PyRotationWarper(warp, f), ocv.PyRotationWarper(warp, f)
    _, warped = o.warp(img, K, R, ocv.INTER_LINEAR, ocv.BORDER_REFLECT)
    for interp in (cv.INTER_NEAREST, cv.INTER_LINEAR):
        bg = g.warpBackward(warped, K, R, interp, cv.BORDER_REFLECT, (w, h))
        bo = o.warpBackward(warped, K, R, interp, ocv.BORDER_REFLECT, (w, h))
        assert bg.shape == img.shape and np.array_equal(bg, bo), (warp, interp)
    back = g.warpBackward(warped, K, R, cv.INTER_LINEAR, cv.BORDER_REFLECT, (w, h)).astype(np.int32)
    inner = (slice(4, h - 4), slice(4, w - 4))
    assert np.mean(np.abs(back[inner] - img[inner].astype(np.int32))) < 6.0     # two bilinear passes blur the stars a little
    with pytest.raises(cv.error):
        g.warpBackward(warped[:-1], K, R, cv.INTER_LINEAR, cv.BORDER_REFLECT, (w, h))


def test_stream_lifetimes_and_cross_stream_frees():
    """Objects created while another composer's stream is current, and destroyed after that stream is gone: the pool re-homes what was
    allocated under a destroyed stream, and frees under a different stream wait for the stream the block was allocated under."""
    import gc
    rig, frames, seams = _rig_small(2, 8, 3)
    dev = [cv.UMat(f) for f in frames]
    kw = dict(blend=rig.blend, num_bands=3, want_result_s16=True)
    a = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), own_stream=True, **kw)
    a.run(dev)                                              # a's stream is now the current one
    plain = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), **kw)      # created meanwhile
    extra = cv.UMat(frames[0])                              # allocated under whatever stream is current
    plain.run(dev)
    want = [u.get() for u in plain.result()]
    got = [u.get() for u in a.result()]
    assert all(np.array_equal(x, y) for x, y in zip(got, want))
    del a
    gc.collect()                                            # a's stream is destroyed here
    plain.run(dev)
    assert all(np.array_equal(u.get(), y) for u, y in zip(plain.result(), want))
    del plain, extra, dev
    gc.collect()


@pytest.mark.parametrize("home", ["as_is", "null_stream"])
def test_inputs_dropped_while_another_stream_still_reads_them(home):
    """The pool's cross-stream guard: frames uploaded on the home stream are warped by composer `a` on ITS stream and released right
    after `a.run`, while `b`'s stream is current; the blocks go straight back into circulation (new uploads of other data reuse
    them).  `a`'s result must still be the panorama of the frames it was given.  Large frames so that `a`'s warp is still in flight
    when the blocks come back.  `null_stream`: the same after ssp_set_stream(NULL) -- HIP's null stream as the home stream is what
    bench.py and HipStripExchange get from torch's default stream, and a non-blocking composer stream does not synchronise with it
    implicitly (the guard used to read `home == nullptr` as "not a pool block" and registered no reader)."""
    import gc
    if home == "null_stream":
        from opencv_starry_sky_panorama_stitcher_amd import _lib as _l
        _l.check(_l.lib().ssp_set_stream(None))
    rig = starfield.make_rig(2, scale_div=2, n_override=3)
    frames = starfield.make_frames(rig)
    other = [np.full_like(f, 200) for f in frames]
    kw = dict(blend=rig.blend, num_bands=5, want_result_s16=True)
    solo = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), **kw)
    solo.run([cv.UMat(f) for f in frames])
    want = [u.get() for u in solo.result()]
    del solo
    a = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), own_stream=True, **kw)
    b = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), own_stream=True, **kw)
    from opencv_starry_sky_panorama_stitcher_amd import _lib
    for _ in range(3):
        _lib.check(_lib.lib().ssp_use_stream(None))
        da = [cv.UMat(f) for f in frames]               # uploaded on the home stream
        db = [cv.UMat(f) for f in other]
        a.run(da)
        b.run(db)                                       # b's stream is current from here on
        del da
        gc.collect()                                    # a's inputs are released while a's warp may still be reading them
        clobber = [cv.UMat(f) for f in other]           # same sizes: the pool hands the released blocks out again at once
        got = [u.get() for u in a.result()]
        assert all(np.array_equal(g, w) for g, w in zip(got, want))
        del db, clobber
    _lib.check(_lib.lib().ssp_use_stream(None))


@pytest.mark.parametrize("n", [4, 12])
def test_full_size_4k_gain_blocks_compose_matches_oracle(n):
    """BASELINE config 3 at full frame size (four of its 3840x2160 frames, and all twelve: the bench workload itself): seam-scale GAIN_BLOCKS feed, gains applied inside the fused
    warp, mask preparation, 5-band multiband -- against the oracle's call sequence.  The gains carry a 1e-9 relative difference (double
    sums in another order), so the north_star tolerance applies: +-1 LSB, on < 0.01 % of the samples."""
    rig = starfield.make_rig(3, scale_div=1, n_override=n)
    frames, seams = starfield.make_frames(rig, want_seam=True)
    comp = cv.detail.ExposureCompensator_createDefault(rig.expos_comp)
    ws = cv.PyRotationWarper(rig.warp, rig.focal * rig.seam_scale)
    cs, ims, mks = [], [], []
    for i in range(rig.n):
        K = rig.Ks[i].copy(); K[0, 0] *= rig.seam_scale; K[0, 2] *= rig.seam_scale; K[1, 1] *= rig.seam_scale; K[1, 2] *= rig.seam_scale
        cnr, im = ws.warp(seams[i], K, rig.Rs[i], cv.INTER_AREA, cv.BORDER_REFLECT)
        _, mk = ws.warp(255 * np.ones(seams[i].shape[:2], np.uint8), K, rig.Rs[i], cv.INTER_NEAREST, cv.BORDER_CONSTANT)
        cs.append(cnr); ims.append(im); mks.append(mk)
    comp.feed(corners=cs, images=ims, masks=mks)
    c = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), blend="multiband", num_bands=5, mask_prep=True, seam_size=rig.seam_size,
                     seam_aspect=rig.seam_scale, want_result_s16=True)
    c.set_compensator(comp)
    c.run([cv.UMat(f) for f in frames])
    mo, mk, rs = [u.get() for u in c.result()]
    ref = cmp.compose_panorama(ocv, frames, rig.Ks, rig.Rs, warp=rig.warp, warper_scale=rig.focal, blend="multiband", num_bands=5, expos_comp=rig.expos_comp,
                               seam_frames=seams, seam_aspect=rig.seam_scale)
    assert c.pano_roi() == ref.pano_roi and np.array_equal(mk, ref.result_mask)
    diff = np.abs(mo.astype(np.int16) - ref.mosaic.astype(np.int16))
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-4
    assert mo.shape[1] > (20000 if n == 12 else 8000)


def _bench_rehearsal(world, steps=3):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SSP_DIST_BACKEND="gloo", SSP_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 34500 + (os.getpid() % 2000) + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", str(steps), "--warmup", "1", "--scale-div", "4"]
    r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_bench_two_ranks_rehearsal_prints_one_line(tmp_path):
    """bench.py's N>1 contract end to end (launcher command line of the driver, two ranks sharing this GPU over gloo, frames at 1/4
    size): every rank takes part in every exchange step -- timed, warm-up and profiled ones -- and rank 0 prints the one JSON line."""
    out = _bench_rehearsal(2)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0 and out["cpu_baseline"] is None
    assert out["config"]["exchange_bytes_rank0"] > 0 and out["roofline"]["frac"] > 0
    # the line explains its exchange: per rank what went to / came from which neighbour, how long the collapse waited for it, which protocol ran
    ex = out["exchange"]
    assert ex["protocol"] == "all-level strips" and ex["double_buffered"] is True and ex["backend"] == "gloo" and len(ex["per_rank"]) == 2
    for r, pr in enumerate(ex["per_rank"]):
        other = str(1 - r)
        assert pr["rank"] == r and pr["sent_bytes"][other] > 0 and pr["recv_bytes"][other] > 0 and pr["strips_sent"] > 0 and pr["strips_received"] > 0
        assert pr["panoramas_timed"] >= 3 and pr["recv_wait_stream_ms"] >= 0 and pr["recv_wait_host_ms"] >= 0
        assert len(pr["owned"]) == 4 and len(pr["region"]) == 4 and pr["feed_units"] == 6
    assert ex["per_rank"][0]["sent_bytes"]["1"] == ex["per_rank"][1]["recv_bytes"]["0"]


def test_bench_four_ranks_rehearsal_closed_rings(tmp_path):
    """The N = 4 line of the driver's scaling run, rehearsed with four ranks sharing this GPU over gloo: 2 rows x 12 yaw positions at 30 degrees -- two
    CLOSED rings, the first and the last rank hold the frames that straddle u = +-pi*scale (two feed units each: 8 units for 6 frames) -- through
    HipStripPipeline (torch tensors as strip buffers, the double-buffered step), which the emulation tests do not run."""
    out = _bench_rehearsal(4, steps=2)
    assert out["n_gpus"] == 4 and out["value"] > 0
    ex = out["exchange"]
    assert len(ex["per_rank"]) == 4
    units = [pr["feed_units"] for pr in ex["per_rank"]]
    assert units == [8, 6, 6, 8]
    pano_w = out["config"]["pano"][2]
    for r, pr in enumerate(ex["per_rank"]):
        assert pr["region"][2] < 0.6 * pano_w and pr["strips_received"] > 0 and pr["panoramas_timed"] >= 2
    # the ring closes: the first and the last rank are neighbours through their straddling frames' far units
    assert ex["per_rank"][0]["sent_bytes"].get("3", 0) > 0 and ex["per_rank"][3]["sent_bytes"].get("0", 0) > 0


@pytest.mark.parametrize("ctype", [1, 2, 3, 4])
def test_compensator_set_mat_gains_round_trip(ctype):
    """cv2's getMatGains / setMatGains pair: a second compensator that receives the first one's gains (no feed) applies identically."""
    rig = starfield.make_rig(3, scale_div=8, n_override=4)
    frames, seams = starfield.make_frames(rig, want_seam=True)
    ws = cv.PyRotationWarper(rig.warp, rig.focal)
    cs, ims, mks = [], [], []
    for i in range(rig.n):
        cnr, im = ws.warp(frames[i], rig.Ks[i], rig.Rs[i], cv.INTER_LINEAR, cv.BORDER_REFLECT)
        _, mk = ws.warp(255 * np.ones(frames[i].shape[:2], np.uint8), rig.Ks[i], rig.Rs[i], cv.INTER_NEAREST, cv.BORDER_CONSTANT)
        cs.append(cnr); ims.append(im); mks.append(mk)
    a = cv.detail.ExposureCompensator_createDefault(ctype)
    a.feed(corners=cs, images=ims, masks=mks)
    b = cv.detail.ExposureCompensator_createDefault(ctype)
    b.setMatGains(a.getMatGains())
    changed = 0
    for i in range(rig.n):
        x, y = ims[i].copy(), ims[i].copy()
        a.apply(i, cs[i], x, mks[i]); b.apply(i, cs[i], y, mks[i])
        assert np.array_equal(x, y)
        changed += int(not np.array_equal(x, ims[i]))
    assert changed >= 2 and np.array_equal(a.gains(), b.gains())
    with pytest.raises(cv.error):
        cv.detail.ExposureCompensator_createDefault(0).setMatGains([np.ones((1, 1))])


@pytest.mark.parametrize("seed", list(range(16)))
def test_fuzz_strip_exchange(seed):
    """Random grids of frames (rows, columns, yaw / pitch steps, frame size, warp), random band counts and rank counts, block or
    interleaved ownership: the strip protocol with every rank emulated on this GPU must reproduce the single-composer panorama bit
    for bit on every owned pixel, and the owned sets must tile the blended area."""
    from opencv_starry_sky_panorama_stitcher_amd import parallel
    from opencv_starry_sky_panorama_stitcher_amd.starfield import Rig, _finish
    rng = np.random.default_rng(7000 + seed)
    rows, cols = int(rng.integers(1, 3)), int(rng.integers(2, 6))
    w, h = int(rng.integers(20, 40)) * 8, int(rng.integers(12, 24)) * 8
    ystep, pstep = float(rng.uniform(16, 30)), float(rng.uniform(10, 20))
    warp = ["spherical", "cylindrical", "spherical", "mercator"][seed % 4]
    yaws, pitches = [], []
    for r in range(rows):
        for c in range(cols):
            yaws.append((c - (cols - 1) / 2.0) * ystep + float(rng.uniform(-2, 2)))
            pitches.append((r - (rows - 1) / 2.0) * pstep + float(rng.uniform(-2, 2)))
    nb = int(rng.integers(2, 5))
    rig = _finish(Rig(f"fuzz {rows}x{cols}", 4, w, h, 60.0, yaws, pitches, warp, "multiband", nb))
    n = rig.n
    world = int(rng.integers(2, min(4, n) + 1))
    if seed % 3 == 0:
        owner = [int(i % world) for i in range(n)]                              # interleaved
    else:
        owner = sorted(int(v) for v in (np.arange(n) * world // n))             # contiguous runs
    frames = [rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8) for _ in range(n)]
    wr = cv.PyRotationWarper(rig.warp, rig.focal)
    rois = [wr.warpRoi((w, h), rig.Ks[i], rig.Rs[i]) for i in range(n)]
    try:
        plan = parallel.plan_strips([r[:2] for r in rois], [r[2:] for r in rois], owner, world, nb, levels=seed % 5 != 4)    # (every fifth: level-0 strips)
    except ValueError as exc:
        # interleaved ownership of heavily overlapping frames can leave a rank without a cell of its own: the plan refuses it
        assert "owns no part" in str(exc)
        pytest.skip(str(exc))
    dev = [cv.UMat(f) for f in frames]
    full = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (w, h), num_bands=nb, want_result_s16=True)
    assert full.pano_roi() == plan.pano_roi
    full.run(dev)
    ref_mos, ref_mask, ref_res = [u.get() for u in full.result()]
    exs, per_rank = [], []
    for r in range(world):
        idx = [i for i in range(n) if owner[i] == r]
        c = cmp.Composer(rig.warp, rig.focal, [rig.Ks[i] for i in idx], [rig.Rs[i] for i in idx], (w, h), num_bands=nb, want_result_s16=True)
        exs.append(parallel.StripExchangeBase(c, plan, r, parallel._umat_alloc))
        per_rank.append([dev[i] for i in idx])
    for _ in range(2):                                                          # the second pass reuses every buffer
        parallel.emulate_strip_exchange(exs, per_rank)
    own = parallel.strip_owner_map(plan)
    assert np.all((own >= 0) | (ref_mask == 0))
    covered = 0
    for r in range(world):
        mos, mk, rs = [u.get() for u in exs[r].c.result()]
        x0, y0 = plan.region[r][0], plan.region[r][1]
        hh, ww = mk.shape
        sel = own[y0:y0 + hh, x0:x0 + ww] == r
        assert int(sel.sum()) == int((own == r).sum())
        covered += int(sel.sum())
        assert np.array_equal(mk[sel], ref_mask[y0:y0 + hh, x0:x0 + ww][sel])
        assert np.array_equal(rs[sel], ref_res[y0:y0 + hh, x0:x0 + ww][sel])
        assert np.array_equal(mos[sel], ref_mos[y0:y0 + hh, x0:x0 + ww][sel])
    assert covered == int((own >= 0).sum())


@pytest.mark.parametrize("seed", list(range(24)))
def test_fuzz_helpers(seed):
    """Random sizes and ratios for the kernels around the loop: resize INTER_LINEAR_EXACT (down, near 1, the 4-byte window path of
    large magnifications and its fallback at row ends), the fused resize+and, INTER_AREA decimation, dilate, Voronoi seams, feather
    and NO blenders, float multiband -- against the oracle, bit for bit (float: 1e-3)."""
    rng = np.random.default_rng(9000 + seed)
    sw, sh = int(rng.integers(2, 90)), int(rng.integers(2, 70))
    m = (rng.uniform(size=(sh, sw)) > rng.uniform(0.2, 0.8)).astype(np.uint8) * 255
    if seed % 2:
        m = rng.integers(0, 256, size=(sh, sw), dtype=np.uint8)
    assert np.array_equal(cv.dilate(m, None), ocv.dilate(m, None))
    for _ in range(3):
        fx, fy = float(rng.choice([0.4, 0.9, 1.0, 1.7, 2.9, 3.0, 3.3, 4.0, 7.5, 18.3])), float(rng.choice([0.5, 1.0, 2.2, 3.1, 9.0, 17.0]))
        dsize = (max(1, int(sw * fx) + int(rng.integers(0, 3))), max(1, int(sh * fy) + int(rng.integers(0, 3))))
        want = ocv.resize(m, dsize, 0, 0, ocv.INTER_LINEAR_EXACT)
        assert np.array_equal(cv.resize(m, dsize, 0, 0, cv.INTER_LINEAR_EXACT), want), (sw, sh, dsize)
    # decimation (the frame prologue)
    iw, ih = int(rng.integers(40, 400)), int(rng.integers(30, 300))
    img = rng.integers(0, 256, size=(ih, iw, 3), dtype=np.uint8)
    f = float(rng.uniform(0.05, 0.95))
    if int(round(iw * f)) >= 1 and int(round(ih * f)) >= 1:
        assert np.array_equal(cv.resize(img, None, fx=f, fy=f, interpolation=cv.INTER_AREA), ocv.resize(img, None, fx=f, fy=f, interpolation=ocv.INTER_AREA)), (iw, ih, f)
    # seams + blenders on a random layout
    n = int(rng.integers(2, 5))
    imgs, masks, tls = [], [], []
    for i in range(n):
        w, h = int(rng.integers(8, 120)), int(rng.integers(8, 90))
        imgs.append(rng.integers(-200, 500, size=(h, w, 3)).astype(np.int16))
        mk = np.zeros((h, w), np.uint8)
        mk[int(rng.integers(0, h // 3)):h - int(rng.integers(0, h // 3)), int(rng.integers(0, w // 3)):w - int(rng.integers(0, w // 3))] = 255
        masks.append(mk)
        tls.append((int(rng.integers(-60, 60)), int(rng.integers(-40, 40))))
    want = ocv.detail.SeamFinder_createDefault(1).find(None, tls, [mk.copy() for mk in masks])
    got = cv.detail.SeamFinder_createDefault(1).find(None, tls, [mk.copy() for mk in masks])
    assert all(np.array_equal(a, b) for a, b in zip(got, want))
    sharp = float(rng.uniform(0.01, 0.5))
    for mg, mo in ((lambda: cv.detail_FeatherBlender(sharp), lambda: ocv.detail_FeatherBlender(sharp)),
                   (lambda: cv.detail.Blender_createDefault(cv.detail.Blender_NO), lambda: ocv.detail.Blender_createDefault(ocv.detail.Blender_NO))):
        (rg, kg), (ro, ko) = _blend_both(mg, mo, imgs, masks, tls)
        assert np.array_equal(kg, ko) and np.array_equal(rg, ro)
    nbf = int(rng.integers(1, 5))
    fimgs = [im.astype(np.float32) for im in imgs]
    (rg, kg), (ro, ko) = _blend_both(lambda: cv.detail_MultiBandBlender(num_bands=nbf, float_pyramids=True),
                                     lambda: ocv.detail_MultiBandBlender(num_bands=nbf, float_pyramids=True), fimgs, masks, tls)
    assert np.array_equal(kg, ko) and np.max(np.abs(rg - ro)) <= 1e-3


@pytest.mark.parametrize("seed", list(range(16)))
def test_fuzz_compensators(seed):
    """Random image counts, sizes, positions (partial and no overlaps, an isolated image), block sizes and feed counts for the four
    compensators: gains at 1e-9 (scalar) / 1e-6 (float32 maps) relative, applied images +-1 LSB on < 0.1 % of the samples."""
    rng = np.random.default_rng(11000 + seed)
    ctype = 1 + seed % 4
    n = int(rng.integers(2, 6))
    base = star_patch(400, 160, seed=seed + 300)
    corners, images, masks = [], [], []
    for i in range(n):
        w, h = int(rng.integers(24, 120)), int(rng.integers(20, 90))
        x0, y0 = int(rng.integers(0, 400 - w)), int(rng.integers(0, 160 - h))
        im = np.clip(np.rint(base[y0:y0 + h, x0:x0 + w].astype(np.float32) * rng.uniform(0.6, 1.4)), 0, 255).astype(np.uint8)
        mk = 255 * np.ones((h, w), np.uint8)
        mk[:, : int(rng.integers(0, 6))] = 0
        mk[rng.integers(0, h, 20), rng.integers(0, w, 20)] = rng.integers(0, 256, 20)
        corners.append((x0 - 11, y0 + 5)); images.append(np.ascontiguousarray(im)); masks.append(mk)
    bw, bh = int(rng.choice([8, 16, 32])), int(rng.choice([8, 16, 32]))
    feeds = int(rng.integers(1, 3))
    cg = cv.detail.ExposureCompensator_createDefault(ctype)
    cg.setNrFeeds(feeds)
    if ctype in (2, 4):
        cg.setBlockSize(bw, bh)
    co = ocv._Comp(ctype, bw, bh, feeds, 2)
    cg.feed(corners=corners, images=images, masks=masks)
    co.feed(corners, images, masks)
    if ctype in (1, 3):
        assert np.allclose(cg.gains(), co.gains(), rtol=1e-9, atol=0)
    else:
        for i in range(n):
            assert np.allclose(cg.gainMap(i), co.gainMap(i), rtol=1e-6, atol=0)
    for i in range(n):
        a, b = images[i].copy(), images[i].copy()
        cg.apply(i, corners[i], a, None)
        co.apply(i, corners[i], b, None)
        d = np.abs(a.astype(np.int16) - b.astype(np.int16))
        assert d.max() <= 1 and (d > 0).mean() < 1e-3


@pytest.mark.parametrize("world", [2, 4, 8])
def test_bench_block_layouts_emulated(world):
    """The exact layouts bench.py runs on 2, 4 and 8 GPUs (2x3 blocks of a 2x6 / 2x12 / 4x12 grid, SURVEY's 30 degree yaw steps: with 12
    columns the rings close and the outer columns' frames straddle u = +-pi*scale -- planned by their live parts; frames at 1/8
    size, 3 bands so that the geometry scales with them): all ranks emulated on this GPU, every owned pixel equal to the one-composer
    panorama of all 6*world frames."""
    import importlib.util
    from opencv_starry_sky_panorama_stitcher_amd import parallel
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    div, nb = 8, 3
    rigs = [bench.block_rig(starfield, world, r, div)[0] for r in range(world)]
    w, h = rigs[0].width, rigs[0].height
    rng = np.random.default_rng(world)
    Ks, Rs, owner = [], [], []
    for r, rg in enumerate(rigs):
        Ks += rg.Ks; Rs += rg.Rs; owner += [r] * rg.n
    frames = [cv.UMat(rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)) for _ in owner]
    fp = parallel.feed_parts(cv, rigs[0].warp, rigs[0].focal, (w, h), Ks, Rs, owner, nb)
    assert len(fp.corners) == len(owner) + (0 if world == 2 else world)      # 12 columns: one straddling frame per row and end
    plan = parallel.plan_strips(fp.corners, fp.sizes, fp.owner, world, nb, pano_roi=fp.pano_roi)
    full = cmp.Composer(rigs[0].warp, rigs[0].focal, Ks, Rs, (w, h), num_bands=nb, want_result_s16=True)
    full.run(frames)
    ref_mos, ref_mask, ref_res = [u.get() for u in full.result()]
    exs, per_rank = [], []
    for r in range(world):
        idx = [i for i in range(len(owner)) if owner[i] == r]
        c = cmp.Composer(rigs[0].warp, rigs[0].focal, [Ks[i] for i in idx], [Rs[i] for i in idx], (w, h), num_bands=nb, want_result_s16=True)
        exs.append(parallel.StripExchangeBase(c, plan, r, parallel._umat_alloc))
        per_rank.append([frames[i] for i in idx])
    parallel.emulate_strip_exchange(exs, per_rank)
    own = parallel.strip_owner_map(plan)
    covered = 0
    for r in range(world):
        mos, mk, rs = [u.get() for u in exs[r].c.result()]
        x0, y0 = plan.region[r][0], plan.region[r][1]
        hh, ww = mk.shape
        sel = own[y0:y0 + hh, x0:x0 + ww] == r
        assert int(sel.sum()) == int((own == r).sum()) > 0
        covered += int(sel.sum())
        assert np.array_equal(mk[sel], ref_mask[y0:y0 + hh, x0:x0 + ww][sel])
        assert np.array_equal(rs[sel], ref_res[y0:y0 + hh, x0:x0 + ww][sel])
        assert np.array_equal(mos[sel], ref_mos[y0:y0 + hh, x0:x0 + ww][sel])
    assert covered == int((own >= 0).sum()) and np.all((own >= 0) | (ref_mask == 0))


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_tall_frames_through_the_staged_pyramid_kernels(seed):
    """The LDS-staged pyrDown kernels (level 0: k_pyr_down_strip_lds, levels >= 1 of 8-bit fed pyramids: k_pyr_down_strip_pk_lds) only
    take levels of at least 512 rows, which the small fuzz rigs never reach: portrait frames of random odd sizes (1.0-1.3 k x 2.5-2.9 k),
    random cameras, 5 or 6 bands, with and without mask preparation / exposure compensation -- the Composer against the oracle's call
    sequence, bit for bit (gains: +-1 LSB on < 0.01 % of the samples)."""
    from opencv_starry_sky_panorama_stitcher_amd.starfield import Rig, _finish
    rng = np.random.default_rng(77000 + seed)
    w, h = int(rng.integers(1000, 1300)), int(rng.integers(2500, 2900))
    n = 2 + seed % 2
    step = float(rng.uniform(10, 22))
    yaws = [(i - (n - 1) / 2.0) * step + float(rng.uniform(-2, 2)) for i in range(n)]
    pitches = [float(rng.uniform(-5, 5)) for _ in range(n)]
    comp_kind = [0, 2, 0, 1, 2, 0][seed % 6]
    rig = _finish(Rig("tall", 90 + seed, w, h, float(rng.uniform(28, 40)), yaws, pitches, "spherical" if seed % 3 else "cylindrical", "multiband", 5 + seed % 2,
                      expos_comp=comp_kind, exposure_spread=(0.8, 1.25)))
    frames, seams = starfield.make_frames(rig, want_seam=True)
    prep = seed % 2 == 0 or comp_kind != 0
    c = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), blend="multiband", num_bands=rig.num_bands, mask_prep=prep, seam_size=rig.seam_size,
                     seam_aspect=rig.seam_scale, want_result_s16=True)
    if comp_kind:
        comp = cv.detail.ExposureCompensator_createDefault(comp_kind)
        ws = cv.PyRotationWarper(rig.warp, rig.focal * rig.seam_scale)
        cs, ims, mks = [], [], []
        for i in range(rig.n):
            K = rig.Ks[i].copy(); K[0, 0] *= rig.seam_scale; K[0, 2] *= rig.seam_scale; K[1, 1] *= rig.seam_scale; K[1, 2] *= rig.seam_scale   # noqa: E702
            cnr, im = ws.warp(seams[i], K, rig.Rs[i], cv.INTER_AREA, cv.BORDER_REFLECT)
            _, mk = ws.warp(255 * np.ones(seams[i].shape[:2], np.uint8), K, rig.Rs[i], cv.INTER_NEAREST, cv.BORDER_CONSTANT)
            cs.append(cnr); ims.append(im); mks.append(mk)                                                                                     # noqa: E702
        comp.feed(corners=cs, images=ims, masks=mks)
        c.set_compensator(comp)
    dev = [cv.UMat(f) for f in frames]
    for _ in range(2):          # second panorama: the composer's steady state (tables, records, rest plan known)
        c.run(dev)
    mo, mk, rs = [u.get() for u in c.result()]
    ref = cmp.compose_panorama(ocv, frames, rig.Ks, rig.Rs, warp=rig.warp, warper_scale=rig.focal, blend="multiband", num_bands=rig.num_bands, expos_comp=comp_kind,
                               seam_frames=seams if prep else None, seam_aspect=rig.seam_scale, mask_prep=prep)
    assert c.pano_roi() == ref.pano_roi and np.array_equal(mk, ref.result_mask)
    assert mo.shape[0] >= 2048
    if comp_kind:
        d = np.abs(mo.astype(np.int16) - ref.mosaic.astype(np.int16))
        assert d.max() <= 1 and (d > 0).mean() < 1e-4
    else:
        assert np.array_equal(mo, ref.mosaic) and np.array_equal(rs, ref.result)


# (config 4 -- 48 frames on eight emulated ranks against the oracle, at 1/8 and at full frame size -- lives in tests/test_closed_ring.py since its
# rig returned to SURVEY's 30 degree steps: test_config4_closed_layout_eight_ranks_against_the_oracle)


@pytest.mark.parametrize("seed", list(range(12)))
def test_fuzz_composer_float_rigs(seed):
    """The config-5 path on random small rigs: float32 frames (sizes on either side of the 62-output pyrDown waves and the 256-column
    warp groups), float pyramids of 1..6 bands, with and without mask preparation.  Mask bit-exact, result within 1e-3 grey levels of
    the oracle (float sums in feed order on both sides; the float warp itself is bit-exact)."""
    from opencv_starry_sky_panorama_stitcher_amd.starfield import Rig, _finish
    rng = np.random.default_rng(13000 + seed)
    w, h = int(rng.integers(40, 420)), int(rng.integers(30, 200))
    n = int(rng.integers(1, 5))
    step = float(rng.uniform(12, 35))
    yaws = [float((i - (n - 1) / 2) * step + rng.uniform(-3, 3)) for i in range(n)]
    pitches = [float(rng.uniform(-5, 5)) for _ in range(n)]
    warp = ["spherical", "cylindrical", "mercator"][seed % 3]
    bands = int(rng.integers(1, 7))
    rig = _finish(Rig(f"fuzz f32 {seed}", 9, w, h, 60.0, yaws, pitches, warp, "multiband", bands, dtype="f32"))
    frames, seams = starfield.make_frames(rig, want_seam=True)
    assert frames[0].dtype == np.float32
    prep = bool(seed % 2)
    c = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), blend="multiband", num_bands=bands, float_frames=True, mask_prep=prep,
                     seam_size=rig.seam_size, seam_aspect=rig.seam_scale, want_result_s16=True)
    c.run([cv.UMat(f) for f in frames])
    c.run([cv.UMat(f) for f in frames])            # a second run reuses the planes
    mo, mk, rs = [u.get() for u in c.result()]
    ref = cmp.compose_panorama(ocv, frames, rig.Ks, rig.Rs, warp=rig.warp, warper_scale=rig.focal, blend="multiband", num_bands=bands,
                               seam_frames=seams if prep else None, seam_aspect=rig.seam_scale, mask_prep=prep, float_pyramids=True)
    assert c.pano_roi() == ref.pano_roi and np.array_equal(mk, ref.result_mask), (seed, w, h, n, warp, bands, prep)
    assert rs.dtype == np.float32 and np.max(np.abs(rs - ref.result)) <= 1e-3, (seed, w, h, n, warp, bands, prep)


@pytest.mark.parametrize("seed", list(range(16)))
def test_fuzz_composer_other_projections_and_rings(seed):
    """The Composer outside its batched fast path: the 13 projections without separable tables (generic warp kernel, per-image feed),
    feather and NO blenders, and spherical / cylindrical rings wide enough that a frame straddles u = +-pi*scale (OpenCV's by-border roi
    then spans the whole surface) -- against the oracle running the reference's call sequence, bit for bit."""
    from opencv_starry_sky_panorama_stitcher_amd.starfield import Rig, _finish
    rng = np.random.default_rng(15000 + seed)
    others = ["plane", "fisheye", "stereographic", "compressedPlaneA2B1", "compressedPlaneA1.5B1", "compressedPlanePortraitA2B1",
              "compressedPlanePortraitA1.5B1", "paniniA2B1", "paniniA1.5B1", "paniniPortraitA2B1", "paniniPortraitA1.5B1", "transverseMercator"]
    ring = seed % 4 == 3
    if ring:
        warp = ["spherical", "cylindrical"][(seed // 4) % 2]
        n = int(rng.integers(7, 10))
        step = 360.0 / n                                  # the ring closes: the frames at the ends straddle +-180 degrees
        yaws = [float((i - (n - 1) / 2) * step) for i in range(n)]
        pitches = [float(rng.uniform(-4, 4)) for _ in range(n)]
        w, h = int(rng.integers(50, 90)), int(rng.integers(36, 60))
    else:
        warp = others[seed % len(others)]
        n = int(rng.integers(2, 4))
        step = float(rng.uniform(10, 22))                 # plane-like projections blow up towards 90 degrees: keep the rig narrow
        yaws = [float((i - (n - 1) / 2) * step + rng.uniform(-2, 2)) for i in range(n)]
        pitches = [float(rng.uniform(-8, 8)) for _ in range(n)]
        w, h = int(rng.integers(60, 200)), int(rng.integers(40, 140))
    blend = ["multiband", "feather", "no", "multiband"][(seed // 2) % 4]
    bands = int(rng.integers(2, 5))
    rig = _finish(Rig(f"fuzz other {seed}", 9, w, h, 60.0, yaws, pitches, warp, blend, bands))
    frames, seams = starfield.make_frames(rig, want_seam=True)
    kwc = dict(blend=blend, mask_prep=bool(seed % 2), seam_size=rig.seam_size, seam_aspect=rig.seam_scale, want_result_s16=True)
    kwo = dict(warp=rig.warp, warper_scale=rig.focal, blend=blend, seam_frames=seams if seed % 2 else None, seam_aspect=rig.seam_scale, mask_prep=bool(seed % 2))
    if blend == "multiband":
        kwc["num_bands"] = bands; kwo["num_bands"] = bands
    if blend != "multiband":
        kwo["num_bands"] = None
    if blend == "feather":
        kwo["blend_strength"] = 5.0
    ref = cmp.compose_panorama(ocv, frames, rig.Ks, rig.Rs, **kwo)
    if ring:
        assert ref.pano_roi[2] > 3 * w                    # the straddling frame's roi spans the surface
    if blend == "feather":                                # sde.py:1808-1819: sharpness = 1 / (sqrt(pano area) * strength / 100)
        kwc["sharpness"] = float(1.0 / (np.sqrt(ref.pano_roi[2] * ref.pano_roi[3]) * 5.0 / 100))
    c = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), **kwc)
    c.run([cv.UMat(f) for f in frames])
    mo, mk, rs = [u.get() for u in c.result()]
    assert c.pano_roi() == ref.pano_roi
    assert np.array_equal(mk, ref.result_mask) and np.array_equal(rs, ref.result) and np.array_equal(mo, ref.mosaic), (seed, warp, blend, n, w, h)


@pytest.mark.parametrize("world,owner,nb", [(2, [0, 0, 1, 1], 4), (3, [0, 0, 1, 1, 2, 2], 3), (2, [0, 0, 0, 1, 1, 1], 2)])
def test_strip_exchange_float_frames_emulated(world, owner, nb):
    """BASELINE config 5 on several GPUs: float32 frames, float pyramids.  The strips are float32 level-0 planes (12 B/px + the 8-bit
    mask); with the global feed order restored the float sums run in the same order as on one GPU, so every owned pixel of the
    float result equals the single-composer panorama BIT FOR BIT."""
    from opencv_starry_sky_panorama_stitcher_amd import parallel
    rig = starfield.make_rig(5, scale_div=32, n_override=len(owner))
    n = len(owner)
    rig.yaws_deg, rig.pitches_deg, rig.Ks, rig.Rs = rig.yaws_deg[:n], rig.pitches_deg[:n], rig.Ks[:n], rig.Rs[:n]
    frames, seams = starfield.make_frames(rig, want_seam=True)
    assert frames[0].dtype == np.float32
    w = cv.PyRotationWarper(rig.warp, rig.focal)
    rois = [w.warpRoi((rig.width, rig.height), rig.Ks[i], rig.Rs[i]) for i in range(n)]
    plan = parallel.plan_strips([r[:2] for r in rois], [r[2:] for r in rois], owner, world, nb)
    dev = [cv.UMat(f) for f in frames]
    kw = dict(blend="multiband", num_bands=nb, float_frames=True, mask_prep=True, seam_size=rig.seam_size, seam_aspect=rig.seam_scale, want_result_s16=True)
    full = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), **kw)
    assert full.pano_roi() == plan.pano_roi
    full.run(dev)
    ref_mos, ref_mask, ref_res = [u.get() for u in full.result()]
    assert ref_res.dtype == np.float32
    exs, per_rank = [], []
    for r in range(world):
        idx = [i for i in range(n) if owner[i] == r]
        c = cmp.Composer(rig.warp, rig.focal, [rig.Ks[i] for i in idx], [rig.Rs[i] for i in idx], (rig.width, rig.height), **kw)
        # the seam-scale masks belong to the frames: a rank's composer derives its own from its cameras, as the full one does
        exs.append(parallel.StripExchangeBase(c, plan, r, parallel._umat_alloc))
        per_rank.append([dev[i] for i in idx])
    for _ in range(2):
        parallel.emulate_strip_exchange(exs, per_rank)
    own = parallel.strip_owner_map(plan)
    covered = 0
    for r in range(world):
        mos, mk, rs = [u.get() for u in exs[r].c.result()]
        x0, y0 = plan.region[r][0], plan.region[r][1]
        hh, ww = mk.shape
        sel = own[y0:y0 + hh, x0:x0 + ww] == r
        assert int(sel.sum()) == int((own == r).sum()) > 0
        covered += int(sel.sum())
        assert np.array_equal(mk[sel], ref_mask[y0:y0 + hh, x0:x0 + ww][sel])
        assert np.array_equal(rs[sel].view(np.uint32), ref_res[y0:y0 + hh, x0:x0 + ww][sel].view(np.uint32))
        assert np.array_equal(mos[sel], ref_mos[y0:y0 + hh, x0:x0 + ww][sel])
    assert covered == int((own >= 0).sum())


def test_level_strip_entry_points_refuse_what_they_cannot_do():
    """ssp_blender_export_level_strips / ssp_blender_feed_level_strips: pyramids not built yet, a band count the buffers were not sized for,
    rectangles off the 2^bands grid or outside the image's planes, an origin right of the strip -- errors, never a copy."""
    import ctypes as C
    from opencv_starry_sky_panorama_stitcher_amd import parallel
    L, chk = cv._lib.lib(), cv._lib.check
    owner, nb = [0, 0, 1, 1], 3
    rig, frames, _ = _rig_small(3, 8, len(owner))
    w = cv.PyRotationWarper(rig.warp, rig.focal)
    rois = [w.warpRoi((rig.width, rig.height), rig.Ks[i], rig.Rs[i]) for i in range(rig.n)]
    plan = parallel.plan_strips([r[:2] for r in rois], [r[2:] for r in rois], owner, 2, nb)
    assert plan.levels and plan.sends(0)
    c = cmp.Composer(rig.warp, rig.focal, rig.Ks[:2], rig.Rs[:2], (rig.width, rig.height), num_bands=nb)
    ex = parallel.StripExchangeBase(c, plan, 0, parallel._umat_alloc)
    c.feed_planes([cv.UMat(f) for f in frames[:2]])
    with pytest.raises(Exception, match="not built yet"):
        ex.export_all()                                   # level strips are cut from finished pyramids
    c.feed_pyramids()
    out = ex.export_all()
    i, d, r, buf, _ = out[0]
    blender = c.blender_handle()
    one = lambda v: (C.c_int * 1)(v)                      # noqa: E731
    rect = lambda *v: (C.c_int * 4)(*v)                   # noqa: E731
    ptr = (C.c_void_p * 1)(buf[1])
    with pytest.raises(Exception, match="sized for"):
        chk(L.ssp_blender_export_level_strips(blender, 1, one(ex.local[i]), rect(*r), ptr, plan.nb + 1))
    with pytest.raises(Exception, match="aligned to"):
        chk(L.ssp_blender_export_level_strips(blender, 1, one(ex.local[i]), rect(r[0] + 4, r[1], r[2], r[3]), ptr, plan.nb))
    far = plan.prect[[k for k in range(rig.n) if owner[k] == 0 and k != i][0]]
    if not (far[0] <= r[0] and r[0] + r[2] <= far[0] + far[2]):
        with pytest.raises(Exception, match="inside the image's padded rectangle"):
            chk(L.ssp_blender_export_level_strips(blender, 1, one(1 - ex.local[i]), rect(*r), ptr, plan.nb))
    with pytest.raises(Exception, match="no fed image"):
        chk(L.ssp_blender_export_level_strips(blender, 1, one(7), rect(*r), ptr, plan.nb))
    with pytest.raises(Exception, match="left of its image"):
        chk(L.ssp_blender_feed_level_strips(blender, 1, rect(*r), one(r[0] + (1 << plan.nb)), ptr, plan.nb))
    with pytest.raises(Exception, match="sized for"):
        chk(L.ssp_blender_feed_level_strips(blender, 1, rect(*r), one(plan.prect[i][0]), ptr, plan.nb - 1))
    size = C.c_size_t()
    with pytest.raises(Exception, match="not a multiple"):
        chk(L.ssp_level_strip_buffer_bytes(r[2] + 1, r[3], plan.nb, 0, C.byref(size)))
    chk(L.ssp_level_strip_buffer_bytes(r[2], r[3], plan.nb, 0, C.byref(size)))
    assert size.value == ex.level_buffer_bytes(r) > 4 * r[2] * r[3]


@pytest.mark.parametrize("ctype", [1, 2, 3, 4])
def test_strip_exchange_with_exposure_compensation(ctype):
    """the gains come from ONE solve over all frames; a rank's composer gets the gains of its own frames
    (parallel.subset_compensator, cv2's getMatGains / setMatGains).  With them the strip exchange reproduces the single-composer
    panorama -- same gains, so bit for bit."""
    from opencv_starry_sky_panorama_stitcher_amd import parallel
    owner, nb, world = [0, 0, 0, 1, 1], 4, 2
    rig = starfield.make_rig(3, scale_div=8, n_override=len(owner))
    frames, seams = starfield.make_frames(rig, want_seam=True)
    ws = cv.PyRotationWarper(rig.warp, rig.focal * rig.seam_scale)
    cs, ims, mks = [], [], []
    for i in range(rig.n):
        K = rig.Ks[i].copy(); K[0, 0] *= rig.seam_scale; K[0, 2] *= rig.seam_scale; K[1, 1] *= rig.seam_scale; K[1, 2] *= rig.seam_scale
        cnr, im = ws.warp(seams[i], K, rig.Rs[i], cv.INTER_AREA, cv.BORDER_REFLECT)
        _, mk = ws.warp(255 * np.ones(seams[i].shape[:2], np.uint8), K, rig.Rs[i], cv.INTER_NEAREST, cv.BORDER_CONSTANT)
        cs.append(cnr); ims.append(im); mks.append(mk)
    full_comp = cv.detail.ExposureCompensator_createDefault(ctype)
    full_comp.feed(corners=cs, images=ims, masks=mks)
    w = cv.PyRotationWarper(rig.warp, rig.focal)
    rois = [w.warpRoi((rig.width, rig.height), rig.Ks[i], rig.Rs[i]) for i in range(rig.n)]
    plan = parallel.plan_strips([r[:2] for r in rois], [r[2:] for r in rois], owner, world, nb)
    dev = [cv.UMat(f) for f in frames]
    kw = dict(num_bands=nb, mask_prep=True, seam_size=rig.seam_size, seam_aspect=rig.seam_scale, want_result_s16=True)
    full = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), **kw)
    full.set_compensator(full_comp)
    full.run(dev)
    ref_mos, ref_mask, ref_res = [u.get() for u in full.result()]
    plain = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), **kw)
    plain.run(dev)
    assert not np.array_equal(plain.result()[0].get(), ref_mos)              # the gains do something
    exs, per_rank, keep = [], [], []
    for r in range(world):
        idx = [i for i in range(rig.n) if owner[i] == r]
        c = cmp.Composer(rig.warp, rig.focal, [rig.Ks[i] for i in idx], [rig.Rs[i] for i in idx], (rig.width, rig.height), **kw)
        local = parallel.subset_compensator(cv, full_comp, idx)
        c.set_compensator(local)
        keep.append(local)
        exs.append(parallel.StripExchangeBase(c, plan, r, parallel._umat_alloc))
        per_rank.append([dev[i] for i in idx])
    parallel.emulate_strip_exchange(exs, per_rank)
    own = parallel.strip_owner_map(plan)
    for r in range(world):
        mos, mk, rs = [u.get() for u in exs[r].c.result()]
        x0, y0 = plan.region[r][0], plan.region[r][1]
        hh, ww = mk.shape
        sel = own[y0:y0 + hh, x0:x0 + ww] == r
        assert int(sel.sum()) > 0
        assert np.array_equal(mk[sel], ref_mask[y0:y0 + hh, x0:x0 + ww][sel])
        assert np.array_equal(rs[sel], ref_res[y0:y0 + hh, x0:x0 + ww][sel])
        assert np.array_equal(mos[sel], ref_mos[y0:y0 + hh, x0:x0 + ww][sel])


def _dist_comp_process(rank, world, port, owner, ctype, outdir):
    import os
    import sys
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import numpy as np
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, root)
        import opencv_starry_sky_panorama_stitcher_amd as cv
        from opencv_starry_sky_panorama_stitcher_amd import parallel, starfield
        cv._lib.check(cv._lib.lib().ssp_init(0))
        rig = starfield.make_rig(3, scale_div=8, n_override=len(owner))
        frames, seams = starfield.make_frames(rig, want_seam=True)
        ws = cv.PyRotationWarper(rig.warp, rig.focal * rig.seam_scale)

        def seam_warp(i):
            K = rig.Ks[i].copy(); K[0, 0] *= rig.seam_scale; K[0, 2] *= rig.seam_scale; K[1, 1] *= rig.seam_scale; K[1, 2] *= rig.seam_scale
            cnr, im = ws.warp(seams[i], K, rig.Rs[i], cv.INTER_AREA, cv.BORDER_REFLECT)
            _, mk = ws.warp(255 * np.ones(seams[i].shape[:2], np.uint8), K, rig.Rs[i], cv.INTER_NEAREST, cv.BORDER_CONSTANT)
            return cnr, im, mk
        mine = [i for i in range(rig.n) if owner[i] == rank]
        loc = [seam_warp(i) for i in mine]                                   # a rank only warps ITS frames ...
        full, local = parallel.distributed_compensator(cv, dist, ctype, owner, [t[0] for t in loc], [t[1] for t in loc], [t[2] for t in loc])
        ref = cv.detail.ExposureCompensator_createDefault(ctype)             # ... the reference feed here sees all of them
        everything = [seam_warp(i) for i in range(rig.n)]
        ref.feed(corners=[t[0] for t in everything], images=[t[1] for t in everything], masks=[t[2] for t in everything])
        same = all(np.array_equal(a, b) for a, b in zip(full.getMatGains(), ref.getMatGains()))
        sub = all(np.array_equal(a, ref.getMatGains()[g]) for a, g in zip(local.getMatGains(), mine))
        np.save(os.path.join(outdir, f"comp_{rank}.npy"), np.array([int(same), int(sub), len(mine)]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("ctype", [1, 2])
def test_distributed_compensator_two_processes(tmp_path, ctype):
    """parallel.distributed_compensator with one process per rank: the seam-scale warps are gathered, every rank solves the same
    system and gets exactly the gains a single process computes; each keeps those of its own frames."""
    import torch.multiprocessing as mp
    owner = [0, 1, 0, 1, 1]
    port = 35500 + (os.getpid() % 2000)
    mp.spawn(_dist_comp_process, args=(2, port, owner, ctype, str(tmp_path)), nprocs=2, join=True)
    stats = [np.load(tmp_path / f"comp_{r}.npy") for r in range(2)]
    assert all(int(s[0]) == 1 and int(s[1]) == 1 for s in stats), stats
    assert sum(int(s[2]) for s in stats) == len(owner)


# ---- the reference's own known-answer tests through the HIP path -----------------------------------------------------------------
def _kat_doc():
    import json

    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat.json")))


@pytest.mark.parametrize("k", _kat_doc()["kats"], ids=lambda k: f"kat{k['id']:02d}-{k['warp']}")
def test_hip_warp_roi_on_recorded_runs(k):
    """The 34 recorded panorama sizes (tests/golden/kat.json: cameras + config + final JPEG size of the reference's runs) through
    camera.prepare_compose_cameras -> HIP PyRotationWarper.warpRoi -> HIP detail.resultRoi: the GPU mirror of
    tests/test_oracle_geometry.py::test_kat_panorama_size."""
    from opencv_starry_sky_panorama_stitcher_amd import camera as cam

    cams = cam.cameras_from_dicts(_kat_doc()["camera_sets"][k["camera_set"]])
    fw, fh = k["full_size"]
    ws = cam.scale_for_megapix(k["work_megapix"], fw, fh)
    g = cam.prepare_compose_cameras(cams, [(fw, fh)] * len(cams), ws, k["compose_megapix"], k["wave_correct"], k["mirror_pano"], k["rotate_pano_rad"])
    w = cv.PyRotationWarper(k["warp"], g.warper_scale)
    rois = [w.warpRoi(sz, K, R) for sz, K, R in zip(g.sizes, g.Ks, g.Rs)]
    pano = cv.detail.resultRoi([r[:2] for r in rois], [r[2:] for r in rois])
    assert list(pano[2:]) == k["golden_pano_size"]
    wo = ocv.PyRotationWarper(k["warp"], g.warper_scale)
    assert [tuple(r) for r in rois] == [tuple(wo.warpRoi(sz, K, R)) for sz, K, R in zip(g.sizes, g.Ks, g.Rs)]


def test_integration_md_section_2_verbatim():
    """INTEGRATION.md section 2 executed as printed: raw ctypes on libssp_hip.so with HOST pointers (ssp_warper_create / _roi / _warp),
    against the oracle."""
    import ctypes as C

    W, H = 211, 140
    img = star_patch(W, H, seed=23)
    K, R, scale = camera(W, H, 60.0, 9.0, -4.0, 2.0)
    K, R = np.ascontiguousarray(K, np.float32), np.ascontiguousarray(R, np.float32)
    lib = C.CDLL(cv._lib.LIB_PATH)
    lib.ssp_last_error.restype = C.c_char_p

    def chk(rc):
        if rc:
            raise RuntimeError(lib.ssp_last_error().decode())

    w = C.c_void_p()
    chk(lib.ssp_warper_create(b"spherical", C.c_float(scale), C.byref(w)))
    roi = (C.c_int * 4)()
    fp = C.POINTER(C.c_float)
    chk(lib.ssp_warper_roi(w, W, H, K.ctypes.data_as(fp), R.ctypes.data_as(fp), roi))
    dst = np.empty((roi[3], roi[2], 3), np.uint8)
    corner = (C.c_int * 2)()
    chk(lib.ssp_warper_warp(w, C.c_void_p(img.ctypes.data), W, H, 3, 0, K.ctypes.data_as(fp), R.ctypes.data_as(fp), 1, 2, C.c_void_p(dst.ctypes.data), roi[2], roi[3], corner))
    chk(lib.ssp_warper_destroy(w))
    o = ocv.PyRotationWarper("spherical", scale)
    assert tuple(roi) == tuple(o.warpRoi((W, H), K, R))
    co, do = o.warp(img, K, R, ocv.INTER_LINEAR, ocv.BORDER_REFLECT)
    assert tuple(corner) == tuple(co) and np.array_equal(dst, do)
    # a wrongly sized destination is an error, not a crash
    w2 = C.c_void_p()
    chk(lib.ssp_warper_create(b"spherical", C.c_float(scale), C.byref(w2)))
    bad = np.empty((roi[3] + 1, roi[2], 3), np.uint8)
    assert lib.ssp_warper_warp(w2, C.c_void_p(img.ctypes.data), W, H, 3, 0, K.ctypes.data_as(fp), R.ctypes.data_as(fp), 1, 2, C.c_void_p(bad.ctypes.data), roi[2], roi[3] + 1, corner) != 0
    chk(lib.ssp_warper_destroy(w2))


def test_feed_element_types_follow_cv2():
    """cv2 asserts CV_16SC3 in Blender / FeatherBlender::feed and takes CV_16SC3 or CV_8UC3 in MultiBandBlender::feed."""
    img8, mask = np.full((8, 8, 3), 7, np.uint8), np.full((8, 8), 255, np.uint8)
    for make in (lambda: cv.detail.Blender_createDefault(0), lambda: cv.detail_FeatherBlender(0.1)):
        b = make()
        b.prepare((0, 0, 8, 8))
        with pytest.raises(cv.error):
            b.feed(img8, mask, (0, 0))
        b.feed(img8.astype(np.int16), mask, (0, 0))
        r, k = b.blend(None, None)
        assert r.dtype == np.int16 and k.min() == 255
    b = cv.detail_MultiBandBlender(num_bands=2)
    b.prepare((0, 0, 8, 8))
    b.feed(img8, mask, (0, 0))
    r8, _ = b.blend(None, None)
    b = cv.detail_MultiBandBlender(num_bands=2)
    b.prepare((0, 0, 8, 8))
    b.feed(img8.astype(np.int16), mask, (0, 0))
    r16, _ = b.blend(None, None)
    assert np.array_equal(r8, r16)


def test_float_composer_refuses_gains():
    """cv2's compensators work on 8-bit images; a float composer must not silently drop them."""
    rig = starfield.make_rig(5, scale_div=32, n_override=3)
    comp = cv.detail.ExposureCompensator_createDefault(cv.detail.ExposureCompensator_GAIN)
    comp.setMatGains([np.array([[1.1]]) for _ in range(len(rig.Ks))])
    c = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), blend="multiband", num_bands=3, float_frames=True)
    with pytest.raises(cv.error):
        c.set_compensator(comp)
    c.set_compensator(cv.detail.ExposureCompensator_createDefault(0))   # the identity is fine
    c.set_compensator(None)


def _kernel_families_of(fn):
    """names of the kernel families fn() launches (the library's hipEvent profile)"""
    import ctypes as C
    L = cv._lib.lib()
    cv._lib.check(L.ssp_profile_reset())
    cv._lib.check(L.ssp_profile_enable(1))
    try:
        fn()
        cv._lib.check(L.ssp_sync())
    finally:
        cv._lib.check(L.ssp_profile_enable(0))
    n = C.c_int()
    cv._lib.check(L.ssp_profile_count(C.byref(n)))
    out = {}
    for i in range(n.value):
        name, launches, ms, ab = C.create_string_buffer(64), C.c_int(), C.c_float(), C.c_double()
        cv._lib.check(L.ssp_profile_get(i, name, 64, C.byref(launches), C.byref(ms), C.byref(ab)))
        out[name.value.decode()] = launches.value
    return out


def test_rest_launch_goes_away_once_the_composer_knows_its_geometry():
    """The strip warp puts tiles it cannot stage on a list for a second launch.  A composer's geometry is fixed, so after its first
    panorama it knows how many such tiles there are; when they are few the strip kernel does them inline from then on and the rest
    launch disappears -- with identical panoramas."""
    rig, frames, seams = _rig_small(3, 4, 4)
    c = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), blend=rig.blend, num_bands=4, mask_prep=True, seam_size=rig.seam_size,
                     seam_aspect=rig.seam_scale, want_result_s16=True)
    dev = [cv.UMat(f) for f in frames]
    fams, outs = [], []
    for _ in range(3):
        fams.append(_kernel_families_of(lambda: c.run(dev)))
        outs.append([u.get() for u in c.result()])
    import ctypes as C
    state, count = C.c_int(), C.c_int()
    cv._lib.check(cv._lib.lib().ssp_composer_warp_rest_tiles(c._h, C.byref(state), C.byref(count)))
    assert fams[0].get("warp_rest", 0) == 1 and fams[0].get("warp_fused", 0) == 1
    assert state.value == 2 and 0 <= count.value <= 64, (state.value, count.value)
    assert fams[2].get("warp_rest", 0) == 0 and fams[2].get("warp_fused", 0) == 1, fams
    assert fams[0].get("warp_prep", 0) == 1 and fams[2].get("warp_prep", 0) == 0, fams      # tables depend on the geometry only
    for o in outs[1:]:
        assert all(np.array_equal(a, b) for a, b in zip(o, outs[0]))
    ref = cmp.compose_panorama(ocv, frames, rig.Ks, rig.Rs, warp=rig.warp, warper_scale=rig.focal, blend=rig.blend, num_bands=4, seam_frames=seams,
                               seam_aspect=rig.seam_scale, mask_prep=True)
    assert np.array_equal(outs[2][0], ref.mosaic) and np.array_equal(outs[2][1], ref.result_mask) and np.array_equal(outs[2][2], ref.result)
    # a compensator arrives: what the composer knew is void (gain rows decide stageability, gain tables are prep output); it relearns
    comp = cv.detail.ExposureCompensator_createDefault(cv.detail.ExposureCompensator_GAIN_BLOCKS)
    ws = cv.PyRotationWarper(rig.warp, rig.focal * rig.seam_scale)
    cs, ims, mks = [], [], []
    for i in range(rig.n):
        K = rig.Ks[i].copy(); K[0, 0] *= rig.seam_scale; K[0, 2] *= rig.seam_scale; K[1, 1] *= rig.seam_scale; K[1, 2] *= rig.seam_scale   # noqa: E702
        cnr, im = ws.warp(seams[i], K, rig.Rs[i], cv.INTER_AREA, cv.BORDER_REFLECT)
        _, mk = ws.warp(255 * np.ones(seams[i].shape[:2], np.uint8), K, rig.Rs[i], cv.INTER_NEAREST, cv.BORDER_CONSTANT)
        cs.append(cnr); ims.append(im); mks.append(mk)                                                                                     # noqa: E702
    comp.feed(corners=cs, images=ims, masks=mks)
    c.set_compensator(comp)
    gfams, gouts = [], []
    for _ in range(4):
        gfams.append(_kernel_families_of(lambda: c.run(dev)))
        gouts.append([u.get() for u in c.result()])
    assert gfams[0].get("warp_prep", 0) == 1 and gfams[0].get("warp_rest", 0) == 1 and gfams[3].get("warp_prep", 0) == 0 and gfams[3].get("warp_rest", 0) == 0, gfams
    assert all(np.array_equal(a, b) for o in gouts[1:] for a, b in zip(o, gouts[0]))
    assert not np.array_equal(gouts[0][0], outs[0][0])                                        # the gains did something


def test_a_fresh_composers_rest_count_stays_inside_the_launch():
    """The guard behind the round-2 fault (gpurun_out/r2h: the list counter started from pool garbage): after a fresh composer's first
    panorama the device-side count of non-stageable tiles lies in [0, tiles of the launch]; a count beyond the list's capacity is an
    error (SSP_ERR_STATE from ssp_composer_warp_rest_tiles / the next run), never a silently clamped list."""
    for config, div, n in ((3, 4, 4), (2, 8, 3)):
        rig, frames, seams = _rig_small(config, div, n)
        c = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), blend=rig.blend, num_bands=4, mask_prep=True, seam_size=rig.seam_size,
                         seam_aspect=rig.seam_scale)
        assert c.warp_rest_tiles() == (0, -1)
        c.run([cv.UMat(f) for f in frames])
        state, count = c.warp_rest_tiles()                 # waits for the read-back
        rois = [c.image_roi(i) for i in range(rig.n)]
        tiles = rig.n * ((max(r[2] for r in rois) + 3 + 63) // 64) * ((max(r[3] for r in rois) + 15) // 16)
        assert state == 2 and 0 <= count <= tiles, (state, count, tiles)


def test_forget_geometry_rebuilds_tables_and_list_with_identical_panoramas():
    """ssp_composer_forget_geometry: the prep launch (tables) and the rest launch (list) run again on the next panorama -- bench.py's
    `tables_rebuilt`, the like-for-like cost against cv2's per-call buildMaps -- and the panorama does not change."""
    rig, frames, seams = _rig_small(3, 4, 4)
    c = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), blend=rig.blend, num_bands=4, mask_prep=True, seam_size=rig.seam_size,
                     seam_aspect=rig.seam_scale, want_result_s16=True)
    dev = [cv.UMat(f) for f in frames]
    for _ in range(3):
        c.run(dev)
    want = [u.get() for u in c.result()]
    fam_steady = _kernel_families_of(lambda: c.run(dev))
    c.forget_geometry()
    fam_rebuilt = _kernel_families_of(lambda: c.run(dev))
    got = [u.get() for u in c.result()]
    assert fam_steady.get("warp_prep", 0) == 0 and fam_steady.get("warp_rest", 0) == 0
    assert fam_rebuilt.get("warp_prep", 0) == 1 and fam_rebuilt.get("warp_rest", 0) == 1, fam_rebuilt
    assert all(np.array_equal(a, b) for a, b in zip(got, want))


def test_composer_fed_by_wrapped_tight_pitch_frames_at_an_odd_address():
    """frames borrowed with UMat.wrap_device (tight pitch 3 w, any base -- e.g. torch tensors) used to take the LDS-staged warp,
    whose 16-byte chunk loads assume a 16-byte aligned base and pitch: with 3 w % 4 != 0 the last row's final chunk left the buffer range and
    came back as zeros.  They are repacked into pool images for the call now.  485 px wide (pitch 1455), base address odd."""
    from opencv_starry_sky_panorama_stitcher_amd.starfield import Rig, _finish, _ring
    rig = _finish(Rig("odd width", 2, 485, 270, 60.0, _ring(3, 45.0), [0.0] * 3, "spherical", "multiband", 5))
    frames = starfield.make_frames(rig)
    c = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), blend="multiband", num_bands=5, want_result_s16=True)
    c.run([cv.UMat(f) for f in frames])
    want = [u.get() for u in c.result()]
    keep, wrapped = [], []
    for f in frames:
        # a caller's linear device buffer holding the frame one byte in: a one-row image is contiguous in the pool
        flat = np.concatenate([np.zeros(1, np.uint8), np.ascontiguousarray(f).reshape(-1), np.full(15, 255, np.uint8)])
        buf = cv.UMat(flat[None, :])
        keep.append(buf)
        wrapped.append(cv.UMat.wrap_device(buf.info()[5] + 1, rig.width, rig.height, 3, np.uint8))
        assert wrapped[-1].info()[4] == 3 * rig.width and (buf.info()[5] + 1) % 2 == 1
        assert np.array_equal(wrapped[-1].get(), f)
    c.run(wrapped)
    got = [u.get() for u in c.result()]
    assert all(np.array_equal(a, b) for a, b in zip(got, want))
    ref = cmp.compose_panorama(ocv, frames, rig.Ks, rig.Rs, warp=rig.warp, warper_scale=rig.focal, blend="multiband", num_bands=5)
    assert np.array_equal(got[0], ref.mosaic)
    del keep


def test_stored_rest_list_when_the_geometry_leaves_many_rest_tiles():
    """Two rows of frames at +-28 degrees pitch: many tiles have footprints too large to stage, so the composer keeps the list
    mode -- but the list of its first panorama is stored and walked again (no re-listing, no counter to zero, so no prep launch
    either).  Panoramas identical to the first one and to the oracle."""
    from opencv_starry_sky_panorama_stitcher_amd.starfield import Rig, _finish
    import ctypes as C
    yaws = [-26.0, 0.0, 26.0, -26.0, 0.0, 26.0]
    pitches = [28.0, 28.0, 28.0, -28.0, -28.0, -28.0]
    rig = _finish(Rig("two rows", 9, 960, 540, 60.0, yaws, pitches, "spherical", "multiband", 4))
    frames, seams = starfield.make_frames(rig, want_seam=True)
    c = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), blend="multiband", num_bands=4, mask_prep=True, seam_size=rig.seam_size,
                     seam_aspect=rig.seam_scale, want_result_s16=True)
    dev = [cv.UMat(f) for f in frames]
    fams, outs = [], []
    for _ in range(4):
        fams.append(_kernel_families_of(lambda: c.run(dev)))
        outs.append([u.get() for u in c.result()])
    state, count = C.c_int(), C.c_int()
    cv._lib.check(cv._lib.lib().ssp_composer_warp_rest_tiles(c._h, C.byref(state), C.byref(count)))
    assert state.value == 2 and count.value > 64, (state.value, count.value)
    assert fams[0].get("warp_prep", 0) == 1 and fams[0].get("warp_rest", 0) == 1
    assert fams[3].get("warp_prep", 0) == 0 and fams[3].get("warp_rest", 0) == 1, fams
    assert all(np.array_equal(a, b) for o in outs[1:] for a, b in zip(o, outs[0]))
    ref = cmp.compose_panorama(ocv, frames, rig.Ks, rig.Rs, warp=rig.warp, warper_scale=rig.focal, blend="multiband", num_bands=4, seam_frames=seams,
                               seam_aspect=rig.seam_scale, mask_prep=True)
    assert np.array_equal(outs[3][0], ref.mosaic) and np.array_equal(outs[3][1], ref.result_mask) and np.array_equal(outs[3][2], ref.result)
