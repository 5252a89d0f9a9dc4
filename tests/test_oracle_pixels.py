"""CPU tests of the oracle's pixel arithmetic: (a) against the independent numpy restatement (oracle/np_ref.py),
(b) analytic properties, (c) against the committed golden vectors (tests/golden/pixels.npz), (d) glibc vs the
deterministic transcendentals."""
import os
import sys

import numpy as np
import pytest

import oracle_cv as ocv
from util import camera, star_patch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
import np_ref  # noqa: E402

orc = ocv.orc
GOLD = np.load(os.path.join(HERE, "golden", "pixels.npz"))


# ---- (a) C oracle vs numpy restatement --------------------------------------------------------------------------------
@pytest.mark.parametrize("interp", [0, 1])
@pytest.mark.parametrize("border", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("cn", [1, 3])
def test_remap_u8_matches_numpy(interp, border, cn):
    rng = np.random.default_rng(interp * 10 + border)
    src = star_patch(37, 29, seed=3, cn=cn)
    xm = rng.uniform(-60, 100, (41, 53)).astype(np.float32)
    ym = rng.uniform(-50, 80, (41, 53)).astype(np.float32)
    xm[0, :5] = [-1, 36.0, 36.5, 35.984375, 1e9]       # behind-camera marker, last column, exact ties, overflow
    ym[0, :5] = [-1, 28.0, 27.5, 0.015625, -1e9]
    xm[1, 0], ym[1, 0] = np.nan, 3.0
    assert np.array_equal(orc.remap(src, xm, ym, interp, border), np_ref.remap_u8(src, xm, ym, interp, border))


@pytest.mark.parametrize("shape", [(40, 56), (33, 47), (2, 2), (5, 3), (64, 1)])
def test_pyr_down_matches_numpy(shape):
    rng = np.random.default_rng(1)
    a = rng.integers(-3000, 3000, shape + (3,)).astype(np.int16)
    assert np.array_equal(orc.pyrDown(a), np_ref.pyr_down(a))
    f = rng.uniform(0, 1, shape).astype(np.float32)
    assert np.array_equal(orc.pyrDown(f).view(np.uint32), np_ref.pyr_down(f).view(np.uint32))


@pytest.mark.parametrize("shape", [(20, 28), (17, 9), (1, 6), (7, 1), (2, 2)])
def test_pyr_up_matches_numpy(shape):
    rng = np.random.default_rng(2)
    a = rng.integers(-3000, 3000, shape + (3,)).astype(np.int16)
    assert np.array_equal(orc.pyrUp(a), np_ref.pyr_up(a))


@pytest.mark.parametrize("bands", [1, 3, 5])
def test_multiband_matches_numpy(bands):
    rng = np.random.default_rng(bands)
    imgs, masks, tls = [], [], [(-20, 4), (31, -3), (70, 6)]
    for i in range(3):
        imgs.append((star_patch(61, 43, seed=bands * 7 + i).astype(np.int32) + rng.integers(-30, 30, (43, 61, 3))).astype(np.int16))
        mk = np.zeros((43, 61), np.uint8)
        mk[2:-4 - i, 3 + i:-2] = 255
        mk[rng.integers(0, 43, 15), rng.integers(0, 61, 15)] = rng.integers(0, 256, 15)
        masks.append(mk)
    roi = orc.resultRoi(tls, [(61, 43)] * 3)
    b = ocv.detail_MultiBandBlender(num_bands=bands)
    b.prepare(roi)
    for im, mk, tl in zip(imgs, masks, tls):
        b.feed(im, mk, tl)
    r, k = b.blend(None, None)
    r2, k2 = np_ref.multiband_blend(imgs, masks, tls, roi, bands)
    assert np.array_equal(k, k2) and np.array_equal(r, r2)


def test_helpers_match_numpy():
    rng = np.random.default_rng(5)
    m = (rng.uniform(size=(19, 26)) > 0.5).astype(np.uint8) * 255
    assert np.array_equal(orc.dilate(m), np_ref.dilate3(m))
    for dsize in [(61, 40), (26, 19), (27, 20), (100, 21), (13, 10)]:
        assert np.array_equal(orc.resize_linear_exact(m, dsize), np_ref.resize_linear_exact(m, dsize)), dsize
    d = orc.distance_l1(m)
    assert np.array_equal(d, np_ref.distance_l1(m))
    assert np.all(orc.distance_l1(255 * np.ones((5, 7), np.uint8)) == 65534.0)  # no zero pixel: the image border is not a zero


# ---- (b) analytic properties ---------------------------------------------------------------------------------------------
def test_remap_identity_and_half_pixel():
    src = star_patch(40, 30, seed=9)
    yy, xx = np.mgrid[0:30, 0:40].astype(np.float32)
    assert np.array_equal(orc.remap(src, xx, yy, 1, 2), src)
    assert np.array_equal(orc.remap(src, xx, yy, 0, 0), src)
    half = orc.remap(src, xx[:, :-1] + 0.5, yy[:, :-1], 1, 2).astype(np.int32)
    avg = (src[:, :-1].astype(np.int32) + src[:, 1:].astype(np.int32) + 1) >> 1   # (a*16384 + b*16384 + 16384) >> 15
    assert np.array_equal(half, avg)


def test_pyramids_preserve_constants_and_sizes():
    c = np.full((33, 47, 3), 123, np.int16)
    d = orc.pyrDown(c)
    assert d.shape == (17, 24, 3) and np.all(d == 123)
    assert np.all(orc.pyrUp(d) == 123)
    f = np.full((20, 20), 0.75, np.float32)
    assert np.all(orc.pyrDown(f) == np.float32(0.75))


def test_single_image_multiband_reconstructs_within_truncation():
    """One image, full mask: the collapsed pyramid returns the image up to the truncating casts (<= bands + 1)."""
    img = star_patch(96, 64, seed=2).astype(np.int16)
    mk = 255 * np.ones((64, 96), np.uint8)
    b = ocv.detail_MultiBandBlender(num_bands=4)
    b.prepare((10, 20, 96, 64))
    b.feed(img, mk, (10, 20))
    r, k = b.blend(None, None)
    assert np.all(k == 255)
    assert np.abs(r.astype(np.int32) - img).max() <= 5


def test_no_blender_last_writer_wins_and_feather_weights():
    a = np.full((10, 10, 3), 50, np.int16)
    bimg = np.full((10, 10, 3), 90, np.int16)
    mk = 255 * np.ones((10, 10), np.uint8)
    nb = ocv.detail.Blender_createDefault(0)
    nb.prepare((0, 0, 15, 10))
    nb.feed(a, mk, (0, 0))
    nb.feed(bimg, mk, (5, 0))
    r, k = nb.blend(None, None)
    assert np.all(r[:, :5] == 50) and np.all(r[:, 5:] == 90) and np.all(k == 255)
    fb = ocv.detail_FeatherBlender(1.0)   # sharpness 1: weight = min(1, L1 distance)
    fb.prepare((0, 0, 10, 10))
    m2 = mk.copy()
    m2[:, 0] = 0
    fb.feed(a, m2, (0, 0))
    r, k = fb.blend(None, None)
    assert np.all(k[:, 0] == 0) and np.all(k[:, 1:] == 255) and np.all(r[:, 1:] == 49)  # (short)(50/(1+1e-5)) = 49


def _gain_solve_precision_bound(A, bvec, g, images, mk):
    """test_gain_solve_precision_bound (oracle/orc_comp.c, open item [CV-U]): an OpenCV built with HAVE_EIGEN solves the same
    system with a single-precision Cholesky.  Its gains differ from the double LU ones in the 7th digit and apply() by <= 1 grey
    level, on few pixels."""
    A32, b32 = A.astype(np.float32), bvec.astype(np.float32)
    L = np.zeros_like(A32)
    for i in range(len(b32)):                       # float32 LLT, operation by operation
        for j in range(i + 1):
            sacc = np.float32(0)
            for k in range(j):
                sacc = np.float32(sacc + np.float32(L[i, k] * L[j, k]))
            L[i, j] = np.float32(np.sqrt(np.float32(A32[i, i] - sacc))) if i == j else np.float32(np.float32(A32[i, j] - sacc) / L[j, j])
    y = np.zeros_like(b32)
    for i in range(len(b32)):
        y[i] = np.float32((b32[i] - np.float32(np.dot(L[i, :i], y[:i]))) / L[i, i])
    x = np.zeros_like(b32)
    for i in reversed(range(len(b32))):
        x[i] = np.float32((y[i] - np.float32(np.dot(L[i + 1:, i], x[i + 1:]))) / L[i, i])
    assert np.max(np.abs(x.astype(np.float64) / g - 1)) < 5e-6
    for k, im in enumerate(images):              # apply(): multiply(image, gain, image) on 8-bit pixels
        ia = np.clip(np.rint(im.astype(np.float64) * g[k]), 0, 255)
        ib = np.clip(np.rint(im.astype(np.float64) * float(x[k])), 0, 255)
        d = np.abs(ia - ib)
        assert d.max() <= 1 and np.count_nonzero(d) <= 1e-3 * d.size


def test_gain_compensator_recovers_exposure_ratio():
    base = star_patch(140, 60, seed=4, n_stars=60).astype(np.float32) + 40
    i0 = np.clip(np.rint(base[:, :90]), 0, 255).astype(np.uint8)
    i1 = np.clip(np.rint(base[:, 50:] * 1.2), 0, 255).astype(np.uint8)
    mk = 255 * np.ones((60, 90), np.uint8)
    c = ocv.detail.ExposureCompensator_createDefault(1)
    c.feed([(0, 0), (50, 0)], [i0, i1], [mk, mk])
    g = c.gains()
    # Brown-Lowe gains with the prior (alpha = 0.01, beta = 100): restate the normal equations in numpy and compare
    ov0, ov1 = i0[:, 50:].astype(np.float64), i1[:, :40].astype(np.float64)
    N = np.array([[i0.shape[0] * i0.shape[1], ov0.shape[0] * ov0.shape[1]], [ov0.shape[0] * ov0.shape[1], i1.shape[0] * i1.shape[1]]], float)
    I = np.array([[np.sqrt((i0.astype(np.float64) ** 2).sum(2)).mean(), np.sqrt((ov0 ** 2).sum(2)).mean()],
                  [np.sqrt((ov1 ** 2).sum(2)).mean(), np.sqrt((i1.astype(np.float64) ** 2).sum(2)).mean()]])
    A, bvec = np.zeros((2, 2)), np.zeros(2)
    for i in range(2):
        for j in range(2):
            bvec[i] += 100 * N[i, j]
            A[i, i] += 100 * N[i, j]
            if i != j:
                A[i, i] += 2 * 0.01 * I[i, j] ** 2 * N[i, j]
                A[i, j] -= 2 * 0.01 * I[i, j] * I[j, i] * N[i, j]
    assert np.allclose(g, np.linalg.solve(A, bvec), rtol=1e-10)
    assert 1.05 < g[0] / g[1] < 1.2   # pulled towards 1 by the prior, but in the right direction
    _gain_solve_precision_bound(A, bvec, g, [i0, i1], mk)
    blk = ocv._Comp(2, 16, 16, 1, 2)
    blk.feed([(0, 0), (50, 0)], [i0, i1], [mk, mk])
    assert blk.gainMap(0).shape == (4, 6, 1)


# ---- (c) committed golden vectors ------------------------------------------------------------------------------------------
def test_oracle_reproduces_golden_vectors():
    img = GOLD["img"]
    K, R, f = GOLD["cam_K"], GOLD["cam_R"], float(GOLD["cam_f"])
    for warp in ("spherical", "cylindrical", "fisheye", "plane", "paniniA2B1", "transverseMercator"):
        o = ocv.PyRotationWarper(warp, f)
        c, d = o.warp(img, K, R, ocv.INTER_LINEAR, ocv.BORDER_REFLECT)
        _, m = o.warp(255 * np.ones(img.shape[:2], np.uint8), K, R, ocv.INTER_NEAREST, ocv.BORDER_CONSTANT)
        assert tuple(c) == tuple(GOLD[f"warp_{warp}_corner"])
        assert np.array_equal(d, GOLD[f"warp_{warp}_img"]) and np.array_equal(m, GOLD[f"warp_{warp}_mask"])
    roi = tuple(int(v) for v in GOLD["blend_roi"])
    for name, make in (("mb3", lambda: ocv.detail_MultiBandBlender(num_bands=3)), ("feather", lambda: ocv.detail_FeatherBlender(0.08)),
                       ("no", lambda: ocv.detail.Blender_createDefault(0))):
        b = make()
        b.prepare(roi)
        for im, mk, tl in zip(GOLD["blend_imgs"], GOLD["blend_masks"], GOLD["blend_tls"]):
            b.feed(im, mk, tuple(int(v) for v in tl))
        r, k = b.blend(None, None)
        assert np.array_equal(r, GOLD[f"blend_{name}_result"]) and np.array_equal(k, GOLD[f"blend_{name}_mask"])
    assert np.array_equal(ocv.dilate(GOLD["mask_small"], None), GOLD["mask_dilated"])
    assert np.array_equal(ocv.resize(GOLD["mask_small"], (77, 52), 0, 0, 5), GOLD["mask_resized_77x52"])


# ---- (d) distance between "our spec" and a glibc-linked OpenCV build ---------------------------------------------------------
@pytest.mark.parametrize("warp", ["spherical", "fisheye", "mercator", "paniniA2B1"])
def test_libm_build_differs_by_at_most_one_ulp_and_rarely_in_pixels(warp):
    """The oracle built with glibc's sinf/cosf/atan2f/... (what a glibc-linked OpenCV calls) against the deterministic
    correctly-rounded functions: maps agree within a few float32 ULP; that moves the 1/32-pixel quantisation for a small
    fraction of the pixels only.  This bounds how far 'our spec' is from such an OpenCV build."""
    w, h = 320, 200
    img = star_patch(w, h, seed=13)
    K, R, f = camera(w, h, 60.0, 10.0, -4.0, 2.0)
    a = orc.PyRotationWarper(warp, f, libm=False)
    b = orc.PyRotationWarper(warp, f, libm=True)
    ra, xa, ya = a.buildMaps((w, h), K, R)
    rb, xb, yb = b.buildMaps((w, h), K, R)
    assert ra == rb
    inside = (xa > 0) & (xa < w) & (ya > 0) & (ya < h)
    assert np.abs(xa - xb)[inside].max() < 2e-3 and np.abs(ya - yb)[inside].max() < 2e-3   # << 1/32 pixel
    _, da = a.warp(img, K, R, 1, 2)
    _, db = b.warp(img, K, R, 1, 2)
    diff = np.abs(da.astype(np.int32) - db.astype(np.int32))
    assert (diff > 0).mean() < 0.02 and diff.max() <= 12


# ---- frame prologue (sde.py:1699-1711): INTER_AREA decimation + black / white point stretch ---------------------------------
@pytest.mark.parametrize("f", [0.37, 0.1829, 0.5, 0.25, 1.0 / 3.0, 0.9])
def test_resize_area_matches_numpy(f):
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, size=(47, 61, 3), dtype=np.uint8)
    assert np.array_equal(orc.resize_area(img, f, f), np_ref.resize_area(img, f, f)), f
    g = img[:, :, 0].copy()
    assert np.array_equal(orc.resize_area(g, f, f * 0.8), np_ref.resize_area(g, f, f * 0.8)), f


def test_resize_area_properties():
    # a constant image stays constant; the mean is preserved up to rounding; an exact 2x2 block average rounds half up
    c = np.full((40, 52, 3), 137, np.uint8)
    assert np.all(orc.resize_area(c, 0.31, 0.31) == 137)
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, size=(120, 160, 3), dtype=np.uint8)
    small = orc.resize_area(img, 0.2, 0.2)
    assert small.shape == (24, 32, 3)
    assert abs(float(small.mean()) - float(img.mean())) < 0.6
    q = np.array([[1, 0], [0, 1]], np.uint8)          # sum 2 -> (2 + 2) >> 2 = 1, not round-half-even 0
    assert orc.resize_area(q, 0.5, 0.5)[0, 0] == 1
    # dsize follows cvRound(size * f): 7 * 0.5 = 3.5 -> 4, the last cell averages the one column that exists
    r = orc.resize_area(np.arange(14, dtype=np.uint8).reshape(2, 7), 0.5, 0.5)
    assert r.shape == (1, 4) and r[0, 3] == np_ref.cv_round(np.float32(6 + 13) / np.float32(2))


@pytest.mark.parametrize("tpl", [(0, 150), (10, 200), (30, 90), (0, 255), None])
def test_black_and_white_point_matches_reference_expression(tpl):
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, size=(33, 41, 3), dtype=np.uint8)
    want = np_ref.adjust_black_and_white_point(img, tpl)
    got = ocv.adjust_black_and_white_point(img, tpl)
    assert np.array_equal(got, want)
    if tpl:
        lut = orc.bw_point_lut(*tpl)
        assert lut[tpl[0]] == 0 and lut[tpl[1]] == 255 and np.all(np.diff(lut.astype(int)) >= 0)
        # fused prologue == the two steps of the reference one after the other
        assert np.array_equal(orc.resize_area(img, 0.4, 0.4, tpl), np_ref.adjust_black_and_white_point(np_ref.resize_area(img, 0.4, 0.4), tpl))


# ---- seam finder and timelapser ------------------------------------------------------------------------
def _seam_case(seed, n=4):
    rng = np.random.default_rng(seed)
    masks, corners = [], []
    for i in range(n):
        w, h = int(rng.integers(40, 90)), int(rng.integers(30, 70))
        m = np.zeros((h, w), np.uint8)
        m[int(rng.integers(0, 6)):h - int(rng.integers(0, 6)), int(rng.integers(0, 6)):w - int(rng.integers(0, 6))] = 255
        m[rng.integers(0, h, 12), rng.integers(0, w, 12)] = 0          # holes
        masks.append(m)
        corners.append((int(rng.integers(-20, 60)), int(rng.integers(-15, 40))))
    return corners, masks


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_voronoi_seam_matches_scipy_restatement(seed):
    corners, masks = _seam_case(seed)
    got = ocv.detail.SeamFinder_createDefault(ocv.detail.SeamFinder_VORONOI_SEAM).find(None, corners, masks)
    want = np_ref.seam_voronoi(corners, masks)
    assert all(np.array_equal(a, b) for a, b in zip(got, want))
    # properties: masks only lose pixels; where two images' rectangles overlap at most one keeps a pixel ...
    for a, m in zip(got, masks):
        assert np.all(a <= m)
    x0 = min(c[0] for c in corners); y0 = min(c[1] for c in corners)
    x1 = max(c[0] + m.shape[1] for c, m in zip(corners, masks)); y1 = max(c[1] + m.shape[0] for c, m in zip(corners, masks))
    cover = np.zeros((y1 - y0, x1 - x0), int)
    before = np.zeros_like(cover)
    for (cx, cy), a, m in zip(corners, got, masks):
        cover[cy - y0:cy - y0 + a.shape[0], cx - x0:cx - x0 + a.shape[1]] += a != 0
        before[cy - y0:cy - y0 + a.shape[0], cx - x0:cx - x0 + a.shape[1]] += m != 0
    assert cover.max() <= 1
    # ... and no pixel that some image covered with a valid pixel alone is lost
    assert np.all(cover[before == 1] == 1)
    assert ocv.detail.SeamFinder_createDefault(ocv.detail.SeamFinder_NO).find(None, corners, masks)[0] is not None


def test_voronoi_seam_symmetric_pair_splits_in_the_middle():
    a = 255 * np.ones((20, 40), np.uint8)
    b = 255 * np.ones((20, 40), np.uint8)
    ga, gb = ocv.detail.SeamFinder_createDefault(1).find(None, [(0, 0), (20, 0)], [a, b])
    # overlap columns 20..39 of a: ties (dist1 == dist2) go to the second image
    assert np.all(ga[:, :20] == 255) and np.all(gb[:, 20:] == 255)
    assert int((ga[:, 20:] != 0).sum(axis=1)[0]) + int((gb[:, :20] != 0).sum(axis=1)[0]) == 20


@pytest.mark.parametrize("kind", [0, 1])
def test_timelapser_semantics(kind):
    corners, sizes = [(-5, 3), (20, -4), (8, 10)], [(40, 30), (35, 32), (50, 28)]
    t = ocv.detail.Timelapser_createDefault(kind)
    t.initialize(corners, sizes)
    rng = np.random.default_rng(kind)
    x0, y0, w, h = t.roi
    if kind == 0:
        assert (x0, y0, w, h) == ocv.detail.resultRoi(corners, sizes)
    else:
        assert (x0, y0, w, h) == (20, 10, 35 - 20, 28 - 10)
    for (cx, cy), (sw, sh) in zip(corners, sizes):
        img = rng.integers(-300, 300, size=(sh, sw, 3)).astype(np.int16)
        t.process(img, None, (cx, cy))
        d = t.getDst()
        canvas = np.zeros((h, w, 3), np.int16)
        for y in range(sh):
            for x in range(sw):
                X, Y = cx + x - x0, cy + y - y0
                if 0 <= X < w and 0 <= Y < h:
                    canvas[Y, X] = img[y, x]
        assert np.array_equal(d, canvas)


def test_openmp_flavour_is_bit_identical():
    """liborc_omp.so (row loops in parallel: bench.py's multi-core CPU baseline) against the serial oracle on a compose-loop sample:
    warp + mask, mask preparation, multiband feed and blend, feather, INTER_AREA."""
    os.environ.setdefault("OMP_NUM_THREADS", "4")
    K, R, f = camera(200, 140, 60.0, yaw=12.0, pitch=-5.0)
    src = star_patch(200, 140, seed=21)
    seam = (np.random.default_rng(2).uniform(size=(35, 50)) > 0.3).astype(np.uint8) * 255

    def run():
        w = orc.PyRotationWarper("spherical", float(f))
        c, im = w.warp(src, K, R, 1, 2)
        _, mk = w.warp(255 * np.ones(src.shape[:2], np.uint8), K, R, 0, 0)
        up = orc.resize_linear_exact(orc.dilate(seam), (mk.shape[1], mk.shape[0])) & mk
        outs = [im, mk, up, orc.resize_area(src, 0.37, 0.37)]
        for make in (lambda: ocv.detail_MultiBandBlender(num_bands=4), lambda: ocv.detail_FeatherBlender(0.05)):
            b = make()
            b.prepare((c[0] - 30, c[1] - 10, im.shape[1] + 70, im.shape[0] + 25))
            b.feed(im.astype(np.int16), up, c)
            b.feed(im[::-1].astype(np.int16).copy(), mk, (c[0] + 40, c[1] + 9))
            outs += list(b.blend(None, None))
        return outs
    serial = run()
    orc.use_openmp(True)
    try:
        parallel = run()
    finally:
        orc.use_openmp(False)
    assert len(serial) == len(parallel) and all(np.array_equal(a, b) for a, b in zip(serial, parallel))
