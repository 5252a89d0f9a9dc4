"""Optional pin of the oracle against a real OpenCV ("if cv2 happens to exist ... an auto-skipping test diffs
against it; nothing depends on that").  opencv-python is NOT part of this image, so the whole module skips here and on the GPU box;
on a machine that has cv2 (ideally 4.6.0, the version the reference pins) it compares the oracle with the library the reference calls,
through the same calls compose_imgs_to_panorama makes (sde.py:1545-1930).

Integer paths (remap fixed point, pyramids, blenders, dilate, resize, INTER_AREA) are compared bit for bit.  The warp geometry goes
through float32 transcendentals: the oracle's glibc flavour (liborc_libm.so) is what a glibc-linked OpenCV computes, rois must agree
exactly there and warped pixels may differ on the rare pixel whose 1/32-pixel coordinate rounds differently (bounded below)."""
import numpy as np
import pytest

cv2 = pytest.importorskip("cv2")

import oracle_cv as ocv  # noqa: E402
from util import camera, star_patch  # noqa: E402

orc = ocv.orc

WARPERS = ["plane", "cylindrical", "spherical", "fisheye", "stereographic", "compressedPlaneA2B1", "compressedPlaneA1.5B1",
           "compressedPlanePortraitA2B1", "compressedPlanePortraitA1.5B1", "paniniA2B1", "paniniA1.5B1", "paniniPortraitA2B1",
           "paniniPortraitA1.5B1", "mercator", "transverseMercator"]


@pytest.mark.parametrize("name", WARPERS)
def test_warp_roi_and_pixels(name):
    w, h = 160, 120
    K, R, f = camera(w, h, 60.0, yaw=17.0, pitch=-6.0, roll=3.0)
    src = star_patch(w, h, seed=5)
    theirs = cv2.PyRotationWarper(name, float(f))
    ours = orc.PyRotationWarper(name, float(f), libm=True)
    assert tuple(theirs.warpRoi((w, h), K, R)) == tuple(ours.warpRoi((w, h), K, R))
    c1, im1 = theirs.warp(src, K, R, cv2.INTER_LINEAR, cv2.BORDER_REFLECT)
    c2, im2 = ours.warp(src, K, R, cv2.INTER_LINEAR, cv2.BORDER_REFLECT)
    assert tuple(c1) == tuple(c2) and im1.shape == im2.shape
    assert (im1 != im2).any(axis=2).mean() < 1e-3
    _, m1 = theirs.warp(255 * np.ones((h, w), np.uint8), K, R, cv2.INTER_NEAREST, cv2.BORDER_CONSTANT)
    _, m2 = ours.warp(255 * np.ones((h, w), np.uint8), K, R, cv2.INTER_NEAREST, cv2.BORDER_CONSTANT)
    assert (m1 != m2).mean() < 1e-3


@pytest.mark.parametrize("border", [cv2.BORDER_CONSTANT, cv2.BORDER_REPLICATE, cv2.BORDER_REFLECT, cv2.BORDER_WRAP, cv2.BORDER_REFLECT_101])
@pytest.mark.parametrize("interp", [cv2.INTER_NEAREST, cv2.INTER_LINEAR])
def test_remap(interp, border):
    rng = np.random.default_rng(interp * 10 + border)
    src = star_patch(37, 29, seed=3)
    xm = rng.uniform(-60, 100, (41, 53)).astype(np.float32)
    ym = rng.uniform(-50, 80, (41, 53)).astype(np.float32)
    assert np.array_equal(cv2.remap(src, xm, ym, interp, borderMode=border), orc.remap(src, xm, ym, interp, border))
    srcf = src.astype(np.float32)
    a, b = cv2.remap(srcf, xm, ym, interp, borderMode=border), orc.remap(srcf, xm, ym, interp, border)
    assert np.max(np.abs(a - b)) <= 1e-4


def test_pyramids_dilate_resize():
    rng = np.random.default_rng(1)
    a = rng.integers(-3000, 3000, (41, 57, 3)).astype(np.int16)
    assert np.array_equal(cv2.pyrDown(a), orc.pyrDown(a))
    assert np.array_equal(cv2.pyrUp(a), orc.pyrUp(a))
    m = (rng.uniform(0, 1, (31, 45)) > 0.6).astype(np.uint8) * 255
    assert np.array_equal(cv2.dilate(m, None), orc.dilate(m))
    assert np.array_equal(cv2.resize(m, (301, 207), interpolation=cv2.INTER_LINEAR_EXACT), orc.resize_linear_exact(m, (301, 207)))
    img = star_patch(400, 300, seed=2)
    for s in (0.5, 0.37, 0.25):
        assert np.array_equal(cv2.resize(img, None, fx=s, fy=s, interpolation=cv2.INTER_AREA), orc.resize_area(img, s, s))


def _three(seed, dtype=np.int16):
    rng = np.random.default_rng(seed)
    imgs, masks, tls = [], [], [(-20, 4), (31, -3), (70, 6)]
    for i in range(3):
        imgs.append((star_patch(61, 43, seed=seed * 7 + i).astype(np.int32) + rng.integers(-30, 30, (43, 61, 3))).astype(dtype))
        mk = np.zeros((43, 61), np.uint8)
        mk[2:-4 - i, 3 + i:-2] = 255
        mk[rng.integers(0, 43, 15), rng.integers(0, 61, 15)] = rng.integers(0, 256, 15)
        masks.append(mk)
    return imgs, masks, tls


@pytest.mark.parametrize("kind", ["no", "feather", "multiband3", "multiband5"])
def test_blenders(kind):
    imgs, masks, tls = _three(4)
    sizes = [(m.shape[1], m.shape[0]) for m in masks]
    roi = cv2.detail.resultRoi(corners=tls, sizes=sizes)
    assert tuple(roi) == tuple(ocv.detail.resultRoi(tls, sizes))
    if kind == "no":
        a, b = cv2.detail.Blender_createDefault(cv2.detail.Blender_NO), ocv.detail.Blender_createDefault(0)
    elif kind == "feather":
        a, b = cv2.detail_FeatherBlender(0.1), ocv.detail_FeatherBlender(0.1)
    else:
        nb = int(kind[-1])
        a, b = cv2.detail_MultiBandBlender(0, nb), ocv.detail_MultiBandBlender(num_bands=nb)
    a.prepare(roi); b.prepare(roi)
    for im, mk, tl in zip(imgs, masks, tls):
        a.feed(im, mk, tl); b.feed(im, mk, tl)
    r1, k1 = a.blend(None, None)
    r2, k2 = b.blend(None, None)
    assert np.array_equal(k1, k2) and np.array_equal(r1, r2)


@pytest.mark.parametrize("ctype", [1, 2, 3, 4])
def test_compensators(ctype):
    imgs, masks, tls = _three(9, np.uint8)
    masks = [np.where(m > 0, 255, 0).astype(np.uint8) for m in masks]
    imgs = [np.clip(im.astype(np.float32) * g, 0, 255).astype(np.uint8) for im, g in zip(imgs, (0.8, 1.0, 1.25))]
    a = cv2.detail.ExposureCompensator_createDefault(ctype)
    b = ocv.detail.ExposureCompensator_createDefault(ctype)
    a.feed(corners=tls, images=[cv2.UMat(i) for i in imgs], masks=[cv2.UMat(m) for m in masks])
    b.feed(corners=tls, images=imgs, masks=masks)
    for i in range(3):
        x, y = imgs[i].copy(), imgs[i].copy()
        a.apply(i, tls[i], x, masks[i]); b.apply(i, tls[i], y, masks[i])
        d = np.abs(x.astype(np.int16) - y.astype(np.int16))
        assert d.max() <= 1 and (d > 0).mean() < 1e-3      # gains are double sums in another order: +-1 LSB on rare ties
