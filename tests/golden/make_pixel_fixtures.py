#!/usr/bin/env python3
"""Generate tests/golden/pixels.npz: small seeded inputs and the CPU oracle's outputs for them.

    python tests/golden/make_pixel_fixtures.py

The reference pins no pixel values (it has no tests), so these vectors come from the repo's own C
oracle after it was cross-checked against the independent numpy restatement (oracle/np_ref.py, tests/test_oracle_*.py).
They freeze the semantics: a later change of the oracle or of the HIP kernels that alters any value is caught.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_cv as ocv  # noqa: E402
from util import camera, star_patch  # noqa: E402


def main():
    out = {}
    w, h = 64, 48
    img = star_patch(w, h, seed=1)
    out["img"] = img
    for warp in ("spherical", "cylindrical", "fisheye", "plane", "paniniA2B1", "transverseMercator"):
        K, R, f = camera(w, h, 60.0, 14.0, -6.0, 3.0)
        o = ocv.PyRotationWarper(warp, f)
        c, d = o.warp(img, K, R, ocv.INTER_LINEAR, ocv.BORDER_REFLECT)
        _, m = o.warp(255 * np.ones((h, w), np.uint8), K, R, ocv.INTER_NEAREST, ocv.BORDER_CONSTANT)
        out[f"warp_{warp}_corner"] = np.array(c)
        out[f"warp_{warp}_img"] = d
        out[f"warp_{warp}_mask"] = m
    out["cam_K"], out["cam_R"], out["cam_f"] = K, R, np.float32(f)
    # blenders
    rng = np.random.default_rng(7)
    imgs, masks, tls = [], [], [(-12, 3), (20, -2), (47, 5)]
    for i in range(3):
        im = star_patch(56, 40, seed=20 + i).astype(np.int16)
        mk = np.zeros((40, 56), np.uint8)
        mk[2 + i:-3, 4:-2 - i] = 255
        mk[rng.integers(0, 40, 12), rng.integers(0, 56, 12)] = rng.integers(0, 256, 12)
        imgs.append(im)
        masks.append(mk)
    out["blend_imgs"], out["blend_masks"], out["blend_tls"] = np.stack(imgs), np.stack(masks), np.array(tls)
    roi = ocv.detail.resultRoi(tls, [(56, 40)] * 3)
    out["blend_roi"] = np.array(roi)
    for name, make in (("mb3", lambda: ocv.detail_MultiBandBlender(num_bands=3)), ("feather", lambda: ocv.detail_FeatherBlender(0.08)),
                       ("no", lambda: ocv.detail.Blender_createDefault(0))):
        b = make()
        b.prepare(roi)
        for im, mk, tl in zip(imgs, masks, tls):
            b.feed(im, mk, tl)
        r, k = b.blend(None, None)
        out[f"blend_{name}_result"], out[f"blend_{name}_mask"] = r, k
    # mask helpers
    m = (rng.uniform(size=(23, 31)) > 0.55).astype(np.uint8) * 255
    out["mask_small"] = m
    out["mask_dilated"] = ocv.dilate(m, None)
    out["mask_resized_77x52"] = ocv.resize(m, (77, 52), 0, 0, ocv.INTER_LINEAR_EXACT)
    # compensator
    corners = [(0, 0), (30, 4), (58, -3)]
    cimgs = [np.clip(np.rint(star_patch(50, 36, seed=40 + i).astype(np.float32) * g), 0, 255).astype(np.uint8) for i, g in enumerate((0.8, 1.0, 1.25))]
    cmasks = [255 * np.ones((36, 50), np.uint8) for _ in range(3)]
    out["comp_imgs"], out["comp_corners"] = np.stack(cimgs), np.array(corners)
    for t in (1, 2, 3, 4):
        c = ocv._Comp(t, 16, 16, 1, 2)
        c.feed(corners, cimgs, cmasks)
        big = star_patch(100, 72, seed=60)
        c.apply(1, corners[1], big, None)
        out[f"comp_{t}_applied"] = big
        if t in (1, 3):
            out[f"comp_{t}_gains"] = c.gains()
        else:
            out[f"comp_{t}_gainmap1"] = c.gainMap(1)
    path = os.path.join(HERE, "pixels.npz")
    np.savez_compressed(path, **out)
    print(f"{len(out)} arrays -> {path} ({os.path.getsize(path)} bytes)")


if __name__ == "__main__":
    main()
