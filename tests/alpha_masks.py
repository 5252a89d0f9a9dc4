"""The alpha channel of the reference's recorded lossless timelapse canvases (tests/golden/alpha_masks.npz, made by
make_alpha_fixtures.py) against the warped masks a cv2-shaped namespace produces -- shared by the CPU (oracle) and GPU (HIP)
flavours of tests/test_alpha_masks.py."""
import json
import os
from functools import lru_cache

import numpy as np

from opencv_starry_sky_panorama_stitcher_amd import camera as cam

HERE = os.path.dirname(os.path.abspath(__file__))
INTER_NEAREST, BORDER_CONSTANT = 0, 0


@lru_cache(maxsize=1)
def fixture():
    fx = np.load(os.path.join(HERE, "golden", "alpha_masks.npz"))
    doc = json.load(open(os.path.join(HERE, "golden", "kat.json")))
    return fx, doc


def kat_ids():
    return [int(i) for i in fixture()[0]["kat_ids"]]


def n_frames(kat_id: int) -> int:
    return len(fixture()[0][f"names_{kat_id}"])


@lru_cache(maxsize=8)
def geometry(kat_id: int):
    _, doc = fixture()
    k = [k for k in doc["kats"] if k["id"] == kat_id][0]
    cams = cam.cameras_from_dicts(doc["camera_sets"][k["camera_set"]])
    fw, fh = k["full_size"]
    ws = cam.scale_for_megapix(k["work_megapix"], fw, fh)                                       # sde.py:751-752
    g = cam.prepare_compose_cameras(cams, [(fw, fh)] * len(cams), ws, k["compose_megapix"], k["wave_correct"], k["mirror_pano"],
                                    k["rotate_pano_rad"])                                        # sde.py:1373-1535, :1677-1695
    return k, g


def recorded(kat_id: int, idx: int):
    """-> (box (x, y, w, h) in panorama pixels, bool h x w alpha inside it, count of alpha pixels of the whole canvas)"""
    fx, _ = fixture()
    box = [int(v) for v in fx[f"box_{kat_id}_{idx:02d}"]]
    bits = np.unpackbits(fx[f"bits_{kat_id}_{idx:02d}"], axis=1)[:, :box[2]].astype(bool)
    return box, bits, int(fx[f"count_{kat_id}_{idx:02d}"])


def mask_difference(cv, kat_id: int, idx: int):
    """Frame idx of the run through `warper.warp(mask, NEAREST, CONSTANT)` (sde.py:1739-1745), pasted where the mask timelapser pastes it
    (sde.py:1847-1851): -> (pixels that differ from the recorded alpha over the whole canvas, mask pixels, panorama size ok)."""
    k, g = geometry(kat_id)
    warper = cv.PyRotationWarper(k["warp"], g.warper_scale)
    r = [tuple(warper.warpRoi(sz, K, R)) for sz, K, R in zip(g.sizes, g.Ks, g.Rs)]               # sde.py:1696
    pano = tuple(cv.detail.resultRoi([x[:2] for x in r], [x[2:] for x in r]))                    # sde.py:1807
    w, h = g.sizes[idx]
    corner, m = warper.warp(np.full((h, w), 255, np.uint8), g.Ks[idx], g.Rs[idx], INTER_NEAREST, BORDER_CONSTANT)
    m = np.asarray(m.get() if hasattr(m, "get") else m) != 0
    assert tuple(corner) == tuple(r[idx][:2])
    (bx, by, bw, bh), want, count = recorded(kat_id, idx)
    assert count == int(np.count_nonzero(want))
    canvas = np.zeros((pano[3], pano[2]), bool)
    x0, y0 = corner[0] - pano[0], corner[1] - pano[1]
    canvas[y0:y0 + m.shape[0], x0:x0 + m.shape[1]] = m
    rec = np.zeros_like(canvas)
    rec[by:by + bh, bx:bx + bw] = want
    return int(np.count_nonzero(canvas != rec)), count, list(pano[2:]) == list(k["golden_pano_size"])
