"""The reference's recorded warped masks of all 16 projections (tests/golden/footprints.npz, made by make_footprint_fixtures.py)
against the masks a cv2-shaped namespace produces -- shared by the CPU (oracle) and GPU (HIP) flavours."""
import io
import json
import os
from functools import lru_cache

import numpy as np
from PIL import Image

from opencv_starry_sky_panorama_stitcher_amd import camera as cam

HERE = os.path.dirname(os.path.abspath(__file__))
INTER_NEAREST, BORDER_CONSTANT = 0, 0


@lru_cache(maxsize=1)
def fixture():
    fx = np.load(os.path.join(HERE, "golden", "footprints.npz"))
    doc = json.load(open(os.path.join(HERE, "golden", "kat.json")))
    return fx, doc


def kat_ids():
    return [int(i) for i in fixture()[0]["kat_ids"]]


def footprints(cv, kat_id: int):
    """-> [(ours bool HxW, recorded bool at our size)] for the run's frames: sde.py:1677-1695 (compose-scale cameras), :1740-1745."""
    fx, doc = fixture()
    k = [k for k in doc["kats"] if k["id"] == kat_id][0]
    cams = cam.cameras_from_dicts(doc["camera_sets"][k["camera_set"]])
    fw, fh = k["full_size"]
    ws = cam.scale_for_megapix(k["work_megapix"], fw, fh)
    g = cam.prepare_compose_cameras(cams, [(fw, fh)] * len(cams), ws, k["compose_megapix"], k["wave_correct"], k["mirror_pano"], k["rotate_pano_rad"])
    warper = cv.PyRotationWarper(k["warp"], g.warper_scale)
    out = []
    for i in range(len(cams)):
        w, h = g.sizes[i]
        _, m = warper.warp(np.full((h, w), 255, np.uint8), g.Ks[i], g.Rs[i], INTER_NEAREST, BORDER_CONSTANT)
        m = np.asarray(m.get() if hasattr(m, "get") else m)
        rec = Image.open(io.BytesIO(fx[f"m_{kat_id}_{i}"].tobytes())).convert("L")
        aspect_ok = abs(rec.size[1] - rec.size[0] * m.shape[0] / m.shape[1]) <= 1.5          # the recording kept the aspect ratio
        rec_full = np.asarray(rec.resize((m.shape[1], m.shape[0]), Image.BILINEAR)) >= 128
        out.append((m > 0, rec_full, aspect_ok, k["warp"]))
    return out


def agreement(ours: np.ndarray, rec: np.ndarray):
    inter, union = np.count_nonzero(ours & rec), np.count_nonzero(ours | rec)
    return inter / max(union, 1)
