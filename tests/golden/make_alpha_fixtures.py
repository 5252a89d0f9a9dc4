#!/usr/bin/env python3
"""Generate tests/golden/alpha_masks.npz from the ALPHA channel of the reference's recorded lossless timelapse canvases (data).

Run in the build container (needs /root/reference and Pillow):  python tests/golden/make_alpha_fixtures.py

`..._07_timelapse/transparent_fixed_<name>.png` is written at stitching_detailed_enhanced.py:1869-1879; its fourth channel is
the mask timelapser's canvas = `warper.warp(mask, K, R, INTER_NEAREST, BORDER_CONSTANT)` of one frame (sde.py:1739-1745) pasted
at the frame's corner (sde.py:1847-1851), at FULL panorama size and lossless.  A warped mask is a function of cameras, frame
size, projection and compose scale only, so the runs whose photographs are not under /root/reference pin it just as well:

* example_03, two runs (KAT 6 / 7 of kat.json): 8 frames of 5184x3456 through **cylindrical** -- the only lossless pixel-level
  recording of `CylindricalWarper` (warpRoi by border, mapBackward, nearest remap) the reference holds; the second run adds
  waveCorrect(HORIZ);
* example_02 (KAT 2), example_05 (KAT 24) and example_06 (KAT 34): 21 frames each through fisheye with three further camera
  sets (the photographs of KAT 2 / 24 are absent, so the colour channels of these canvases cannot be replayed).

Kept per canvas: the bounding box of the non-zero alpha and the bit-packed alpha inside it (np.packbits), a few kB after
compression.  `spherical` has no lossless artifact anywhere in the reference (example_04 kept 700-px JPEG thumbnails only).
"""
import json
import os

import numpy as np
from PIL import Image

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "alpha_masks.npz")
# KAT id -> directory of its timelapse canvases (the KAT's run name up to the projection suffix + "_07_timelapse")
RUNS = {
    6: "example_03_waviness_correction/2022-12-30_12h33m29s_no-waviness-correction",
    7: "example_03_waviness_correction/2022-12-30_19h50m51s_horizontal-waviness-correction",
    2: "example_02_colorized_seams_and_edges/2022-12-30_12h33m24s_",
    24: "example_05_demonstrate_rotation/2022-12-30_12h34m05s_",
    34: "example_06_star_polygon_matcher_outperforms_orb_matcher_on_dawn_images/2022-12-30_12h34m14s_",
}


def main():
    kat = json.load(open(os.path.join(HERE, "kat.json")))
    out = {"kat_ids": np.array(sorted(RUNS), np.int32)}
    for kid, run in sorted(RUNS.items()):
        k = [k for k in kat["kats"] if k["id"] == kid][0]
        assert k["run"].startswith(run), (k["run"], run)
        cfg = json.load(open(os.path.join(REF, k["run"] + ".txt")))
        names = cfg["img_names"]
        out[f"names_{kid}"] = np.array(names)
        for i, n in enumerate(names):
            png = np.asarray(Image.open(os.path.join(REF, run + "_07_timelapse", f"transparent_fixed_{n}.png")))
            assert png.shape == (k["golden_pano_size"][1], k["golden_pano_size"][0], 4), (png.shape, k["golden_pano_size"])
            alpha = png[:, :, 3]
            assert set(np.unique(alpha).tolist()) <= {0, 255}
            nz = np.argwhere(alpha)
            (y0, x0), (y1, x1) = nz.min(axis=0), nz.max(axis=0) + 1
            out[f"box_{kid}_{i:02d}"] = np.array([x0, y0, x1 - x0, y1 - y0], np.int32)
            out[f"bits_{kid}_{i:02d}"] = np.packbits(alpha[y0:y1, x0:x1] != 0, axis=1)
            out[f"count_{kid}_{i:02d}"] = np.int64(np.count_nonzero(alpha))
    np.savez_compressed(OUT, **out)
    print(f"{OUT}: {os.path.getsize(OUT) / 1e6:.2f} MB; KATs {sorted(RUNS)}")


if __name__ == "__main__":
    main()
