#!/usr/bin/env python3
"""tests/golden/footprints.npz: the warped validity masks the reference recorded for ALL 16 projections (data, not code).

example_04_demonstrate_all_projections ran the same 3 frames (5184x3456, cameras of `2022-12-30_19h46m42s__cylindrical_...
CameraParams.json`) through every `warp` string of stitching_detailed_enhanced.py:218-237 and wrote, per frame,
`..._05_masks_untouched/masks_<name>_0_untouched_mask.jpg` = `warper.warp(mask, K, R, INTER_NEAREST, BORDER_CONSTANT)` (sde.py:1740-1752).
The copies kept in the reference repository are shrunk to 700 px width and JPEG coded; the photographs themselves are not there --
but a footprint needs none: it is a function of the cameras, the frame size and the projection.  Run in the build container:
    python tests/golden/make_footprint_fixtures.py
"""
import glob
import json
import os

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "footprints.npz")


def main():
    kat = json.load(open(os.path.join(HERE, "kat.json")))
    out, ids = {}, []
    for k in kat["kats"]:
        if not k["run"].startswith("example_04_demonstrate_all_projections/"):
            continue
        prefix = os.path.join(REF, k["run"][:k["run"].index("_" + k["warp"] + "_multiband")])
        cfg = json.load(open(os.path.join(REF, k["run"] + ".txt")))
        files = [os.path.join(prefix + "_05_masks_untouched", f"masks_{n}_0_untouched_mask.jpg") for n in cfg["img_names"]]
        assert all(os.path.exists(f) for f in files), files
        ids.append(k["id"])
        for i, f in enumerate(files):
            out[f"m_{k['id']}_{i}"] = np.frombuffer(open(f, "rb").read(), dtype=np.uint8)
    out["kat_ids"] = np.array(ids, np.int32)
    np.savez_compressed(OUT, **out)
    print(f"{OUT}: {os.path.getsize(OUT) / 1e3:.0f} kB, {len(ids)} projections x 3 frames")


if __name__ == "__main__":
    main()
