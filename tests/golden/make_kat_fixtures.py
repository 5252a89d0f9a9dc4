#!/usr/bin/env python3
"""Generate tests/golden/kat.json from the reference's RECORDED RUN ARTIFACTS (data, not code).

Run in the build container (needs /root/reference):  python tests/golden/make_kat_fixtures.py

Each recorded run of the reference left three data files:
  *.CameraParams.json  cameras after bundle adjustment, written at stitching_detailed_enhanced.py:1122-1156
  *.jpg.txt            the full config dump, written at sde.py:1945-1952
  *.jpg                the final panorama, whose pixel size equals resultRoi(corners, sizes)
                       (sde.py:1807, :1930-1944)
The fixture keeps: the cameras, the handful of config fields that reach the warper, the full-image
size of the input set and the panorama size.  That pins warpRoi for all 16 projections, resultRoi,
waveCorrect and the mirror/rotate composition (geometry only; pixel values stay unpinned).
"""
import glob
import json
import os

from PIL import Image

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat.json")

# full-resolution input size per image set: read from the set when present, else from a sibling set shot
# with the same camera (all night sets are 5184x3456: consistent with ppx=671, ppy=447 at 1.2 MP)
def _set_size(input_dir):
    files = sorted(glob.glob(os.path.join(REF, input_dir, "*.jpg")))
    if files:
        return list(Image.open(files[0]).size)
    return [5184, 3456]


def _load_cams(path):
    doc = json.load(open(path))
    cams = doc[doc.index("list_of_camera_params_for_disk_output:") + 1]
    return [{"R": c["R"], "aspect": c["aspect"], "focal": c["focal"], "ppx": c["ppx"], "ppy": c["ppy"]} for c in cams]


def main():
    camera_sets = {}
    kats = []
    for d in sorted(glob.glob(os.path.join(REF, "example_0*"))):
        if not os.path.isdir(d):
            continue
        cam_files = sorted(glob.glob(os.path.join(d, "*.CameraParams.json")))
        for txt in sorted(glob.glob(os.path.join(d, "*.jpg.txt"))):
            jpg = txt[:-4]
            if not os.path.exists(jpg):
                continue
            cfg = json.load(open(txt))
            # cameras recorded for this run: the newest CameraParams.json not newer than the run with
            # the same number of cameras and the same matcher family (re-compose runs reload that state)
            stamp = os.path.basename(txt)[:20]
            cands = [c for c in cam_files if os.path.basename(c)[:20] <= stamp]
            if not cands:
                continue
            cam_file = cands[-1]
            cams = _load_cams(cam_file)
            if len(cams) != len(cfg["img_names"]) and len(cfg["img_names"]) > 0:
                # subset runs keep only the biggest component; sizes must still match the recorded cameras
                pass
            # example_06 re-runs (19h55m..19h57m) used other matchers whose cameras were not recorded
            if "example_06" in d and not os.path.basename(cam_file)[:20] == stamp:
                continue
            set_name = os.path.relpath(cam_file, REF)
            camera_sets.setdefault(set_name, cams)
            kats.append(
                {
                    "id": len(kats) + 1,
                    "run": os.path.relpath(jpg, REF),
                    "camera_set": set_name,
                    "full_size": _set_size(cfg["input_dir"]),
                    "work_megapix": cfg["work_megapix"],
                    "compose_megapix": cfg["compose_megapix"],
                    "warp": cfg["warp"],
                    "wave_correct": cfg["wave_correct"],
                    "mirror_pano": cfg["mirror_pano"],
                    "rotate_pano_rad": cfg["rotate_pano_rad"],
                    "blend": cfg["blend"],
                    "blend_strength": float(cfg["blend_strength"]),
                    "golden_pano_size": list(Image.open(jpg).size),
                }
            )
    json.dump({"camera_sets": camera_sets, "kats": kats}, open(OUT, "w"), indent=0, separators=(",", ":"))
    print(f"{len(kats)} KATs, {len(camera_sets)} camera sets -> {OUT} ({os.path.getsize(OUT)} bytes)")
    for k in kats:
        print(k["id"], k["run"], k["warp"], k["wave_correct"], k["mirror_pano"], round(k["rotate_pano_rad"], 3), k["golden_pano_size"])


if __name__ == "__main__":
    main()
