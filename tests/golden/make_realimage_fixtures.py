#!/usr/bin/env python3
"""Generate tests/golden/real_kat26.npz from the reference's RECORDED RUN ARTIFACTS of its daylight example (data, not code).

Run in the build container (needs /root/reference and Pillow):  python tests/golden/make_realimage_fixtures.py

The run `example_01_stitching_daylight_images/2022-12-30_12h33m17s_ORB-BruteForceMatcher_*` (KAT 26 of kat.json: 21 frames of
2592x1728, fisheye warp, waveCorrect HORIZ, mirror "x,y", compose_megapix 0.6, 9-band multiband, timelapse "as_is") left, besides
the cameras and the config that kat.json already holds:

* the 21 input photographs (`img_autumn_forest_a_8+8+4+1_shots/*.jpg`);
* `..._07_timelapse/transparent_fixed_<name>.png` -- written at stitching_detailed_enhanced.py:1869-1879: the timelapser's canvas
  after `process(bitwise_and(image_warped_s, mask=masks_warped_untouched[idx]))` concatenated with the mask timelapser's canvas,
  channels B, G, R, mask, at FULL panorama size (2676x2688) in a LOSSLESS format.  That is OpenCV 4.6's own output of
  imread -> resize(INTER_AREA) -> PyRotationWarper("fisheye").warp(INTER_LINEAR, BORDER_REFLECT) -> warp(mask, INTER_NEAREST,
  BORDER_CONSTANT) -> bitwise_and -> Timelapser.process, pixel for pixel (sde.py:1701-1707, :1731-1746, :1838-1851);
* the final panorama JPEG (sde.py:1938; lossy, blended with dp_colorgrad seams);
* `..._06_masks_warped_seamed/*.jpg`: the seamed compose-scale masks (sde.py:1772-1780), shrunk to <= 700 px and JPEG-coded.

The fixture keeps the encoded input photographs (bytes; decoded with Pillow at test time -- the decode is part of what the lossless
frames pin), for a subset of frames the bounding-box crop of the lossless canvas, for ALL the others every third pixel of it in
both directions (still lossless values, 1/9 of the bytes) with the exact count of mask pixels and the channel sums of the whole
crop, the panorama JPEG and the seamed-mask JPEGs.
"""
import io
import json
import os

import numpy as np
from PIL import Image

REF = "/root/reference"
RUN = "example_01_stitching_daylight_images/2022-12-30_12h33m17s_ORB-BruteForceMatcher"
SHOTS = "img_autumn_forest_a_8+8+4+1_shots"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "real_kat26.npz")
# frames whose lossless canvas is kept: horizon north / south-east, both elevated rings, zenith
LOSSLESS = [0, 3, 8, 12, 16, 20]
RUN2 = "example_01_stitching_daylight_images/2022-12-30_12h33m14s_cv.detail_BestOf2NearestMatcher"
RUN2_FRAMES = [1, 10, 18]
SUBSAMPLE = 3     # the other frames: every third pixel of the lossless canvas in x and y


def _bytes(path):
    return np.frombuffer(open(path, "rb").read(), dtype=np.uint8)


def main():
    kat = json.load(open(os.path.join(HERE, "kat.json")))
    k = [k for k in kat["kats"] if k["run"].startswith(RUN)][0]
    cfg = json.load(open(os.path.join(REF, RUN + "_fisheye_multiband-042.jpg.txt")))
    names = cfg["img_names"]
    out = {"kat_id": np.int32(k["id"]), "names": np.array(names), "lossless": np.array(LOSSLESS, np.int32)}
    for i, n in enumerate(names):
        out[f"jpeg_{i:02d}"] = _bytes(os.path.join(REF, SHOTS, n))
        out[f"seam_{i:02d}"] = _bytes(os.path.join(REF, RUN + "_06_masks_warped_seamed", f"masks_{n}_3_mask_warped_and_seamed.jpg"))
    out["pano_jpeg"] = _bytes(os.path.join(REF, RUN + "_fisheye_multiband-042.jpg"))
    for i in LOSSLESS:
        png = np.asarray(Image.open(os.path.join(REF, RUN + "_07_timelapse", f"transparent_fixed_{names[i]}.png")))
        assert png.shape == (k["golden_pano_size"][1], k["golden_pano_size"][0], 4), png.shape
        nz = np.argwhere(png.any(axis=2))
        (y0, x0), (y1, x1) = nz.min(axis=0), nz.max(axis=0) + 1
        crop = png[y0:y1, x0:x1]
        # PIL hands the cv2-written BGRA file back as R, G, B, A: store B, G, R, mask as the reference held them
        out[f"tl_{i:02d}"] = np.ascontiguousarray(crop[:, :, [2, 1, 0, 3]])
        out[f"tl_box_{i:02d}"] = np.array([x0, y0, x1 - x0, y1 - y0], np.int32)      # in panorama pixels; everything outside is zero
    for i in range(len(names)):
        if i in LOSSLESS:
            continue
        png = np.asarray(Image.open(os.path.join(REF, RUN + "_07_timelapse", f"transparent_fixed_{names[i]}.png")))
        nz = np.argwhere(png.any(axis=2))
        (y0, x0), (y1, x1) = nz.min(axis=0), nz.max(axis=0) + 1
        crop = png[y0:y1, x0:x1][:, :, [2, 1, 0, 3]]
        out[f"ts_{i:02d}"] = np.ascontiguousarray(crop[::SUBSAMPLE, ::SUBSAMPLE])
        out[f"ts_box_{i:02d}"] = np.array([x0, y0, x1 - x0, y1 - y0], np.int32)
        out[f"ts_sums_{i:02d}"] = np.array([int(np.count_nonzero(crop[:, :, 3]))] + [int(crop[:, :, c].astype(np.int64).sum()) for c in range(3)], np.int64)
    out["subsample"] = np.int32(SUBSAMPLE)
    # ---- a SECOND recorded run on the same photographs: other cameras (BestOf2NearestMatcher instead of the brute-force matcher),
    # compose_megapix 1 instead of 0.6 (another INTER_AREA factor), no mirroring; it left no final panorama (so it is not a KAT of
    # kat.json) but the same lossless timelapse canvases, 3494x3453.  Kept: cameras, the few config fields the geometry needs, and
    # every third pixel of three canvases with their mask counts and channel sums.
    cfg2 = json.load(open(os.path.join(REF, RUN2 + "_fisheye_multiband-042.jpg.txt")))
    doc2 = json.load(open(os.path.join(REF, RUN2 + "_fisheye_multiband-042.CameraParams.json")))
    cams2 = doc2[doc2.index("list_of_camera_params_for_disk_output:") + 1]
    assert cfg2["img_names"] == names
    out["run2_json"] = np.array(json.dumps({
        "cameras": [{"R": c["R"], "aspect": c["aspect"], "focal": c["focal"], "ppx": c["ppx"], "ppy": c["ppy"]} for c in cams2],
        "warp": cfg2["warp"], "work_megapix": cfg2["work_megapix"], "compose_megapix": cfg2["compose_megapix"], "wave_correct": cfg2["wave_correct"],
        "mirror_pano": cfg2["mirror_pano"], "rotate_pano_rad": cfg2["rotate_pano_rad"], "full_size": [2592, 1728]}))
    for i in RUN2_FRAMES:
        png = np.asarray(Image.open(os.path.join(REF, RUN2 + "_07_timelapse", f"transparent_fixed_{names[i]}.png")))
        if i == RUN2_FRAMES[0]:
            out["run2_pano_size"] = np.array([png.shape[1], png.shape[0]], np.int32)
        nz = np.argwhere(png.any(axis=2))
        (y0, x0), (y1, x1) = nz.min(axis=0), nz.max(axis=0) + 1
        crop = png[y0:y1, x0:x1][:, :, [2, 1, 0, 3]]
        out[f"r2_{i:02d}"] = np.ascontiguousarray(crop[::SUBSAMPLE, ::SUBSAMPLE])
        out[f"r2_box_{i:02d}"] = np.array([x0, y0, x1 - x0, y1 - y0], np.int32)
        out[f"r2_sums_{i:02d}"] = np.array([int(np.count_nonzero(crop[:, :, 3]))] + [int(crop[:, :, c].astype(np.int64).sum()) for c in range(3)], np.int64)
    out["run2_frames"] = np.array(RUN2_FRAMES, np.int32)
    for i, n in enumerate(names):      # the second run's seamed masks (dp_colorgrad again, on other cameras and another compose scale)
        out[f"seam2_{i:02d}"] = _bytes(os.path.join(REF, RUN2 + "_06_masks_warped_seamed", f"masks_{n}_3_mask_warped_and_seamed.jpg"))
    np.savez_compressed(OUT, **out)
    print(f"{OUT}: {os.path.getsize(OUT) / 1e6:.1f} MB; KAT {k['id']}, {len(names)} frames, lossless canvases of {LOSSLESS}")


# ---- the NIGHT run (example_06, KAT with cameras `2022-12-30_12h34m14s__fisheye_multiband-042.CameraParams.json`): 21 frames of
# 5184x3456 taken at dusk with stars; only four of its photographs are under /root/reference.  A frame's timelapse canvas depends on
# that frame, the recorded cameras and the geometry only, so two of them are kept: the encoded photographs (4 MB each) and every
# third pixel of their lossless canvases.
NIGHT_RUN = "example_06_star_polygon_matcher_outperforms_orb_matcher_on_dawn_images/2022-12-30_12h34m14s_"
NIGHT_SHOTS = "img_eisenberg_18h06m_ISO1600_10s"
NIGHT_FRAMES = ["17-alt2-n.jpg", "21-zenith.jpg"]
NIGHT_OUT = os.path.join(HERE, "real_night.npz")


def night():
    kat = json.load(open(os.path.join(HERE, "kat.json")))
    k = [k for k in kat["kats"] if k["run"].startswith(NIGHT_RUN)][0]
    cfg = json.load(open(os.path.join(REF, NIGHT_RUN + "_fisheye_multiband-042.jpg.txt")))
    names = cfg["img_names"]
    out = {"kat_id": np.int32(k["id"]), "names": np.array(names), "subsample": np.int32(SUBSAMPLE),
           "frames": np.array([names.index(n) for n in NIGHT_FRAMES], np.int32),
           # the run stretched its frames before warping them (sde.py:1711, image_processors.py:32-41): recorded config field
           "bw_point": np.array(cfg["black_and_white_point_adjustment"]["final_panorama"], np.int32)}
    for n in NIGHT_FRAMES:
        i = names.index(n)
        out[f"jpeg_{i:02d}"] = _bytes(os.path.join(REF, NIGHT_SHOTS, n))
        png = np.asarray(Image.open(os.path.join(REF, NIGHT_RUN + "_07_timelapse", f"transparent_fixed_{n}.png")))
        assert png.shape == (k["golden_pano_size"][1], k["golden_pano_size"][0], 4), png.shape
        nz = np.argwhere(png.any(axis=2))
        (y0, x0), (y1, x1) = nz.min(axis=0), nz.max(axis=0) + 1
        crop = png[y0:y1, x0:x1][:, :, [2, 1, 0, 3]]
        out[f"ts_{i:02d}"] = np.ascontiguousarray(crop[::SUBSAMPLE, ::SUBSAMPLE])
        out[f"ts_box_{i:02d}"] = np.array([x0, y0, x1 - x0, y1 - y0], np.int32)
        out[f"ts_sums_{i:02d}"] = np.array([int(np.count_nonzero(crop[:, :, 3]))] + [int(crop[:, :, c].astype(np.int64).sum()) for c in range(3)], np.int64)
    np.savez_compressed(NIGHT_OUT, **out)
    print(f"{NIGHT_OUT}: {os.path.getsize(NIGHT_OUT) / 1e6:.1f} MB; KAT {k['id']}, frames {NIGHT_FRAMES}")


if __name__ == "__main__":
    main()
    night()
