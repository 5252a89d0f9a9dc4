"""The reference's recorded daylight run (tests/golden/real_kat26.npz, made by tests/golden/make_realimage_fixtures.py) replayed
through a cv2-shaped namespace -- the CPU oracle (tests/oracle_cv.py) or the HIP package -- following the reference's own call
sequence (stitching_detailed_enhanced.py:1673-1944).  Shared by the CPU and the GPU flavour of tests/test_real_images.py."""
import io
import json
import os
from functools import lru_cache

import numpy as np
from PIL import Image

from opencv_starry_sky_panorama_stitcher_amd import camera as cam
from opencv_starry_sky_panorama_stitcher_amd import compose as cmp

HERE = os.path.dirname(os.path.abspath(__file__))
INTER_NEAREST, INTER_LINEAR = 0, 1
BORDER_CONSTANT, BORDER_REFLECT = 0, 2
TIMELAPSER_AS_IS = 0


@lru_cache(maxsize=3)
def fixture(run: int = 1):
    """run 1: the recorded run of kat.json (brute-force matcher cameras, compose_megapix 0.6, mirrored); run 2: the second recorded
    run on the same photographs (BestOf2NearestMatcher cameras, compose_megapix 1, not mirrored; it left no final panorama);
    run 3: the night run of example_06 (5184x3456 frames with stars; two of its photographs are kept, real_night.npz)."""
    fx = np.load(os.path.join(HERE, "golden", "real_night.npz" if run == 3 else "real_kat26.npz"))
    if run == 2:
        k = json.loads(str(fx["run2_json"]))
        cams = cam.cameras_from_dicts(k["cameras"])
    else:
        doc = json.load(open(os.path.join(HERE, "golden", "kat.json")))
        k = [k for k in doc["kats"] if k["id"] == int(fx["kat_id"])][0]
        cams = cam.cameras_from_dicts(doc["camera_sets"][k["camera_set"]])
    fw, fh = k["full_size"]
    ws = cam.scale_for_megapix(k["work_megapix"], fw, fh)                                     # sde.py:751-752
    g = cam.prepare_compose_cameras(cams, [(fw, fh)] * len(cams), ws, k["compose_megapix"], k["wave_correct"], k["mirror_pano"],
                                    k["rotate_pano_rad"])                                      # sde.py:1373-1535, :1677-1695
    return fx, k, g


def decode_bgr(jpeg_bytes: np.ndarray) -> np.ndarray:
    """cv.imread(name): 8-bit BGR."""
    return np.ascontiguousarray(np.asarray(Image.open(io.BytesIO(jpeg_bytes.tobytes())).convert("RGB"))[:, :, ::-1])


def decode_gray(jpeg_bytes: np.ndarray) -> Image.Image:
    return Image.open(io.BytesIO(jpeg_bytes.tobytes())).convert("L")


def rois(cv, run: int = 1):
    _, k, g = fixture(run)
    w = cv.PyRotationWarper(k["warp"], g.warper_scale)                                         # sde.py:1684-1688
    r = [tuple(w.warpRoi(sz, K, R)) for sz, K, R in zip(g.sizes, g.Ks, g.Rs)]                  # sde.py:1696
    return r, tuple(cv.detail.resultRoi([x[:2] for x in r], [x[2:] for x in r]))


def _host(a):
    return np.asarray(a.get() if hasattr(a, "get") else a)


def timelapse_canvas(cv, idx: int, run: int = 1):
    """What the reference wrote to `transparent_fixed_<name>.png` for frame idx (sde.py:1699-1707, :1731-1746, :1838-1851, :1869-1879):
    B, G, R of the image timelapser's canvas and channel 0 of the mask timelapser's canvas, saturated to 8 bits by imwrite."""
    fx, k, g = fixture(run)
    r, pano = rois(cv, run)
    corners, sizes = [x[:2] for x in r], [x[2:] for x in r]
    bw = tuple(int(v) for v in fx["bw_point"]) if "bw_point" in fx.files else None             # black / white point of the run, if any
    img = cv.prepare_frame(decode_bgr(fx[f"jpeg_{idx:02d}"]), g.compose_scale, bw)             # :1701-1711
    warper = cv.PyRotationWarper(k["warp"], g.warper_scale)
    corner, image_warped = warper.warp(img, g.Ks[idx], g.Rs[idx], INTER_LINEAR, BORDER_REFLECT)  # :1731
    mask = 255 * np.ones((img.shape[0], img.shape[1]), np.uint8)
    _, mask_warped = warper.warp(mask, g.Ks[idx], g.Rs[idx], INTER_NEAREST, BORDER_CONSTANT)   # :1740
    assert tuple(corner) == tuple(corners[idx])
    image_warped_s = _host(image_warped).astype(np.int16)                                      # :1755
    mask_warped = _host(mask_warped)
    ma_tones = np.ones(image_warped_s.shape[:2], np.uint8)
    t_img = cv.detail.Timelapser_createDefault(TIMELAPSER_AS_IS)
    t_img.initialize(corners, sizes)
    t_msk = cv.detail.Timelapser_createDefault(TIMELAPSER_AS_IS)
    t_msk.initialize(corners, sizes)
    t_img.process(cv.bitwise_and(image_warped_s, image_warped_s, mask=mask_warped), ma_tones, corners[idx])        # :1840-1844
    t_msk.process(np.repeat(mask_warped.astype(np.int16)[:, :, np.newaxis], 3, axis=2), ma_tones, corners[idx])   # :1847-1851
    canvas = np.concatenate((_host(t_img.getDst()), _host(t_msk.getDst())), axis=2)[:, :, 0:4]                      # :1878
    return np.clip(canvas, 0, 255).astype(np.uint8), pano


def compare_with_recorded_canvas(canvas: np.ndarray, idx: int):
    """-> (samples compared, samples that differ, max |diff|, mask pixels that differ, non-zero samples outside the recorded box)."""
    fx, _, _ = fixture()
    x0, y0, w, h = [int(v) for v in fx[f"tl_box_{idx:02d}"]]
    want = fx[f"tl_{idx:02d}"]
    got = canvas[y0:y0 + h, x0:x0 + w]
    outside = canvas.copy()
    outside[y0:y0 + h, x0:x0 + w] = 0
    both = (got[..., 3] != 0) & (want[..., 3] != 0)
    d = np.abs(got[..., :3].astype(np.int16) - want[..., :3].astype(np.int16))[both]
    return int(d.size), int((d > 0).sum()), int(d.max()) if d.size else 0, int((got[..., 3] != want[..., 3]).sum()), int(np.count_nonzero(outside))


def flip_attribution(orc, ocv, idx: int):
    """Every sample of frame idx that differs from OpenCV's recorded canvas, attributed: the recorded B, G, R must equal the oracle's
    own remap (fixed-point bilinear, BORDER_REFLECT) evaluated at the oracle's quantised map coordinate moved by one 1/32-px step,
    (sx + dx, sy + dy) with sx = cvRound(32 xmap), dx, dy in {-1, 0, 1} -- i.e. the difference is the map's last ulp (the libm of the
    recording machine) and nothing in remap, INTER_AREA, the tables or the rounding rules.
    -> (pixels that differ, pixels explained, {(dx, dy): pixels first explained by that step})."""
    fx, k, g = fixture()
    canvas, pano = timelapse_canvas(ocv, idx)
    x0, y0, w, h = [int(v) for v in fx[f"tl_box_{idx:02d}"]]
    want = fx[f"tl_{idx:02d}"]
    got = canvas[y0:y0 + h, x0:x0 + w]
    both = (got[..., 3] != 0) & (want[..., 3] != 0)
    ys, xs = np.nonzero(both & (got[..., :3] != want[..., :3]).any(axis=2))
    r, _ = rois(ocv)
    cx, cy = r[idx][0] - pano[0], r[idx][1] - pano[1]                                          # the frame's corner on the canvas
    img = ocv.prepare_frame(decode_bgr(fx[f"jpeg_{idx:02d}"]), g.compose_scale, None)
    _, xmap, ymap = ocv.PyRotationWarper(k["warp"], g.warper_scale).buildMaps((img.shape[1], img.shape[0]), g.Ks[idx], g.Rs[idx])
    wy, wx = y0 + ys - cy, x0 + xs - cx
    sx = np.rint(xmap[wy, wx] * np.float32(32)).astype(np.int64)                                # cvRound(x * 32), float product exact
    sy = np.rint(ymap[wy, wx] * np.float32(32)).astype(np.int64)
    rec = want[ys, xs, :3]
    explained = np.zeros(len(ys), bool)
    first = {}
    for dx, dy in ((0, 0), (-1, 0), (1, 0), (0, -1), (0, 1), (-1, -1), (1, -1), (-1, 1), (1, 1)):
        mx = ((sx + dx) / 32.0).astype(np.float32)[None, :]                                     # exact: |sx| < 2^24
        my = ((sy + dy) / 32.0).astype(np.float32)[None, :]
        hit = (orc.remap(img, mx, my, INTER_LINEAR, BORDER_REFLECT)[0] == rec).all(axis=1)
        first[(dx, dy)] = int((hit & ~explained).sum())
        explained |= hit
    return len(ys), int(explained.sum()), first


def compare_with_recorded_subsample(canvas: np.ndarray, idx: int, key: str = "ts"):
    """The frames whose lossless canvas is kept as every third pixel (plus the exact mask count and channel sums of the whole crop):
    -> (samples compared, samples that differ, max |diff|, mask samples that differ, non-zero samples outside the box,
        mask-pixel count difference over the whole crop, largest relative channel-sum difference over the whole crop)."""
    fx, _, _ = fixture(3 if key == "night" else 1)
    key = "ts" if key == "night" else key
    st = int(fx["subsample"])
    x0, y0, w, h = [int(v) for v in fx[f"{key}_box_{idx:02d}"]]
    want = fx[f"{key}_{idx:02d}"]
    crop = canvas[y0:y0 + h, x0:x0 + w]
    got = crop[::st, ::st]
    outside = canvas.copy()
    outside[y0:y0 + h, x0:x0 + w] = 0
    both = (got[..., 3] != 0) & (want[..., 3] != 0)
    d = np.abs(got[..., :3].astype(np.int16) - want[..., :3].astype(np.int16))[both]
    sums = fx[f"{key}_sums_{idx:02d}"]
    mine = [int(np.count_nonzero(crop[:, :, 3]))] + [int(crop[:, :, c].astype(np.int64).sum()) for c in range(3)]
    rel = max(abs(mine[c] - int(sums[c])) / max(int(sums[c]), 1) for c in (1, 2, 3))
    return (int(d.size), int((d > 0).sum()), int(d.max()) if d.size else 0, int((got[..., 3] != want[..., 3]).sum()), int(np.count_nonzero(outside)),
            mine[0] - int(sums[0]), rel)


def recorded_seam_masks(cv, run: int = 1):
    """The run's `masks_warped_and_seamed` (sde.py:1772-1780), recorded shrunk to <= 700 px and JPEG-coded: brought back to the
    warped size (bilinear, threshold 128).  Accurate to about +-2 px along the seams."""
    fx, _, _ = fixture(run)
    r, _ = rois(cv, run)
    out = []
    for i, roi in enumerate(r):
        m = decode_gray(fx[f"seam{'2' if run == 2 else ''}_{i:02d}"]).resize((roi[2], roi[3]), Image.BILINEAR)
        out.append(((np.asarray(m) >= 128) * 255).astype(np.uint8))
    return out


def panorama(cv):
    """All 21 frames through the compose loop with the recorded seams and the reference's blender set-up (sde.py:1805-1820 with
    blend_strength 42: 9 bands): -> ComposeResult."""
    fx, k, g = fixture()
    frames = [cv.prepare_frame(decode_bgr(fx[f"jpeg_{i:02d}"]), g.compose_scale) for i in range(len(g.Ks))]
    frames = [_host(f) for f in frames]
    return cmp.compose_panorama(cv, frames, g.Ks, g.Rs, warp=k["warp"], warper_scale=g.warper_scale, blend=k["blend"], num_bands=None,
                                blend_strength=k["blend_strength"], blend_masks=recorded_seam_masks(cv))


def recorded_panorama() -> np.ndarray:
    fx, _, _ = fixture()
    return decode_bgr(fx["pano_jpeg"])


def jpeg_roundtrip(bgr: np.ndarray) -> np.ndarray:
    """cv.imwrite(name.jpg, img) defaults: quality 95, 4:2:0 chroma."""
    buf = io.BytesIO()
    Image.fromarray(np.ascontiguousarray(bgr[:, :, ::-1])).save(buf, format="JPEG", quality=95, subsampling=2)
    return decode_bgr(np.frombuffer(buf.getvalue(), np.uint8))


def psnr(a, b, sel) -> float:
    d = (a.astype(np.float64) - b.astype(np.float64))[sel]
    return float(10 * np.log10(255.0 ** 2 / (d ** 2).mean()))


def panorama_agreement(mosaic: np.ndarray, result_mask: np.ndarray):
    """Figures of the blended panorama against the recorded JPEG (see test_real_images.py for the bars)."""
    from scipy import ndimage as ndi

    want = recorded_panorama()
    assert want.shape == mosaic.shape, (want.shape, mosaic.shape)
    valid = ndi.binary_erosion(result_mask > 0, iterations=8)
    out = {"psnr": psnr(mosaic, want, valid), "psnr_after_same_jpeg": psnr(jpeg_roundtrip(mosaic), want, valid)}
    a = ndi.gaussian_filter(mosaic.astype(np.float64), (3, 3, 0))
    b = ndi.gaussian_filter(want.astype(np.float64), (3, 3, 0))
    out["rms_blur3"] = float(np.sqrt(((a - b)[valid] ** 2).mean()))
    # sub-pixel registration: the best cubic-shift match of the green channel must sit at (0, 0)
    g1, g2 = mosaic[..., 1].astype(np.float64), want[..., 1].astype(np.float64)
    H, W = g1.shape
    shifts = []
    for cy, cx in ((H // 2, W // 2), (H // 3, W // 3), (2 * H // 3, W // 3), (H // 3, 2 * W // 3), (2 * H // 3, 2 * W // 3)):
        sl = (slice(cy - 100, cy + 100), slice(cx - 100, cx + 100))
        best = None
        for sy in np.arange(-0.75, 0.76, 0.25):
            for sx in np.arange(-0.75, 0.76, 0.25):
                e = ((ndi.shift(g1[sl], (sy, sx), order=3, mode="nearest") - g2[sl])[8:-8, 8:-8] ** 2).mean()
                if best is None or e < best[0]:
                    best = (e, float(sy), float(sx))
        shifts.append(best[1:])
    out["best_shifts"] = shifts
    return out
