"""Pixel-level pin against what the reference itself recorded (tests/golden/real_kat26.npz, see make_realimage_fixtures.py).

(a) LOSSLESS.  The reference's daylight run wrote, per frame, the timelapser canvas of the masked warped frame plus the warped
    mask as a full-size PNG (stitching_detailed_enhanced.py:1869-1879).  That is OpenCV 4.6's own output of
    imread -> resize(INTER_AREA) -> fisheye buildMaps -> remap(INTER_LINEAR, BORDER_REFLECT) -> remap(mask, NEAREST, CONSTANT) ->
    bitwise_and -> Timelapser.process, preceded by waveCorrect + mirror on the cameras.  Bar: identical masks, identical
    placement, and >= 99.9 % of the image samples bit-identical, the rest within 8 grey levels.  The remainder (measured
    0.05-0.08 %) is the distance between this repo's correctly rounded transcendentals (include/ssp_math.h) and the libm of
    the machine that recorded the run: a 1-ULP difference in sinf / cosf / atan2f moves a source coordinate by ~2e-4 px, which
    flips the 1/32-px quantisation of cv::remap for that share of samples by one step (<= gradient / 32 grey levels).
(b) LOSSY.  The final 2676x2688 panorama (9-band multiband) exists as a JPEG produced with dp_colorgrad seams; the seamed masks
    were recorded shrunk.  With the recorded seams brought back to size, the blended panorama must agree with the recorded JPEG
    to within JPEG noise, register at (0, 0) +- 0.25 px, and move closer when it goes through the same JPEG coding.

The same bodies run against the CPU oracle (here, `not gpu`) and against the HIP library (`gpu`).
"""
import numpy as np
import pytest

import real_images as ri

LOSSLESS = [int(i) for i in ri.fixture()[0]["lossless"]]


def _check_canvas(cv, idx):
    canvas, pano = ri.timelapse_canvas(cv, idx)
    _, k, _ = ri.fixture()
    assert list(pano[2:]) == k["golden_pano_size"]
    n, ndiff, dmax, mask_diff, outside = ri.compare_with_recorded_canvas(canvas, idx)
    assert outside == 0, "pixels written outside the recorded frame box"
    assert mask_diff == 0, f"warped mask differs from OpenCV's in {mask_diff} pixels"
    assert n > 1_000_000
    assert ndiff <= 1e-3 * n, f"{ndiff} of {n} samples differ from OpenCV's recorded warp ({100.0 * ndiff / n:.3f} %)"
    assert dmax <= 8
    return ndiff / n


SUBSAMPLED = [i for i in range(21) if i not in LOSSLESS]


def _check_subsampled_canvas(cv, idx, run=1):
    """the other 15 frames (and the three kept of the second recorded run): every third pixel of OpenCV's lossless canvas, the exact
    number of mask pixels, the channel sums"""
    canvas, pano = ri.timelapse_canvas(cv, idx, run)
    if run == 2:
        assert [int(v) for v in pano[2:]] == [int(v) for v in ri.fixture()[0]["run2_pano_size"]]
    if run == 3:
        assert [int(v) for v in pano[2:]] == ri.fixture(3)[1]["golden_pano_size"]
    n, ndiff, dmax, mask_diff, outside, mask_count_diff, sum_rel = ri.compare_with_recorded_subsample(canvas, idx, {1: "ts", 2: "r2", 3: "night"}[run])
    assert outside == 0 and mask_diff == 0 and mask_count_diff == 0, (outside, mask_diff, mask_count_diff)
    assert n > 100_000
    # measured over the 15 frames: 0.03 % ... 0.22 % of the samples differ (by <= 8), channel sums of the whole crop within 2.1e-6
    assert ndiff <= 3e-3 * n and dmax <= 8, f"{ndiff} of {n} samples differ ({100.0 * ndiff / n:.3f} %), max {dmax}"
    assert sum_rel < (1e-4 if run == 3 else 5e-6), sum_rel     # the night frames are dark: small sums, larger relative differences
    return ndiff / n


def _check_panorama(res):
    a = ri.panorama_agreement(res.mosaic, res.result_mask)
    assert res.num_bands == 9 and tuple(res.pano_roi[2:]) == (2676, 2688)
    # JPEG noise of this content at quality 95 measures 37.9 dB (panorama vs its own round trip)
    assert a["psnr"] >= 37.0, a
    assert a["psnr_after_same_jpeg"] >= 41.5 and a["psnr_after_same_jpeg"] > a["psnr"] + 3.0, a
    assert a["rms_blur3"] <= 0.8, a
    assert all(abs(sy) < 0.25 and abs(sx) < 0.25 for sy, sx in a["best_shifts"]), a
    return a


# ---- CPU: the oracle against the reference's recordings (this is what pins the oracle's pixel arithmetic) -------------------------
@pytest.mark.parametrize("idx", LOSSLESS)
def test_oracle_reproduces_opencv_recorded_warp(oracle, idx):
    import oracle_cv as ocv

    _check_canvas(ocv, idx)


@pytest.mark.parametrize("idx", LOSSLESS)
def test_residual_against_opencv_recorded_warp_is_the_maps_last_ulp_and_nothing_else(oracle, idx):
    """Attribution of the 0.03-0.22 % of samples that differ from the recording: every one of them is reproduced exactly,
    all three channels, by the oracle's own fixed-point remap at its own quantised coordinate moved one 1/32-px step; zero unexplained.
    Measured: 534/534, 641/641, 291/291, 688/688, 703/703, 544/544 on frames 0, 3, 8, 12, 16, 20 -- so remap's fixed point, INTER_AREA,
    the projector and the rounding rules are pinned bit for bit; what is left is the recording machine's sinf / cosf / atan2f."""
    import oracle_cv as ocv

    n, explained, first = ri.flip_attribution(oracle, ocv, idx)
    assert 0 < n < 2000, n
    assert first[(0, 0)] == 0                       # a differing sample is not explained by the unmoved coordinate, by definition
    assert explained == n, f"frame {idx}: {n - explained} of {n} differing samples are NOT a one-step move of the quantised map: {first}"


def test_oracle_panorama_agrees_with_recorded_jpeg(oracle):
    import oracle_cv as ocv

    _check_panorama(ri.panorama(ocv))


def test_float64_eigensolver_would_miss_the_recorded_warp(oracle, monkeypatch):
    """Why camera.eigen_symmetric_f32 restates cv::eigen's binary32 Jacobi sweep: with a binary64 eigh the wave-correction rotation
    lands 1.5e-5 away and about a fifth of the samples leave OpenCV's output."""
    import oracle_cv as ocv
    from opencv_starry_sky_panorama_stitcher_amd import camera as cam

    def eigh64(m):
        vals, vecs = np.linalg.eigh(np.asarray(m, np.float64))
        order = np.argsort(-vals)
        return vals[order].astype(np.float32), vecs[:, order].T.astype(np.float32)

    monkeypatch.setattr(cam, "eigen_symmetric_f32", eigh64)
    ri.fixture.cache_clear()
    try:
        canvas, _ = ri.timelapse_canvas(ocv, 0)
        n, ndiff, _, _, _ = ri.compare_with_recorded_canvas(canvas, 0)
    finally:
        monkeypatch.undo()
        ri.fixture.cache_clear()
    assert ndiff > 0.05 * n


@pytest.mark.parametrize("idx", SUBSAMPLED[::4])
def test_oracle_reproduces_opencv_recorded_warp_other_frames(oracle, idx):
    """a few of the remaining frames on the CPU (the GPU flavour below runs all 15)"""
    import oracle_cv as ocv
    _check_subsampled_canvas(ocv, idx)


RUN2 = [int(i) for i in ri.fixture()[0]["run2_frames"]]


@pytest.mark.parametrize("idx", RUN2[:2])
def test_oracle_reproduces_the_second_recorded_run(oracle, idx):
    """other cameras (another matcher), compose_megapix 1 (another INTER_AREA factor), no mirroring, 3494x3453 panorama"""
    import oracle_cv as ocv
    _check_subsampled_canvas(ocv, idx, run=2)


NIGHT = [int(i) for i in ri.fixture(3)[0]["frames"]]


@pytest.mark.parametrize("idx", NIGHT[:1])
def test_oracle_reproduces_the_recorded_night_run(oracle, idx):
    """example_06: 5184x3456 photographs taken at dusk (stars), decimated by INTER_AREA to 0.6 MPix and stretched with the run's
    black / white point (12, 100) (sde.py:1711), the run's own 21 cameras: 99.99 % of the samples identical to OpenCV's output"""
    import oracle_cv as ocv
    _check_subsampled_canvas(ocv, idx, run=3)


# ---- GPU: the HIP library through the same bodies, and bit for bit against the oracle -------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("idx", LOSSLESS)
def test_hip_reproduces_opencv_recorded_warp(oracle, idx):
    import opencv_starry_sky_panorama_stitcher_amd as cv
    import oracle_cv as ocv

    _check_canvas(cv, idx)
    assert np.array_equal(ri.timelapse_canvas(cv, idx)[0], ri.timelapse_canvas(ocv, idx)[0])


@pytest.mark.gpu
@pytest.mark.parametrize("idx", SUBSAMPLED)
def test_hip_reproduces_opencv_recorded_warp_other_frames(idx):
    import opencv_starry_sky_panorama_stitcher_amd as cv

    _check_subsampled_canvas(cv, idx)


@pytest.mark.gpu
@pytest.mark.parametrize("idx", RUN2)
def test_hip_reproduces_the_second_recorded_run(idx):
    import opencv_starry_sky_panorama_stitcher_amd as cv

    _check_subsampled_canvas(cv, idx, run=2)


@pytest.mark.gpu
@pytest.mark.parametrize("idx", NIGHT)
def test_hip_reproduces_the_recorded_night_run(idx):
    import opencv_starry_sky_panorama_stitcher_amd as cv

    _check_subsampled_canvas(cv, idx, run=3)


@pytest.mark.gpu
def test_hip_panorama_agrees_with_recorded_jpeg(oracle):
    import opencv_starry_sky_panorama_stitcher_amd as cv
    import oracle_cv as ocv

    res = ri.panorama(cv)
    _check_panorama(res)
    ref = ri.panorama(ocv)
    assert np.array_equal(res.result_mask, ref.result_mask) and np.array_equal(res.mosaic, ref.mosaic)
