"""Footprints of all 16 projections against the reference's recorded warped masks (example_04: 16 runs x 3 frames,
`..._05_masks_untouched/*.jpg` = `warper.warp(mask, K, R, INTER_NEAREST, BORDER_CONSTANT)`, sde.py:1740-1752, kept shrunk to 700 px and
JPEG coded).  A footprint is a function of cameras, frame size and projection only, so the missing photographs do not matter.
Beyond the roi sizes of the KATs this pins, for every projection, WHERE mapBackward lands inside the frame (the validity region
of the warp: horizon cut-offs of the plane-like projections, the curved outlines of fisheye / stereographic / transverse Mercator).
Bar: intersection over union >= 0.999 (the recording's accuracy is about a pixel of outline at its 700-px size)."""
import pytest

import footprints as fp


def _check(cv, kat_id):
    res = fp.footprints(cv, kat_id)
    assert len(res) == 3
    for ours, rec, aspect_ok, warp in res:
        assert aspect_ok, (warp, ours.shape)
        iou = fp.agreement(ours, rec)
        assert iou >= 0.999, (warp, ours.shape, iou)


@pytest.mark.parametrize("kat_id", fp.kat_ids())
def test_oracle_footprints_match_the_recorded_masks(oracle, kat_id):
    import oracle_cv as ocv
    _check(ocv, kat_id)


@pytest.mark.gpu
@pytest.mark.parametrize("kat_id", fp.kat_ids())
def test_hip_footprints_match_the_recorded_masks(kat_id):
    import opencv_starry_sky_panorama_stitcher_amd as cv
    _check(cv, kat_id)
