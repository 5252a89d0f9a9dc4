"""The reference's own call sequence (stitching_detailed_enhanced.py:1731-1930) object by object on device-resident arrays.

With ``UMat`` operands the calls of the loop return deferred arrays and ``blender.blend`` runs the recognised sequence as one Composer plan
(opencv_starry_sky_panorama_stitcher_amd/deferred.py).  These tests pin: the plan is taken for the reference's sequence, its output equals
the call-by-call evaluation (SSP_EAGER=1) and the oracle; everything that is NOT that sequence still evaluates correctly call by call.
Run on the GPU box:  python -m pytest tests -m gpu -x -q
"""
import numpy as np
import pytest

import opencv_starry_sky_panorama_stitcher_amd as cv
from opencv_starry_sky_panorama_stitcher_amd import compose as cmp
from opencv_starry_sky_panorama_stitcher_amd import deferred, starfield
from opencv_starry_sky_panorama_stitcher_amd.starfield import Rig, _finish, _ring

import oracle_cv as ocv

pytestmark = pytest.mark.gpu


def _rig(n=5, step=27.0, div=8, warp="spherical", blend="multiband", bands=4, expos_comp=0, dtype="u8"):
    return _finish(Rig("drop-in", 3, 3840 // div, 2160 // div, 60.0, _ring(n, step), [0.0, 2.0, -3.0, 1.0, -1.0, 0.5, 2.5, -2.0, 0.0, 1.0, -1.5, 0.0][:n], warp, blend, bands,
                       expos_comp=expos_comp, exposure_spread=(0.8, 1.25) if expos_comp else (1.0, 1.0), dtype=dtype))


def _run(cvmod, rig, frames, seams, **kw):
    return cmp.compose_panorama(cvmod, frames, rig.Ks, rig.Rs, warp=rig.warp, warper_scale=rig.focal, blend=rig.blend,
                                num_bands=rig.num_bands if rig.blend == "multiband" else None, blend_strength=5.0 if rig.blend == "feather" else None,
                                expos_comp=rig.expos_comp, seam_frames=seams, seam_aspect=rig.seam_scale, **kw)


def _same(a, b, exact=True):
    assert tuple(a.pano_roi) == tuple(b.pano_roi) and [tuple(c) for c in a.corners] == [tuple(c) for c in b.corners]
    g = [x.get() if hasattr(x, "get") else x for x in (a.result, a.result_mask, a.mosaic)]
    w = [x.get() if hasattr(x, "get") else x for x in (b.result, b.result_mask, b.mosaic)]
    assert np.array_equal(g[1], w[1])
    if exact:
        assert np.array_equal(g[0], w[0]) and np.array_equal(g[2], w[2])
    else:
        d = np.abs(g[2].astype(np.int16) - w[2].astype(np.int16))
        assert d.max() <= 1 and (d > 0).mean() < 1e-4


@pytest.mark.parametrize("warp,blend,comp,seam,n,step", [("spherical", "multiband", 2, "no", 5, 27.0), ("spherical", "multiband", 0, "voronoi", 4, 27.0),
                                                        ("cylindrical", "multiband", 1, "no", 4, 25.0), ("fisheye", "multiband", 4, "no", 3, 20.0),
                                                        ("spherical", "feather", 3, "no", 3, 27.0), ("plane", "no", 0, "no", 2, 15.0),
                                                        ("spherical", "multiband", 2, "dp_colorgrad", 4, 27.0), ("spherical", "multiband", 2, "no", 12, 30.0)])
def test_reference_sequence_on_umats_runs_as_one_plan(monkeypatch, warp, blend, comp, seam, n, step):
    """compose_panorama -- the reference's sequence call for call -- with UMat frames: blend() takes the plan; output == call-by-call == oracle
    (gains: +-1 LSB against the oracle, as everywhere; plan against call-by-call: exact)."""
    rig = _rig(n=n, step=step, warp=warp, blend=blend, expos_comp=comp, bands=4)
    frames, seams = starfield.make_frames(rig, want_seam=True)
    dev, dev_s = [cv.UMat(f) for f in frames], [cv.UMat(s) for s in seams]
    before = dict(deferred.stats)
    got = _run(cv, rig, dev, dev_s, seam=seam)
    assert deferred.stats["planned"] == before["planned"] + 1 and deferred.stats["call_by_call"] == before["call_by_call"]
    monkeypatch.setenv("SSP_EAGER", "1")
    eager = _run(cv, rig, dev, dev_s, seam=seam)
    assert deferred.stats["planned"] == before["planned"] + 1
    monkeypatch.delenv("SSP_EAGER")
    _same(got, eager)
    ref = _run(ocv, rig, frames, seams, seam=seam)
    _same(got, ref, exact=(comp == 0))
    # a second panorama through the same sequence reuses the cached plan
    again = _run(cv, rig, dev, dev_s, seam=seam)
    _same(again, got)


def test_float_frames_sequence_runs_as_one_plan():
    rig = _rig(n=3, div=16, bands=3, dtype="f32")
    frames, seams = starfield.make_frames(rig, want_seam=True)
    before = deferred.stats["planned"]
    got = _run(cv, rig, [cv.UMat(f) for f in frames], [cv.UMat(s) for s in seams], float_pyramids=True)
    assert deferred.stats["planned"] == before + 1
    ref = _run(ocv, rig, frames, seams, float_pyramids=True)
    assert np.array_equal(got.result_mask.get(), ref.result_mask) and np.max(np.abs(got.result.get() - ref.result)) <= 1e-3


def test_everything_else_evaluates_call_by_call():
    """Deferred arrays outside the reference's sequence: read back, handed to other functions, fed in other shapes -- same values as eager."""
    rig = _rig(n=3, bands=3)
    frames = starfield.make_frames(rig)
    w = cv.PyRotationWarper(rig.warp, rig.focal)
    K, R = rig.Ks[1], rig.Rs[1]
    f = cv.UMat(frames[1])
    corner, d = w.warp(f, K, R, cv.INTER_LINEAR, cv.BORDER_REFLECT)
    assert isinstance(d, deferred.DeferredUMat) and d.pending and d.dtype == np.uint8
    c2, want = w.warp(frames[1], K, R, cv.INTER_LINEAR, cv.BORDER_REFLECT)          # ndarray in -> ndarray out, eager
    assert corner == c2 and d.shape == want.shape
    assert np.array_equal(d.get(), want) and not d.pending
    ones = cv.UMat(np.full(frames[1].shape[:2], 255, np.uint8))
    _, m = w.warp(ones, K, R, cv.INTER_NEAREST, cv.BORDER_CONSTANT)
    _, m_want = w.warp(np.full(frames[1].shape[:2], 255, np.uint8), K, R, cv.INTER_NEAREST, cv.BORDER_CONSTANT)
    dil = cv.dilate(m, None)                       # deferred of deferred
    rs = cv.resize(dil, (m.shape[1] // 2, m.shape[0] // 2), 0, 0, cv.INTER_LINEAR_EXACT)
    assert np.array_equal(rs.get(), cv.resize(cv.dilate(m_want, None), (m.shape[1] // 2, m.shape[0] // 2), 0, 0, cv.INTER_LINEAR_EXACT))
    a16 = d.astype(np.int16)
    assert np.array_equal(a16.get(), want.astype(np.int16))
    # a blender fed with a mask that is NOT the warp of an all-255 array: no plan, evaluated call by call, same as eager ndarrays
    half = np.full(frames[1].shape[:2], 255, np.uint8); half[:, : half.shape[1] // 2] = 0
    _, mh = w.warp(cv.UMat(half), K, R, cv.INTER_NEAREST, cv.BORDER_CONSTANT)
    _, mh_want = w.warp(half, K, R, cv.INTER_NEAREST, cv.BORDER_CONSTANT)
    roi = w.warpRoi((rig.width, rig.height), K, R)
    _, d2 = w.warp(f, K, R, cv.INTER_LINEAR, cv.BORDER_REFLECT)
    before = dict(deferred.stats)
    outs = []
    for img, mask in ((d2.astype(np.int16), mh), (want.astype(np.int16), mh_want)):
        b = cv.detail_MultiBandBlender(num_bands=3)
        b.prepare(roi)
        b.feed(img, mask, corner)
        outs.append(b.blend(None, None))
        with pytest.raises(cv.error):
            b.blend(None, None)                   # consumed
    assert deferred.stats["call_by_call"] == before["call_by_call"] + 1 and deferred.stats["planned"] == before["planned"]
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


def test_operand_overwritten_before_evaluation_raises():
    rig = _rig(n=2, bands=3, expos_comp=1)
    frames, seams = starfield.make_frames(rig, want_seam=True)
    comp, masks = cmp.seam_stage(cv, [cv.UMat(s) for s in seams], rig.Ks, rig.Rs, rig.warp, rig.focal, rig.seam_scale, expos_comp=1)
    w = cv.PyRotationWarper(rig.warp, rig.focal)
    _, img = w.warp(cv.UMat(frames[0]), rig.Ks[0], rig.Rs[0], cv.INTER_LINEAR, cv.BORDER_REFLECT)
    img = img.materialize()
    d16 = deferred.DeferredUMat("astype", (img, np.dtype(np.int16)), img.shape[1], img.shape[0], 3, np.int16)
    comp.apply(0, (0, 0), img, None)              # in place, after the deferred astype read it
    with pytest.raises(cv.error):
        d16.get()


def test_int16_feed_of_an_8bit_image_takes_the_8bit_path_bit_for_bit(monkeypatch):
    """blender.feed(astype(np.int16) of an 8-bit UMat) -- every image the reference feeds, sde.py:1755 -- runs the 8-bit pyramid path
    (ssp_image origin tracking) and defers its pyramids to blend(); identical to the round-3 behaviour (SSP_EAGER_FEED: int16 levels, one chain
    per feed) and to int16 ndarrays; an image modified after astype is fed as it is."""
    monkeypatch.setenv("SSP_EAGER", "1")          # plain eager UMats: the C path is under test here
    rig = _rig(n=4, bands=4)
    frames = starfield.make_frames(rig)
    w = cv.PyRotationWarper(rig.warp, rig.focal)
    rois = [w.warpRoi((rig.width, rig.height), rig.Ks[i], rig.Rs[i]) for i in range(rig.n)]
    pano = cv.detail.resultRoi([r[:2] for r in rois], [r[2:] for r in rois])
    warped, masks = [], []
    for i in range(rig.n):
        c, im, mk = w.warpWithMask(cv.UMat(frames[i]), rig.Ks[i], rig.Rs[i], cv.BORDER_REFLECT)
        warped.append(im); masks.append(mk)

    def blend(feed_imgs):
        b = cv.detail_MultiBandBlender(num_bands=4)
        b.prepare(pano)
        for i, im in enumerate(feed_imgs):
            b.feed(im, masks[i], rois[i][:2])
        return b.blend(None, None)
    via_origin = blend([u.astype(np.int16) for u in warped])
    host16 = blend([u.get().astype(np.int16) for u in warped])
    assert np.array_equal(via_origin[0], host16[0]) and np.array_equal(via_origin[1], host16[1])
    # modified after the conversion: the int16 image no longer equals its origin and must be fed as it is
    s16 = [u.astype(np.int16) for u in warped]
    gain = cv.detail.ExposureCompensator_createDefault(1)
    gain.setMatGains([np.array([[1.5]])] * rig.n)
    gain.apply(0, (0, 0), warped[0], None)        # overwrites the 8-bit original in place
    after = blend(s16)
    assert np.array_equal(after[0], host16[0])
