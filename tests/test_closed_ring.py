"""Closed 360-degree rings (config 3 = 12 frames at 30 degree yaw steps, config 4 = 4 rows x 12 yaw positions).

A frame that straddles u = +-pi*scale gets OpenCV's full-circle roi from warpRoi (sde.py:1696); the Composer feeds such a frame as its two
live column ranges (``Composer.parts``) while corners / sizes / resultRoi stay OpenCV's.  Every test compares with the ORACLE running the
reference's call sequence on the WHOLE rois, bit for bit (+-1 LSB where double-summed gains enter, as everywhere else in the suite).
Run on the GPU box:  python -m pytest tests -m gpu -x -q
"""
import os

import numpy as np
import pytest

import opencv_starry_sky_panorama_stitcher_amd as cv
from opencv_starry_sky_panorama_stitcher_amd import compose as cmp
from opencv_starry_sky_panorama_stitcher_amd import parallel, starfield
from opencv_starry_sky_panorama_stitcher_amd.starfield import Rig, _finish, _ring

import oracle_cv as ocv

pytestmark = pytest.mark.gpu


def _ring_rig(div, n=12, step=30.0, pitches=None, warp="spherical", bands=5, expos_comp=0, config_id=3):
    return _finish(Rig(f"closed ring {n} x {step} deg", config_id, 3840 // div, 2160 // div, 60.0, _ring(n, step), pitches or [0.0] * n, warp, "multiband", bands,
                       expos_comp=expos_comp, exposure_spread=(0.8, 1.25) if expos_comp else (1.0, 1.0)))


def _fed_compensator(rig, seams):
    """sde.py:1543-1613 on the HIP path: seam-scale warps, compensator.feed."""
    comp = cv.detail.ExposureCompensator_createDefault(rig.expos_comp)
    ws = cv.PyRotationWarper(rig.warp, rig.focal * rig.seam_scale)
    cs, ims, mks = [], [], []
    for i in range(rig.n):
        K = rig.Ks[i].copy(); K[0, 0] *= rig.seam_scale; K[0, 2] *= rig.seam_scale; K[1, 1] *= rig.seam_scale; K[1, 2] *= rig.seam_scale
        cnr, im = ws.warp(seams[i], K, rig.Rs[i], cv.INTER_AREA, cv.BORDER_REFLECT)
        _, mk = ws.warp(255 * np.ones(seams[i].shape[:2], np.uint8), K, rig.Rs[i], cv.INTER_NEAREST, cv.BORDER_CONSTANT)
        cs.append(cnr); ims.append(im); mks.append(mk)
    comp.feed(corners=cs, images=ims, masks=mks)
    return comp


def test_live_parts_of_a_straddling_frame():
    """warpRoi of a straddling frame spans the circle (OpenCV's by-border roi); liveParts returns its two ends, which hold every set mask
    pixel with the blender's reach to spare, and the roi itself for an ordinary frame."""
    rig = _ring_rig(8, bands=4)
    w = cv.PyRotationWarper(rig.warp, rig.focal)
    size = (rig.width, rig.height)
    circle = 2 * np.pi * rig.focal
    for i in range(rig.n):
        roi = w.warpRoi(size, rig.Ks[i], rig.Rs[i])
        parts = w.liveParts(size, rig.Ks[i], rig.Rs[i], 4)
        if i in (0, rig.n - 1):
            assert roi[2] > 0.98 * circle and len(parts) == 2
            a, b = parts
            assert a[0] == roi[0] and b[0] + b[2] == roi[0] + roi[2] and a[1] == b[1] == roi[1] and a[3] == b[3] == roi[3]
            assert a[2] + b[2] < 0.45 * roi[2]
            _, mask = w.warp(255 * np.ones((rig.height, rig.width), np.uint8), rig.Ks[i], rig.Rs[i], cv.INTER_NEAREST, cv.BORDER_CONSTANT)
            cols = np.nonzero(mask.any(axis=0))[0] + roi[0]
            reach = 4 * 16
            inside = np.zeros(len(cols), bool)
            for p in parts:           # a set column lies at least `reach` inside a part, or the part ends where the roi ends
                inside |= ((cols - reach >= p[0]) | (p[0] == roi[0])) & ((cols + reach < p[0] + p[2]) | (p[0] + p[2] == roi[0] + roi[2]))
            assert inside.all() and len(cols) > 100
        else:
            assert parts == [tuple(roi)]
    # more bands than the frame can bear (reach beyond the dead run), or an unknown band count: the whole roi
    assert w.liveParts(size, rig.Ks[0], rig.Rs[0], 9) == [tuple(w.warpRoi(size, rig.Ks[0], rig.Rs[0]))]


@pytest.mark.parametrize("bands,comp,prep,warp", [(3, 0, False, "spherical"), (5, 2, True, "spherical"), (4, 1, True, "cylindrical"), (4, 4, True, "spherical"),
                                                   (5, 3, False, "mercator")])
def test_closed_ring_composer_feeds_live_parts_and_matches_oracle(bands, comp, prep, warp):
    """Config 3's closed ring at 1/8 frame size: 14 feed units for 12 frames, the caller-visible geometry unchanged, mosaic / mask / int16
    result against the oracle (which warps and feeds the two full-circle rois whole, as OpenCV does)."""
    rig = _ring_rig(8, bands=bands, expos_comp=comp, warp=warp, pitches=[0.0, 2.0, -3.0, 1.0, 0.0, -2.0, 3.0, 0.0, 1.0, -1.0, 2.0, 0.0])
    frames, seams = starfield.make_frames(rig, want_seam=True)
    c = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), blend="multiband", num_bands=bands, mask_prep=prep, seam_size=rig.seam_size,
                     seam_aspect=rig.seam_scale, want_result_s16=True)
    if comp:
        c.set_compensator(_fed_compensator(rig, seams))
    parts = c.parts()
    assert [p[0] for p in parts] == [0, 0] + list(range(1, 11)) + [11, 11]
    for k in (0, 1, 12, 13):
        assert parts[k][1][2] < 0.3 * c.image_roi(parts[k][0])[2]
    dev = [cv.UMat(f) for f in frames]
    for _ in range(3):           # first panorama learns the rest list, later ones run the steady-state launches
        c.run(dev)
    mo, mk, rs = [u.get() for u in c.result()]
    ref = cmp.compose_panorama(ocv, frames, rig.Ks, rig.Rs, warp=rig.warp, warper_scale=rig.focal, blend="multiband", num_bands=bands, expos_comp=comp,
                               seam_frames=seams if (prep or comp) else None, seam_aspect=rig.seam_scale, mask_prep=prep)
    assert c.pano_roi() == ref.pano_roi and [c.image_roi(i) for i in range(rig.n)] == [tuple(cn) + tuple(sz) for cn, sz in zip(ref.corners, ref.sizes)]
    assert np.array_equal(mk, ref.result_mask)
    if comp == 0:
        assert np.array_equal(rs, ref.result) and np.array_equal(mo, ref.mosaic)
    else:
        diff = np.abs(mo.astype(np.int16) - ref.mosaic.astype(np.int16))
        assert diff.max() <= 1 and (diff > 0).mean() < 1e-4


def test_closed_ring_split_equals_unsplit_bit_for_bit(monkeypatch):
    """The same closed ring through a composer that feeds whole rois (SSP_NO_SPLIT, the round-3 behaviour): identical outputs, with exposure
    compensation too (no tolerance: both sides apply the same gains)."""
    rig = _ring_rig(8, bands=5, expos_comp=2)
    frames, seams = starfield.make_frames(rig, want_seam=True)
    comp = _fed_compensator(rig, seams)
    dev = [cv.UMat(f) for f in frames]
    outs = []
    for no_split in (False, True):
        if no_split:
            monkeypatch.setenv("SSP_NO_SPLIT", "1")
        c = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), blend="multiband", num_bands=5, mask_prep=True, seam_size=rig.seam_size,
                         seam_aspect=rig.seam_scale, want_result_s16=True)
        c.set_compensator(comp)
        assert len(c.parts()) == (12 if no_split else 14)
        c.run(dev); c.run(dev)
        outs.append([u.get() for u in c.result()])
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


def test_unwritten_plane_bytes_never_reach_the_panorama():
    """far tiles leave the image bytes of the blender's planes unwritten.  With every pool block poisoned (0xFF: -1 / NaN / 255) a far
    tile that did reach the panorama would show; a child process (the switch is read once per process) runs the closed ring both ways."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, hashlib, numpy as np\n"
        f"sys.path.insert(0, {root!r}); sys.path.insert(0, {os.path.join(root, 'tests')!r})\n"
        "import opencv_starry_sky_panorama_stitcher_amd as cv\n"
        "from opencv_starry_sky_panorama_stitcher_amd import compose as cmp, starfield\n"
        "from opencv_starry_sky_panorama_stitcher_amd.starfield import Rig, _finish, _ring\n"
        "rig = _finish(Rig('ring', 3, 480, 270, 60.0, _ring(12, 30.0), [0.0] * 12, 'spherical', 'multiband', 4))\n"
        "frames = starfield.make_frames(rig)\n"
        "c = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), num_bands=4, want_result_s16=True)\n"
        "dev = [cv.UMat(f) for f in frames]\n"
        "c.run(dev); c.run(dev)\n"
        "print('HASH', hashlib.sha256(b''.join(u.get().tobytes() for u in c.result())).hexdigest())\n")
    hashes = []
    for poison in (False, True):
        env = dict(os.environ)
        env.pop("SSP_POOL_POISON", None)
        if poison:
            env["SSP_POOL_POISON"] = "1"
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        hashes.append([ln for ln in r.stdout.splitlines() if ln.startswith("HASH")][0])
    assert hashes[0] == hashes[1]


@pytest.mark.parametrize("world,levels", [(2, True), (3, True), (4, False)])
def test_closed_ring_strip_exchange_is_bit_exact(world, levels):
    """The closed 12-frame ring sharded over 2 / 3 / 4 emulated ranks (contiguous runs of frames): a rank that holds a straddling frame has a
    feed unit at either end of the panorama; plan_strips owns by the ranks' main clusters.  Every owned pixel against the single composer."""
    nb = 3
    rig = _ring_rig(8, bands=nb)
    frames = starfield.make_frames(rig)
    owner = [i * world // rig.n for i in range(rig.n)]
    fp = parallel.feed_parts(cv, rig.warp, rig.focal, (rig.width, rig.height), rig.Ks, rig.Rs, owner, nb)
    assert len(fp.corners) == 14 and fp.image == [0, 0] + list(range(1, 11)) + [11, 11]
    plan = parallel.plan_strips(fp.corners, fp.sizes, fp.owner, world, nb, levels=levels, pano_roi=fp.pano_roi)
    dev = [cv.UMat(f) for f in frames]
    full = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), num_bands=nb, want_result_s16=True)
    assert full.pano_roi() == plan.pano_roi
    full.run(dev)
    ref_mos, ref_mask, ref_res = [u.get() for u in full.result()]
    exs, per_rank = [], []
    for r in range(world):
        idx = [i for i in range(rig.n) if owner[i] == r]
        c = cmp.Composer(rig.warp, rig.focal, [rig.Ks[i] for i in idx], [rig.Rs[i] for i in idx], (rig.width, rig.height), num_bands=nb, want_result_s16=True)
        exs.append(parallel.StripExchangeBase(c, plan, r, parallel._umat_alloc))
        per_rank.append([dev[i] for i in idx])
    parallel.emulate_strip_exchange(exs, per_rank)
    own = parallel.strip_owner_map(plan)
    assert np.all((own >= 0) | (ref_mask == 0))
    covered = 0
    for r in range(world):
        assert plan.region[r][2] < 0.8 * plan.padded[0]          # nobody collapses the whole circle
        mos, mk, rs = [u.get() for u in exs[r].c.result()]
        x0, y0 = plan.region[r][0], plan.region[r][1]
        hh, ww = mk.shape
        sel = own[y0:y0 + hh, x0:x0 + ww] == r
        assert int(sel.sum()) == int((own == r).sum()) > 0
        covered += int(sel.sum())
        assert np.array_equal(mk[sel], ref_mask[y0:y0 + hh, x0:x0 + ww][sel])
        assert np.array_equal(rs[sel], ref_res[y0:y0 + hh, x0:x0 + ww][sel])
        assert np.array_equal(mos[sel], ref_mos[y0:y0 + hh, x0:x0 + ww][sel])
    assert covered == int((own >= 0).sum())


@pytest.mark.parametrize("div,nb", [(8, 3), (1, 5)])
def test_config4_closed_layout_eight_ranks_against_the_oracle(div, nb):
    """SURVEY 8(d)'s config 4 as written: 48 frames = 4 rows (pitch -30, -10, 10, 30 degrees) x 12 yaw positions at 30 degree steps -- four
    closed rings, eight frames straddling u = +-pi*scale -- a 2x3 block of 6 frames per GPU (bench.py's N = 8 layout), at 1/8 frame size and at
    FULL size (48 x 3840x2160, 5 bands).  The eight ranks of the strip exchange emulated on this GPU, every owned pixel (mosaic, mask, int16
    result) against the ORACLE's panorama of all 48 frames through the reference's call sequence."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod_ring4", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    world = 8
    rigs = [bench.block_rig(starfield, world, r, div)[0] for r in range(world)]
    w, h = rigs[0].width, rigs[0].height
    Ks, Rs, owner, frames = [], [], [], []
    for r, rg in enumerate(rigs):
        Ks += rg.Ks; Rs += rg.Rs; owner += [r] * rg.n
        rg.config_id = 40 + r
        frames += starfield.make_frames(rg)
    assert len(frames) == 48
    yaws = sorted({round(y, 3) for rg in rigs for y in rg.yaws_deg})
    assert len(yaws) == 12 and abs((yaws[1] - yaws[0]) - 30.0) < 1e-6 and abs(yaws[0] + 165.0) < 1e-6      # the closed ring of SURVEY 8(d)
    fp = parallel.feed_parts(cv, rigs[0].warp, rigs[0].focal, (w, h), Ks, Rs, owner, nb)
    assert len(fp.corners) == 48 + 8
    plan = parallel.plan_strips(fp.corners, fp.sizes, fp.owner, world, nb, pano_roi=fp.pano_roi)
    dev = [cv.UMat(f) for f in frames]
    exs, per_rank = [], []
    for r in range(world):
        idx = [i for i in range(len(owner)) if owner[i] == r]
        c = cmp.Composer(rigs[0].warp, rigs[0].focal, [Ks[i] for i in idx], [Rs[i] for i in idx], (w, h), num_bands=nb, want_result_s16=True)
        exs.append(parallel.StripExchangeBase(c, plan, r, parallel._umat_alloc))
        per_rank.append([dev[i] for i in idx])
    parallel.emulate_strip_exchange(exs, per_rank)
    ref = cmp.compose_panorama(ocv, frames, Ks, Rs, warp=rigs[0].warp, warper_scale=rigs[0].focal, blend="multiband", num_bands=nb, mask_prep=False)
    assert tuple(ref.pano_roi) == tuple(plan.pano_roi) and ref.num_bands == nb
    own = parallel.strip_owner_map(plan)
    covered = 0
    for r in range(world):
        assert plan.region[r][2] < 0.6 * plan.padded[0]
        mos, mk, rs = [u.get() for u in exs[r].c.result()]
        x0, y0 = plan.region[r][0], plan.region[r][1]
        hh, ww = mk.shape
        sel = own[y0:y0 + hh, x0:x0 + ww] == r
        assert int(sel.sum()) == int((own == r).sum()) > 0
        covered += int(sel.sum())
        assert np.array_equal(mk[sel], ref.result_mask[y0:y0 + hh, x0:x0 + ww][sel])
        assert np.array_equal(rs[sel], ref.result[y0:y0 + hh, x0:x0 + ww][sel])
        assert np.array_equal(mos[sel], ref.mosaic[y0:y0 + hh, x0:x0 + ww][sel])
    assert covered == int((own >= 0).sum()) and np.all((own >= 0) | (ref.result_mask == 0))
