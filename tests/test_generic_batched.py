"""The batched (LDS-staged) warp for the thirteen projections without separable tables -- `fisheye` first, the reference's default
(stitching_detailed_enhanced.py:237) -- through per-part coordinate planes (csrc/ssp_warp.hip k_warp_cmap_batch), and the same planes for the
separable projections (SSP_WARP_CMAP=1).  Everything against the oracle running the reference's call sequence, bit for bit.
Run on the GPU box:  python -m pytest tests -m gpu -x -q
"""
import numpy as np
import pytest

import opencv_starry_sky_panorama_stitcher_amd as cv
from opencv_starry_sky_panorama_stitcher_amd import compose as cmp
from opencv_starry_sky_panorama_stitcher_amd import parallel, starfield
from opencv_starry_sky_panorama_stitcher_amd.starfield import Rig, _finish

import oracle_cv as ocv

pytestmark = pytest.mark.gpu

GENERIC = ["fisheye", "stereographic", "plane", "compressedPlaneA2B1", "compressedPlaneA1.5B1", "compressedPlanePortraitA2B1", "compressedPlanePortraitA1.5B1",
           "paniniA2B1", "paniniA1.5B1", "paniniPortraitA2B1", "paniniPortraitA1.5B1", "transverseMercator"]


def _block(div, warp, bands=4, expos_comp=0, yaw_step=22.0, pitches=(-9.0, 9.0), cols=3):
    yaws, pts = [], []
    for p in pitches:
        for c in range(cols):
            yaws.append((c - (cols - 1) / 2.0) * yaw_step); pts.append(p)
    return _finish(Rig(f"block {warp}", 6, 3840 // div, 2160 // div, 60.0, yaws, pts, warp, "multiband", bands, expos_comp=expos_comp,
                       exposure_spread=(0.8, 1.25) if expos_comp else (1.0, 1.0)))


def _fed_compensator(rig, seams):
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


@pytest.mark.parametrize("warp,comp,prep,div", [("fisheye", 2, True, 8), ("fisheye", 0, False, 4), ("stereographic", 1, True, 8), ("transverseMercator", 4, True, 8),
                                               ("paniniA2B1", 0, True, 8), ("compressedPlanePortraitA1.5B1", 3, False, 8), ("plane", 2, True, 8)])
def test_generic_projection_composer_is_batched_and_matches_oracle(warp, comp, prep, div):
    """2 x 3 frames at 1/8 (1/4) of 4K through the batched path: one fused warp launch with the map read from coordinate planes; gains and mask
    preparation in its epilogue; three panoramas (first: planes, records, rest list; then the steady state)."""
    rig = _block(div, warp, bands=4, expos_comp=comp, yaw_step=14.0 if warp == "plane" else 22.0)
    frames, seams = starfield.make_frames(rig, want_seam=True)
    c = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), blend="multiband", num_bands=4, mask_prep=prep, seam_size=rig.seam_size,
                     seam_aspect=rig.seam_scale, want_result_s16=True)
    if comp:
        c.set_compensator(_fed_compensator(rig, seams))
    dev = [cv.UMat(f) for f in frames]
    names = _profiled_kernels(lambda: [c.run(dev) for _ in range(3)])
    assert "warp_fused" in names and "warp_cmap" in names and "warp_generic" not in names        # the LDS-staged kernel, not the per-frame one
    mo, mk, rs = [u.get() for u in c.result()]
    ref = cmp.compose_panorama(ocv, frames, rig.Ks, rig.Rs, warp=rig.warp, warper_scale=rig.focal, blend="multiband", num_bands=4, expos_comp=comp,
                               seam_frames=seams if (prep or comp) else None, seam_aspect=rig.seam_scale, mask_prep=prep)
    assert c.pano_roi() == ref.pano_roi and np.array_equal(mk, ref.result_mask)
    if comp == 0:
        assert np.array_equal(rs, ref.result) and np.array_equal(mo, ref.mosaic)
    else:
        d = np.abs(mo.astype(np.int16) - ref.mosaic.astype(np.int16))
        assert d.max() <= 1 and (d > 0).mean() < 1e-4


def _profiled_kernels(fn):
    import ctypes as C
    L = cv._lib.lib()
    cv._lib.check(L.ssp_profile_reset()); cv._lib.check(L.ssp_profile_enable(1))
    try:
        fn()
        cv._lib.check(L.ssp_sync())
    finally:
        cv._lib.check(L.ssp_profile_enable(0))
    n = C.c_int()
    cv._lib.check(L.ssp_profile_count(C.byref(n)))
    names = set()
    for i in range(n.value):
        name = C.create_string_buffer(64)
        launches, ms, ab = C.c_int(), C.c_float(), C.c_double()
        cv._lib.check(L.ssp_profile_get(i, name, 64, C.byref(launches), C.byref(ms), C.byref(ab)))
        if launches.value:
            names.add(name.value.decode())
    return names


@pytest.mark.parametrize("seed", list(range(24)))
def test_fuzz_generic_batched(seed):
    """Random small rigs through every non-separable projection (two seeds each): frame sizes either side of the 64 x 16 tiles, 1-4 frames,
    1-5 bands, with / without mask preparation; mosaic, mask and int16 result bit for bit."""
    rng = np.random.default_rng(21000 + seed)
    warp = GENERIC[seed % len(GENERIC)]
    w, h = int(rng.integers(60, 420)), int(rng.integers(40, 260))
    n = int(rng.integers(1, 5))
    step = float(rng.uniform(8, 18))
    yaws = [float((i - (n - 1) / 2) * step + rng.uniform(-2, 2)) for i in range(n)]
    pitches = [float(rng.uniform(-7, 7)) for _ in range(n)]
    bands = int(rng.integers(1, 6))
    prep = bool(seed % 2)
    rig = _finish(Rig(f"fuzz generic {seed}", 9, w, h, 60.0, yaws, pitches, warp, "multiband", bands))
    frames, seams = starfield.make_frames(rig, want_seam=True)
    c = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (w, h), blend="multiband", num_bands=bands, mask_prep=prep, seam_size=rig.seam_size, seam_aspect=rig.seam_scale,
                     want_result_s16=True)
    dev = [cv.UMat(f) for f in frames]
    c.run(dev); c.run(dev)
    mo, mk, rs = [u.get() for u in c.result()]
    ref = cmp.compose_panorama(ocv, frames, rig.Ks, rig.Rs, warp=warp, warper_scale=rig.focal, blend="multiband", num_bands=bands, seam_frames=seams if prep else None,
                               seam_aspect=rig.seam_scale, mask_prep=prep)
    assert c.pano_roi() == ref.pano_roi, (seed, warp)
    assert np.array_equal(mk, ref.result_mask) and np.array_equal(rs, ref.result) and np.array_equal(mo, ref.mosaic), (seed, warp, w, h, n, bands, prep)


@pytest.mark.parametrize("warp", ["spherical", "cylindrical", "mercator"])
def test_separable_projections_through_coordinate_planes(monkeypatch, warp):
    """SSP_WARP_CMAP=1: the separable projections with the map read from coordinate planes instead of computed from their tables -- identical
    output (closed ring with two straddling frames, gain blocks, mask preparation)."""
    from opencv_starry_sky_panorama_stitcher_amd.starfield import _ring
    rig = _finish(Rig("ring", 3, 480, 270, 60.0, _ring(12, 30.0), [0.0, 1.0, -2.0, 0.0, 2.0, -1.0, 0.0, 1.0, 0.0, -1.0, 2.0, 0.0], warp, "multiband", 4, expos_comp=2,
                      exposure_spread=(0.8, 1.25)))
    frames, seams = starfield.make_frames(rig, want_seam=True)
    comp = _fed_compensator(rig, seams)
    dev = [cv.UMat(f) for f in frames]
    outs = []
    for planes in (False, True):
        if planes:
            monkeypatch.setenv("SSP_WARP_CMAP", "1")
        c = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), blend="multiband", num_bands=4, mask_prep=True, seam_size=rig.seam_size,
                         seam_aspect=rig.seam_scale, want_result_s16=True)
        c.set_compensator(comp)
        names = _profiled_kernels(lambda: [c.run(dev) for _ in range(3)])
        assert ("warp_cmap" in names) == planes
        outs.append([u.get() for u in c.result()])
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


def test_coordinate_planes_by_configuration():
    """ssp_compose_config.coordinate_planes (Composer(coordinate_planes=True)): the choice for many panoramas with the same cameras, without the
    environment switch -- the plane kernel runs, same mosaic, mask and int16 result as the table kernel's."""
    rig = _finish(Rig("planes", 3, 480, 270, 60.0, [-30.0, 0.0, 30.0], [1.0, -2.0, 0.5], "spherical", "multiband", 4))
    frames, seams = starfield.make_frames(rig, want_seam=True)
    dev = [cv.UMat(f) for f in frames]
    outs = []
    for planes in (False, True):
        c = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), blend="multiband", num_bands=4, mask_prep=True, seam_size=rig.seam_size,
                         seam_aspect=rig.seam_scale, want_result_s16=True, coordinate_planes=planes)
        names = _profiled_kernels(lambda: [c.run(dev) for _ in range(2)])
        assert ("warp_cmap" in names) == planes
        outs.append([u.get() for u in c.result()])
    assert all(np.array_equal(a, b) for a, b in zip(*outs))


@pytest.mark.parametrize("world,levels", [(2, True), (3, False)])
def test_fisheye_strip_exchange_is_bit_exact(world, levels):
    """The multi-GPU emulation with the reference's default projection: ssp_composer_feed_planes takes fisheye frames (batched path), every
    owned pixel equals the single composer's."""
    nb = 3
    rig = _block(8, "fisheye", bands=nb)
    frames = starfield.make_frames(rig)
    owner = [i * world // rig.n for i in range(rig.n)]
    fp = parallel.feed_parts(cv, rig.warp, rig.focal, (rig.width, rig.height), rig.Ks, rig.Rs, owner, nb)
    plan = parallel.plan_strips(fp.corners, fp.sizes, fp.owner, world, nb, levels=levels, pano_roi=fp.pano_roi)
    dev = [cv.UMat(f) for f in frames]
    full = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), num_bands=nb, want_result_s16=True)
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
    covered = 0
    for r in range(world):
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
