"""The N>1 path on CPU: world_size-2 ``gloo`` run of the overlap-band exchange protocol (parallel.plan_exchange /
run_exchange) with the CPU oracle as the compute backend, checked against the single-process oracle panorama.
Also unit tests of the sharding/plan geometry."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from opencv_starry_sky_panorama_stitcher_amd import parallel, starfield  # noqa: E402


def test_shard_indices_cover_everything():
    for n in (1, 5, 6, 12, 48, 96):
        for world in (1, 2, 4, 8):
            if world > n:
                continue
            got = [i for r in range(world) for i in parallel.shard_indices(n, world, r)]
            assert got == list(range(n))
            sizes = [len(parallel.shard_indices(n, world, r)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def test_padded_rect_properties():
    pano = (-500, 100, 1837, 611)
    for nb in (1, 3, 5):
        m = 1 << nb
        pw, ph = parallel.padded_pano_size(pano, nb)
        assert pw % m == 0 and ph % m == 0 and 0 <= pw - pano[2] < m
        for corner, size in [((-500, 100), (300, 200)), ((900, 400), (437, 311)), ((100, 250), (640, 360))]:
            x, y, w, h = parallel.padded_rect(corner, size, pano, nb)
            assert x % m == 0 and y % m == 0 and w % m == 0 and h % m == 0
            assert x >= 0 and y >= 0 and x + w <= pw and y + h <= ph
            # the image itself is inside its padded rectangle
            ix, iy = corner[0] - pano[0], corner[1] - pano[1]
            assert x <= ix and y <= iy and x + w >= ix + size[0] and y + h >= iy + size[1]


def test_plan_pairs_symmetric_and_aligned():
    corners = [(0, 0), (300, 10), (600, -5), (900, 7)]
    sizes = [(400, 300)] * 4
    plan = parallel.plan_exchange(corners, sizes, [0, 0, 1, 1], 2, 3)
    assert plan.nb == 3 and plan.pano_roi == (0, -5, 1300, 315)
    assert {(s, d) for s, d, _ in plan.pairs} == {(0, 1), (1, 0)}
    r01 = [r for s, d, r in plan.pairs if (s, d) == (0, 1)][0]
    r10 = [r for s, d, r in plan.pairs if (s, d) == (1, 0)][0]
    assert r01 == r10 and all(v % 8 == 0 for v in r01)
    own = parallel.owner_map(plan)
    assert own.shape == (315, 1300) and set(np.unique(own)) <= {-1, 0, 1}
    assert plan.bytes_sent(0) == plan.bytes_sent(1) > 0


class _OracleBackend:
    """export/import on the oracle blender's full-level accumulators."""

    def __init__(self, blender, torch):
        self.b, self.torch = blender, torch

    def export(self, lvl, rect):
        lap, w = self.b.level(lvl)
        x, y, rw, rh = [v >> lvl for v in rect]
        t = self.torch
        return (t.from_numpy(np.ascontiguousarray(lap[y:y + rh, x:x + rw]).reshape(-1)), t.from_numpy(np.ascontiguousarray(w[y:y + rh, x:x + rw]).reshape(-1)))

    def import_(self, lvl, rect, lap, w):
        full_lap, full_w = self.b.level(lvl)
        x, y, rw, rh = [v >> lvl for v in rect]
        add_l = np.zeros(full_lap.shape, np.int32)
        add_w = np.zeros(full_w.shape, np.float32)
        add_l[y:y + rh, x:x + rw] = lap.numpy().reshape(rh, rw, 3)
        add_w[y:y + rh, x:x + rw] = w.numpy().reshape(rh, rw)
        self.b.addPartial(lvl, add_l, add_w)


def _worker(rank, world, port, tmpdir):
    import torch
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_cv as ocv
    from opencv_starry_sky_panorama_stitcher_amd import compose as cmp

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rig = starfield.make_rig(3, scale_div=8, n_override=5)   # 5 overlapping frames, 27 degree steps
        frames = starfield.make_frames(rig)
        nbands = 3
        owner = [0, 0, 0, 1, 1] if world == 2 else [0] * 5
        warper = ocv.PyRotationWarper(rig.warp, rig.focal)
        rois = [warper.warpRoi((rig.width, rig.height), rig.Ks[i], rig.Rs[i]) for i in range(rig.n)]
        corners, sizes = [r[:2] for r in rois], [r[2:] for r in rois]
        plan = parallel.plan_exchange(corners, sizes, owner, world, nbands)
        assert plan.pano_roi == ocv.detail.resultRoi(corners, sizes)
        # single-process reference (every rank computes it: the rig is tiny)
        ref = cmp.compose_panorama(ocv, frames, rig.Ks, rig.Rs, warp=rig.warp, warper_scale=rig.focal, blend="multiband", num_bands=nbands)
        # this rank's share
        blender = ocv.detail_MultiBandBlender(num_bands=nbands)
        blender.prepare(plan.pano_roi)
        for i in range(rig.n):
            if owner[i] != rank:
                continue
            _, img = warper.warp(frames[i], rig.Ks[i], rig.Rs[i], ocv.INTER_LINEAR, ocv.BORDER_REFLECT)
            _, msk = warper.warp(255 * np.ones(frames[i].shape[:2], np.uint8), rig.Ks[i], rig.Rs[i], ocv.INTER_NEAREST, ocv.BORDER_CONSTANT)
            blender.feed(img.astype(np.int16), msk, corners[i])
        backend = _OracleBackend(blender, torch)

        def make_buffers(lvl, rect):
            n = (rect[2] >> lvl) * (rect[3] >> lvl)
            return torch.empty(n * 3, dtype=torch.int16), torch.empty(n, dtype=torch.float32)

        parallel.run_exchange(plan, rank, backend, dist, make_buffers)
        result, mask = blender.blend(None, None)
        own = parallel.owner_map(plan) == rank
        assert own.any()
        # masks exact; values: integer sums are order independent, the f32 weight sums may differ by 1 ULP in association
        assert np.array_equal(mask[own], ref.result_mask[own])
        d = np.abs(result.astype(np.int32) - ref.result.astype(np.int32))[own]
        assert d.max() <= 1, f"rank {rank}: max diff {d.max()}"
        frac = float((d > 0).mean())
        assert frac < 1e-3, f"rank {rank}: {frac:.2e} of the owned samples differ"
        np.save(os.path.join(tmpdir, f"ok_{rank}.npy"), np.array([d.max(), frac, own.mean()]))
    finally:
        dist.destroy_process_group()


def test_overlap_exchange_world2_gloo(tmp_path):
    import torch.multiprocessing as mp

    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    stats = [np.load(tmp_path / f"ok_{r}.npy") for r in range(2)]
    assert all(s[0] <= 1 for s in stats)
    assert sum(s[2] for s in stats) > 0.8  # the two ranks together own (almost) the whole panorama
