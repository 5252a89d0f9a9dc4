"""Compute part of ONE rank's multi-GPU step on one GPU (no RCCL): feed own frames, export the strips its neighbours need, feed
the strips it would receive (buffer contents are irrelevant for timing), order, collapse its region.

    python tools/bench_strip_rank.py --world 8 --rank 5 [--steps 10]
"""
import argparse
import ctypes as C
import importlib.util
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import opencv_starry_sky_panorama_stitcher_amd as cv  # noqa: E402
from opencv_starry_sky_panorama_stitcher_amd import compose as cmp, parallel, starfield  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=5)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--paired", type=int, default=0)
    ap.add_argument("--profile", type=int, default=1, help="0: time without per-kernel events")
    ap.add_argument("--levels", type=int, default=1, help="1: all-level strips (the receiver builds nothing), 0: level-0 strips (it rebuilds their pyramids)")
    a = ap.parse_args()
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    L = cv._lib.lib()
    rig, layout = bench.block_rig(starfield, a.world, a.rank, 1)
    frames = [cv.UMat(f) for f in starfield.make_frames(rig)]
    # what the composers feed (bench.py does the same): a frame's roi, or the two live ends of a frame that straddles u = +-pi*scale -- the 4 rows x 12
    # yaw positions at 30 degrees of world 8 are four closed rings
    all_Ks, all_Rs, frame_owner = [], [], []
    for r in range(a.world):
        rr, _ = bench.block_rig(starfield, a.world, r, 1)
        all_Ks += rr.Ks; all_Rs += rr.Rs; frame_owner += [r] * rr.n
    fparts = parallel.feed_parts(cv, rig.warp, rig.focal, (rig.width, rig.height), all_Ks, all_Rs, frame_owner, rig.num_bands)
    plan = parallel.plan_strips(fparts.corners, fparts.sizes, fparts.owner, a.world, rig.num_bands, levels=bool(a.levels), pano_roi=fparts.pano_roi)
    comp = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), blend=rig.blend, num_bands=rig.num_bands, mask_prep=True,
                        seam_size=rig.seam_size, seam_aspect=rig.seam_scale)
    ex = parallel.StripExchangeBase(comp, plan, a.rank, parallel._umat_alloc)
    if a.paired and not a.levels:
        # the double-buffered order of parallel.HipStripPipeline: two composers alternate, the strips' pyramids of one panorama and
        # the own pyramids of the next are one chain of launches
        comp2 = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), blend=rig.blend, num_bands=rig.num_bands, mask_prep=True,
                             seam_size=rig.seam_size, seam_aspect=rig.seam_scale)
        exs = [ex, parallel.StripExchangeBase(comp2, plan, a.rank, parallel._umat_alloc)]
        k = [0]

        def step():
            cur, prev = exs[k[0] & 1], exs[(k[0] & 1) ^ 1]
            cur.c.feed_planes(frames)
            cur.export_all()
            started = k[0] > 0
            if started:
                prev.import_strips()
            cv._lib.check(L.ssp_blender_feed_end_pair(cur.c.blender_handle(), prev.c.blender_handle() if started else None))
            if started:
                prev.collapse()
            k[0] += 1
    elif a.levels:
        def step():
            comp.feed_planes(frames)
            comp.feed_pyramids()
            ex.export_all()
            ex.finish(ex.recv_slots())
    else:
        def step():
            comp.feed_planes(frames)
            ex.export_all()
            comp.feed_pyramids()
            ex.finish(ex.recv_slots())

    for _ in range(2):
        step()
    L.ssp_sync()
    cv._lib.check(L.ssp_profile_reset()); cv._lib.check(L.ssp_profile_enable(1 if a.profile else 0))
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    L.ssp_sync()
    dt = (time.perf_counter() - t0) / a.steps
    cv._lib.check(L.ssp_profile_enable(0))
    n = C.c_int()
    cv._lib.check(L.ssp_profile_count(C.byref(n)))
    kern = []
    for i in range(n.value):
        name = C.create_string_buffer(64)
        launches, ms, ab = C.c_int(), C.c_float(), C.c_double()
        cv._lib.check(L.ssp_profile_get(i, name, 64, C.byref(launches), C.byref(ms), C.byref(ab)))
        if launches.value:
            kern.append((name.value.decode(), launches.value / a.steps, round(ms.value * 1e3 / a.steps, 1)))
    # the same box's figure for the 6-frame block alone (bench.py's scale_base): the weak-scaling model is alone / rank step
    rig1, _ = bench.block_rig(starfield, 1, 0, 1)
    frames1 = [cv.UMat(f) for f in starfield.make_frames(rig1)]
    comp1 = cmp.Composer(rig1.warp, rig1.focal, rig1.Ks, rig1.Rs, (rig1.width, rig1.height), blend=rig1.blend, num_bands=rig1.num_bands, mask_prep=True,
                         seam_size=rig1.seam_size, seam_aspect=rig1.seam_scale)
    cv._lib.check(L.ssp_profile_enable(0))
    for _ in range(3):
        comp1.run(frames1)
    L.ssp_sync()
    t1 = time.perf_counter()
    for _ in range(a.steps):
        comp1.run(frames1)
    L.ssp_sync()
    alone = (time.perf_counter() - t1) / a.steps
    print(json.dumps({"world": a.world, "rank": a.rank, "levels": a.levels, "block_alone_ms": round(alone * 1e3, 4), "model_efficiency": round(alone / dt, 3), "ms_per_step_compute_only": round(dt * 1e3, 4), "sent_MB": round(plan.bytes_sent(a.rank) / 1e6, 1),
                      "recv_strips": len(plan.recvs(a.rank)), "region": plan.region[a.rank], "kernels_us_per_step": sorted(kern, key=lambda k: -k[2])}))


if __name__ == "__main__":
    main()
