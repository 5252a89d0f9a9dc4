"""Times the DROP-IN form of the loop -- the reference's own call sequence (compose.compose_panorama, object by object, sde.py:1537-1944)
against this package as `cv` -- next to the batched Composer, on the bench rig (6 x 4K).  A side measurement.

    python tools/bench_dropin.py [--reps 5]
"""
import argparse
import importlib.util
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import opencv_starry_sky_panorama_stitcher_amd as cv  # noqa: E402
from opencv_starry_sky_panorama_stitcher_amd import compose as cmp, starfield  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rig, _ = bench.block_rig(starfield, 1, 0, 1)
    frames, seams = starfield.make_frames(rig, want_seam=True)
    L = cv._lib.lib()
    kw = dict(warp=rig.warp, warper_scale=rig.focal, blend="multiband", num_bands=rig.num_bands, seam_frames=seams, seam_aspect=rig.seam_scale, mask_prep=True)
    res = {}
    # (1) ndarray in, ndarray out: exactly what the reference's loop does with cv2 (every call crosses PCIe)
    cmp.compose_panorama(cv, frames, rig.Ks, rig.Rs, **kw)
    t0 = time.perf_counter()
    for _ in range(a.reps):
        r = cmp.compose_panorama(cv, frames, rig.Ks, rig.Rs, **kw)
    res["object_api_host_arrays_ms"] = round((time.perf_counter() - t0) / a.reps * 1e3, 2)
    # (1b) the same calls with the frames (and so every intermediate) as UMats: nothing crosses PCIe
    dev_frames, dev_seams = [cv.UMat(f) for f in frames], [cv.UMat(f) for f in seams]
    kwd = dict(kw, seam_frames=dev_seams)
    rd = cmp.compose_panorama(cv, dev_frames, rig.Ks, rig.Rs, **kwd)
    L.ssp_sync()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        rd = cmp.compose_panorama(cv, dev_frames, rig.Ks, rig.Rs, **kwd)
    L.ssp_sync()
    res["object_api_umat_ms"] = round((time.perf_counter() - t0) / a.reps * 1e3, 2)
    res["umat_same_mosaic"] = bool(np.array_equal(rd.mosaic.get(), r.mosaic))
    # (2) the batched composer on device-resident frames
    c = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), num_bands=rig.num_bands, mask_prep=True, seam_size=rig.seam_size,
                     seam_aspect=rig.seam_scale)
    dev = [cv.UMat(f) for f in frames]
    c.run(dev); L.ssp_sync()
    t0 = time.perf_counter()
    for _ in range(20):
        c.run(dev)
    L.ssp_sync()
    res["composer_device_resident_ms"] = round((time.perf_counter() - t0) / 20 * 1e3, 3)
    mo = c.result()[0].get()
    res["same_mosaic"] = bool(np.array_equal(mo, r.mosaic))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
