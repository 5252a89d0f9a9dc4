"""Times the frame prologue (INTER_AREA decimation + black / white point stretch, sde.py:1699-1711) on one GPU with the
library's hipEvent profile, next to the CPU oracle.  A side measurement: bench.py's JSON line stays the headline metric.

    python tools/bench_prologue.py [--w 5184 --h 3456 --megapix 0.6 --reps 50]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import opencv_starry_sky_panorama_stitcher_amd as cv  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--w", type=int, default=5184)
    ap.add_argument("--h", type=int, default=3456)
    ap.add_argument("--megapix", type=float, default=0.6)
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    from util import big_frame
    img = big_frame(a.w, a.h, seed=1)
    scale = min(1.0, float(np.sqrt(a.megapix * 1e6 / (a.w * a.h))))
    L = cv._lib.lib()
    dev = cv.UMat(img)
    out = cv.prepare_frame(dev, scale, (0, 150))
    L.ssp_sync()
    cv._lib.check(L.ssp_profile_reset())
    cv._lib.check(L.ssp_profile_enable(1))
    for _ in range(a.reps):
        out = cv.prepare_frame(dev, scale, (0, 150))
    L.ssp_sync()
    cv._lib.check(L.ssp_profile_enable(0))
    n = C.c_int()
    cv._lib.check(L.ssp_profile_count(C.byref(n)))
    res = {"workload": f"prologue {a.w}x{a.h} -> {out.get().shape[1]}x{out.get().shape[0]} (compose_scale {scale:.4f}) + bw point (0,150)", "kernels": []}
    for i in range(n.value):
        name = C.create_string_buffer(64)
        launches, ms, ab = C.c_int(), C.c_float(), C.c_double()
        cv._lib.check(L.ssp_profile_get(i, name, 64, C.byref(launches), C.byref(ms), C.byref(ab)))
        if launches.value:
            us = ms.value * 1e3 / launches.value
            res["kernels"].append({"name": name.value.decode(), "us": round(us, 2), "algo_GBps": round(ab.value / launches.value / us / 1e3, 1),
                                   "MPix_per_s_source": round(a.w * a.h / us, 1)})
    if not a.no_cpu:
        import oracle_cv as ocv
        t0 = time.perf_counter()
        ref = ocv.prepare_frame(img, scale, (0, 150))
        dt = time.perf_counter() - t0
        res["cpu_oracle_ms"] = round(dt * 1e3, 1)
        res["bit_exact"] = bool(np.array_equal(ref, out.get()))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
