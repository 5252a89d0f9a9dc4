"""DpSeamFinder('COLOR_GRAD') on the reference's recorded daylight run (21 seam-scale frames): HIP library against the scalar oracle.
    python tools/bench_seam_dp.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import opencv_starry_sky_panorama_stitcher_amd as cv  # noqa: E402
import oracle_cv as ocv  # noqa: E402
import test_seam_dp as t  # noqa: E402

corners, images, masks = t.recorded_seam_inputs(ocv)
imf = [im.astype(np.float32) for im in images]
t0 = time.time(); want = ocv.detail_DpSeamFinder("COLOR_GRAD").find(imf, corners, masks); t_or = time.time() - t0
dev_i, dev_m = [cv.UMat(im) for im in imf], None      # float32, as the reference passes them (sde.py:1601-1604)
best = 1e9
for _ in range(3):
    dev_m = [cv.UMat(m) for m in masks]
    cv._lib.check(cv._lib.lib().ssp_sync())
    t0 = time.time(); cv.detail_DpSeamFinder("COLOR_GRAD").find(dev_i, corners, dev_m); cv._lib.check(cv._lib.lib().ssp_sync()); best = min(best, time.time() - t0)
same = all(np.array_equal(a.get(), b) for a, b in zip(dev_m, want))
print(f"21 frames ~{images[0].shape[1]}x{images[0].shape[0]}: oracle (1 core) {t_or * 1e3:.0f} ms, HIP library (device-resident images and masks) {best * 1e3:.0f} ms, identical masks: {same}")
