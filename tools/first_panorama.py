"""What a one-shot caller pays (the reference stitches once per set of frames): device first use, composer creation, and a composer's FIRST
panorama -- which builds the projection / resize tables, the tile records and the rest list that later panoramas reuse -- against its second and
third.   python tools/first_panorama.py   (GPU box; BASELINE config 3 without the compensator)"""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import opencv_starry_sky_panorama_stitcher_amd as cv
from opencv_starry_sky_panorama_stitcher_amd import compose as cmp, starfield
L = cv._lib.lib()
rig = starfield.make_rig(3, scale_div=1)
frames, seams = starfield.make_frames(rig, want_seam=True)
t0 = time.perf_counter(); dev = [cv.UMat(f) for f in frames]; L.ssp_sync(); t1 = time.perf_counter()
print("upload 12 frames (first use of the device: context, pool)", round((t1 - t0) * 1e3, 1), "ms")
for k in range(3):
    t0 = time.perf_counter()
    c = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), blend="multiband", num_bands=5, mask_prep=True, seam_size=rig.seam_size, seam_aspect=rig.seam_scale)
    L.ssp_sync(); t1 = time.perf_counter()
    c.run(dev); mo = c.result()[0]; L.ssp_sync(); t2 = time.perf_counter()
    c.run(dev); mo = c.result()[0]; L.ssp_sync(); t3 = time.perf_counter()
    c.run(dev); mo = c.result()[0]; L.ssp_sync(); t4 = time.perf_counter()
    print("composer", k, "create", round((t1 - t0) * 1e3, 2), "first run", round((t2 - t1) * 1e3, 2), "second", round((t3 - t2) * 1e3, 2), "third", round((t4 - t3) * 1e3, 2), "ms")
    del c
