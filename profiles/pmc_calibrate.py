import sys, ctypes as C
sys.path.insert(0, '.')
import opencv_starry_sky_panorama_stitcher_amd as cv
L = cv._lib.lib(); cv._lib.check(L.ssp_init(0))
for w in (4, 8, 16):
    cv._lib.check(L.ssp_calibrate_stream(w, 512 << 20, 2))
print("calibration streams done")
