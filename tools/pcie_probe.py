import sys, time, os, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import opencv_starry_sky_panorama_stitcher_amd as cv
L = cv._lib.lib()
hip = C.CDLL("libamdhip64.so")
frames = [np.random.default_rng(i).integers(0, 255, (2160, 3840, 3), dtype=np.uint8) for i in range(12)]
def up():
    t0 = time.perf_counter(); d = [cv.UMat(f) for f in frames]; L.ssp_sync(); return (time.perf_counter() - t0) * 1e3, d
for i in range(3):
    ms, d = up(); print("upload 12 x 24.9 MB pageable: %.2f ms = %.1f GB/s" % (ms, 298.6 / ms))
big = cv.UMat(np.zeros((2089, 20895, 3), np.uint8))
for i in range(3):
    t0 = time.perf_counter(); h = big.get(); ms = (time.perf_counter() - t0) * 1e3; print("download 131 MB into a fresh array: %.2f ms = %.1f GB/s" % (ms, 130.9 / ms))
out = np.empty((2089, 20895, 3), np.uint8); out[:] = 1
info = big.info()
for i in range(3):
    t0 = time.perf_counter(); rc = L.ssp_image_download(big._h, out.ctypes.data_as(C.c_void_p)); ms = (time.perf_counter() - t0) * 1e3; print("download into a touched array: %.2f ms = %.1f GB/s" % (ms, 130.9 / ms), rc)
for f in frames:
    rc = hip.hipHostRegister(C.c_void_p(f.ctypes.data), C.c_size_t(f.nbytes), C.c_uint(0)); assert rc == 0, rc
for i in range(3):
    ms, d = up(); print("upload 12 x 24.9 MB registered (pinned): %.2f ms = %.1f GB/s" % (ms, 298.6 / ms))
rc = hip.hipHostRegister(C.c_void_p(out.ctypes.data), C.c_size_t(out.nbytes), C.c_uint(0)); assert rc == 0, rc
for i in range(3):
    t0 = time.perf_counter(); rc = L.ssp_image_download(big._h, out.ctypes.data_as(C.c_void_p)); ms = (time.perf_counter() - t0) * 1e3; print("download into a registered array: %.2f ms = %.1f GB/s" % (ms, 130.9 / ms), rc)
