"""Where a wave of the coordinate-plane strip kernel spends its time: s_memtime stamps written by a debug build (csrc built with -DWS_TRACE into
build/v/libssp_trace.so, never shipped), config 3 closed ring, third panorama.   tools/build_trace_lib.sh here, then on the GPU box: python tools/trace_warp.py   (SSP_WARP_CMAP=0: the table kernel)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SSP_LIB", os.path.join(ROOT, "build/v/libssp_trace.so"))
os.environ.setdefault("SSP_WARP_CMAP", "1")       # SSP_WARP_CMAP=0: the table kernel (k_warp_strip_batch), whose stamps sit at other places
TABLES = os.environ["SSP_WARP_CMAP"] == "0"
if TABLES:
    del os.environ["SSP_WARP_CMAP"]
import numpy as np  # noqa: E402

import opencv_starry_sky_panorama_stitcher_amd as cv  # noqa: E402
from opencv_starry_sky_panorama_stitcher_amd import compose as cmp, starfield  # noqa: E402

L = cv._lib.lib()
rig = starfield.make_rig(int(sys.argv[1]) if len(sys.argv) > 1 else 3, scale_div=1)
frames, seams = starfield.make_frames(rig, want_seam=True)
dev = [cv.UMat(f) for f in frames]
c = cmp.Composer(rig.warp, rig.focal, rig.Ks, rig.Rs, (rig.width, rig.height), blend="multiband", num_bands=rig.num_bands, mask_prep=True, seam_size=rig.seam_size,
                 seam_aspect=rig.seam_scale)
for _ in range(3):
    c.run(dev)
L.ssp_sync()
n = 32768 * 4
buf = np.zeros((n, 32), np.uint32)
rc = L.ssp_debug_warp_trace(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.nbytes))
assert rc == 0, rc
t = buf.astype(np.int64)
live = (t[:, 0] > 0) & (t[:, 3] != 0)
t = t[live]
# 32-bit stamps: differences modulo 2^32
def d(a, b):
    return (a - b) & 0xffffffff
print("waves traced", len(t))
tick = 1.0
def stat(name, v):
    v = v[(v >= 0) & (v < 10_000_000)]
    print(f"  {name:34s} mean {v.mean() * tick:9.1f}  median {np.median(v) * tick:9.1f}  p90 {np.percentile(v, 90) * tick:9.1f}  n {len(v)}")
stat("whole wave", d(t[:, 3], t[:, 0]))
stat("set-up to the fence", d(t[:, 1], t[:, 0]))
stat("slots, first copies issued", d(t[:, 2], t[:, 1]))
for k in range(4):
    A, B, Cc, D, E, F = (t[:, 4 + 6 * k + j] for j in range(6))
    has = (A > 0) & (F > 0)
    prev = t[:, 2] if k == 0 else t[:, 9 + 6 * (k - 1)]
    if not has.any():
        continue
    print(f" tile {k}: {int(has.sum())} waves")
    if TABLES:
        stat("map arithmetic (column tables, divisions)", d(A, prev)[has])
        stat("wait vmcnt(0): rectangle + previous stores", d(B, A)[has])
        stat("barrier", d(Cc, B)[has])
    else:
        stat("wait: coordinates + rectangle", d(A, prev)[has])
        stat("barrier", d(B, A)[has])
        stat("issue next copy / coordinates, previous stores", d(Cc, B)[has])
    lv = has & (D > 0)
    stat("taps from LDS", d(D, Cc)[lv])
    stat("gain, packing, mask preparation", d(E, D)[lv])
    stat("next copy issued, stores" if TABLES else "to the end of the tile", d(F, E)[has & (E > 0)])
