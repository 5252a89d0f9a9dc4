#!/bin/bash
# The debug build tools/trace_warp.py loads: the library with ssp_warp.hip compiled with -DWS_TRACE (s_memtime stamps in the strip kernels), as
# build/v/libssp_trace.so.  Never shipped, never loaded unless SSP_LIB names it.   tools/build_trace_lib.sh   (here, no GPU needed)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
cd "$root/opencv_starry_sky_panorama_stitcher_amd/csrc"
make > /dev/null
mkdir -p "$root/build/v"
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fvisibility=hidden -Wno-unused-function -fno-slp-vectorize"
/opt/rocm/bin/hipcc $F -DWS_TRACE -c ssp_warp.hip -o /tmp/ssp_warp_trace.o
objs=$(ls *.o | grep -v '^ssp_warp.o$' | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/build/v/libssp_trace.so" $objs /tmp/ssp_warp_trace.o
echo "$root/build/v/libssp_trace.so"
