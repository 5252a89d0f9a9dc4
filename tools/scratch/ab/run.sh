#!/bin/bash
# A/B of two library builds on the same box: tools/scratch/ab/lib_{A,B}.so, alternating, N rounds
set -e
L=opencv_starry_sky_panorama_stitcher_amd/libssp_hip.so
cp $L /tmp/lib_orig.so
mkdir -p gpurun_out/ab
for r in 1 2 3; do
  for v in A B; do
    cp tools/scratch/ab/lib_$v.so $L
    timeout -k 10 300 python bench.py --cpu-baseline-frames 0 "$@" > gpurun_out/ab/bench_${v}_$r.json 2> gpurun_out/ab/bench_${v}_$r.err
  done
done
cp /tmp/lib_orig.so $L
python - <<'PY'
import json, glob
for v in "AB":
    rows = []
    for f in sorted(glob.glob(f"gpurun_out/ab/bench_{v}_*.json")):
        d = json.loads(open(f).read().strip().splitlines()[-1])
        rows.append((d["ms_per_step"], d["in_flight_2"]["ms_per_step"], (d.get("scale_base") or {}).get("ms_per_step"), {k["kernel"]: round(k["avg_us"], 1) for k in d["kernels"]}))
    for r in rows: print(v, r[0], r[1], r[2], r[3])
PY
