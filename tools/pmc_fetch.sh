#!/bin/bash
# HBM-side read / write bytes per launch of the kernels matching a substring: rocprofv3 --pmc FETCH_SIZE, then WRITE_SIZE (own passes, kernel trace
# only; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950) on a short bench run.  Environment variables pass through (kernel variants).
#   tools/pmc_fetch.sh <kernel substring> [bench.py arguments ...]
match=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  d=$(mktemp -d /tmp/ssp_fs_XXXX)
  timeout -k 10 150 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $d -o p -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-traffic --no-scale-base --no-self-check --frame-sets 1 "$@" > $root/gpurun_out/pmc_fetch.log 2>&1
  python3 - $d $ctr "$match" <<'PY'
import csv, sys, os
d, ctr, match = sys.argv[1:4]
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(os.path.join(d, "p_counter_collection.csv"))) if r["Counter_Name"] == ctr and match in r["Kernel_Name"]]
t = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(os.path.join(d, "p_kernel_trace.csv"))) if match in r["Kernel_Name"]]
f = 2048.0 if ctr == "FETCH_SIZE" else 1024.0
print(f"{match} {ctr}: n={len(v)} {'read' if ctr == 'FETCH_SIZE' else 'write'} {sum(v) / max(len(v), 1) * f / 1e6:.1f} MB per launch, avg {sum(t) / max(len(t), 1):.1f} us")
PY
  rm -rf $d
done
