#!/bin/bash
# Per-dispatch durations (kernel x grid size), which separates the pyramid levels that share a kernel:   tools/per_dispatch.sh <tag> [bench.py arguments ...]
#   gpurun_out/<tag>_dispatch.txt
set -o pipefail
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rm -rf "$out/${tag}_dprof"
rocprofv3 --kernel-trace -d "$out/${tag}_dprof" -o p --output-format csv -- python3 "$root/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-profile --no-traffic --no-scale-base --no-self-check --quick 1 "$@" > "$out/${tag}_dprof.log" 2>&1 || { tail -5 "$out/${tag}_dprof.log"; exit 1; }
python3 - "$(find "$out/${tag}_dprof" -name 'p_kernel_trace.csv' | head -1)" > "$out/${tag}_dispatch.txt" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
g = collections.OrderedDict()
for r in rows:
    key = (r["Kernel_Name"].split("(")[0][:60], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"])
    g.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("%-60s %22s %6s %6s %9s %9s" % ("kernel", "grid (work-items)", "wg", "calls", "avg us", "min us"))
for k, v in g.items():
    v2 = v[len(v) // 4:]          # the first quarter: warm-up
    print("%-60s %22s %6s %6d %9.1f %9.1f" % (k[0], "x".join(k[1:4]), k[4], len(v), sum(v2) / len(v2), min(v2)))
PY
rm -rf "$out/${tag}_dprof"
tail -40 "$out/${tag}_dispatch.txt"
