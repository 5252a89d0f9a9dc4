#!/bin/bash
# texture-addresser busy share per kernel: rocprofv3 --pmc TA_TA_BUSY_sum (own pass, kernel trace only) on a short bench run
#   tools/pmc_ta.sh <out.txt> [bench.py arguments ...]
out=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
d=$(mktemp -d /tmp/ssp_ta_XXXX)
timeout -k 10 150 rocprofv3 --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum --kernel-trace --output-format csv -d $d -o p -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-traffic --no-scale-base --no-self-check --frame-sets 1 "$@" > $root/gpurun_out/pmc_ta.log 2>&1
python3 - $d "$root/$out" <<'PY'
import csv, sys, collections, os
d, out = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for r in csv.DictReader(open(os.path.join(d, "p_counter_collection.csv"))):
    agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for r in csv.DictReader(open(os.path.join(d, "p_kernel_trace.csv"))):
    dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
lines = ["ta_busy = TA_TA_BUSY_sum / (256 CUs x duration x 2.4 GHz); wavefront counts are wave-level vector-memory instructions per launch"]
for k in sorted(agg, key=lambda k: -sum(dur[k])):
    m = {c: sum(v) / len(v) for c, v in agg[k].items()}; us = sum(dur[k]) / len(dur[k])
    lines.append(f"{k.split('(')[0][:44]:44s} n={len(dur[k]):3d} avg={us:8.1f}us ta_busy={m.get('TA_TA_BUSY_sum', 0) / (256 * us * 2.4e3):.2f} flat_rd={m.get('TA_FLAT_READ_WAVEFRONTS_sum', 0):9.0f} flat_wr={m.get('TA_FLAT_WRITE_WAVEFRONTS_sum', 0):9.0f} buffer={m.get('TA_BUFFER_WAVEFRONTS_sum', 0):9.0f}")
open(out, "w").write("\n".join(lines) + "\n"); print("\n".join(lines))
PY
rm -rf $d
