#!/bin/bash
# Any set of PMC counters (one rocprofv3 --pmc pass, kernel trace only) for the kernels matching a substring, averaged per launch:
#   tools/pmc_custom.sh "SQ_WAVES SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM" k_warp_strip [bench.py arguments ...]      (environment passes through)
ctrs=$1; match=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
d=$(mktemp -d /tmp/ssp_pc_XXXX)
timeout -k 10 200 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $d -o p -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-traffic --no-scale-base --no-self-check --frame-sets 1 "$@" > $root/gpurun_out/pmc_custom.log 2>&1
python3 - $d "$match" <<'PY'
import csv, sys, os, collections
d, match = sys.argv[1:3]
agg = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for r in csv.DictReader(open(os.path.join(d, "p_counter_collection.csv"))):
    if match in r["Kernel_Name"]: agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for r in csv.DictReader(open(os.path.join(d, "p_kernel_trace.csv"))):
    if match in r["Kernel_Name"]: dur[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in agg:
    print(k, "n=%d avg=%.1fus" % (len(dur[k]), sum(dur[k]) / len(dur[k])), " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(agg[k].items())))
PY
rm -rf $d
