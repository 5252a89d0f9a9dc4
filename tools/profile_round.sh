#!/bin/bash
# The standard evidence set of a build, run on the GPU box:   tools/profile_round.sh <tag> [bench.py arguments ...]
#   gpurun_out/<tag>_bench.json         the bench line (with roofline, cpu_baseline, kernels[], PMC traffic)
#   gpurun_out/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the same command (shorter run, no CPU baseline)
#   gpurun_out/<tag>_pmc_sq.txt         SQ / LDS counters per kernel (tools/pmc_kernels.py)
#   gpurun_out/<tag>_dispatch.txt       per launch: kernel x grid size (tools/per_dispatch.sh)
#   gpurun_out/<tag>_ta.txt             texture-addresser busy share per kernel (tools/pmc_ta.sh)
# Copy what is to be judged from gpurun_out/ into profiles/.
set -o pipefail
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
python3 "$root/bench.py" --steps 20 --warmup 5 "$@" > "$out/${tag}_bench.json" 2> "$out/${tag}_bench.err" || { tail -5 "$out/${tag}_bench.err"; exit 1; }
rm -rf "$out/${tag}_prof"
rocprofv3 --kernel-trace --stats -d "$out/${tag}_prof" -o p --output-format csv -- python3 "$root/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-profile --no-traffic --no-scale-base --no-self-check "$@" > "$out/${tag}_prof.log" 2>&1 || { tail -5 "$out/${tag}_prof.log"; exit 1; }
cp "$(find "$out/${tag}_prof" -name 'p_kernel_stats.csv' | head -1)" "$out/${tag}_kernel_stats.csv"
python3 "$root/tools/pmc_kernels.py" --out "$out/${tag}_pmc_sq.txt" -- "$@" > /dev/null 2> "$out/${tag}_pmc.err" || { tail -5 "$out/${tag}_pmc.err"; exit 1; }
# per launch (separates the pyramid levels that share a kernel) and the texture addresser's busy share; both optional evidence: failures are reported, not fatal
"$root/tools/per_dispatch.sh" "$tag" "$@" > /dev/null 2>&1 || echo "per_dispatch.sh failed (see $out/${tag}_dprof.log)"
"$root/tools/pmc_ta.sh" "gpurun_out/${tag}_ta.txt" "$@" > /dev/null 2>&1 || echo "pmc_ta.sh failed (see $out/pmc_ta.log)"
cd /tmp
python3 - "$out/${tag}_bench.json" <<'PY'
import json, sys
b = json.load(open(sys.argv[1]))
print("ms_per_step", b["ms_per_step"], "value", b["value"], "in_flight_2", b.get("in_flight_2"), "scale_base", (b.get("scale_base") or {}).get("ms_per_step"))
print("roofline", {k: b["roofline"][k] for k in ("kernel", "achieved", "frac", "avg_us", "traffic")} if b.get("roofline") else None)
for k in b["kernels"]:
    print("  %-14s x%-4.1f %8.1f us  %7.0f GB/s  traffic %s" % (k["kernel"], k["launches_per_step"], k["avg_us"], k["achieved_GBps"], k.get("hbm_traffic_bytes_per_launch")))
PY
