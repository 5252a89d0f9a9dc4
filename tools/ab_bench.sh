#!/bin/bash
# A/B two builds of libssp_hip.so on ONE box (boxes differ by +-4 %): alternating bench runs, the shipped library is never overwritten
# (the variant is selected with SSP_LIB).   usage: tools/ab_bench.sh <libA.so> <libB.so> [rounds] [bench args...]
set -u
A=$1; B=$2; R=${3:-3}; shift 3 2>/dev/null || shift $#
for i in $(seq 1 "$R"); do
  for v in A B; do
    lib=$A; [ $v = B ] && lib=$B
    out=$(SSP_LIB=$lib timeout -k 10 300 python bench.py --quick 1 "$@" 2>/dev/null | tail -1)
    echo "$v $i $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "self_check_ok" if (d.get("self_check") or {}).get("mosaic_identical") else "SELF_CHECK_FAILED", {k["kernel"]: round(k["avg_us"] * k["launches_per_step"],1) for k in d.get("kernels",[])})' 2>/dev/null || echo FAILED)"
  done
done
