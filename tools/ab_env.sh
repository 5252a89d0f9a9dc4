#!/bin/bash
# usage: ab_env.sh VAR valA valB rounds [bench args]
V=$1; A=$2; B=$3; R=$4; shift 4
for i in $(seq 1 $R); do for x in $A $B; do
  out=$(env $V=$x timeout -k 10 300 python bench.py --quick 1 "$@" 2>/dev/null | tail -1)
  echo "$V=$x $i $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ok" if (d.get("self_check") or {}).get("mosaic_identical") else "SELF_CHECK_FAILED", {k["kernel"]: round(k["avg_us"] * k["launches_per_step"],1) for k in d.get("kernels",[])})' 2>/dev/null || echo FAILED)"
done; done
