#!/usr/bin/env python3
"""Per-kernel SQ / LDS counters of a bench.py run, one table.

    python tools/pmc_kernels.py [--out FILE] [--match warp] -- [bench.py arguments ...]

Runs `rocprofv3 --pmc <set> --kernel-trace` once per counter set (8 SQ slots per pass; no tracing domain besides the kernel
trace) on `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-traffic --no-scale-base --no-self-check --frame-sets 1 <args>`
and prints, per kernel: launches, average duration, waves, VALU / SALU / VMEM / LDS instructions per wave, the share of the
kernel's wave-cycles spent waiting (SQ_WAIT_ANY), issue-stalled (SQ_WAIT_INST_ANY) or issuing, VALU busy share, LDS bank-conflict
cycles per LDS-active cycle.  Environment variables (SSP_WARP_VARIANT ...) pass through."""
import collections
import csv
import os
import shutil
import subprocess
import sys
import tempfile

SETS = [
    ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU"],
    ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_LDS", "SQ_BUSY_CYCLES"],
]


def main():
    argv = sys.argv[1:]
    out, match = None, ""
    while argv and argv[0] != "--":
        if argv[0] == "--out":
            out = argv[1]; argv = argv[2:]
        elif argv[0] == "--match":
            match = argv[1]; argv = argv[2:]
        else:
            raise SystemExit(__doc__)
    bench_args = argv[1:] if argv else []
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = shutil.which("rocprofv3")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for ctrs in SETS:
        d = tempfile.mkdtemp(prefix="ssp_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", *ctrs, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, os.path.join(root, "bench.py"), "--steps", "3",
               "--warmup", "1", "--no-cpu-baseline", "--no-profile", "--no-traffic", "--no-scale-base", "--no-self-check", "--frame-sets", "1", *bench_args]
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=dict(os.environ, TMPDIR="/tmp"), timeout=300)
        for r in csv.DictReader(open(os.path.join(d, "p_counter_collection.csv"))):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for r in csv.DictReader(open(os.path.join(d, "p_kernel_trace.csv"))):
            dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        shutil.rmtree(d, ignore_errors=True)
    lines = ["rocprofv3 --pmc (two passes) --kernel-trace -- bench.py --steps 3 --warmup 1 " + " ".join(bench_args) + "   env: " +
             " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("SSP_")),
             "wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES (waves parked on s_waitcnt / barrier), stall = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES, valu_busy = 4 SQ_ACTIVE_INST_VALU / "
             "(duration x 2.4 GHz x 1024 SIMDs), lds_conf = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (durations are those of the profiled passes)"]
    for k in sorted(agg, key=lambda k: -sum(dur[k])):
        if match and match not in k:
            continue
        m = {c: sum(v) / len(v) for c, v in agg[k].items()}
        us = sum(dur[k]) / len(dur[k])
        w = max(m.get("SQ_WAVES", 0), 1.0)
        wc = max(m.get("SQ_WAVE_CYCLES", 0), 1.0)
        lines.append(f"{k.split('(')[0][:44]:44s} n={len(dur[k]) // len(SETS):3d} avg={us:8.1f}us waves={w:8.0f} VALU/w={m.get('SQ_INSTS_VALU', 0) / w:6.0f} SALU/w={m.get('SQ_INSTS_SALU', 0) / w:5.0f} "
                     f"VMEM_RD/w={m.get('SQ_INSTS_VMEM_RD', 0) / w:5.1f} VMEM_WR/w={m.get('SQ_INSTS_VMEM_WR', 0) / w:4.1f} LDS/w={m.get('SQ_INSTS_LDS', 0) / w:5.1f} "
                     f"wait={m.get('SQ_WAIT_ANY', 0) / wc:.2f} stall={m.get('SQ_WAIT_INST_ANY', 0) / wc:.2f} valu_busy={4 * m.get('SQ_ACTIVE_INST_VALU', 0) / (us * 2.4e3 * 1024):.2f} "
                     f"lds_conf={m.get('SQ_LDS_BANK_CONFLICT', 0) / max(m.get('SQ_LDS_IDX_ACTIVE', 0), 1):.2f} lds_active={m.get('SQ_LDS_IDX_ACTIVE', 0) / (us * 2.4e3 * 256):.2f} "
                     f"wavecyc/w={4 * wc / w:7.0f}")
    text = "\n".join(lines)
    print(text)
    if out:
        open(out, "w").write(text + "\n")


if __name__ == "__main__":
    main()
