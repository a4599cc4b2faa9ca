#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
bash scripts/pmc_kernel.sh maze_m9 maze_step scripts/bench_families.py --families maze64_m9 --steps 120 --warmup 10 2>&1 | grep -A45 "maze_step" | grep "SQ_\|VGPR\|hbm\|LDS\|Scratch\|maze_step\|GRBM" | head -45
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/pmc_maze_m9_g0/**/*kernel_trace.csv", recursive=True):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in d.items():
        if "maze" in k:
            print("%-70s n=%d avg %.2f us" % (k[:70], len(v), sum(v) / len(v) / 1e3))
PY
