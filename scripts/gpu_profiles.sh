#!/bin/bash
# Kernel-trace summaries of every family (bench_families) and the PMC traffic of the AnyMDP step kernel in the
# shared-task configuration (2b); the 2a traffic comes from gpu_pmc.sh with its defaults.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof_fam
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_fam -o fam -- python3 scripts/bench_families.py --steps 200 --warmup 20 > gpurun_out/families.jsonl 2> gpurun_out/families.err
echo "families rc=$?"; cat gpurun_out/families.jsonl | cut -c1-300
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/prof_fam/fam_kernel_stats.csv")))
keep = [r for r in rows if any(k in r["Name"] for k in ("anymdp", "linds", "maze", "cartpole", "acrobot"))]
with open("gpurun_out/families_kernel_stats.csv", "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(keep)
for r in keep:
    print("%-90s calls %6s avg %12.1f ns" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])))
PY
bash scripts/gpu_pmc.sh "--steps 200 --warmup 20 --no-cpu-baseline --tasks 1024" 2b
