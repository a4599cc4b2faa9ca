#!/bin/bash
# round 6, visit zz23: rocprofv3 --kernel-trace --stats of the families' bench on the final tree
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
rm -rf $O/prof_fam
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fam -o fam -- python3 scripts/bench_families.py --steps 200 --warmup 20 > $O/zz23_bench_families.jsonl 2> $O/zz23_bench_families.err
echo "families rc=$?"
S=$(ls $O/prof_fam/*kernel_stats.csv $O/prof_fam/*/*kernel_stats.csv 2>/dev/null | head -1)
python3 - "$S" $O/zz23_kernel_stats_families.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if any(k in r["Name"] for k in ("anymdp", "linds", "maze", "cartpole", "acrobot", "mixed"))]
with open(sys.argv[2], "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(keep)
for r in keep[:12]:
    print("  %-80s calls %6s avg %12.1f ns" % (r["Name"][:80], r["Calls"], float(r["AverageNs"])))
PY
rm -rf $O/prof_fam
