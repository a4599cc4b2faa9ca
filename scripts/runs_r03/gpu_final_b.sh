#!/bin/bash
# round 3, the bench lines of record (after gpu_final.sh): default flags, the driver's flags, graph forced on / off, and
# the rocprofv3 kernel stats of the same command.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${1:-finalb}
echo "== bench default"; timeout 900 python bench.py --fused > gpurun_out/${T}_bench_2a.json 2> gpurun_out/${T}_bench_2a.err; echo "rc=$?"
echo "== bench driver flags"; timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench_2a_steps20.json 2> gpurun_out/${T}_bench_2a_steps20.err; echo "rc=$?"
echo "== bench plain launches (2000 steps)"; timeout 900 python bench.py --graph off --no-cpu-baseline --no-families > gpurun_out/${T}_bench_2a_plain.json 2>/dev/null; echo "rc=$?"
echo "== bench driver flags, graph off"; timeout 900 python bench.py --steps 20 --warmup 5 --graph off --no-cpu-baseline --no-families > gpurun_out/${T}_bench_2a_steps20_plain.json 2>/dev/null; echo "rc=$?"
python - <<PY
import json
for f in ("bench_2a", "bench_2a_steps20", "bench_2a_plain", "bench_2a_steps20_plain"):
    try:
        d = json.load(open("gpurun_out/${T}_%s.json" % f))
    except Exception as ex:
        print(f, "unreadable", ex); continue
    r = d["roofline"]
    print("%-22s value %.4e ms/step %.5f kernel us %.3f frac %.3f traffic %s launch: %s" % (f, d["value"], d["ms_per_step"], r["avg_launch_us"], r["frac"], r.get("traffic"), d["config"]["launch"][-40:]), d.get("fused_rollout_env_steps_per_s_rank0"))
    print("   ", r["kernel"][:60], (d.get("cpu_baseline") or {}).get("value"))
    for k, v in (d.get("families") or {}).items():
        print("   ", k, {a: v.get(a) for a in ("ms_per_step", "env_steps_per_s", "wall_s", "error")}, (v.get("roofline") or {}).get("frac"))
PY
echo "== rocprof bench (default flags minus the CPU leg)"
rm -rf gpurun_out/prof_b
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b -o b -- python3 bench.py --no-cpu-baseline --no-families > gpurun_out/${T}_prof_b.json 2> gpurun_out/${T}_prof_b.err; echo "rc=$?"
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/prof_b/**/b_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
keep = [r for r in rows if "anymdp" in r["Name"]]
with open("gpurun_out/${T}_kernel_stats_anymdp_2a.csv", "w", newline="") as o:
    w = csv.DictWriter(o, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(keep)
for r in keep:
    if "step" in r["Name"]:
        print("%-100s calls %6s avg %10.1f ns" % (r["Name"][:100], r["Calls"], float(r["AverageNs"])))
PY
grep -o '"avg_launch_us": [0-9.]*' gpurun_out/${T}_prof_b.json
