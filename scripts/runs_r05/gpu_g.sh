#!/bin/bash
# round 5, call G: the bench line with the overlapped step_many (default and driver flags), rocprof stats + overlap summary
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_g
mkdir -p $O
timeout 900 python bench.py > $O/bench_2a.json 2> $O/bench_2a.err; echo "bench rc=$?"
python3 - $O/bench_2a.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1])
print("value %.4g  ms/step %.5f  launch: %s" % (d["value"], d["ms_per_step"], d["config"]["launch"][:90]))
print("roofline", {k: d["roofline"][k] for k in ("frac", "frac_wall", "frac_traffic", "avg_launch_us", "traffic")})
print("variants", json.dumps(d.get("search_variants"))[:900])
print("errors", d["config"]["device_error_flags"], "families", list((d.get("families") or {}).keys()))
PY
timeout 600 python bench.py --steps 20 --warmup 5 --no-families --no-cpu-baseline > $O/bench_2a_steps20.json 2> $O/bench_2a_steps20.err; echo "bench20 rc=$?"
python3 - $O/bench_2a_steps20.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1])
print("steps20 value %.4g  ms/step %.5f  overlap %s" % (d["value"], d["ms_per_step"], d["config"]["overlap"]))
PY
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o b -- python3 bench.py --no-cpu-baseline --no-families --no-variants --sustain-seconds 0 --steps 640 --warmup 64 --repeats 3 > $O/prof_bench.json 2> $O/prof_bench.err
echo "prof rc=$?"
S=$(ls $O/prof/*kernel_stats.csv $O/prof/*/*kernel_stats.csv 2>/dev/null | head -1); T=$(ls $O/prof/*kernel_trace.csv $O/prof/*/*kernel_trace.csv 2>/dev/null | head -1)
python3 - "$S" $O/kernel_stats_anymdp_2a_overlap.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if "anymdp" in r["Name"]]
with open(sys.argv[2], "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(keep)
for r in keep:
    print("%-100s calls %6s avg %10.1f ns" % (r["Name"][:100], r["Calls"], float(r["AverageNs"])))
PY
python3 scripts/devtools/trace_overlap.py "$T" --match "true, 1, true>" --skip 70 --out $O/trace_overlap_2a.json
rm -rf $O/prof
