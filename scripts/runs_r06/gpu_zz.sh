#!/bin/bash
# round 6, the record on the FINAL tree (after the ray-caster steps, the device-wide overlap slot, two launches may fill the device)
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06_zz
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" $O/pytest_gpu.txt | tail -2
timeout 300 python __graft_entry__.py smoke > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.txt
# PMC traffic first: the bench lines fall back to it where rocprofv3 child passes are not possible, and the CPU test pins its hash
bash scripts/gpu_pmc.sh > $O/pmc.log 2>&1; tail -3 $O/pmc.log
cp gpurun_out/pmc_traffic_anymdp_2a.json $O/pmc_traffic_anymdp_2a_bucket.json
S0=$(date +%s); timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_2a_steps20.json 2> $O/bench_2a_steps20.err; echo "bench steps20 (the driver's command) rc=$? wall $(( $(date +%s) - S0 )) s"
timeout 900 python bench.py > $O/bench_2a.json 2> $O/bench_2a.err; echo "bench 2a rc=$?"
timeout 600 python bench.py --tasks 1024 --no-cpu-baseline --no-families > $O/bench_2b.json 2> $O/bench_2b.err; echo "bench 2b rc=$?"
timeout 600 python bench.py --workload mixed > $O/bench_mixed_n1.json 2> $O/bench_mixed_n1.err; echo "bench mixed rc=$?"
for f in bench_2a_steps20 bench_2a bench_2b bench_mixed_n1; do python3 - $O/$f.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1])
r = d.get("roofline") or {}
print("%-28s value %.4g ms/step %.5f frac %s survey %s traffic %s overlap %s errs %s" % (sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"],
      r.get("frac"), r.get("frac_survey_bytes"), r.get("traffic"), d["config"].get("overlap"), d["config"].get("device_error_flags")))
lc = d.get("long_call")
if lc:
    for m in ("one_stream", "overlapped", "fused_rollout"):
        row = lc[m]
        print("   long_call %-14s %.3f us events %.3f wall %.4g env-steps/s state %s frac %s" % (m, row["us_per_step"], row["wall_us_per_step"], row["env_steps_per_s"], row["overlap_state"], row["roofline"]["frac"]))
    print("   sustain/long_call", lc.get("sustain_over_long_call"))
fam = d.get("families") or {}
for k in ("linds", "mazeworld_64", "mazeworld_256", "mixed_share", "anymdp_refdist", "anymdp_tok_refdist", "python_loop"):
    if k in fam:
        print("   families.%-18s ms/step %s frac %s" % (k, fam[k].get("ms_per_step"), (fam[k].get("roofline") or {}).get("frac")))
PY
done
rm -rf $O/prof_off
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_off -o b -- python3 bench.py --no-cpu-baseline --no-families --no-variants --no-live-pmc --sustain-seconds 0 --long-steps 0 --steps 640 --warmup 64 --repeats 3 --overlap off > $O/prof_bench_off.json 2> $O/prof_bench_off.err
echo "rocprof rc=$?"
S=$(ls $O/prof_off/*kernel_stats.csv $O/prof_off/*/*kernel_stats.csv 2>/dev/null | head -1)
python3 - "$S" $O/kernel_stats_anymdp_2a_one_stream.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if "anymdp" in r["Name"]]
with open(sys.argv[2], "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(keep)
for r in keep:
    if "step_kernel" in r["Name"]: print("  %-90s calls %6s avg %10.1f ns" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])))
PY
rm -rf $O/prof_off $O/prof_fam
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fam -o fam -- python3 scripts/bench_families.py --steps 200 --warmup 20 > $O/bench_families.jsonl 2> $O/bench_families.err
echo "families rc=$?"
S=$(ls $O/prof_fam/*kernel_stats.csv $O/prof_fam/*/*kernel_stats.csv 2>/dev/null | head -1)
python3 - "$S" $O/kernel_stats_families.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if any(k in r["Name"] for k in ("anymdp", "linds", "maze", "cartpole", "acrobot", "mixed"))]
with open(sys.argv[2], "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(keep)
for r in keep[:10]:
    print("  %-80s calls %6s avg %12.1f ns" % (r["Name"][:80], r["Calls"], float(r["AverageNs"])))
PY
rm -rf $O/prof_fam
timeout 900 python bench.py --sweep-envs 4096,16384,32768,65536,98304,131072 --steps 640 --warmup 64 --sweep-out $O/anymdp_envs_sweep.json > /dev/null 2> $O/sweep.err; echo "sweep rc=$?"
python3 -c "
import json
d=json.load(open('$O/anymdp_envs_sweep.json'))
for r in d['rows']: print('  envs', r['envs'], r.get('search'), 'us/step %.3f' % r.get('us_per_step',0), 'overlapped', r.get('overlapped'), 'fused', r.get('fused_rollout_us_per_step'))
"
