#!/bin/bash
# round 5, the record on the final tree: GPU suite, smoke, counter passes of the current kernel source (merged into the
# profile the bench looks up), then the bench lines, rocprof statistics, families, sweep
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_z4
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -a "passed\|failed" $O/pytest_gpu.txt | tail -2
timeout 300 python __graft_entry__.py smoke > $O/smoke.txt 2>&1; echo "smoke rc=$?"; grep -a "smoke ok" $O/smoke.txt
# PMC traffic of the step kernel (separate passes per counter), one stream, then overlapped; merged
bash scripts/gpu_pmc.sh "--steps 200 --warmup 20 --no-cpu-baseline --no-families --no-variants --sustain-seconds 0 --overlap off" 2a > $O/pmc_off.log 2>&1
cp gpurun_out/pmc_traffic_anymdp_2a.json $O/pmc_traffic_anymdp_2a_bucket_one_stream.json
timeout 900 bash scripts/gpu_pmc.sh "--steps 200 --warmup 20 --no-cpu-baseline --no-families --no-variants --sustain-seconds 0 --overlap on" 2a > $O/pmc_on.log 2>&1
cp gpurun_out/pmc_traffic_anymdp_2a.json $O/pmc_traffic_anymdp_2a_bucket_overlap.json
python3 scripts/merge_pmc.py $O/pmc_traffic_anymdp_2a_bucket_one_stream.json $O/pmc_traffic_anymdp_2a_bucket_overlap.json $O/pmc_traffic_anymdp_2a_bucket.json
cp $O/pmc_traffic_anymdp_2a_bucket.json profiles/r05_z_pmc_traffic_anymdp_2a_bucket.json      # what bench.py looks up (same file is committed)
timeout 900 python bench.py > $O/bench_2a.json 2> $O/bench_2a.err; echo "bench 2a rc=$?"
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_2a_steps20.json 2> $O/bench_2a_steps20.err; echo "bench steps20 rc=$?"
timeout 600 python bench.py --tasks 1024 --no-cpu-baseline --no-families > $O/bench_2b.json 2> $O/bench_2b.err; echo "bench 2b rc=$?"
timeout 600 python bench.py --workload mixed > $O/bench_mixed_n1.json 2> $O/bench_mixed_n1.err; echo "bench mixed rc=$?"
timeout 600 python bench.py --workload mixed --overlap off --no-cpu-baseline > $O/bench_mixed_n1_one_stream.json 2> $O/bench_mixed_n1_one_stream.err; echo "bench mixed one stream rc=$?"
for f in bench_2a bench_2a_steps20 bench_2b bench_mixed_n1 bench_mixed_n1_one_stream; do python3 - $O/$f.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1])
r = d.get("roofline") or {}
print("%-34s value %.4g ms/step %.5f frac %.3f frac_traffic %s overlap %s errs %s gather %s" % (sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"],
      r.get("frac", 0), r.get("frac_traffic"), d["config"].get("overlap"), d["config"].get("device_error_flags"), (d.get("with_allgather") or {}).get("value")))
PY
done
# rocprof: per-kernel statistics, one stream and overlapped
for ov in off on; do
  rm -rf $O/prof_$ov
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$ov -o b -- python3 bench.py --no-cpu-baseline --no-families --no-variants --sustain-seconds 0 --steps 640 --warmup 64 --repeats 3 --overlap $ov > $O/prof_bench_$ov.json 2> $O/prof_bench_$ov.err
  echo "rocprof overlap=$ov rc=$?"
  S=$(ls $O/prof_$ov/*kernel_stats.csv $O/prof_$ov/*/*kernel_stats.csv 2>/dev/null | head -1); T=$(ls $O/prof_$ov/*kernel_trace.csv $O/prof_$ov/*/*kernel_trace.csv 2>/dev/null | head -1)
  python3 - "$S" $O/kernel_stats_anymdp_2a_overlap_$ov.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if "anymdp" in r["Name"]]
with open(sys.argv[2], "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(keep)
for r in keep:
    if "step_kernel" in r["Name"]: print("  %-90s calls %6s avg %10.1f ns" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])))
PY
  python3 scripts/devtools/trace_overlap.py "$T" --match "step_kernel<false, 1, false, true, 1" --skip 70 --out $O/trace_overlap_2a_overlap_$ov.json
  rm -rf $O/prof_$ov
done
# families kernel statistics
rm -rf $O/prof_fam
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fam -o fam -- python3 scripts/bench_families.py --steps 200 --warmup 20 > $O/bench_families.jsonl 2> $O/bench_families.err
echo "families rc=$?"
S=$(ls $O/prof_fam/*kernel_stats.csv $O/prof_fam/*/*kernel_stats.csv 2>/dev/null | head -1)
python3 - "$S" $O/kernel_stats_families.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if any(k in r["Name"] for k in ("anymdp", "linds", "maze", "cartpole", "acrobot", "mixed"))]
with open(sys.argv[2], "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(keep)
PY
rm -rf $O/prof_fam
timeout 900 python bench.py --sweep-envs 4096,16384,32768,65536,131072 --steps 640 --warmup 64 --sweep-out $O/anymdp_envs_sweep.json > /dev/null 2> $O/sweep.err; echo "sweep rc=$?"
python3 -c "
import json
d=json.load(open('$O/anymdp_envs_sweep.json'))
for r in d['rows']: print('  envs', r['envs'], r.get('search'), 'us/step %.3f' % r.get('us_per_step',0), 'overlapped', r.get('overlapped'))
"
timeout 600 python scripts/devtools/probe_set_task.py --tasks 1024 > $O/set_task_probe.json 2>/dev/null; cat $O/set_task_probe.json
PYTHONPATH=.:tests timeout 400 python tests/soak_anymdp.py 300 > $O/soak_anymdp.txt 2>&1; echo "soak anymdp rc=$?"; tail -1 $O/soak_anymdp.txt
