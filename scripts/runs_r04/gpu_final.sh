#!/bin/bash
# round 4, the record: whole GPU suite, smoke, the bench line (default flags, the driver's flags, fence), rocprofv3 kernel
# stats of the headline (AUTO and fence) and of every family, PMC traffic of the AnyMDP step kernel for both searches (keyed
# on its source hash), floor probe, the two-ranks-on-one-GPU functional run started WITHOUT a launcher.  -> gpurun_out/.
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${1:-r04_z}
echo "== pytest -m gpu"; timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/${T}_pytest_gpu.log 2>&1; echo "rc=$?"; grep -n "passed\|failed\|Error" gpurun_out/${T}_pytest_gpu.log | head -5
tail -3 gpurun_out/${T}_pytest_gpu.log > gpurun_out/${T}_pytest_gpu_tail.txt
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
echo "== floor probe"; timeout 600 python scripts/devtools/floor_probe.py gpurun_out/${T}_floor_probe.json | cut -c1-400
echo "== bench default"; timeout 900 python bench.py --fused > gpurun_out/${T}_bench_2a.json 2> gpurun_out/${T}_bench_2a.err; echo "rc=$?"
echo "== bench driver flags"; timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench_2a_steps20.json 2> gpurun_out/${T}_bench_2a_steps20.err; echo "rc=$?"
echo "== bench 2b"; timeout 900 python bench.py --tasks 1024 --no-cpu-baseline --no-families --fused > gpurun_out/${T}_bench_2b.json 2>/dev/null; echo "rc=$?"
python - <<PY
import json
for f in ("bench_2a", "bench_2a_steps20", "bench_2b"):
    try:
        d = json.loads([l for l in open("gpurun_out/${T}_%s.json" % f) if l.startswith('{"metric"')][-1])
    except Exception as ex:
        print(f, "unreadable", ex); continue
    r = d["roofline"]
    print("%-18s value %.4e ms/step %.5f kernel us %.3f search %s primary %s frac %.3f frac_traffic %s frac_of_floor %s" % (
        f, d["value"], d["ms_per_step"], r["avg_launch_us"], d["config"]["search"], r["primary"], r["frac"], r["frac_traffic"], r["frac_of_floor"]),
        d.get("fused_rollout_env_steps_per_s_rank0"), d.get("search_variants"), d.get("sustain"))
    for k, v in (d.get("families") or {}).items():
        print("   ", k, {a: v.get(a) for a in ("ms_per_step", "env_steps_per_s", "wall_s", "error", "auto_search", "us_per_step", "us_per_vector_step")}, (v.get("roofline") or {}).get("frac"))
PY
for S in auto fence; do
  echo "== rocprof bench --search $S"
  rm -rf gpurun_out/prof_$S
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$S -o st -- python3 bench.py --steps 500 --warmup 50 --repeats 10 --no-cpu-baseline --no-families --search $S > /dev/null 2> gpurun_out/${T}_prof_$S.err
  f=$(find gpurun_out/prof_$S -name "*kernel_stats.csv" | head -1)
  head -1 $f > gpurun_out/${T}_kernel_stats_anymdp_2a_$S.csv; grep anymdp $f >> gpurun_out/${T}_kernel_stats_anymdp_2a_$S.csv
  grep step_kernel gpurun_out/${T}_kernel_stats_anymdp_2a_$S.csv | cut -c1-200
done
echo "== rocprof families"
rm -rf gpurun_out/prof_fam
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_fam -o fam -- python3 scripts/bench_families.py --families linds,cartpole,acrobot,maze64,maze64_f32,mixed,anymdp_tok,anymdp_tok_refdist,anymdp_refdist --steps 200 --warmup 20 > gpurun_out/${T}_families_prof.jsonl 2> gpurun_out/${T}_families_prof.err; echo "rc=$?"
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/prof_fam/**/fam_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
keep = [r for r in rows if any(k in r["Name"] for k in ("anymdp", "linds", "maze", "cartpole", "acrobot", "mixed"))]
with open("gpurun_out/${T}_kernel_stats_families.csv", "w", newline="") as o:
    w = csv.DictWriter(o, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(keep)
for r in keep:
    if "step" in r["Name"] or "raycast" in r["Name"] or "rollout" in r["Name"]:
        print("%-100s calls %6s avg %10.1f ns" % (r["Name"][:100], r["Calls"], float(r["AverageNs"])))
PY
echo "== families clean"; timeout 900 python scripts/bench_families.py --families linds,cartpole,acrobot,maze64,maze64_f32,maze256,maze256_f32,mixed,anymdp_tok,anymdp_tok_refdist,anymdp_refdist,python_loop,teacher > gpurun_out/${T}_bench_families.jsonl 2> gpurun_out/${T}_families.err; echo "rc=$?"; cut -c1-420 gpurun_out/${T}_bench_families.jsonl
echo "== PMC anymdp traffic (AUTO)"; bash scripts/gpu_pmc.sh "--steps 200 --warmup 20 --no-cpu-baseline --no-families" 2a > gpurun_out/${T}_pmc_anymdp_auto.log 2>&1; tail -3 gpurun_out/${T}_pmc_anymdp_auto.log | cut -c1-300
cp gpurun_out/pmc_traffic_anymdp_2a.json gpurun_out/${T}_pmc_traffic_anymdp_2a_bucket.json
echo "== PMC anymdp traffic (fence)"; bash scripts/gpu_pmc.sh "--steps 200 --warmup 20 --no-cpu-baseline --no-families --search fence" 2a > gpurun_out/${T}_pmc_anymdp_fence.log 2>&1; tail -3 gpurun_out/${T}_pmc_anymdp_fence.log | cut -c1-300
cp gpurun_out/pmc_traffic_anymdp_2a.json gpurun_out/${T}_pmc_traffic_anymdp_2a_fence.json
echo "== PMC raycast f32 (texture path)"
PMC_EXTRA="TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum|TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" bash scripts/pmc_kernel.sh raycast_f32_${T} maze_raycast scripts/bench_families.py --families maze64_f32 --steps 200 > gpurun_out/${T}_pmc_raycast_f32.log 2>&1; tail -2 gpurun_out/${T}_pmc_raycast_f32.log | cut -c1-300
echo "== n2 functional (two ranks sharing the GPU, no launcher)"
bash scripts/gpu_n2_functional.sh 2>&1 | tail -1 | python -c "
import sys, json
l = sys.stdin.read().strip()
try:
    d = json.loads(l[l.index('{'):]); print({k: d.get(k) for k in ('n_gpus', 'value', 'rccl', 'rccl_ranks', 'transport', 'transport_requested', 'transport_note', 'allgather_timeout')}, d['config']['exchange'][:160])
except Exception as ex:
    print('n2 line unreadable:', ex, l[:300])
" | tee gpurun_out/${T}_n2_functional.txt
