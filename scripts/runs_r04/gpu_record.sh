#!/bin/bash
# round 4, the record at the final kernel source: bench lines, rocprofv3 kernel stats (AUTO, fence), PMC traffic (both searches)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${1:-r04_z}
echo "== PMC anymdp traffic (AUTO)"; bash scripts/gpu_pmc.sh "--steps 200 --warmup 20 --no-cpu-baseline --no-families" 2a > gpurun_out/${T}_pmc_anymdp_auto.log 2>&1; tail -2 gpurun_out/${T}_pmc_anymdp_auto.log | cut -c1-300
cp gpurun_out/pmc_traffic_anymdp_2a.json gpurun_out/${T}_pmc_traffic_anymdp_2a_bucket.json
cp gpurun_out/pmc_traffic_anymdp_2a.json profiles/${T}_pmc_traffic_anymdp_2a_bucket.json      # so that the bench lines below find it
echo "== PMC anymdp traffic (fence)"; bash scripts/gpu_pmc.sh "--steps 200 --warmup 20 --no-cpu-baseline --no-families --search fence" 2a > gpurun_out/${T}_pmc_anymdp_fence.log 2>&1; tail -2 gpurun_out/${T}_pmc_anymdp_fence.log | cut -c1-300
cp gpurun_out/pmc_traffic_anymdp_2a.json gpurun_out/${T}_pmc_traffic_anymdp_2a_fence.json
cp gpurun_out/pmc_traffic_anymdp_2a.json profiles/${T}_pmc_traffic_anymdp_2a_fence.json
echo "== PMC anymdp traffic 2b"; bash scripts/gpu_pmc.sh "--steps 200 --warmup 20 --no-cpu-baseline --no-families --tasks 1024" 2b > gpurun_out/${T}_pmc_anymdp_2b.log 2>&1; tail -2 gpurun_out/${T}_pmc_anymdp_2b.log | cut -c1-300
cp gpurun_out/pmc_traffic_anymdp_2b.json gpurun_out/${T}_pmc_traffic_anymdp_2b.json
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
    print("%-18s value %.4e ms/step %.5f kernel us %.3f search %s primary %s frac %.3f frac_traffic %s frac_of_floor %s traffic_src %s" % (
        f, d["value"], d["ms_per_step"], r["avg_launch_us"], d["config"]["search"], r["primary"], r["frac"], r["frac_traffic"], r["frac_of_floor"], r["traffic_source"]),
        d.get("fused_rollout_env_steps_per_s_rank0"), d.get("search_variants"))
PY
for S in auto fence; do
  echo "== rocprof bench --search $S"
  rm -rf gpurun_out/prof_$S
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$S -o st -- python3 bench.py --steps 500 --warmup 50 --repeats 10 --no-cpu-baseline --no-families --search $S > /dev/null 2> gpurun_out/${T}_prof_$S.err
  f=$(find gpurun_out/prof_$S -name "*kernel_stats.csv" | head -1)
  head -1 $f > gpurun_out/${T}_kernel_stats_anymdp_2a_$S.csv; grep anymdp $f >> gpurun_out/${T}_kernel_stats_anymdp_2a_$S.csv
  grep step_kernel gpurun_out/${T}_kernel_stats_anymdp_2a_$S.csv | cut -c1-200
done
timeout 600 python -m pytest tests/test_gpu_anymdp.py -q -m gpu 2>&1 | tail -2
