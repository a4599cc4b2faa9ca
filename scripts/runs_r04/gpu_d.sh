# round 4, trip d: captured loops with the tick batch, bench (AUTO rule with auto_limit), rocprof stats + PMC traffic of the headline
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_capture.py tests/test_gpu_mixed.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r04_d_pytest.txt
cat gpurun_out/r04_d_pytest.txt
timeout 300 python scripts/bench_families.py --families python_loop,anymdp_refdist --steps 400 > gpurun_out/r04_d_families.jsonl 2> gpurun_out/r04_d_families.err
cat gpurun_out/r04_d_families.jsonl | cut -c1-1500; tail -3 gpurun_out/r04_d_families.err
timeout 900 python bench.py > gpurun_out/r04_d_bench_2a.json 2> gpurun_out/r04_d_bench_2a.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04_d_bench_2a.json') if l.startswith('{"metric"')][-1])
print(d['value'], d['ms_per_step'], d['config']['search'], d['config']['bucket_census'])
print(json.dumps(d['roofline'])[:1500]); print(d.get('search_variants')); print(d.get('sustain'))
PY
# kernel stats of the AUTO (bucket) and the FENCE kernel at 2a
for S in auto fence; do
  rm -rf gpurun_out/prof_$S
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$S -o st -- python3 bench.py --steps 500 --warmup 50 --repeats 10 --no-cpu-baseline --no-families --search $S > gpurun_out/r04_d_prof_$S.json 2> gpurun_out/r04_d_prof_$S.err
  f=$(find gpurun_out/prof_$S -name "*kernel_stats.csv" | head -1)
  head -1 $f > gpurun_out/r04_d_kernel_stats_anymdp_2a_$S.csv; grep anymdp $f >> gpurun_out/r04_d_kernel_stats_anymdp_2a_$S.csv
  cat gpurun_out/r04_d_kernel_stats_anymdp_2a_$S.csv | cut -c1-220
done
bash scripts/gpu_pmc.sh "--steps 200 --warmup 20 --no-cpu-baseline --no-families" 2a 2>&1 | tail -6
cp gpurun_out/pmc_traffic_anymdp_2a.json gpurun_out/r04_d_pmc_traffic_anymdp_2a_auto.json
bash scripts/gpu_pmc.sh "--steps 200 --warmup 20 --no-cpu-baseline --no-families --search fence" 2a 2>&1 | tail -6
cp gpurun_out/pmc_traffic_anymdp_2a.json gpurun_out/r04_d_pmc_traffic_anymdp_2a_fence.json
