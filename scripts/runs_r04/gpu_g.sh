# round 4, trip g: LinDS paired command rows — parity, step time, traffic counters
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_linds.py tests/test_gpu_mixed.py tests/test_gpu_fullsize.py -x -q -m gpu -k "linds or mixed or config_3 or config_5" 2>&1 | tail -6 > gpurun_out/r04_g_pytest.txt
cat gpurun_out/r04_g_pytest.txt
for i in 1 2; do timeout 300 python scripts/bench_families.py --families linds_mfma --steps 800 2>/dev/null | cut -c1-400; done | tee gpurun_out/r04_g_linds.jsonl
rm -rf gpurun_out/prof_l
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_l -o l -- python3 scripts/bench_families.py --families linds_mfma --steps 800 > /dev/null 2>&1
f=$(find gpurun_out/prof_l -name "*kernel_stats.csv" | head -1); head -1 $f > gpurun_out/r04_g_kernel_stats_linds.csv; grep linds_step $f >> gpurun_out/r04_g_kernel_stats_linds.csv; cat gpurun_out/r04_g_kernel_stats_linds.csv | cut -c1-200
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmcl_$c
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmcl_$c -o pmc -- python3 scripts/bench_families.py --families linds_mfma --steps 200 > /dev/null 2>&1
  python3 - <<PY
import csv, glob, collections
fs = glob.glob("gpurun_out/pmcl_$c/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if r["Counter_Name"] == "$c": agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if "linds_step" in k: print("$c", k[:70], "avg KB", sum(v)/len(v), "n", len(v))
PY
done 2>&1 | tee gpurun_out/r04_g_pmc_linds.txt
