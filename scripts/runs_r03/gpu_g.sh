#!/bin/bash
# round 3, visit G: AnyMDP with the 8-byte env record and 16-byte reset units, S up to 512; LinDS flags on squares
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest anymdp + linds + fullsize"; timeout 1800 python -m pytest tests/test_gpu_anymdp.py tests/test_gpu_anymdp_tok.py tests/test_gpu_linds.py tests/test_gpu_fullsize.py tests/test_gpu_mixed.py tests/test_gpu_sampler.py -q -x > gpurun_out/g_pytest.log 2>&1; echo "rc=$?"; tail -8 gpurun_out/g_pytest.log
echo "== bench"; timeout 900 python bench.py --no-cpu-baseline --no-families > gpurun_out/g_bench.json 2> gpurun_out/g_bench.err; echo "rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/g_bench.json"))
print("value %.4e ms/step %.5f avg_launch_us %.3f frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"]))
PY
echo "== bench fence"; timeout 900 python bench.py --no-cpu-baseline --no-families --search fence 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fence: value %.4e avg_launch_us %.3f' % (d['value'], d['roofline']['avg_launch_us']))"
echo "== rocprof bench"
rm -rf gpurun_out/prof_g
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_g -o g -- python3 bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-families > gpurun_out/g_prof_bench.json 2> gpurun_out/g_prof.err; echo "rc=$?"
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_g/**/g_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
keep = [r for r in rows if "anymdp" in r["Name"]]
with open("gpurun_out/g_kernel_stats_anymdp_2a.csv", "w", newline="") as o:
    w = csv.DictWriter(o, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(keep)
for r in keep:
    print("%-100s calls %6s avg %12.1f ns" % (r["Name"][:100], r["Calls"], float(r["AverageNs"])))
PY
echo "== linds"; timeout 300 python scripts/bench_families.py --families linds_mfma 2>/dev/null | cut -c1-400
