#!/bin/bash
# round 2, second visit: LinDS after the reorder / fast Box-Muller / slot layout; full-size parity again
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== pytest linds + fullsize + mixed"; timeout 1500 python -m pytest tests/test_gpu_linds.py tests/test_gpu_fullsize.py tests/test_gpu_mixed.py -m gpu -x -q > gpurun_out/pytest_b.log 2>&1; echo "rc=$?"; tail -6 gpurun_out/pytest_b.log
echo "== families linds"; timeout 600 python scripts/bench_families.py --families linds --steps 400 --warmup 40 2>&1 | cut -c1-600
echo "== linds counters"
bash scripts/pmc_kernel.sh linds_b linds_step scripts/bench_families.py --families linds --steps 60 --warmup 10 2>&1 | grep -A60 "mfma" | grep "SQ_WAVE_CYCLES\|_over_\|per_wave\|VGPR\|hbm_bytes\|dispatches\|^void"
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/pmc_linds_b_g0/**/*kernel_trace.csv", recursive=True):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in d.items():
        if "linds_step" in k:
            print("%-70s n=%d avg %.2f us" % (k[:70], len(v), sum(v) / len(v) / 1e3))
PY
