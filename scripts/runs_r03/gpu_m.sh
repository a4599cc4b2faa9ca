#!/bin/bash
# round 3, visit M: LinDS state loads non-temporal (A/B), with FETCH / WRITE counters
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/m_linds_variants.jsonl
for v in default m12 m13; do
  if [ $v = default ]; then unset XV_LIB_PATH; else export XV_LIB_PATH=$PWD/scripts/devtools/_build/libxeno_$v.so; fi
  echo "== $v"
  timeout 300 python -m pytest tests/test_gpu_linds.py -x -q 2>&1 | tail -1
  timeout 300 python scripts/bench_families.py --families linds_mfma --steps 400 --warmup 40 2>/dev/null | cut -c1-330 | tee -a gpurun_out/m_linds_variants.jsonl
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc_m_$c
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_m_$c -o pmc -- python3 scripts/bench_families.py --families linds_mfma --steps 200 --warmup 20 > /dev/null 2>&1
    python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/pmc_m_$c/**/*counter_collection.csv", recursive=True)[0]
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == "$c" and "linds_step_mfma" in r["Kernel_Name"]]
print("$v $c KB avg %.1f over %d" % (sum(v) / len(v), len(v)))
PY
  done
done
