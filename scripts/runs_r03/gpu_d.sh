#!/bin/bash
# round 3, visit D: LinDS — arithmetic task id, restart word from the noise call, deferred command select; A/B of
# non-temporal output stores and of the restart-row touch
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest linds + fullsize + mixed"; timeout 900 python -m pytest tests/test_gpu_linds.py tests/test_gpu_fullsize.py tests/test_gpu_mixed.py -x -q > gpurun_out/d_pytest_linds.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/d_pytest_linds.log
rm -f gpurun_out/d_linds_variants.jsonl
for v in default nt warm ntwarm; do
  if [ $v = default ]; then unset XV_LIB_PATH; else export XV_LIB_PATH=$PWD/scripts/devtools/_build/libxeno_$v.so; fi
  echo "== $v"
  if [ $v != default ]; then timeout 300 python -m pytest tests/test_gpu_linds.py -x -q 2>&1 | tail -1; fi
  timeout 300 python scripts/bench_families.py --families linds,linds_mfma --steps 400 --warmup 40 2>/dev/null | cut -c1-500 | tee -a gpurun_out/d_linds_variants.jsonl
done
