#!/bin/bash
# round 3, visit C: LinDS — compact command rows; A/B of non-temporal output stores and of fewer resident waves; batch sweep
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest linds"; timeout 900 python -m pytest tests/test_gpu_linds.py tests/test_gpu_fullsize.py -x -q -k "linds or config3" > gpurun_out/c_pytest_linds.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/c_pytest_linds.log
for v in default nt occ2 occ3; do
  if [ $v = default ]; then unset XV_LIB_PATH; else export XV_LIB_PATH=$PWD/scripts/devtools/_build/libxeno_$v.so; fi
  echo "== $v"; timeout 300 python scripts/bench_families.py --families linds_mfma,linds_sweep --steps 400 --warmup 40 2>/dev/null | cut -c1-900 | tee -a gpurun_out/c_linds_variants.jsonl
done
