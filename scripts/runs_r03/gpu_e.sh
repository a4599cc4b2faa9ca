#!/bin/bash
# round 3, visit E: LinDS with non-temporal output stores as the default; A/B of more non-temporal traffic; counters
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/e_linds_variants.jsonl
for v in default m1 m2 m3 m4; do
  if [ $v = default ]; then unset XV_LIB_PATH; else export XV_LIB_PATH=$PWD/scripts/devtools/_build/libxeno_$v.so; fi
  echo "== $v"
  timeout 300 python -m pytest tests/test_gpu_linds.py -x -q 2>&1 | tail -1
  timeout 300 python scripts/bench_families.py --families linds_mfma,linds_sweep --steps 400 --warmup 40 2>/dev/null | cut -c1-1200 | tee -a gpurun_out/e_linds_variants.jsonl
done
unset XV_LIB_PATH
bash scripts/pmc_kernel.sh linds_r03e linds_step_mfma scripts/bench_families.py --families linds_mfma --steps 300 --warmup 30 > gpurun_out/e_pmc.log 2>&1
python - <<'PY'
import json
d = json.load(open("gpurun_out/pmc_linds_r03e.json"))
for k, v in d["kernels"].items():
    print(k, {x: v[x] for x in ("FETCH_SIZE", "WRITE_SIZE", "hbm_bytes_per_launch_corrected", "SQ_INSTS_VALU_per_wave", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY_over_WAVE_CYCLES", "SQ_WAIT_INST_ANY_over_WAVE_CYCLES", "TCC_HIT_sum", "TCC_MISS_sum") if x in v})
PY
