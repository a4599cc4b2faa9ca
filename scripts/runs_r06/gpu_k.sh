#!/bin/bash
# round 6, visit k: ray caster with the weight sum in the packed float32 chain and no clamp instructions (KAPPA 3.0e-6):
# parity, soak, timing, counters; the new mixed-shard test
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_maze.py tests/test_gpu_maze_agent.py tests/test_gpu_mixed_shard.py tests/test_gpu_fullsize.py -x -q --timeout 600 > $O/k_pytest.log 2>&1; echo "rc=$?"; tail -4 $O/k_pytest.log
PYTHONPATH=.:tests timeout 400 python tests/soak_maze.py 300 > $O/k_soak_maze.txt 2>&1; echo "soak rc=$?"; tail -2 $O/k_soak_maze.txt
for rep in 1 2; do
  timeout 600 python scripts/bench_families.py --families maze64,maze256 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('new', d['workload'][-14:], d['filter'], {k: round(x, 1) for k, x in d['us_per_step'].items()}, d.get('valu_issue'))
"
  XV_LIB_PATH=$PWD/scripts/devtools/_build/libxeno_mzold.so timeout 600 python scripts/bench_families.py --families maze64,maze256 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('round-5 filter', d['workload'][-14:], d['filter'], {k: round(x, 1) for k, x in d['us_per_step'].items()})
"
done | tee $O/k_maze_ab.txt
export PMC_EXTRA="TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr|SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32"
XV_MAZE_STEPS=12 bash scripts/pmc_kernel.sh raycast_spec32_64 maze_raycast scripts/bench_families.py --families maze64 > $O/k_pmc_64.log 2>&1; tail -2 $O/k_pmc_64.log
XV_MAZE_STEPS=6 bash scripts/pmc_kernel.sh raycast_spec32_256 maze_raycast scripts/bench_families.py --families maze256 > $O/k_pmc_256.log 2>&1; tail -2 $O/k_pmc_256.log
python - <<'PY'
import json
for r in (64, 256):
    d = json.load(open("gpurun_out/pmc_raycast_spec32_%d.json" % r))
    for k, v in d["kernels"].items():
        print(r, k[:60], "VALU/pixel %.1f" % (v["SQ_INSTS_VALU"] * 64 / (r * r * 16384)))
PY
