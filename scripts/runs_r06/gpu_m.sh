#!/bin/bash
# round 6, visit m: ray caster, fast reciprocals in the speculation (v_rcp_f64 + one Newton step for 10 / d2 and 1 / sw): parity,
# soak, A/B against the previous commit, counters
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_maze.py tests/test_gpu_maze_agent.py tests/test_gpu_fullsize.py -x -q --timeout 600 > $O/m_pytest.log 2>&1; echo "rc=$?"; tail -4 $O/m_pytest.log
PYTHONPATH=.:tests timeout 500 python tests/soak_maze.py 400 > $O/m_soak_maze.txt 2>&1; echo "soak rc=$?"; tail -2 $O/m_soak_maze.txt
for rep in 1 2; do
  for v in new prev; do
    if [ $v = prev ]; then export XV_LIB_PATH=$PWD/scripts/devtools/_build/libxeno_mzprev.so; else unset XV_LIB_PATH; fi
    timeout 600 python scripts/bench_families.py --families maze64,maze256 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$v', d['workload'][-14:], d['filter'], {k: round(x, 1) for k, x in d['us_per_step'].items()})
"
  done
done | tee $O/m_maze_ab.txt
unset XV_LIB_PATH
export PMC_EXTRA="TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr|SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32"
XV_MAZE_STEPS=12 bash scripts/pmc_kernel.sh raycast_spec32_64 maze_raycast scripts/bench_families.py --families maze64 > $O/m_pmc_64.log 2>&1; tail -2 $O/m_pmc_64.log
XV_MAZE_STEPS=6 bash scripts/pmc_kernel.sh raycast_spec32_256 maze_raycast scripts/bench_families.py --families maze256 > $O/m_pmc_256.log 2>&1; tail -2 $O/m_pmc_256.log
python - <<'PY'
import json
for r in (64, 256):
    d = json.load(open("gpurun_out/pmc_raycast_spec32_%d.json" % r))
    for k, v in d["kernels"].items():
        print(r, k[:60], "VALU/pixel %.1f" % (v["SQ_INSTS_VALU"] * 64 / (r * r * 16384)))
PY
