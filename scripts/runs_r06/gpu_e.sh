#!/bin/bash
# round 6, visit e: ray caster with float32 colour sums in the speculation (XV_MAZE_SPEC32) — parity, A/B against the round-5
# filter, counters; LinDS: what the noise and the command table cost
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
echo "== maze parity"
timeout 900 python -m pytest tests/test_gpu_maze.py tests/test_gpu_maze_agent.py -x -q > $O/e_pytest_maze.log 2>&1; echo "rc=$?"; tail -4 $O/e_pytest_maze.log
PYTHONPATH=.:tests timeout 300 python tests/soak_maze.py 120 > $O/e_soak_maze.txt 2>&1; echo "soak rc=$?"; tail -3 $O/e_soak_maze.txt
echo "== maze A/B (new = product build, old = -DXV_MAZE_SPEC32=0)"
for rep in 1 2; do
for v in new old; do
  if [ $v = old ]; then export XV_LIB_PATH=$PWD/scripts/devtools/_build/libxeno_mzold.so; else unset XV_LIB_PATH; fi
  timeout 600 python scripts/bench_families.py --families maze64,maze256,maze64_f32,maze256_f32 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$v', d['workload'][-14:], d['filter'], {k: round(x, 1) for k, x in d['us_per_step'].items()})
"
done
done | tee $O/e_maze_ab.txt
unset XV_LIB_PATH
echo "== linds A/B"
for rep in 1 2; do
for v in base nonoise nobm cmdoff; do
  unset XV_LIB_PATH XV_LINDS_AB_CMD_TABLE_OFF
  [ $v = nonoise ] && export XV_LIB_PATH=$PWD/scripts/devtools/_build/libxeno_lnz1.so
  [ $v = nobm ] && export XV_LIB_PATH=$PWD/scripts/devtools/_build/libxeno_lnz2.so
  [ $v = cmdoff ] && export XV_LINDS_AB_CMD_TABLE_OFF=1
  timeout 600 python scripts/bench_families.py --families linds_mfma --steps 2000 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$v', {k: round(x, 2) for k, x in d['us_per_step'].items()})
"
done
done | tee $O/e_linds_ab.txt
unset XV_LIB_PATH XV_LINDS_AB_CMD_TABLE_OFF
echo "== raycast counters (current source)"
export PMC_EXTRA="TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr|SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32"
XV_MAZE_STEPS=12 bash scripts/pmc_kernel.sh raycast_spec32_64 maze_raycast scripts/bench_families.py --families maze64 > $O/e_pmc_64.log 2>&1; tail -3 $O/e_pmc_64.log
XV_MAZE_STEPS=6 bash scripts/pmc_kernel.sh raycast_spec32_256 maze_raycast scripts/bench_families.py --families maze256 > $O/e_pmc_256.log 2>&1; tail -3 $O/e_pmc_256.log
XV_MAZE_STEPS=12 bash scripts/pmc_kernel.sh raycast_f32_64 maze_raycast scripts/bench_families.py --families maze64_f32 > $O/e_pmc_64f.log 2>&1; tail -3 $O/e_pmc_64f.log
ls $O/pmc_raycast*.json
