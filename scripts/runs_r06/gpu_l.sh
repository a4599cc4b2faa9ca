#!/bin/bash
# round 6, visit l: the ray caster's re-run pixels spread over the wave (XV_MAZE_REDO_SPREAD): parity, soak, A/B, counters
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_maze.py tests/test_gpu_maze_agent.py tests/test_gpu_mixed_shard.py tests/test_gpu_fullsize.py -x -q --timeout 600 > $O/l_pytest.log 2>&1; echo "rc=$?"; tail -4 $O/l_pytest.log
PYTHONPATH=.:tests timeout 400 python tests/soak_maze.py 300 > $O/l_soak_maze.txt 2>&1; echo "soak rc=$?"; tail -2 $O/l_soak_maze.txt
for rep in 1 2; do
  for v in spread nospread; do
    if [ $v = nospread ]; then export XV_LIB_PATH=$PWD/scripts/devtools/_build/libxeno_mznospread.so; else unset XV_LIB_PATH; fi
    timeout 600 python scripts/bench_families.py --families maze64,maze256 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$v', d['workload'][-14:], d['filter'], {k: round(x, 1) for k, x in d['us_per_step'].items()})
"
  done
done | tee $O/l_maze_ab.txt
unset XV_LIB_PATH
export PMC_EXTRA="TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr|SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32"
XV_MAZE_STEPS=12 bash scripts/pmc_kernel.sh raycast_spec32_64 maze_raycast scripts/bench_families.py --families maze64 > $O/l_pmc_64.log 2>&1; tail -2 $O/l_pmc_64.log
XV_MAZE_STEPS=6 bash scripts/pmc_kernel.sh raycast_spec32_256 maze_raycast scripts/bench_families.py --families maze256 > $O/l_pmc_256.log 2>&1; tail -2 $O/l_pmc_256.log
python - <<'PY'
import json
for r in (64, 256):
    d = json.load(open("gpurun_out/pmc_raycast_spec32_%d.json" % r))
    for k, v in d["kernels"].items():
        print(r, k[:60], "VALU/pixel %.1f" % (v["SQ_INSTS_VALU"] * 64 / (r * r * 16384)), "active VALU / wave cycles", v.get("SQ_ACTIVE_INST_VALU_over_WAVE_CYCLES"))
PY
