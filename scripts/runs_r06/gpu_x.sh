#!/bin/bash
# round 6, visit x: the kept fetch (one-phase pair copy, third span as two 16-byte loads): parity, soak, timing, counters
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_maze.py tests/test_gpu_maze_agent.py tests/test_gpu_fullsize.py -x -q --timeout 600 > $O/x_pytest.log 2>&1; echo "rc=$?"; tail -3 $O/x_pytest.log
PYTHONPATH=.:tests timeout 400 python tests/soak_maze.py 300 > $O/x_soak_maze.txt 2>&1; echo "soak rc=$?"; tail -2 $O/x_soak_maze.txt
for rep in 1 2; do
timeout 600 python scripts/bench_families.py --families maze64,maze256 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['workload'][-16:], d['filter'], {k: round(x, 1) for k, x in d['us_per_step'].items()})
"
done | tee $O/x_maze.txt
export PMC_EXTRA="TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr"
XV_MAZE_STEPS=12 bash scripts/pmc_kernel.sh raycast_spec32_64 maze_raycast scripts/bench_families.py --families maze64 > $O/x_pmc_64.log 2>&1; tail -2 $O/x_pmc_64.log
