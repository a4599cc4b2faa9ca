#!/bin/bash
# round 6, visit zz5: the tree after the ray caster's fetch work (third span whole, nontemporal flush, prefetching columns loop):
# the whole GPU suite, maze soak, counters at 64 x 64, the families' bench lines
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu --timeout 900 > $O/zz5_pytest.log 2>&1; echo "rc=$?"; tail -3 $O/zz5_pytest.log
PYTHONPATH=.:tests timeout 400 python tests/soak_maze.py 300 > $O/zz5_soak_maze.txt 2>&1; echo "soak rc=$?"; tail -2 $O/zz5_soak_maze.txt
export PMC_EXTRA="TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr"
XV_MAZE_STEPS=12 bash scripts/pmc_kernel.sh raycast_spec32_64 maze_raycast scripts/bench_families.py --families maze64 > $O/zz5_pmc_64.log 2>&1; tail -2 $O/zz5_pmc_64.log
unset PMC_EXTRA
timeout 900 python scripts/bench_families.py > $O/zz5_bench_families.jsonl 2> $O/zz5_bench_families.err; echo "families rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/zz5_bench_families.jsonl"):
    if l.startswith("{"):
        d = json.loads(l)
        print(d.get("workload", "?")[:70], {k: round(v, 1) for k, v in d.get("us_per_step", {}).items()} if isinstance(d.get("us_per_step"), dict) else d.get("us_per_step"))
PY
