#!/bin/bash
# round 6, visit zz28: AUTO = rows mapping at every size, half-pass chunk at every size: the whole GPU suite, maze soak, counters
# at 64 x 64, the families
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu --timeout 900 > $O/zz28_pytest.log 2>&1; echo "rc=$?"; grep -n "passed\|failed" $O/zz28_pytest.log | tail -2
PYTHONPATH=.:tests timeout 500 python tests/soak_maze.py 400 > $O/zz28_soak_maze.txt 2>&1; echo "soak rc=$?"; tail -2 $O/zz28_soak_maze.txt
export PMC_EXTRA="TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr"
XV_MAZE_STEPS=12 bash scripts/pmc_kernel.sh raycast_spec32_64 maze_raycast scripts/bench_families.py --families maze64 > $O/zz28_pmc_64.log 2>&1; tail -1 $O/zz28_pmc_64.log
unset PMC_EXTRA
for rep in 1 2; do
timeout 600 python scripts/bench_families.py --families maze64,maze256 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['workload'][-16:], d['filter'], {k: round(x, 1) for k, x in d['us_per_step'].items()})
"
done | tee $O/zz28_maze.txt
python - <<'PY'
import json
d = json.load(open("gpurun_out/pmc_raycast_spec32_64.json"))
for k, v in d["kernels"].items():
    px = 64 * 64 * 16384
    cyc = v["GRBM_GUI_ACTIVE"] / 8
    print(k[:60], "VALU/pixel %.1f busy %.3f waves/SIMD %.2f L1 acc/pixel %.2f TCC req/pixel %.2f miss %.3g" % (
        v["SQ_INSTS_VALU"] * 64 / px, v["SQ_INSTS_VALU"] * 4 / 1024 / cyc, v["SQ_WAVE_CYCLES"] * 4 / 1024 / cyc, v["TCP_TOTAL_CACHE_ACCESSES_sum"] / px, v["TCP_TCC_READ_REQ_sum"] / px, v["TCC_MISS_sum"]))
PY
