#!/bin/bash
# round 6, visit zz24: counters of the move kernel (maze_step9_kernel<9>) in the 64 x 64 bench
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
XV_MAZE_STEPS=12 bash scripts/pmc_kernel.sh maze_move maze_step9 scripts/bench_families.py --families maze64 > $O/zz24_pmc_move.log 2>&1; tail -1 $O/zz24_pmc_move.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/pmc_maze_move.json"))
for k, v in d["kernels"].items():
    cyc = v["GRBM_GUI_ACTIVE"] / 8
    print(k[:70])
    print({a: (round(b, 2) if isinstance(b, float) else b) for a, b in v.items()})
    print("cycles/XCD %.4g  waves %d  VALU/wave %.0f  SALU/wave %.0f  LDS/wave %.0f  busy %.3f  avg waves/SIMD %.2f" % (
        cyc, v["SQ_WAVES"], v["SQ_INSTS_VALU"] / v["SQ_WAVES"], v["SQ_INSTS_SALU"] / v["SQ_WAVES"], v["SQ_INSTS_LDS"] / v["SQ_WAVES"],
        v["SQ_INSTS_VALU"] * 4 / 1024 / cyc, v["SQ_WAVE_CYCLES"] * 4 / 1024 / cyc))
PY
