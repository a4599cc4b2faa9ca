#!/bin/bash
# round 6, visit v: counters of the 64 x 64 ray cast under the fetch variants of visit u
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
export PMC_EXTRA="TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr|TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
XV_MAZE_STEPS=12 XV_LIB_PATH=$PWD/scripts/devtools/_build/libxeno_mzspan2.so bash scripts/pmc_kernel.sh raycast_64_pairs6 maze_raycast scripts/bench_families.py --families maze64 > $O/v_pmc_pairs6.log 2>&1; tail -2 $O/v_pmc_pairs6.log
XV_MAZE_STEPS=12 XV_MAZE_FILT=5 bash scripts/pmc_kernel.sh raycast_64_rows maze_raycast scripts/bench_families.py --families maze64 > $O/v_pmc_rows.log 2>&1; tail -2 $O/v_pmc_rows.log
python - <<'PY'
import json
for t in ("pairs6", "rows"):
    d = json.load(open("gpurun_out/pmc_raycast_64_%s.json" % t))
    for k, v in d["kernels"].items():
        px = 64 * 64 * 16384
        cyc = v["GRBM_GUI_ACTIVE"] / 8
        print(t, k[:60])
        print({a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a.startswith(("TCC", "TCP", "TA_"))})
        print("  cycles/XCD %.3g  VALU/pixel %.1f  busy %.3f  TCP acc/CU/cycle %.3f  acc/pixel %.2f  TCC req/pixel %.2f" % (
            cyc, v["SQ_INSTS_VALU"] * 64 / px, v["SQ_INSTS_VALU"] * 4 / 1024 / cyc, v["TCP_TOTAL_CACHE_ACCESSES_sum"] / 256 / cyc,
            v["TCP_TOTAL_CACHE_ACCESSES_sum"] / px, v["TCP_TCC_READ_REQ_sum"] / px))
PY
