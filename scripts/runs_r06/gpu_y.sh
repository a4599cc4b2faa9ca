#!/bin/bash
# round 6, visit y: the frame flush as nontemporal stores (the 201 MB / 3.2 GB of frames a launch writes pass through the L2s that
# hold the 4.3 MB of texels): A/B at 64 x 64 and 256 x 256, L2 counters
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
run() {  # tag families
  timeout 600 python scripts/bench_families.py --families $2 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', d['workload'][-16:], d['filter'], {k: round(x, 1) for k, x in d['us_per_step'].items()})
"
}
for rep in 1 2; do
  unset XV_LIB_PATH
  run base maze64,maze256
  XV_LIB_PATH=$PWD/scripts/devtools/_build/libxeno_mznt.so run nt_flush maze64,maze256
done | tee $O/y_maze_nt_ab.txt
export PMC_EXTRA="TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr"
XV_MAZE_STEPS=12 XV_LIB_PATH=$PWD/scripts/devtools/_build/libxeno_mznt.so bash scripts/pmc_kernel.sh raycast_64_nt maze_raycast scripts/bench_families.py --families maze64 > $O/y_pmc_64.log 2>&1; tail -2 $O/y_pmc_64.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/pmc_raycast_64_nt.json"))
for k, v in d["kernels"].items():
    print(k[:60], {a: round(b) for a, b in v.items() if a.startswith(("TCC", "FETCH", "WRITE", "SQ_WAVE_CYCLES", "GRBM"))})
PY
