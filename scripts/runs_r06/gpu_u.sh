#!/bin/bash
# round 6, visit u: the 64 x 64 ray caster is bound by L1 (TCP) accesses — 8.15 per pixel at 0.96 per CU and cycle
# (r06_m_pmc_raycast_spec32_64.json): 4 dwordx4 + 4 dword loads per pixel, the latter hipcc's narrowing of the third pair span.
# A/B of the window fetch: pair copy as is (8 loads) / pair copy with the third span as two 16-byte loads (6) / row-major copy (4)
# / rows mapping (4, coalesced across lanes)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
run() {  # tag
  timeout 600 python scripts/bench_families.py --families maze64 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', d['workload'][-14:], d['filter'], {k: round(x, 1) for k, x in d['us_per_step'].items()})
"
}
for rep in 1 2; do
  unset XV_LIB_PATH XV_MAZE_FILT XV_MAZE_MAPPING
  run pairs_8_loads
  XV_LIB_PATH=$PWD/scripts/devtools/_build/libxeno_mzspan2.so run pairs_6_loads
  XV_MAZE_FILT=3 run rowmajor_4_loads
  XV_MAZE_FILT=5 run rows_mapping
done | tee $O/u_maze_fetch_ab.txt
