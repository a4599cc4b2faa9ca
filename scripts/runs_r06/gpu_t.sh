#!/bin/bash
# round 6, visit t: what the 64 x 64 ray caster would run at without the LDS frame chunk capping its occupancy (timing-only builds:
# XV_MAZE_TIMING_ALIAS puts every column on the same LDS bytes — wrong frames, same instructions), at 3 and 4 waves per SIMD;
# and the move kernel on compacted batches.  The macro was a throwaway (three #ifdef lines in maze.hip: cstride = 4, the row table at
# LDS byte 2,048, lds_bytes = 2,048 + 16 H); it is not in the tree — record: profiles/r06_t_raycast_64_without_lds_cap_timing_only_not_kept.txt
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
for rep in 1 2; do
  for v in base alias alias4; do
    if [ $v = base ]; then unset XV_LIB_PATH; else export XV_LIB_PATH=$PWD/scripts/devtools/_build/libxeno_mz$v.so; fi
    timeout 600 python scripts/bench_families.py --families maze64 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$v', d['workload'][-14:], d['filter'], {k: round(x, 1) for k, x in d['us_per_step'].items()})
"
  done
done | tee $O/t_maze_alias.txt
unset XV_LIB_PATH
timeout 300 python scripts/devtools/probe_maze_move.py 2>&1 | tail -8 | tee $O/t_maze_move.txt
