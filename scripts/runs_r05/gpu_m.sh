#!/bin/bash
# round 5, call M: ray caster LIST mapping — parity, timing against the other mappings; python loop with slabs
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_m
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_maze.py -x -q -k "mapping" > $O/pytest_maze_mapping.txt 2>&1; echo "pytest mapping rc=$?"; tail -4 $O/pytest_maze_mapping.txt
for m in auto list columns rows; do
  for fam in maze64 maze256; do
    XV_MAZE_MAPPING=$m timeout 600 python scripts/bench_families.py --families $fam 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$m', '$fam', {k: round(x, 1) for k, x in d['us_per_step'].items()})
"
  done
done | tee $O/mapping_ab.txt
PYTHONPATH=.:tests timeout 300 python tests/soak_maze.py 150 > $O/soak_maze.txt 2>&1; echo "soak maze rc=$?"; tail -2 $O/soak_maze.txt
timeout 600 python scripts/bench_families.py --families python_loop > $O/python_loop.jsonl 2> $O/python_loop.err; echo "python_loop rc=$?"; cut -c1-700 $O/python_loop.jsonl
