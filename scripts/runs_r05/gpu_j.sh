#!/bin/bash
# round 5, call J: ray caster, wave-uniform wall filter: timing + parity
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_j
mkdir -p $O
for fam in maze64 maze256; do
  timeout 600 python scripts/bench_families.py --families $fam 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$fam', {k: round(x, 1) for k, x in d['us_per_step'].items()})
"
done
timeout 1200 python -m pytest tests/test_gpu_maze.py -x -q > $O/pytest_maze.txt 2>&1; echo "pytest maze rc=$?"; tail -2 $O/pytest_maze.txt
PYTHONPATH=.:tests timeout 200 python scripts/devtools/soak_spec_filter.py 60 > $O/soak_spec.txt 2>&1; echo "soak spec rc=$?"; tail -1 $O/soak_spec.txt
