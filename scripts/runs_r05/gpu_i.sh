#!/bin/bash
# round 5, call I: ray caster with the constant-weight wall filter — parity (tests, soaks), timing 64x64 / 256x256
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_i
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_maze.py -x -q > $O/pytest_maze.txt 2>&1; echo "pytest maze rc=$?"; tail -4 $O/pytest_maze.txt
for fam in maze64 maze64_direct maze256; do
  timeout 600 python scripts/bench_families.py --families $fam 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$fam', {k: round(x, 1) for k, x in d['us_per_step'].items()})
"
done
PYTHONPATH=.:tests timeout 400 python tests/soak_maze.py 120 > $O/soak_maze.txt 2>&1; echo "soak maze rc=$?"; tail -3 $O/soak_maze.txt
PYTHONPATH=.:tests timeout 400 python scripts/devtools/soak_spec_filter.py 120 > $O/soak_spec.txt 2>&1; echo "soak spec rc=$?"; tail -3 $O/soak_spec.txt
