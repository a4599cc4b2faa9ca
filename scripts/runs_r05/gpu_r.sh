#!/bin/bash
# round 5, call R: quickstart example, soaks on the final tree
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_r
mkdir -p $O
timeout 600 python examples/quickstart.py > $O/quickstart.txt 2>&1; echo "quickstart rc=$?"; tail -6 $O/quickstart.txt | cut -c1-250
PYTHONPATH=.:tests timeout 400 python tests/soak_anymdp.py 240 > $O/soak_anymdp.txt 2>&1; echo "soak anymdp rc=$?"; tail -2 $O/soak_anymdp.txt | cut -c1-300
PYTHONPATH=.:tests timeout 300 python tests/soak_maze.py 150 > $O/soak_maze.txt 2>&1; echo "soak maze rc=$?"; tail -1 $O/soak_maze.txt | cut -c1-300
PYTHONPATH=.:tests timeout 300 python tests/soak_linds.py 120 > $O/soak_linds.txt 2>&1; echo "soak linds rc=$?"; tail -1 $O/soak_linds.txt | cut -c1-300
