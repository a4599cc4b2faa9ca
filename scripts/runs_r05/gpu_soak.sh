#!/bin/bash
# round 5: longer soaks on the final tree
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_soak
mkdir -p $O
PYTHONPATH=.:tests timeout 700 python tests/soak_anymdp.py 600 > $O/soak_anymdp.txt 2>&1; echo "soak anymdp rc=$?"; tail -1 $O/soak_anymdp.txt; grep -c "ov1" $O/soak_anymdp.txt
PYTHONPATH=.:tests timeout 400 python tests/soak_maze.py 300 > $O/soak_maze.txt 2>&1; echo "soak maze rc=$?"; tail -1 $O/soak_maze.txt
PYTHONPATH=.:tests timeout 300 python tests/soak_linds.py 200 > $O/soak_linds.txt 2>&1; echo "soak linds rc=$?"; tail -1 $O/soak_linds.txt
