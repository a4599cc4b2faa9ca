#!/bin/bash
# round 6, visit q: long soaks and two more suite runs on the final tree
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06_q
mkdir -p $O
for i in 1 2; do
  timeout 1200 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $O/pytest_gpu_$i.txt 2>&1; echo "pytest $i rc=$?"; grep -n "passed\|failed" $O/pytest_gpu_$i.txt | tail -1
done
PYTHONPATH=.:tests timeout 700 python tests/soak_anymdp.py 600 > $O/soak_anymdp.txt 2>&1; echo "soak anymdp rc=$?"; tail -1 $O/soak_anymdp.txt | cut -c1-300
PYTHONPATH=.:tests timeout 700 python tests/soak_mixed.py 600 > $O/soak_mixed.txt 2>&1; echo "soak mixed rc=$?"; tail -1 $O/soak_mixed.txt | cut -c1-300
PYTHONPATH=.:tests timeout 700 python tests/soak_maze.py 600 > $O/soak_maze.txt 2>&1; echo "soak maze rc=$?"; tail -1 $O/soak_maze.txt | cut -c1-300
PYTHONPATH=.:tests timeout 400 python tests/soak_linds.py 300 > $O/soak_linds.txt 2>&1; echo "soak linds rc=$?"; tail -1 $O/soak_linds.txt | cut -c1-300
python scripts/devtools/gpu_hog.py 330 > $O/hog.txt 2>&1 &
HOG=$!
sleep 5
PYTHONPATH=.:tests timeout 200 python tests/soak_mixed.py 150 > $O/soak_mixed_beside_hog.txt 2>&1; echo "soak mixed beside hog rc=$?"; tail -1 $O/soak_mixed_beside_hog.txt | cut -c1-300
PYTHONPATH=.:tests timeout 200 python tests/soak_anymdp.py 150 > $O/soak_anymdp_beside_hog.txt 2>&1; echo "soak anymdp beside hog rc=$?"; tail -1 $O/soak_anymdp_beside_hog.txt | cut -c1-300
kill $HOG 2>/dev/null; wait $HOG 2>/dev/null
for k in matmul allgather; do
  PROBE_DUMP_S=100 timeout 150 python scripts/devtools/probe_neighbour.py $k 8 > $O/neighbour_$k.txt 2>&1; echo "neighbour $k rc=$?"; grep "overlap 1 call  7" $O/neighbour_$k.txt
done
