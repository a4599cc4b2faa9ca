# third round of the soaks, at the final source (after ABI 9), other master seeds
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PYTHONPATH=.:tests timeout 1500 python tests/soak_anymdp.py 300 21 > gpurun_out/r04_soak3_anymdp.txt 2>&1; echo "anymdp rc=$?"; tail -1 gpurun_out/r04_soak3_anymdp.txt
PYTHONPATH=.:tests timeout 1500 python tests/soak_maze.py 240 22 > gpurun_out/r04_soak3_maze.txt 2>&1; echo "maze rc=$?"; tail -1 gpurun_out/r04_soak3_maze.txt
PYTHONPATH=.:tests timeout 1500 python tests/soak_linds.py 120 23 > gpurun_out/r04_soak3_linds.txt 2>&1; echo "linds rc=$?"; tail -1 gpurun_out/r04_soak3_linds.txt
