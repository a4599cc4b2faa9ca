# second, longer round of the randomised soaks with other master seeds
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PYTHONPATH=.:tests timeout 1500 python tests/soak_anymdp.py 600 11 > gpurun_out/r04_soak2_anymdp.txt 2>&1; echo "anymdp rc=$?"; tail -1 gpurun_out/r04_soak2_anymdp.txt
PYTHONPATH=.:tests timeout 1500 python tests/soak_maze.py 600 12 > gpurun_out/r04_soak2_maze.txt 2>&1; echo "maze rc=$?"; tail -1 gpurun_out/r04_soak2_maze.txt
PYTHONPATH=.:tests timeout 1500 python tests/soak_linds.py 300 13 > gpurun_out/r04_soak2_linds.txt 2>&1; echo "linds rc=$?"; tail -1 gpurun_out/r04_soak2_linds.txt
