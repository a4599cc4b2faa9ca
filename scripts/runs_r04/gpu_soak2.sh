cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PYTHONPATH=.:tests timeout 1500 python tests/soak_maze.py ${1:-240} 6 > gpurun_out/r04_soak_maze.txt 2>&1; echo "soak maze rc=$?"; tail -4 gpurun_out/r04_soak_maze.txt | cut -c1-300; grep -c "^ok" gpurun_out/r04_soak_maze.txt
