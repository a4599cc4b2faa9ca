cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r04_z_pytest_gpu.log 2>&1; echo "rc=$?"; grep -n "passed\|failed" gpurun_out/r04_z_pytest_gpu.log | tee gpurun_out/r04_z_pytest_gpu_tail.txt
grep -n "Error\|assert" gpurun_out/r04_z_pytest_gpu.log | head -20
