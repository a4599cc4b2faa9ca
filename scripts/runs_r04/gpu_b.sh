# round 4, trip b: device tick + captured loops
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_capture.py tests/test_gpu_anymdp.py -x -q -m gpu 2>&1 | tail -30 > gpurun_out/r04_b_pytest.txt
cat gpurun_out/r04_b_pytest.txt
