# round 4, trip a: cut-line bucket search — parity tests, then fence vs bucket on reference-distribution tasks
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_anymdp.py tests/test_gpu_anymdp_tok.py tests/test_gpu_mixed.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r04_a_pytest.txt
cat gpurun_out/r04_a_pytest.txt
timeout 600 python scripts/devtools/probe_real_tasks.py 2>&1 | tail -30 > gpurun_out/r04_a_probe_real_tasks.txt
cat gpurun_out/r04_a_probe_real_tasks.txt
