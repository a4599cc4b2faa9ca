cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_anymdp_tok.py tests/test_gpu_anymdp.py tests/test_gpu_fullsize.py tests/test_gpu_capture.py -m gpu -q -x > gpurun_out/r04_tokinfo_pytest.log 2>&1; echo "pytest rc=$? $(grep -h 'passed\|failed' gpurun_out/r04_tokinfo_pytest.log | tail -1)"; grep -n "^FAILED\|^E  " gpurun_out/r04_tokinfo_pytest.log | head -5
PYTHONPATH=. timeout 600 python scripts/devtools/probe_python_tok_step.py 2>&1 | grep "us per"
