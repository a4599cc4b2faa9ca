# soak of the speculated filter + the new wide-format token tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_anymdp_tok.py -m gpu -q -x 2>&1 | tail -3
PYTHONPATH=. timeout 900 python scripts/devtools/soak_spec_filter.py 240 > gpurun_out/r04_v_soak_spec_filter.txt 2>&1; echo "soak rc=$?"; tail -4 gpurun_out/r04_v_soak_spec_filter.txt; grep -c pixels gpurun_out/r04_v_soak_spec_filter.txt
