#!/bin/bash
# round 4, last visit: the whole GPU suite and the quick tour at the final tree, then the record (scripts/runs_r04/gpu_record.sh)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r04_z_pytest_gpu.log 2>&1; echo "rc=$?"; grep -n "passed\|failed" gpurun_out/r04_z_pytest_gpu.log | tee gpurun_out/r04_z_pytest_gpu_tail.txt
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
echo "== quickstart"; timeout 600 python examples/quickstart.py 2>&1 | grep -v "amdgpu.ids" | tail -12
bash scripts/runs_r04/gpu_record.sh r04_z
echo "== families (token step, python loop)"; timeout 600 python scripts/bench_families.py --families anymdp_tok,anymdp_tok_refdist,python_loop --steps 800 > gpurun_out/r04_z_bench_tok_python_loop.jsonl 2>/dev/null; cut -c1-600 gpurun_out/r04_z_bench_tok_python_loop.jsonl
rm -rf gpurun_out/prof_tok
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tok -o t -- python3 scripts/bench_families.py --families anymdp_tok --steps 400 > /dev/null 2>&1
f=$(find gpurun_out/prof_tok -name "*kernel_stats.csv" | head -1); head -1 $f > gpurun_out/r04_z_kernel_stats_anymdp_tok.csv; grep tok_step $f >> gpurun_out/r04_z_kernel_stats_anymdp_tok.csv; cut -c1-200 gpurun_out/r04_z_kernel_stats_anymdp_tok.csv
