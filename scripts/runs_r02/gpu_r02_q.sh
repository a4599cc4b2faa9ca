#!/bin/bash
# checkpoint: whole GPU suite + smoke + default bench + driver-flag bench
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_q.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_q.log
tail -4 gpurun_out/pytest_q.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/bench_q.json 2> gpurun_out/bench_q.err
cat gpurun_out/bench_q.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_q20.json 2> gpurun_out/bench_q20.err
cat gpurun_out/bench_q20.json
