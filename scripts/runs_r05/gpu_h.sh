#!/bin/bash
# round 5, call H: config 5 across ranks — shard/exchange parity, the mixed bench line at N = 1 (one-rank communicator),
# two ranks sharing the GPU (functional), the shared-engine fused step
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_h
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_mixed_shard.py tests/test_gpu_capture.py tests/test_gpu_mixed.py -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.txt
timeout 900 python bench.py --workload mixed > $O/bench_mixed_n1.json 2> $O/bench_mixed_n1.err; echo "mixed n1 rc=$?"; cut -c1-2500 $O/bench_mixed_n1.json; tail -3 $O/bench_mixed_n1.err
export XV_BENCH_SHARE_GPU=1
export MASTER_PORT=29533
timeout 900 python bench.py --workload mixed --gpus 2 --steps 320 --warmup 64 --repeats 5 --no-cpu-baseline > $O/bench_mixed_n2_shared_gpu.json 2> $O/bench_mixed_n2_shared_gpu.err; echo "mixed n2 (one GPU shared) rc=$?"
grep '^{"metric"' $O/bench_mixed_n2_shared_gpu.json | cut -c1-2500; tail -5 $O/bench_mixed_n2_shared_gpu.err
export MASTER_PORT=29544
timeout 900 python bench.py --gpus 2 --steps 200 --warmup 20 --tasks 1024 --no-cpu-baseline > $O/bench_n2_shared_gpu.json 2> $O/bench_n2_shared_gpu.err; echo "anymdp n2 (one GPU shared) rc=$?"
grep '^{"metric"' $O/bench_n2_shared_gpu.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('n_gpus',d['n_gpus'],'value %.3g'%d['value'],'rccl',d['rccl'],d['rccl_ranks'],'transport',d['transport'],'with_allgather',d.get('with_allgather'))
print('families.mixed', json.dumps(d.get('families',{}).get('mixed'))[:1500])
"; tail -3 $O/bench_n2_shared_gpu.err
