#!/bin/bash
# functional check of bench.py's all-gather watchdog on a 1-GPU box: two ranks share the GPU, gloo backend, the
# gather pass is forced and given 5 s; whatever happens to it, rank 0 must still print the pass-1 JSON line.
cd $GRAFT_REPO_ROOT
export XV_BENCH_SHARE_GPU=1 XV_BENCH_BACKEND=gloo XV_BENCH_FORCE_GATHER=1
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 200 --warmup 20 --tasks 1024 --no-cpu-baseline --gather-timeout ${GT:-5} 2>&1 | grep -E '^\{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['config']['exchange'], d.get('with_allgather'))"
