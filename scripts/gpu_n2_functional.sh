cd $GRAFT_REPO_ROOT
export XV_BENCH_SHARE_GPU=1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 200 --warmup 20 --tasks 1024 --no-cpu-baseline "$@" 2>&1 | grep '^{"metric"' | tail -1
