# two ranks sharing the box's one GPU (functional, not a measurement), started the way the driver starts N = 1:
# no launcher in front — bench.py starts its own ranks (self_launch)
cd $GRAFT_REPO_ROOT
export XV_BENCH_SHARE_GPU=1
export MASTER_PORT=29511
timeout 600 python bench.py --gpus 2 --steps 200 --warmup 20 --tasks 1024 --no-cpu-baseline "$@" 2>&1 | grep '^{"metric"' | tail -1
