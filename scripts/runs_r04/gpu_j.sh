# round 4, trip j: pair-interleaved packed textures (ray caster) against row-major ones
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
V=$GRAFT_REPO_ROOT/scripts/devtools/_build/libxeno_rows.so
timeout 1200 python -m pytest tests/test_gpu_maze.py tests/test_gpu_maze_agent.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -4
for i in 1 2; do
  for L in pairs rows; do
    if [ $L = rows ]; then export XV_LIB_PATH=$V; else unset XV_LIB_PATH; fi
    timeout 600 python scripts/bench_families.py --families maze64,maze64_f32,maze256,maze256_f32 --steps 400 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print('$L', d['workload'][-14:], d['filter'], {k: round(v,1) for k,v in d['us_per_step'].items()})"
  done
done | tee gpurun_out/r04_j_ab_tex_pairs.txt
