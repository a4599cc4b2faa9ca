# round 4, trip e: observation cut lines (token step), noise distribution test, token benches
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_anymdp_tok.py tests/test_gpu_linds.py tests/test_gpu_maze.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r04_e_pytest.txt
cat gpurun_out/r04_e_pytest.txt
timeout 600 python scripts/bench_families.py --families anymdp_tok_refdist,anymdp_tok,python_loop --steps 400 > gpurun_out/r04_e_families.jsonl 2> gpurun_out/r04_e_families.err
cat gpurun_out/r04_e_families.jsonl | cut -c1-1800; tail -3 gpurun_out/r04_e_families.err
