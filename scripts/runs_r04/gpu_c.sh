# round 4, trip c: remaining anymdp tests, floor probe, bench with the new JSON fields, families incl. refdist + python_loop
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_anymdp.py -x -q -m gpu -k "six_cut or census or auto_keeps or overflow or rebuilding" 2>&1 | tail -8 > gpurun_out/r04_c_pytest.txt
cat gpurun_out/r04_c_pytest.txt
timeout 600 python scripts/devtools/floor_probe.py gpurun_out/r04_c_floor_probe.json 2>&1 | tail -3
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_c_bench_2a_steps20.json 2> gpurun_out/r04_c_bench_2a_steps20.err
tail -c 6000 gpurun_out/r04_c_bench_2a_steps20.json; tail -5 gpurun_out/r04_c_bench_2a_steps20.err
