# final tree: whole GPU suite, smoke, maze soak incl. a resolution on the rows mapping, maze families
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r04_z_pytest_gpu.log 2>&1; echo "rc=$?"; grep -n "passed\|failed" gpurun_out/r04_z_pytest_gpu.log | tee gpurun_out/r04_z_pytest_gpu_tail.txt
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
PYTHONPATH=.:tests timeout 900 python tests/soak_maze.py 240 41 > gpurun_out/r04_soak4_maze.txt 2>&1; echo "soak rc=$?"; tail -1 gpurun_out/r04_soak4_maze.txt; grep -c "res=(160, 120)" gpurun_out/r04_soak4_maze.txt
timeout 900 python scripts/bench_families.py --families maze64,maze64_direct,maze64_f32,maze256,maze256_direct,maze256_f32 > gpurun_out/r04_z_bench_families_maze.jsonl 2>/dev/null; python - <<PY
import json
for l in open("gpurun_out/r04_z_bench_families_maze.jsonl"):
    if l.startswith("{"):
        d = json.loads(l); print(d["filter"], d["workload"][-16:], {k: round(v, 1) for k, v in d["us_per_step"].items()})
PY
