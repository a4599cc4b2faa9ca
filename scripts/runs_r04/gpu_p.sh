cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r04_z_pytest_gpu.log 2>&1; echo "rc=$?"; grep -n "passed\|failed" gpurun_out/r04_z_pytest_gpu.log | tee gpurun_out/r04_z_pytest_gpu_tail.txt
grep -n "^FAILED\|^ERROR" gpurun_out/r04_z_pytest_gpu.log | head -20
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_z2_bench_2a_steps20.json 2>/dev/null
python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r04_z2_bench_2a_steps20.json") if l.startswith('{"metric"')][-1])
r = d["roofline"]
print("value %.4e kernel us %.3f search %s primary %s frac_traffic %s traffic_src %s" % (d["value"], r["avg_launch_us"], d["config"]["search"], r["primary"], r["frac_traffic"], r["traffic_source"]))
for k, v in (d.get("families") or {}).items():
    print("   ", k, {a: v.get(a) for a in ("ms_per_step", "error", "auto_search")}, (v.get("roofline") or {}).get("frac"))
PY
