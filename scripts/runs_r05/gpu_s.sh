#!/bin/bash
# round 5, run s: overlapped xv_mixed_step_many — parity, then the mixed bench with and without it
export TMPDIR=/tmp
mkdir -p gpurun_out/r05_s
O=gpurun_out/r05_s
timeout 600 python -m pytest tests/test_gpu_mixed_shard.py tests/test_gpu_mixed.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.txt
for ov in off on; do
  timeout 300 python bench.py --workload mixed --overlap $ov --steps 2048 --warmup 256 > $O/bench_mixed_$ov.json 2> $O/bench_mixed_$ov.err; echo "bench $ov rc=$?"
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_mixed_$ov.json").read().strip().splitlines()[-1])
    print("$ov", d["value"], d["ms_per_step"], d.get("roofline", {}).get("frac"), d.get("overlap"), d.get("device_errors"))
except Exception as e:
    print("$ov failed", e); print(open("$O/bench_mixed_$ov.err").read()[-2000:])
PY
done
