#!/bin/bash
# round 5, run t: side stream chosen by measurement + cycle gate — parity, then 2a / mixed with an RCCL communicator made first
export TMPDIR=/tmp
O=gpurun_out/r05_t
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_chains.py tests/test_gpu_mixed_shard.py tests/test_gpu_mixed.py tests/test_gpu_capture.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.txt
timeout 100 python scripts/devtools/probe_side_streams.py 2>&1 | grep round > $O/side_streams_plain.txt
timeout 100 python scripts/devtools/probe_side_streams.py --rccl-first 2>&1 | grep round > $O/side_streams_rccl_first.txt
cat $O/side_streams_plain.txt $O/side_streams_rccl_first.txt
for ov in off on; do
  timeout 300 python bench.py --workload mixed --overlap $ov --steps 2048 --warmup 256 > $O/bench_mixed_$ov.json 2> $O/bench_mixed_$ov.err; echo "bench mixed $ov rc=$?"
done
timeout 300 python bench.py --steps 2048 --warmup 256 > $O/bench_2a.json 2> $O/bench_2a.err; echo "bench 2a rc=$?"
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_2a_steps20.json 2> $O/bench_2a_steps20.err; echo "bench 2a steps20 rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05_t/bench_*.json")):
    for l in open(f).read().splitlines():
        if l.startswith("{"):
            d = json.loads(l)
            c = d["config"]
            print(f.split("/")[-1], "value %.4g" % d["value"], "us/step %.3f" % (d["ms_per_step"] * 1e3), "overlap", c.get("overlap"),
                  "flags", c.get("device_error_flags", d.get("device_errors")), "gather", (d.get("with_allgather") or {}).get("value"))
PY
