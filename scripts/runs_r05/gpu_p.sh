#!/bin/bash
# round 5, call P: 20-step bursts as launches without the queue barrier (hipExtAnyOrderLaunch) + record hand-off, one stream
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_p
mkdir -p $O
for v in off on; do
  unset XV_ANYMDP_ANYORDER; [ $v = on ] && export XV_ANYMDP_ANYORDER=1
  timeout 300 python scripts/devtools/probe_chains.py --tag anyorder_$v --ks 1 --overlap --repeats 5 --steps 640 > $O/anyorder_$v.jsonl 2> $O/anyorder_$v.err
  echo "anyorder=$v rc=$?"
  python3 - $O/anyorder_$v.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print("  %-8s us/step %.3f  short %.3f (min %.3f)  err %s state %s" % (d["how"], d["us_per_step"], d["short_us_per_step"], d["short_us_min"], d["device_error_flags"], d["overlap_state"]))
PY
done
export XV_ANYMDP_ANYORDER=1
timeout 600 python -m pytest tests/test_gpu_chains.py -x -q -k "overlapped" > $O/pytest.txt 2>&1; echo "pytest (any-order on short calls) rc=$?"; tail -3 $O/pytest.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-families > $O/bench_steps20_anyorder.json 2> $O/bench_steps20_anyorder.err; echo "bench rc=$?"
python3 - $O/bench_steps20_anyorder.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1])
print("steps20 any-order: value %.4g ms/step %.5f events %.3f overlap %s errs %s" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], d["config"]["overlap"], d["config"]["device_error_flags"]))
PY
