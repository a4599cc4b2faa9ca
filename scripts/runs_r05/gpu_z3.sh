#!/bin/bash
# round 5, the record, bench lines again with the PMC profile of the same kernel source in profiles/ (frac_traffic)
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_z9
mkdir -p $O
timeout 900 python bench.py > $O/bench_2a.json 2> $O/bench_2a.err; echo "bench 2a rc=$?"
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_2a_steps20.json 2> $O/bench_2a_steps20.err; echo "bench steps20 rc=$?"
timeout 600 python bench.py --tasks 1024 --no-cpu-baseline --no-families > $O/bench_2b.json 2> $O/bench_2b.err; echo "bench 2b rc=$?"
timeout 600 python bench.py --workload mixed > $O/bench_mixed_n1.json 2> $O/bench_mixed_n1.err; echo "bench mixed rc=$?"
timeout 600 python bench.py --workload mixed --overlap off --no-cpu-baseline > $O/bench_mixed_n1_one_stream.json 2> $O/bench_mixed_n1_one_stream.err; echo "bench mixed one stream rc=$?"
XV_MIXED_PIPE_MIN_STEPS=32 timeout 600 python bench.py --workload mixed --no-cpu-baseline > $O/bench_mixed_n1_min32.json 2> $O/bench_mixed_n1_min32.err; echo "bench mixed min32 rc=$?"
for f in bench_2a bench_2a_steps20 bench_2b bench_mixed_n1 bench_mixed_n1_one_stream bench_mixed_n1_min32; do python3 - $O/$f.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1])
r = d.get("roofline") or {}
print("%-34s value %.4g ms/step %.5f frac %.3f frac_traffic %s overlap %s errs %s gather %s" % (sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"],
      r.get("frac", 0), r.get("frac_traffic"), d["config"].get("overlap"), d["config"].get("device_error_flags"), (d.get("with_allgather") or {}).get("value")))
PY
done
