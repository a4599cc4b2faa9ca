#!/bin/bash
# round 5, call Q: overlapped token steps — parity, timing
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_q
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_chains.py tests/test_gpu_anymdp_tok.py -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.txt | cut -c1-200
timeout 600 python scripts/bench_families.py --families anymdp_tok_refdist --steps 640 > $O/tok.jsonl 2> $O/tok.err; echo "tok rc=$?"
python3 - $O/tok.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        for k, v in d["variants"].items():
            print(k, {a: (round(b["us_per_step"], 2), b.get("taken")) for a, b in v.items() if isinstance(b, dict) and "us_per_step" in b}, "errs", v["device_error_flags"])
PY
tail -3 $O/tok.err
