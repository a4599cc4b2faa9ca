#!/bin/bash
# device sampler: parity + throughput
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_sampler.py -m gpu -x -q > gpurun_out/pytest_sampler.log 2>&1; echo "rc=$?"; tail -25 gpurun_out/pytest_sampler.log
timeout 600 python - <<'PY'
import sys, time, json
sys.path.insert(0, ".")
import torch
from xenoverse_amd.anymdp import device_sampler as ds
from xenoverse_amd.engine import Engine
eng = Engine("cuda:0")
for S, A, n in ((16, 4, 16384), (64, 8, 4096), (64, 5, 4096)):
    ds.sample_candidates(eng, 1, 0, 256, S, A, tables=True); eng.sync()
    t0 = time.perf_counter()
    r = ds.sample_candidates(eng, 1, 1000, n, S, A, tables=True); eng.sync()
    dt = time.perf_counter() - t0
    st = r["status"].cpu().numpy()
    acc = int((st == 0).sum())
    print(json.dumps({"S": S, "A": A, "candidates": n, "seconds": dt, "candidates_per_s": n / dt, "accepted": acc,
                      "accepted_per_s": acc / dt, "status_hist": [int((st == k).sum()) for k in range(5)]}))
PY
