# round 4, trip f: device sampler up to 256 states
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_sampler.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r04_f_pytest.txt
cat gpurun_out/r04_f_pytest.txt
timeout 900 python - <<'PY' 2>&1 | tee gpurun_out/r04_f_sampler_throughput.jsonl
import json, time, torch
from xenoverse_amd.anymdp import device_sampler as ds
for S, A, n, batch in ((64, 8, 512, 2048), (128, 5, 256, 512), (256, 5, 128, 256)):
    out = ds.sample_tasks_device(n, S, A, seed=1, batch=batch)
    st = out["stats"]
    print(json.dumps({"S": S, "A": A, "accepted": st["accepted"], "candidates": st["candidates"], "seconds": st["seconds"],
                      "accepted_per_s": st["accepted"] / st["seconds"], "candidates_per_s": st["candidates"] / st["seconds"],
                      "status": st["status"], "batch": st["batch"]}), flush=True)
PY
