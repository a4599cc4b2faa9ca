#!/usr/bin/env python3
"""gpurun_out/pmc_{FETCH_SIZE,WRITE_SIZE}/**/counter_collection.csv -> gpurun_out/pmc_traffic_<workload>.json
(the file bench.py's `roofline.traffic` reads once it is copied to profiles/).  Units and correction as
/opt/skills/guides/MI355X_MICROARCH.md prescribes: counter values are KB; on gfx950 FETCH_SIZE tallies wide
coalesced reads at half, so read bytes = 2 x FETCH_SIZE; WRITE_SIZE is exact."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xenoverse_amd.build import source_hash

workload, search, envs, cmd = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
if search == "auto":      # what AUTO resolved to in the profiled run (bench.py's own line, written beside the counters)
    try:
        line = [ln for ln in open("gpurun_out/pmc_FETCH_SIZE.json") if ln.startswith('{"metric"')][-1]
        search = json.loads(line)["config"]["search"]
    except Exception as ex:
        sys.exit("cannot tell which search the profiled run used: %r" % (ex,))
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only beside them) -- " + cmd,
       "units": "counter values are KB; gfx950 correction per MI355X_MICROARCH.md HBM section: read bytes = 2 x FETCH_SIZE, "
                "WRITE_SIZE exact",
       "kernels": {}, "bench_key": {"workload": workload, "search": search, "envs_per_gpu": envs,
                                      "kernel_source_sha16": source_hash(("anymdp.hip", "philox.h", "xv_common.h"))}}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob("gpurun_out/pmc_%s/**/*counter_collection.csv" % c, recursive=True)
    if not fs:
        sys.exit("no counter file for " + c)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] == c:
            agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if not k.startswith(("anymdp", "void anymdp")):
            continue
        d = out["kernels"].setdefault(k, {})
        d[c + "_KB_avg"] = sum(v) / len(v)
        d["dispatches"] = len(v)
for k, d in out["kernels"].items():
    if "step_kernel" in k and "FETCH_SIZE_KB_avg" in d and "WRITE_SIZE_KB_avg" in d:
        d["traffic_bytes_per_launch_corrected"] = (2 * d["FETCH_SIZE_KB_avg"] + d["WRITE_SIZE_KB_avg"]) * 1024
        d["traffic_bytes_per_launch_uncorrected"] = (d["FETCH_SIZE_KB_avg"] + d["WRITE_SIZE_KB_avg"]) * 1024
        d["per_env_step_corrected_B"] = d["traffic_bytes_per_launch_corrected"] / envs
path = "gpurun_out/pmc_traffic_anymdp_%s.json" % workload
json.dump(out, open(path, "w"), indent=1)
print(path)
for k, d in out["kernels"].items():
    print("%-60s %s" % (k[:60], {a: round(b, 1) for a, b in d.items()}))
