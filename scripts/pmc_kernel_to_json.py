#!/usr/bin/env python3
"""gpurun_out/pmc_<tag>_g*/**/counter_collection.csv -> gpurun_out/pmc_<tag>.json: per kernel (name contains the
given substring) the average of every collected counter per dispatch, plus derived figures.  Units as
/opt/skills/guides/MI355X_MICROARCH.md states: FETCH_SIZE / WRITE_SIZE in KB, gfx950 reads of 16 B/lane tallied at half
(read bytes = 2 x FETCH_SIZE); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* in quad-cycles summed over waves."""
import collections
import csv
import glob
import json
import sys

tag, ksub, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
per = collections.defaultdict(lambda: collections.defaultdict(list))
extra = {}
for f in sorted(glob.glob("gpurun_out/pmc_%s_g*/**/*counter_collection.csv" % tag, recursive=True)):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0]
        if ksub not in name:
            continue
        per[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size"):
            if k in r:
                extra.setdefault(name, {})[k] = r[k]
out = {"source": "rocprofv3 --pmc <one group per pass> --kernel-trace -- " + cmd, "kernels": {}}
for name, cs in per.items():
    d = {c: sum(v) / len(v) for c, v in cs.items()}
    d["dispatches"] = max(len(v) for v in cs.values())
    d.update(extra.get(name, {}))
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_bytes_per_launch_corrected"] = (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024
    if d.get("SQ_WAVE_CYCLES"):
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM"):
            if k in d:
                d[k + "_over_WAVE_CYCLES"] = d[k] / d["SQ_WAVE_CYCLES"]
    if d.get("SQ_WAVES"):
        for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS",
                  "SQ_INSTS_MFMA"):
            if k in d:
                d[k + "_per_wave"] = d[k] / d["SQ_WAVES"]
    out["kernels"][name] = d
path = "gpurun_out/pmc_%s.json" % tag
json.dump(out, open(path, "w"), indent=1)
print(path)
for k, d in out["kernels"].items():
    print(k[:100])
    for a, b in sorted(d.items()):
        print("   %-40s %s" % (a, b))
