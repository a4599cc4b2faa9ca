#!/bin/bash
# HBM traffic of the AnyMDP step kernel from PMC counters: separate passes for FETCH_SIZE and WRITE_SIZE
# (TCC slots: they do not fit one pass), kernel-trace only beside them.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
ARGS="${1:---steps 200 --warmup 20 --no-cpu-baseline}"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -o pmc -- python3 bench.py $ARGS > gpurun_out/pmc_$c.json 2> gpurun_out/pmc_$c.err
  echo "$c rc=$?"
done
python3 - <<'PY'
import csv, glob, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob("gpurun_out/pmc_%s/**/*counter_collection.csv" % c, recursive=True)
    if not fs:
        print(c, "no counter file", glob.glob("gpurun_out/pmc_%s/**/*" % c, recursive=True)[:5]); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] == c:
            agg[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:8]:
        print("%-11s %-62s n=%5d avg=%14.1f sum=%16.1f" % (c, k, len(v), sum(v) / len(v), sum(v)))
PY
