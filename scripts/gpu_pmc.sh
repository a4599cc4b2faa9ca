#!/bin/bash
# HBM traffic of the AnyMDP step kernel from PMC counters: separate passes for FETCH_SIZE and WRITE_SIZE
# (TCC slots: they do not fit one pass), kernel-trace only beside them.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
ARGS="${1:---steps 200 --warmup 20 --no-cpu-baseline --no-families --fused}"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -o pmc -- python3 bench.py $ARGS > gpurun_out/pmc_$c.json 2> gpurun_out/pmc_$c.err
  echo "$c rc=$?"
done
WL="${2:-2a}"
python3 scripts/pmc_to_json.py "$WL" auto 65536 "python3 bench.py $ARGS"
