#!/bin/bash
# round 6, visit zz27: exact filter, columns mapping (pair copy, prefetch) against rows mapping with the half-pass chunk at
# 32 / 64 / 96 / 128 / 192 square: where should AUTO switch?
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
export XV_LIB_PATH=$PWD/scripts/devtools/_build/libxeno_mzhalfall.so
for rep in 1 2; do
  TAG=columns XV_MAZE_FILT=0 timeout 600 python scripts/devtools/probe_maze_res.py 32,64,96,128,192 2>/dev/null
  TAG=rows_half XV_MAZE_FILT=5 timeout 600 python scripts/devtools/probe_maze_res.py 32,64,96,128,192 2>/dev/null
done | tee $O/zz27_maze_res_mapping.txt
