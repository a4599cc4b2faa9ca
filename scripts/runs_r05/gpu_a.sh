#!/bin/bash
# round 5, call A: K interleaved chains — parity tests, the A/B at config 2a / 2b, HW-queue settings, one kernel trace
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_a
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_chains.py -x -q > $O/pytest_chains.txt 2>&1
echo "pytest chains rc=$?"; tail -5 $O/pytest_chains.txt
# A/B, config 2a (one task per env), default runtime settings
timeout 900 python scripts/devtools/probe_chains.py --tag 2a_default > $O/chains_2a_default.jsonl 2> $O/chains_2a_default.err
echo "2a default rc=$?"; cut -c1-400 $O/chains_2a_default.jsonl
# more hardware queues for the chains' streams / the graph's branches
GPU_MAX_HW_QUEUES=8 DEBUG_HIP_FORCE_GRAPH_QUEUES=8 timeout 900 python scripts/devtools/probe_chains.py --tag 2a_q8 --ks 1,4,8 > $O/chains_2a_q8.jsonl 2> $O/chains_2a_q8.err
echo "2a q8 rc=$?"; cut -c1-400 $O/chains_2a_q8.jsonl
# 2b (1,024 shared tasks)
timeout 600 python scripts/devtools/probe_chains.py --tag 2b_default --tasks 1024 --ks 1,2,4,8 --hows streams > $O/chains_2b.jsonl 2> $O/chains_2b.err
echo "2b rc=$?"; cut -c1-400 $O/chains_2b.jsonl
# kernel trace of the 4-chain run: do step kernels of different chains overlap?
rm -rf $O/trace
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o c4 -- python3 scripts/devtools/probe_chains.py --ks 4 --hows streams --steps 320 --repeats 1 --short 0 --tag trace > $O/trace_run.jsonl 2> $O/trace_run.err
echo "trace rc=$?"
F=$(ls $O/trace/*kernel_trace.csv $O/trace/*/*kernel_trace.csv 2>/dev/null | head -1)
echo "trace file: $F"; head -2 "$F"
python3 scripts/devtools/trace_overlap.py "$F" --skip 400 --out $O/trace_overlap_c4.json
# keep the trace small: the last 600 step-kernel rows
python3 - "$F" $O/trace_c4_tail.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
k = [r for r in rows if "step_kernel" in r.get("Kernel_Name", r.get("Name", ""))]
k.sort(key=lambda r: int(r["Start_Timestamp"]))
with open(sys.argv[2], "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(k[-600:])
PY
rm -rf $O/trace
