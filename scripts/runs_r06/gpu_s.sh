#!/bin/bash
# round 6, visit s: do the HIP runtime's graph knobs move the 20-step burst (the driver's flags)?
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06_s
mkdir -p $O
run() {
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-families --no-live-pmc --long-steps 0 --sustain-seconds 0 --no-variants 2>/dev/null | python3 -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); print('$1', 'value %.4g wall us/step %.3f events us/step %.3f' % (d['value'], d['ms_per_step'] * 1e3, d['roofline']['avg_launch_us']))
"
}
for rep in 1 2; do
  run default
  DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 run packet_capture_0
  DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 run packet_capture_1
  DEBUG_HIP_GRAPH_BATCH_SIZE=1 run batch_1
  DEBUG_HIP_GRAPH_BATCH_SIZE=8 run batch_8
  DEBUG_HIP_GRAPH_BATCH_SIZE=64 run batch_64
  DEBUG_HIP_FORCE_GRAPH_QUEUES=1 run force_queues_1
  GPU_MAX_HW_QUEUES=8 run hwq_8
  HIP_FORCE_DEV_KERNARG=1 run dev_kernarg_1
  HIP_FORCE_DEV_KERNARG=0 run dev_kernarg_0
done | tee $O/graph_knobs.txt
