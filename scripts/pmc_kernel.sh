#!/bin/bash
# PMC counters of one kernel of any python command: one rocprofv3 pass per counter group (TCC slots do not hold
# FETCH_SIZE and WRITE_SIZE together; SQ has 8 slots), kernel-trace only beside the counters.
#   scripts/pmc_kernel.sh <tag> <kernel name substring> <python script + args ...>
# -> gpurun_out/pmc_<tag>.json  (averages per dispatch of the kernels whose name contains the substring)
set -u
export TMPDIR=/tmp
TAG=$1; KSUB=$2; shift 2
mkdir -p gpurun_out
PMCG=("FETCH_SIZE" "WRITE_SIZE"
        "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES"
        "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA"
        "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE")
# PMC_EXTRA="group one|group two": further passes (e.g. the texture path: TA_* / TCP_* counters)
if [ -n "${PMC_EXTRA:-}" ]; then IFS='|' read -r -a EXTRA <<< "$PMC_EXTRA"; PMCG+=("${EXTRA[@]}"); fi
i=0
for g in "${PMCG[@]}"; do
  d=gpurun_out/pmc_${TAG}_g$i
  rm -rf $d
  timeout 600 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $d -o pmc -- python3 "$@" > $d.out 2> $d.err
  echo "group $i [$g] rc=$?"
  i=$((i+1))
done
python3 scripts/pmc_kernel_to_json.py "$TAG" "$KSUB" "python3 $*"
