#!/bin/bash
# round 6, visit zz11: counters of the 256 x 256 ray cast, staged-per-wave write-out against the LDS chunk
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
export PMC_EXTRA="TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr|SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS"
XV_MAZE_STEPS=6 bash scripts/pmc_kernel.sh rows256_staged maze_raycast scripts/bench_families.py --families maze256 > $O/zz11_pmc_staged.log 2>&1; tail -1 $O/zz11_pmc_staged.log
XV_MAZE_STEPS=6 XV_LIB_PATH=$PWD/scripts/devtools/_build/libxeno_mznodirect.so bash scripts/pmc_kernel.sh rows256_chunk maze_raycast scripts/bench_families.py --families maze256 > $O/zz11_pmc_chunk.log 2>&1; tail -1 $O/zz11_pmc_chunk.log
python - <<'PY'
import json
for t in ("staged", "chunk"):
    d = json.load(open("gpurun_out/pmc_rows256_%s.json" % t))
    for k, v in d["kernels"].items():
        px = 256 * 256 * 16384
        cyc = v["GRBM_GUI_ACTIVE"] / 8
        print(t, k[:60])
        print("  cycles/XCD %.4g  VALU/pixel %.1f  busy %.3f  waves/SIMD %.2f  TCP acc/pixel %.2f  TCC req/pixel %.2f  TCC miss %.3g  FETCH %.4g WRITE %.4g LDS inst/pixel %.2f" % (
            cyc, v["SQ_INSTS_VALU"] * 64 / px, v["SQ_INSTS_VALU"] * 4 / 1024 / cyc, v["SQ_WAVE_CYCLES"] * 4 / 1024 / cyc,
            v["TCP_TOTAL_CACHE_ACCESSES_sum"] / px, v["TCP_TCC_READ_REQ_sum"] / px, v["TCC_MISS_sum"], v["FETCH_SIZE"], v["WRITE_SIZE"], v["SQ_INSTS_LDS"] * 64 / px))
        print("  ", {a: round(b) for a, b in v.items() if a in ("SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INST_CYCLES_VMEM", "SQ_ACTIVE_INST_LDS", "TCP_PENDING_STALL_CYCLES_sum", "TA_BUSY_avr", "SQ_WAIT_INST_ANY", "SQ_INSTS_VMEM_WR", "SQ_INSTS_VMEM_RD", "LDS_Block_Size", "VGPR_Count", "SQ_INSTS_SALU")})
PY
