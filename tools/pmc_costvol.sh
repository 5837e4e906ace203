#!/bin/bash
# rocprofv3 passes for the sampling kernels (run on the GPU box from the repo root).
# Counters are collected in their own runs (never together with sys/hip traces).
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-pmc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/bench_costvol.py --config ${2:-B} --iters 3 --only costvol ${3:+--gen $3}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1 || true
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY --output-format csv -d $OUT/pmc1 -o p -- $CMD > $OUT/pmc1.log 2>&1 || true
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc2 -o p -- $CMD > $OUT/pmc2.log 2>&1 || true
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -o p -- $CMD > $OUT/pmc3.log 2>&1 || true
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc4 -o p -- $CMD > $OUT/pmc4.log 2>&1 || true
find $OUT -name "*.csv" | head -20
