#!/bin/bash
# rocprofv3 counter passes for the Winograd-domain K-Net kernels (wino_pc.hip, wino_dw.hip) on tools/bench_wino.py; run on the GPU box.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-pmc_wino}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/bench_wino.py --config ${2:-B} --iters 2 ${3:+--only $3}"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc1 -o p -- $CMD > $OUT/pmc1.log 2>&1 || true
rocprofv3 --pmc SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc2 -o p -- $CMD > $OUT/pmc2.log 2>&1 || true
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -o p -- $CMD > $OUT/pmc3.log 2>&1 || true
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc4 -o p -- $CMD > $OUT/pmc4.log 2>&1 || true
cd $R
for k in "conv_wino_dw_kernel<false, false, false, false, false>" "conv_wino_dw_kernel<false, false, false, true, false>" "conv_wino_dw_kernel<true, true, false"; do python tools/pmc_summary.py $OUT "$k"; done > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
