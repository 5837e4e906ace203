#!/bin/bash
# rocprofv3 counter passes for the feature-CNN conv kernels (run on the GPU box from the repo root).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-pmc_cnn}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/bench_cnn.py --config ${2:-B}"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc1 -o p -- $CMD > $OUT/pmc1.log 2>&1 || true
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -o p -- $CMD > $OUT/pmc3.log 2>&1 || true
rocprofv3 --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc4 -o p -- $CMD > $OUT/pmc4.log 2>&1 || true
ls $OUT/*/ | head
