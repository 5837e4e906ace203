#!/bin/bash
# copy the judged summaries of a final_check run (gpurun_out/<tag>/) into profiles/r6_*
T=${1:-r6final}; S=gpurun_out/$T; P=profiles
for c in B S K H H_300frames train train_accum1; do cp $S/bench_$c.json $P/r6_bench_$c.json; done
for c in B S K train; do cp $S/bench_${c}_kernel_stats.csv $P/r6_bench_${c}_kernel_stats.csv; done
cp $S/costvol_traffic.json $P/r6_costvol_traffic.json
cp $S/frame_B_dispatches.txt $P/r6_frame_B_dispatches.txt
cp $S/pmc_wino_summary.txt $P/r6_pmc_wino.txt
cp $S/pytest.txt $P/r6_pytest_gpu.txt
cp $S/smoke_kernel_stats.csv $P/r6_smoke_kernel_stats.csv
python tools/summary.py r6 > /dev/null
