# kernel-trace of the hipGraph-replayed bench: per-kernel start/end timestamps -> idle gaps (tools/gaps.py)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python bench.py --steps 2 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_gaps -- python bench.py --steps 4 --warmup 3 --no-cpu-baseline > gpurun_out/prof_gaps.log 2>&1
python tools/gaps.py $(find gpurun_out/prof_gaps -name "*kernel_trace.csv" | head -1) > gpurun_out/gaps.txt 2>&1
find gpurun_out/prof_gaps -name "*kernel_trace.csv" -delete
tail -40 gpurun_out/gaps.txt
