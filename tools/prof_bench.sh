cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-graph > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1b -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-graph > gpurun_out/prof_r1b.log 2>&1
tail -1 gpurun_out/prof_r1b.log | cut -c1-200
find gpurun_out/prof_r1b -name "*kernel_stats.csv" | head
