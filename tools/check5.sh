cd $GRAFT_REPO_ROOT
O=gpurun_out/c5; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-graph > $O/prof.log 2>&1
find $O/prof -name "*kernel_trace.csv" -delete
tail -1 $O/prof.log | cut -c1-200
