cd $GRAFT_REPO_ROOT
O=gpurun_out/c8; mkdir -p $O
python bench.py --mode train --steps 3 --warmup 2 > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --mode train --steps 6 --warmup 2 > $O/prof.log 2>&1
find $O/prof -name "*kernel_trace.csv" -delete
tail -1 $O/prof.log | cut -c1-200
