# first GPU call of a session: tests, default bench, kernel profile of the bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/c1; mkdir -p $O
timeout 1000 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > $O/pytest.txt
python bench.py 2>$O/bench.err | tail -1 > $O/bench_B.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-graph > $O/prof.log 2>&1
find $O/prof -name "*kernel_trace.csv" -delete
cat $O/pytest.txt | tail -4
cut -c1-900 $O/bench_B.json
