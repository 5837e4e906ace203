# round-end verification on the GPU box: tests, smoke, default bench (with CPU baseline), other configs, train line,
# kernel profile of the bench, PMC counters of the Winograd kernels
cd $GRAFT_REPO_ROOT
O=gpurun_out/final; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/pytest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
python bench.py 2>$O/bench.err | tail -1 > $O/bench_B.json
for c in S K H; do python bench.py --config $c 2>/dev/null | tail -1 > $O/bench_$c.json; done
python bench.py --mode train 2>/dev/null | tail -1 > $O/bench_train.json
python bench.py --config H --steps 300 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_H_300frames.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-graph > $O/prof.log 2>&1
find $O/prof -name "*kernel_trace.csv" -delete
bash tools/pmc_wino.sh final/pmc_wino B > /dev/null 2>&1
cat $O/pytest.txt $O/smoke.txt | tail -5
cut -c1-420 $O/bench_B.json
for c in S K H train; do cut -c95-200 $O/bench_$c.json; done
