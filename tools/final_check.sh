# round-end verification on the GPU box: tests, smoke, default bench (with CPU baseline), kernel profile
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/final/pytest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final/smoke.txt 2>&1
python bench.py 2>gpurun_out/final/bench.err | tail -1 > gpurun_out/final/bench_B.json
for c in S K H; do python bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/final/bench_$c.json; done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/prof -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-graph > gpurun_out/final/prof.log 2>&1
find gpurun_out/final/prof -name "*kernel_trace.csv" -delete
cat gpurun_out/final/pytest.txt gpurun_out/final/smoke.txt | tail -5
cut -c1-420 gpurun_out/final/bench_B.json
for c in S K H; do cut -c95-200 gpurun_out/final/bench_$c.json; done
