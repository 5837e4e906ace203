# round-end verification on the GPU box: tests, smoke, default bench (with CPU baseline + parity), other configs, train line,
# kernel profiles of the bench (B, S, K), PMC traffic of the sampling kernel, PMC counters of the Winograd kernels
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6final}; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "amdgpu.ids" > $O/pytest_full.txt; grep -E "passed|failed|error" $O/pytest_full.txt | tail -3 > $O/pytest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
# the kernels smoke() launches (no miopen / Cijk row may appear): a kernel trace of the same call
( cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_smoke -- python -c "import __graft_entry__ as g; g.smoke()" > $O/prof_smoke.log 2>&1; cp $(find $O/prof_smoke -name "*kernel_stats.csv" | head -1) $O/smoke_kernel_stats.csv; rm -rf $O/prof_smoke )
python bench.py 2>$O/bench.err | tail -1 > $O/bench_B.json
for c in S K H; do python bench.py --config $c --no-other-configs 2>/dev/null | tail -1 > $O/bench_$c.json; done
python bench.py --mode train 2>/dev/null | tail -1 > $O/bench_train.json
python bench.py --mode train --accum 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_train_accum1.json
python bench.py --config H --steps 300 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_H_300frames.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in B S K; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$c -- python bench.py --config $c --steps 6 --warmup 2 --no-cpu-baseline --no-graph --no-other-configs > $O/prof_$c.log 2>&1
  cp $(find $O/prof_$c -name "*kernel_stats.csv" | head -1) $O/bench_${c}_kernel_stats.csv
  rm -rf $O/prof_$c
done
bash tools/r6_trace_frame.sh ${1:-r6final} > $O/trace_frame.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_T -- python bench.py --mode train --accum 1 --steps 6 --warmup 2 --no-cpu-baseline --no-graph > $O/prof_T.log 2>&1
cp $(find $O/prof_T -name "*kernel_stats.csv" | head -1) $O/bench_train_kernel_stats.csv; rm -rf $O/prof_T
if [ -z "$SKIP_PMC" ]; then   # SKIP_PMC=1: the committed PMC files stay valid while the sampling / Winograd kernel sources are unchanged
bash tools/pmc_traffic.sh ${1:-r6final}/traffic B S K H > /dev/null 2>&1
cp $O/traffic/costvol_traffic.json $O/costvol_traffic.json; rm -rf $O/traffic/*/fetch $O/traffic/*/write
bash tools/pmc_dw4.sh ${1:-r6final}/pmc_wino B > /dev/null 2>&1
cp $O/pmc_wino/summary.txt $O/pmc_wino_summary.txt
fi
cat $O/pytest.txt $O/smoke.txt | tail -8
cut -c1-420 $O/bench_B.json
for c in S K H train train_accum1 H_300frames; do cut -c95-230 $O/bench_$c.json; done
python -c "
import json; d=json.load(open('$O/bench_B.json')); print(json.dumps(d.get('parity'))[:900]); print(json.dumps(d.get('roofline'))[:600]); print(json.dumps(d.get('roofline_mfma'))[:500]); print(json.dumps(d.get('cpu_baseline'))[:300])"
[ -z "$SKIP_PMC" ] && cat $O/costvol_traffic.json | head -30
