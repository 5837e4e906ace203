cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b7; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_rnet.py tests/test_gpu_cnn.py tests/test_gpu_dist.py -q -x -s 2>&1 | grep -v amdgpu.ids | tail -30 > $O/tests.txt
python bench.py --mode train --accum 1 --no-cpu-baseline 2>$O/train1.err | tail -1 > $O/bench_train_accum1.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_T -- python bench.py --mode train --accum 1 --steps 6 --warmup 2 --no-cpu-baseline --no-graph > $O/prof_T.log 2>&1
cp $(find $O/prof_T -name "*kernel_stats.csv" | head -1) $O/bench_train_kernel_stats.csv; rm -rf $O/prof_T
cat $O/tests.txt; cut -c1-330 $O/bench_train_accum1.json; echo; tail -2 $O/train1.err
python - <<'P'
import csv
rows=list(csv.DictReader(open('gpurun_out/r4b7/bench_train_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
nr=sum(float(r['TotalDurationNs']) for r in rows if 'nrgbd::' in r['Name'])
vend=[r for r in rows if any(k in r['Name'] for k in ('miopen','igemm','Cijk','naive_conv','MIOpen','gemm','Gemm'))]
print('total %.1f ms, nrgbd %.1f %%, vendor rows %d = %.2f %%' % (tot/1e6, 100*nr/tot, len(vend), 100*sum(float(r['TotalDurationNs']) for r in vend)/tot))
for r in vend: print('   ', r['Name'][:90], r['Calls'], r['Percentage'])
print('top non-nrgbd:')
for r in [r for r in rows if 'nrgbd::' not in r['Name']][:14]: print('   %-100s %5s %6s' % (r['Name'][:100], r['Calls'], r['Percentage']))
P
