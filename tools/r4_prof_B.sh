cd $GRAFT_REPO_ROOT
O=gpurun_out/r4prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_B -- python bench.py --config B --steps 8 --warmup 2 --no-cpu-baseline --no-graph --no-live-traffic > $O/prof_B.log 2>&1
cp $(find $O/prof_B -name "*kernel_stats.csv" | head -1) $O/bench_B_kernel_stats.csv; rm -rf $O/prof_B
python - <<'P'
import csv
rows=list(csv.DictReader(open('gpurun_out/r4prof/bench_B_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total ms", tot/1e6)
for r in rows[:40]: print('%-100s %5s %9.1f us avg %7.2f ms %6s%%' % (r['Name'][:100], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, r['Percentage']))
P
