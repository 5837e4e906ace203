# kernel-level profile of the default bench (eager launches so that every kernel is its own record), configs B (+ S, K quick)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3prof; mkdir -p $O
for c in ${CFGS:-B}; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$c -- python bench.py --config $c --steps 6 --warmup 2 --no-cpu-baseline --no-graph > $O/prof_$c.log 2>&1
  f=$(find $O/prof_$c -name "*kernel_stats.csv" | head -1); cp $f $O/bench_${c}_kernel_stats.csv
  find $O/prof_$c -name "*kernel_trace.csv" -delete
  tail -1 $O/prof_$c.log | cut -c1-250
done
python - <<'PY'
import csv,glob
for f in sorted(glob.glob('gpurun_out/r3prof/bench_*_kernel_stats.csv')):
    rows=list(csv.DictReader(open(f)))
    tot=sum(float(r['TotalDurationNs']) for r in rows)
    print(f, 'total ms', tot/1e6)
    for r in rows[:28]:
        print("  %-100s %5s %9.1f us %8.2f ms %5.1f%%"%(r['Name'][:100], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, float(r['TotalDurationNs'])/tot*100))
    non=sum(float(r['TotalDurationNs']) for r in rows if 'nrgbd' not in r['Name'])
    print("  non-nrgbd total ms", non/1e6)
PY
