cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3train; mkdir -p $O
for mode in default native; do
  if [ $mode = native ]; then export NRGBD_TRAIN_CONV=native; else unset NRGBD_TRAIN_CONV; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$mode -- python bench.py --mode train --steps 6 --warmup 3 > $O/prof_$mode.log 2>&1
  cp $(find $O/prof_$mode -name "*kernel_stats.csv" | head -1) $O/train_${mode}_kernel_stats.csv; rm -rf $O/prof_$mode
  tail -1 $O/prof_$mode.log | cut -c1-260
done
python - <<'PY'
import csv
for mode in ('default','native'):
    rows=list(csv.DictReader(open('gpurun_out/r3train/train_%s_kernel_stats.csv'%mode)))
    tot=sum(float(r['TotalDurationNs']) for r in rows); it=9
    print(mode,'total ms/iter',tot/1e6/it)
    for r in rows[:32]:
        print("  %-105s %5s %8.1f us %7.2f ms/it"%(r['Name'][:105], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6/it))
PY
