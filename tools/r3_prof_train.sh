cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3train; mkdir -p $O
export NRGBD_TRAIN_CONV=${MODE:-native}
python bench.py --mode train --no-cpu-baseline --steps 3 --warmup 3 > /dev/null 2>&1    # MIOpen find-mode results cached on disk first
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --mode train --no-cpu-baseline --no-graph --steps 10 --warmup 3 > $O/prof.log 2>&1
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/train_kernel_stats.csv; rm -rf $O/prof
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r3train/train_kernel_stats.csv')))
it=13
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total ms/iter',tot/1e6/it)
for r in rows[:45]:
    print("  %-100s %6.1f/it %8.1f us %7.2f ms/it"%(r['Name'][:100], int(r['Calls'])/it, float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6/it))
PY
