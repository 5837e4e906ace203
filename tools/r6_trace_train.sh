# per-dispatch kernel trace of one training window (eager), grouped by kernel and grid size
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6prof; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/tr_T -- python bench.py --mode train --accum 1 --steps 3 --warmup 2 --no-cpu-baseline --no-graph > $O/tr_T.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r6prof/tr_T/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last window: between the last two adam_kernel bursts
idx = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
# group adam launches into steps (bursts)
bursts = []
for i in idx:
    if not bursts or i - bursts[-1][-1] > 50: bursts.append([i])
    else: bursts[-1].append(i)
s, e = bursts[-2][-1] + 1, bursts[-1][-1] + 1
agg = collections.OrderedDict()
t0 = int(rows[s]["Start_Timestamp"]); t1 = int(rows[e - 1]["End_Timestamp"])
for r in rows[s:e]:
    key = (r["Kernel_Name"][:70], r.get("Grid_Size_X", ""), r.get("Grid_Size_Y", ""))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += d
tot = sum(v[1] for v in agg.values())
with open("gpurun_out/r6prof/train_window_kernels.txt", "w") as out:
    out.write("one training window (eager): %d launches, kernels %.2f ms, span %.2f ms\n" % (e - s, tot / 1e3, (t1 - t0) / 1e6))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
        out.write("%7.1f us total %4d x %7.1f us  grid %8s x %-4s %s\n" % (v[1], v[0], v[1] / v[0], k[1], k[2], k[0]))
print(open("gpurun_out/r6prof/train_window_kernels.txt").read()[:6000])
PY
rm -rf $O/tr_T
