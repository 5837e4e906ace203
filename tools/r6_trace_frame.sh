# per-dispatch kernel trace of config-B frames (eager launches): one line per launch, in order -> gpurun_out/$1/frame_B_dispatches.txt
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6prof}; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/tr_B -- python bench.py --config ${2:-B} --steps 3 --warmup 2 --no-cpu-baseline --no-graph --no-other-configs --no-live-traffic > $O/tr_B.log 2>&1
python - "$O" <<'PY'
import csv, glob, sys
O = sys.argv[1]
f = glob.glob(O + "/tr_B/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "dpv_resample" in n]
starts = [i for i, n in enumerate(names) if "space_to_depth2_rgb" in n]
e = idx[-1]
s = max(i for i in starts if i < e)
tot = {}
with open(O + "/frame_B_dispatches.txt", "w") as out:
    t0 = int(rows[s]["Start_Timestamp"])
    for r in rows[s:e + 1]:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        out.write("%9.1f %8.1f  grid %-8s wg %-5s %s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e3, d,
                  r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")), r["Kernel_Name"][:150]))
        k = r["Kernel_Name"].split("(")[0][:70]
        tot[k] = tot.get(k, [0, 0.0]); tot[k][0] += 1; tot[k][1] += d
    span = (int(rows[e]["End_Timestamp"]) - t0) / 1e3
    busy = sum(v[1] for v in tot.values())
    out.write("# frame span %.1f us, kernel time %.1f us, launches %d\n" % (span, busy, e + 1 - s))
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        out.write("# %8.1f us %4d x  %s\n" % (v[1], v[0], k))
print(open(O + "/frame_B_dispatches.txt").read()[-2500:])
PY
rm -rf $O/tr_B
