# per-dispatch kernel trace of config-B frames (eager launches): one line per launch, in order
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5prof; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/tr_B -- python bench.py --config B --steps 3 --warmup 2 --no-cpu-baseline --no-graph --no-other-configs > $O/tr_B.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r5prof/tr_B/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last complete frame before the kernel timers: find costvol launches; frame = from pack_nhwc-ish start (space_to_depth2_rgb) to dpv_resample
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "dpv_resample" in n]
starts = [i for i, n in enumerate(names) if "space_to_depth2_rgb" in n]
e = idx[-1]
s = max(i for i in starts if i < e)
with open("gpurun_out/r5prof/frame_B_dispatches.txt", "w") as out:
    t0 = int(rows[s]["Start_Timestamp"])
    for r in rows[s:e + 1]:
        out.write("%9.1f %8.1f  grid %-8s wg %-5s %s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                                   r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")), r["Kernel_Name"][:150]))
print(open("gpurun_out/r5prof/frame_B_dispatches.txt").read()[:200])
PY
rm -rf $O/tr_B
