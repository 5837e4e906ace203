#!/usr/bin/env python
"""Per-dispatch averages of rocprofv3 --pmc CSVs for one kernel.

    python tools/pmc_summary.py gpurun_out/<dir> <kernel-name-substring> [title]

Walks <dir>/**/ *counter_collection.csv, keeps the dispatches whose kernel name contains the substring and prints the
mean of every counter plus the kernel's mean duration (End - Start timestamps of the same rows, ns) and register counts.
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root, needle = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else needle
    vals, durs, regs = defaultdict(list), [], set()
    for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if needle not in row["Kernel_Name"]:
                    continue
                vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
                durs.append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
                regs.add((row["VGPR_Count"], row["SGPR_Count"], row["LDS_Block_Size"], row["Grid_Size"], row["Workgroup_Size"]))
    print(title)
    if not vals:
        print("  (no dispatch matched)")
        return
    print("  dispatches/counter %d   mean duration under counters %.1f us   (VGPR, SGPR, LDS, grid, wg) = %s" %
          (min(len(v) for v in vals.values()), sum(durs) / len(durs) / 1e3, sorted(regs)))
    for k in sorted(vals):
        print("  %-28s %16.0f" % (k, sum(vals[k]) / len(vals[k])))


if __name__ == "__main__":
    main()
