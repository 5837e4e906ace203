#!/usr/bin/env python
"""Idle time between consecutive kernels in a rocprofv3 kernel trace (CSV): total, and grouped by the kernel that
FOLLOWS the gap.  Used to see what a frame loses between launches under hipGraph replay."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# keep the last 40 % of the trace (steady-state replayed frames)
rows = rows[int(len(rows) * 0.6):]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
gaps = defaultdict(lambda: [0, 0])
end = int(rows[0]["End_Timestamp"])
for r in rows[1:]:
    g = int(r["Start_Timestamp"]) - end
    if g > 0:
        k = gaps[r["Kernel_Name"][:70]]
        k[0] += g
        k[1] += 1
    end = max(end, int(r["End_Timestamp"]))
print("kernels %d  span %.2f ms  busy %.2f ms  idle %.2f ms (%.1f %%)  mean gap %.1f us"
      % (len(rows), span / 1e6, busy / 1e6, (span - busy) / 1e6, 100.0 * (span - busy) / span, (span - busy) / 1e3 / len(rows)))
for name, (t, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:30]:
    print("%-72s gaps %5d  total %8.3f ms  mean %6.1f us" % (name, n, t / 1e6, t / 1e3 / n))
