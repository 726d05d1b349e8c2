#!/usr/bin/env python3
"""Where the GPU idles inside one bench step: reads a rocprofv3 --kernel-trace CSV (argv[1]) of `tools/phase_steps.py 3`, takes
the LAST step (from the last subsample_conv / prefix_rows launch on) and lists every gap between consecutive kernels above 30 us with
its neighbours, and the sum of busy / idle time."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
last = max(i for i, n in enumerate(names) if "prefix_rows" in n)
rows = rows[last:]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print(f"last step: {len(rows)} kernels, span {span / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms, idle {(span - busy) / 1e6:.2f} ms")
t0 = int(rows[0]["Start_Timestamp"])
for a, b in zip(rows, rows[1:]):
    gap = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    if gap > 30000:
        print(f"  at {(int(a['End_Timestamp']) - t0) / 1e6:8.2f} ms: {gap / 1e3:8.1f} us idle between {a['Kernel_Name'][:50]}  ->  {b['Kernel_Name'][:50]}")
