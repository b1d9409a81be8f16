#!/usr/bin/env python3
"""Print the kernel timeline (start offset, duration, name) of the last N kernels of a rocprofv3 kernel_trace.csv."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s / 1000:9.1f} {e / 1000:9.1f} {(e - s) / 1000:8.1f} us  q={r.get('Queue_Id', '?'):>3} {r['Kernel_Name'][:70]}")
