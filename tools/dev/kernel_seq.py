#!/usr/bin/env python3
"""Print the ordered kernel sequence (name, start offset us, duration us, gap to the previous kernel's end) of the LAST repetition in a
rocprofv3 kernel_trace.csv — used to count the launches and gaps of one chip proof."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:7.1f}  {r['Kernel_Name'][:60]}")
    prev_end = max(prev_end, e)
