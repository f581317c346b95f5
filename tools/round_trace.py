"""Print per-launch durations of the last sumcheck in a rocprofv3 --kernel-trace CSV (tools for DESIGN.md §4).
usage: python tools/round_trace.py <dir-with-*_kernel_trace.csv> [n_last]"""
import csv, glob, sys

d = sys.argv[1]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 40
f = sorted(glob.glob(d + "/**/*_kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
ks = sorted(((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Grid_Size_X"]),
              int(r["VGPR_Count"])) for r in rows), key=lambda x: x[1])
last = ks[-n_last:]
t0 = last[0][1]
for k in last:
    print(f"{k[0][:44]:44s} start {(k[1]-t0)/1e3:9.1f}us dur {(k[2]-k[1])/1e3:8.1f}us grid {k[3]:7d} vgpr {k[4]}")
