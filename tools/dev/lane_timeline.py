#!/usr/bin/env python3
"""per-stream timeline of the LAST chip-proof phase in a rocprofv3 kernel trace of tools/bench_shard.py: busy time, idle gaps, and what
ran around the longest gaps.  usage: lane_timeline.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
# the chip phase of the last run: from the last k_wit_infer burst back ... find the last k_leaf_hash_classes (commit) and the next k_gen_eq (batched main)
commits = [r for r in rows if "k_leaf_hash_classes" in r["Kernel_Name"]]
t0 = commits[-1]["e"]
t1 = min(r["s"] for r in rows if r["s"] > t0 and ("k_gen_eq" in r["Kernel_Name"] or "k_eq_base0" in r["Kernel_Name"]))
ph = [r for r in rows if r["s"] >= t0 and r["e"] <= t1]
first = min(r["s"] for r in ph if "k_wit_infer" in r["Kernel_Name"])
ph = [r for r in ph if r["s"] >= first]
print(f"chip phase {(t1 - first) / 1e3:.0f} us, {len(ph)} kernels")
by = collections.defaultdict(list)
for r in ph:
    by[(r["Queue_Id"], r["Stream_Id"])].append(r)
for k, v in sorted(by.items()):
    busy = sum(r["e"] - r["s"] for r in v)
    span = v[-1]["e"] - v[0]["s"]
    gaps = [(v[i + 1]["s"] - v[i]["e"], i) for i in range(len(v) - 1)]
    big = sorted(gaps, reverse=True)[:4]
    print(f"queue {k[0]} stream {k[1]}: {len(v)} kernels, span {span / 1e3:.0f} us from +{(v[0]['s'] - first) / 1e3:.0f}, busy {busy / 1e3:.0f} us, gaps total {(span - busy) / 1e3:.0f} us; "
          f"gap histogram <5us {sum(1 for g, _ in gaps if g < 5000)}, 5-15 {sum(1 for g, _ in gaps if 5000 <= g < 15000)}, 15-40 {sum(1 for g, _ in gaps if 15000 <= g < 40000)}, >40 {sum(1 for g, _ in gaps if g >= 40000)}")
    for g, i in big:
        print(f"     gap {g / 1e3:6.1f} us after {v[i]['Kernel_Name'][:40]:40s} before {v[i + 1]['Kernel_Name'][:40]}")
