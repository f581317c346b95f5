#!/usr/bin/env python3
"""Summarise rocprofv3 output of `bench.py` into the files kept under profiles/.

usage: tools/pmc_summary.py <trace_dir> <pmc_fetch_dir> <pmc_write_dir> <sumchecks_in_pmc_run> <out_prefix>
 - kernel stats: copied verbatim (rocprofv3 --kernel-trace --stats)
 - traffic: FETCH_SIZE / WRITE_SIZE (KB) of the k_dense launches; FETCH_SIZE is doubled (gfx950: the counter
   reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md §HBM)
"""
import csv
import glob
import json
import shutil
import sys


def counter_sum(d, name):
    """(sum over the dense kernel's launches, their number, number of read-only round-0 launches = sumchecks, sum over the small
    rounds' persistent kernels).  Since round 3 a pipelined dense sumcheck hands its rounds of <= 2^15 pairs to k_mid / k_tail."""
    tot, n, n0, small = 0.0, 0, 0, 0.0
    for fn in glob.glob(d + "/*counter_collection.csv"):
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] != name:
                continue
            k = r["Kernel_Name"]
            if "k_dense" in k:
                tot += float(r["Counter_Value"])
                n += 1
                if "k_dense<3, 0" in k or "k_dense_pf<3, 0" in k:
                    n0 += 1
            elif "k_mid" in k or "k_tail" in k:
                small += float(r["Counter_Value"])
    return tot, n, n0, small


def main():
    trace, fdir, wdir, n_sc, out = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]
    shutil.copy(glob.glob(trace + "/*kernel_stats.csv")[0], out + "_kernel_stats.csv")
    f, nf, nf0, fs = counter_sum(fdir, "FETCH_SIZE")
    w, nw, nw0, ws = counter_sum(wdir, "WRITE_SIZE")
    if n_sc == 0:  # one read-only round-0 launch per sumcheck
        n_sc = nf0
        assert n_sc > 0 and nw0 == nf0 and nw == nf, (nf, nw, nf0, nw0)
    fetch_b = (f + fs) * 1024 * 2  # gfx950 correction
    write_b = (w + ws) * 1024
    res = {
        "kernel": "k_dense<3,*> (+ k_mid / k_tail for the rounds of <= 2^15 pairs: < 0.1 % of the bytes)",
        "sumchecks_profiled": n_sc,
        "launches": nf,
        "fetch_size_kb_raw": f,
        "write_size_kb_raw": w,
        "fetch_bytes_corrected": fetch_b,
        "write_bytes": write_b,
        "hbm_bytes_per_sumcheck": (fetch_b + write_b) / n_sc,
        "hbm_bytes_per_launch": (fetch_b + write_b) / max(nf, 1),
        "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (wide coalesced reads are tallied at half size on gfx950); "
                "WRITE_SIZE checked against k_fill_splitmix (exact)",
    }
    json.dump(res, open(out + "_pmc_traffic.json", "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
