#!/bin/bash
# round-5 evidence for the dense kernel (verdict item 5): do the VALU and the memory pipe of the fold rounds overlap as far as the hardware
# lets them?  Per k_dense instantiation: SQ occupancy / issue / wait counters and the TCP (vector L1) stall counters, for the shipped kernel and
# for the LDS-prefetched variant (CENO_HIP_DENSE_LDS=1: the next iteration's blocks in flight through LDS-DMA while the current one is
# multiplied), plus wall times of both on the same box.  Separate --pmc passes, no trace domains.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r05d; mkdir -p $o; rm -rf $o/*
for v in 0 1 0 1; do
  CENO_HIP_DENSE_LDS=$v python3 bench.py --no-extra --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.read()); print(json.dumps({'CENO_HIP_DENSE_LDS': $v, 'ms_per_step': round(r['ms_per_step'],4), 'ms_per_step_stub': round(r['ms_per_step_stub'],4), 'kernel_ms_per_sumcheck': round(r['roofline']['kernel_ms_per_sumcheck'],4)}))" >> $o/wall.jsonl
done
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
P2="TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"
P3="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"
for v in 0 1; do
  CENO_HIP_DENSE_LDS=$v timeout 300 rocprofv3 --pmc $P1 --output-format csv -d $o/p1_$v -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $o/p1_$v.log 2>&1
  CENO_HIP_DENSE_LDS=$v timeout 300 rocprofv3 --pmc $P2 --output-format csv -d $o/p2_$v -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $o/p2_$v.log 2>&1
  CENO_HIP_DENSE_LDS=$v timeout 300 rocprofv3 --pmc $P3 --output-format csv -d $o/p3_$v -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $o/p3_$v.log 2>&1
done
python3 - $o <<'PY'
import csv, sys, json, glob, collections
o = sys.argv[1]
def per_kernel(d):
    fs = sorted(glob.glob(o + "/" + d + "/**/*counter_collection.csv", recursive=True))
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    if not fs: return {}
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0]
        if "k_dense" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        acc[k]["_rows_" + r["Counter_Name"]] += 1
    return {k: {c: x for c, x in v.items() if not c.startswith("_rows_")} | {"launches": int(max(x for c, x in v.items() if c.startswith("_rows_")))} for k, v in acc.items()}
res = {"question": "round-4 verdict item 5: the dense kernel pays VALU (1.34 ms) plus most of memory (1.67 ms) in its fold rounds instead of their maximum; "
                   "try a producer / consumer split through LDS-DMA, or show with counters that the two pipes overlap as far as the hardware lets them",
       "wall_same_box": [json.loads(l) for l in open(o + "/wall.jsonl")], "counters": {}}
for v, name in ((0, "shipped k_dense<3,2>"), (1, "k_dense_lds<3> (CENO_HIP_DENSE_LDS=1)")):
    m = {}
    for p in ("p1", "p2", "p3"):
        for k, c in per_kernel(f"{p}_{v}").items():
            m.setdefault(k, {}).update(c)
    for k, c in m.items():
        wc = c.get("SQ_WAVE_CYCLES")
        if wc:
            c["valu_active_frac_of_wave_cycles"] = round(c.get("SQ_ACTIVE_INST_VALU", 0) / wc, 4)
            c["wait_inst_any_frac_of_wave_cycles"] = round(c.get("SQ_WAIT_INST_ANY", 0) / wc, 4)
            c["wait_any_frac_of_wave_cycles"] = round(c.get("SQ_WAIT_ANY", 0) / wc, 4)
            c["waves_per_simd_avg"] = round(wc * 4 / (1024.0 * c["SQ_BUSY_CYCLES"] / 8 * 4), 3) if c.get("SQ_BUSY_CYCLES") else None
    res["counters"][name] = m
json.dump(res, open(o + "/r05_dense_overlap.json", "w"), indent=1)
for name, m in res["counters"].items():
    for k, c in m.items():
        print(name, "|", k, {x: c.get(x) for x in ("launches", "valu_active_frac_of_wave_cycles", "wait_inst_any_frac_of_wave_cycles", "wait_any_frac_of_wave_cycles", "TCP_PENDING_STALL_CYCLES", "SQ_BUSY_CYCLES")})
print(res["wall_same_box"])
PY
