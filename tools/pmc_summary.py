#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV passes for the path kernel:
tools/pmc_summary.py gpurun_out/pmc_dir [kernel-substr] [--json out.json --source "text" --bound hbm|valu --limiter "text"]
--json writes what bench.py's roofline object reads (profiles/r2_pmc_roofline.json): HBM-side bytes per launch
(FETCH_SIZE + WRITE_SIZE, KB -> bytes), lane utilisation, executed FP32 operations per launch."""
import csv
import glob
import sys

import json

args = [a for a in sys.argv[1:]]
json_out = source = None
if "--json" in args:
    i = args.index("--json")
    json_out = args[i + 1]
    del args[i:i + 2]
bound = None
if "--bound" in args:  # which limit the launch runs into, decided by whoever read the counters ("hbm", "valu", "l1")
    i = args.index("--bound")
    bound = args[i + 1]
    del args[i:i + 2]
limiter = None
if "--limiter" in args:  # one sentence: what the counters say the launch waits for
    i = args.index("--limiter")
    limiter = args[i + 1]
    del args[i:i + 2]
if "--source" in args:
    i = args.index("--source")
    source = args[i + 1]
    del args[i:i + 2]
d = args[0]
sub = args[1] if len(args) > 1 else "paths_kernel"
vals = {}
dur = {}
for f in glob.glob(d + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for f in glob.glob(d + "/*/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            dur.setdefault(f.split("/")[-3], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
            meta = (r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"], r["Grid_Size_X"])
v = {k: sum(x) / len(x) for k, x in vals.items()}
for k in sorted(v):
    print(f"{k:28s} {v[k]:.6g}")
print("kernel ms per pass:", {k: round(sum(x) / len(x), 3) for k, x in sorted(dur.items())})
print("VGPR,SGPR,LDS,scratch,grid:", meta)
g = v.get
if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
    print("lane utilisation (THREAD_CYCLES_VALU / (ACTIVE_INST_VALU*64)) = %.3f" % (v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_ACTIVE_INST_VALU"] * 64)))
if g("SQ_ACTIVE_INST_VALU") and g("SQ_BUSY_CYCLES"):
    print("SQ_ACTIVE_INST_VALU*4 / SQ_BUSY_CYCLES = %.3f ; WAVE_CYCLES/BUSY = %.2f" % (v["SQ_ACTIVE_INST_VALU"] * 4 / v["SQ_BUSY_CYCLES"], v.get("SQ_WAVE_CYCLES", 0) / v["SQ_BUSY_CYCLES"]))
if g("GRBM_GUI_ACTIVE") and "p1" in dur:
    print("effective clock GHz ~ %.2f" % (v["GRBM_GUI_ACTIVE"] / 8 / (sum(dur["p1"]) / len(dur["p1"]) * 1e-3) / 1e9))

lane_util = None
if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
    lane_util = v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_ACTIVE_INST_VALU"] * 64)
executed = None
if g("SQ_INSTS_VALU_ADD_F32") is not None and lane_util:
    wave_fp = v["SQ_INSTS_VALU_ADD_F32"] + v["SQ_INSTS_VALU_MUL_F32"] + v.get("SQ_INSTS_VALU_TRANS_F32", 0) + 2 * v["SQ_INSTS_VALU_FMA_F32"]
    executed = wave_fp * 64 * lane_util
    print("executed FP32 operations per launch (ADD + MUL + TRANS + 2 FMA wave-instr x 64 x lane utilisation) = %.4g" % executed)
if g("SQ_INSTS_VALU") and "p1" in dur and g("GRBM_GUI_ACTIVE"):
    clock = v["GRBM_GUI_ACTIVE"] / 8 / (sum(dur["p1"]) / len(dur["p1"]) * 1e-3)
    print("VALU wave-instr per SIMD %.4g ; kernel cycles %.4g ; cycles per VALU wave-instr per SIMD %.2f" %
          (v["SQ_INSTS_VALU"] / 1024, clock * sum(dur["p1"]) / len(dur["p1"]) * 1e-3, clock * sum(dur["p1"]) / len(dur["p1"]) * 1e-3 / (v["SQ_INSTS_VALU"] / 1024)))
if json_out:
    out = {"source": source or d, "kernel": sub, "kernel_ms_per_pass": {k: sum(x) / len(x) for k, x in sorted(dur.items())},
           "fetch_size_kb": g("FETCH_SIZE"), "write_size_kb": g("WRITE_SIZE"),
           "traffic_bytes_per_launch": (g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024 if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None else None,
           "l2_hit_rate": (g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))) if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None else None,
           "l2_miss_per_launch": g("TCC_MISS_sum"), "bound": bound or "valu", "limiter": limiter,
           "lane_utilisation": lane_util, "executed_flop_per_launch": executed,
           "valu_wave_instr_per_launch": g("SQ_INSTS_VALU"), "salu_wave_instr_per_launch": g("SQ_INSTS_SALU"),
           "note": "FETCH_SIZE / WRITE_SIZE count bytes at the L2's far side: HBM plus what the 256 MB Infinity Cache served. FETCH_SIZE is "
                   "dominated by 16-byte sky-texel gathers (random reads of one 64-byte sector each, to which the guide's x2 correction for "
                   "wide coalesced streams does not apply); WRITE_SIZE = block sums, exact for 16-byte stores. Counters of the path kernel only."}
    with open(json_out, "w") as f:
        json.dump(out, f, indent=1)
