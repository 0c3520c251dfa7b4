#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV passes for the render kernel: tools/pmc_summary.py gpurun_out/pmc_dir [kernel-substr]"""
import csv
import glob
import sys

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "render_kernel"
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
