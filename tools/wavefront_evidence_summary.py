#!/usr/bin/env python3
"""Summarise gpurun_out/<tag> written by tools/wavefront_evidence.sh into profiles/<name>.txt"""
import csv, glob, re, sys
base, outp = sys.argv[1], sys.argv[2]
lines = ["rocprofv3 --kernel-trace --stats and --pmc FETCH_SIZE / WRITE_SIZE (separate runs), tools/wavefront_evidence.sh, MI355X.",
         "Same scene, seed, samples and (bit-identical) accumulator; 256 spp at 1920x1080 = 530.8 M samples per run.",
         "wavefront = 32 Mi paths per batch; wavefrontbig = the whole render as one batch (512 Mi paths).", ""]
for cfg in ("c2", "c3"):
    for sch in ("regen", "wavefront", "wavefrontbig"):
        st = glob.glob(f"{base}/{cfg}_{sch}_stats/*/*kernel_stats.csv")
        if not st:
            continue
        tot = 0; rows = []
        for r in csv.DictReader(open(st[0])):
            if "pt::" in r["Name"]:
                ms = float(r["TotalDurationNs"]) / 1e6; tot += ms
                m = re.search(r"namespace\)::([A-Za-z_0-9]+(<[^>]*>)?)", r["Name"])
                rows.append((m.group(1) if m else r["Name"][:40], int(r["Calls"]), ms))
        tr = {}
        for kind in ("fetch", "write"):
            s = 0.0
            for f in glob.glob(f"{base}/{cfg}_{sch}_{kind}/*/*counter_collection.csv"):
                for r in csv.DictReader(open(f)):
                    if "pt::" in r["Kernel_Name"]:
                        s += float(r["Counter_Value"])
            tr[kind] = s / 1e6
        t = tr["fetch"] + tr["write"]
        lines.append(f"{cfg} {sch:12s}: kernels {tot:8.2f} ms = {530.8/tot:6.1f} Gsamples/s   FETCH_SIZE {tr['fetch']:6.2f} GB  WRITE_SIZE {tr['write']:6.2f} GB"
                     f"  -> {t*1e3/530.8:6.1f} B/sample, {t/tot:5.2f} TB/s")
        for n, c, ms in rows:
            lines.append(f"      {n:34s} calls {c:4d}  {ms:8.2f} ms")
open(outp, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
