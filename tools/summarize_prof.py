#!/usr/bin/env python3
"""Condense rocprofv3 output under gpurun_out/ into the tracked summaries under profiles/.

usage: summarize_prof.py TAG KT_DIR [FETCH_DIR WRITE_DIR [SQ_DIR]]
  KT_DIR     rocprofv3 --kernel-trace --stats --output-format csv
  FETCH_DIR  rocprofv3 --pmc FETCH_SIZE  (own pass)      WRITE_DIR  rocprofv3 --pmc WRITE_SIZE (own pass)
  SQ_DIR     rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE (own pass)
Writes profiles/TAG_kernel_stats.csv, and profiles/pmc_traffic.json (read by bench.py's roofline.traffic).
HBM bytes per launch = FETCH_SIZE*1024*2 + WRITE_SIZE*1024: FETCH_SIZE/WRITE_SIZE are in KiB, and gfx950's
FETCH_SIZE counts half the bytes of a streamed read (MI355X_MICROARCH.md, section HBM).
"""
import csv, glob, json, os, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, kt = sys.argv[1], sys.argv[2]
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)

def short(name):
    n = name.split("(")[0]
    if n.startswith("void "): n = n[5:]
    return n.split("<")[0][:70]                       # template arguments dropped: one entry per kernel

rows = []
for f in glob.glob(os.path.join(kt, "**", "*_kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((short(r["Name"]), int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"]), float(r["Percentage"]),
                     float(r["MinNs"]), float(r["MaxNs"])))
rows.sort(key=lambda r: -r[2])
with open(os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv"), "w") as o:
    o.write("kernel,calls,total_ns,avg_ns,percent,min_ns,max_ns\n")
    for r in rows:
        o.write("%s,%d,%.0f,%.1f,%.2f,%.0f,%.0f\n" % r)
print("wrote", tag + "_kernel_stats.csv", len(rows), "kernels")

if len(sys.argv) >= 5:
    def counter(d, name):
        acc = defaultdict(lambda: [0.0, 0])
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] != name: continue
                k = short(r["Kernel_Name"])
                acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
        return acc
    fe, wr = counter(sys.argv[3], "FETCH_SIZE"), counter(sys.argv[4], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fe) | set(wr)):
        if not k.startswith("k_"): continue
        f = fe[k][0] / max(fe[k][1], 1); w = wr[k][0] / max(wr[k][1], 1)
        out[k] = {"fetch_size_kib_per_launch": round(f, 1), "write_size_kib_per_launch": round(w, 1),
                  "hbm_bytes_per_launch": int(f * 1024 * 2 + w * 1024), "launches_sampled": fe[k][1],
                  "note": "FETCH_SIZE x2 (gfx950 correction for streamed reads; uncalibrated for byte-wide loads) + WRITE_SIZE"}
    if len(sys.argv) >= 6:
        # vector-ALU issue occupancy: measured cost of a wave64 instruction on a saturated SIMD (tools/valu_calib.hip,
        # profiles/r02_valu_issue_calibration.txt): 2.3 cycles for plain ALU ops, 4.2 for packed / 3-operand / SDWA / dot / perm
        # ones; the chip has 256 CUs x 4 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs.  Bounds = all-cheap .. all-expensive.
        names = ["SQ_INSTS_VALU", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "GRBM_GUI_ACTIVE"]
        sq = {nm: counter(sys.argv[5], nm) for nm in names}
        for k in sorted(set(sq["SQ_INSTS_VALU"])):
            if not k.startswith("k_"): continue
            m = {nm: sq[nm][k][0] / max(sq[nm][k][1], 1) for nm in names}
            cyc = m["GRBM_GUI_ACTIVE"] / 8.0
            e = out.setdefault(k, {})
            e.update({"valu_insts_per_launch": int(m["SQ_INSTS_VALU"]), "waves_per_launch": int(m["SQ_WAVES"]),
                      "busy_cycles_per_launch": int(cyc),
                      "valu_issue_frac_bounds": [round(m["SQ_INSTS_VALU"] * 2.3 / (1024.0 * cyc), 4), round(m["SQ_INSTS_VALU"] * 4.2 / (1024.0 * cyc), 4)] if cyc else None,
                      "wait_any_frac_of_wave_cycles": round(m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], 4) if m["SQ_WAVE_CYCLES"] else None})
    json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
    for k, v in out.items(): print(k, v.get("hbm_bytes_per_launch"), v.get("valu_issue_frac_bounds"))
