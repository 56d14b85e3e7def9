#!/usr/bin/env python3
"""MFMA utilisation and effective clock per kernel from one rocprofv3 --pmc pass
(SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE, csv output, with --kernel-trace so the dispatch durations are there).

usage: python tools/pmc_mfma.py <pmc_dir> > profiles/rNN_mfma_util.json

SQ_VALU_MFMA_BUSY_CYCLES sums, over every SIMD, the cycles its MFMA pipe was busy (32 per v_mfma_f32_32x32x16);
GRBM_GUI_ACTIVE counts shader-clock cycles of the dispatch (MI355X_MICROARCH.md: effective clock = GUI_ACTIVE / wall); rocprofv3
reports it summed over the chip's 8 XCDs (one GRBM each), so it is divided by 8 here.
  mfma_util       = MFMA_BUSY / (GUI_ACTIVE * 256 CUs * 4 SIMDs)   -- fraction of the MFMA issue capacity at the clock it ran
  effective_clock = GUI_ACTIVE / dispatch duration
  frac_of_nominal = mfma_util * effective_clock / 2.4 GHz           -- comparable with bench.py's roofline.frac"""
import csv, glob, json, sys, collections

root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        s, e = r.get("Start_Timestamp"), r.get("End_Timestamp")
        if s and e and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            dur[k].append(float(e) - float(s))
out = {}
for k, cs in sorted(acc.items()):
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in cs or "GRBM_GUI_ACTIVE" not in cs:
        continue
    busy = sum(cs["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(cs["SQ_VALU_MFMA_BUSY_CYCLES"])
    gui = sum(cs["GRBM_GUI_ACTIVE"]) / len(cs["GRBM_GUI_ACTIVE"]) / 8.0      # per XCD = shader cycles of the dispatch
    if busy <= 0 or gui <= 0:
        continue
    rec = {"launches": len(cs["GRBM_GUI_ACTIVE"]), "SQ_VALU_MFMA_BUSY_CYCLES_mean": busy, "GRBM_GUI_ACTIVE_per_xcd_mean": gui,
           "mfma_util": busy / (gui * 256 * 4)}
    if dur[k]:
        ns = sum(dur[k]) / len(dur[k])
        rec["duration_us_mean"] = ns / 1e3
        rec["effective_clock_GHz"] = gui / ns
        rec["frac_of_nominal_peak"] = rec["mfma_util"] * rec["effective_clock_GHz"] / 2.4
    out[k] = rec
json.dump(out, sys.stdout, indent=1)
