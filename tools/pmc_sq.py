#!/usr/bin/env python3
"""Per-kernel means of the SQ counters of one or more rocprofv3 --pmc passes (csv output with --kernel-trace).

usage: python tools/pmc_sq.py <pmc_dir> [<pmc_dir> ...] > profiles/rNN_sq.json

Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_BUSY_CYCLES per
SE; SQ_VALU_MFMA_BUSY_CYCLES / COEXEC cycles summed over SIMDs; SQ_INSTS_* are wave-instructions.  Derived here for the GEMMs:
  valu_per_mfma   = SQ_INSTS_VALU / SQ_INSTS_MFMA          (issue slots the matrix pipe competes with)
  trans_per_mfma  = SQ_INSTS_VALU_TRANS_F32 / SQ_INSTS_MFMA
  coexec_frac     = SQ_VALU_MFMA_COEXEC_CYCLES / SQ_VALU_MFMA_BUSY_CYCLES   (MFMA time that also issued VALU)
  wait_frac       = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES      (share of wave time spent waiting on an instruction's data)
  lds_wait_frac   = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES"""
import collections, csv, glob, json, sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for root in sys.argv[1:]:
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in sorted(acc.items()):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    rec = {"launches": max(len(v) for v in cs.values()), "counters": {c: round(v, 1) for c, v in sorted(m.items())}}
    g = m.get
    if g("SQ_INSTS_MFMA"):
        rec["valu_per_mfma"] = round(g("SQ_INSTS_VALU", 0) / g("SQ_INSTS_MFMA"), 3)
        rec["trans_per_mfma"] = round(g("SQ_INSTS_VALU_TRANS_F32", 0) / g("SQ_INSTS_MFMA"), 3)
        rec["lds_per_mfma"] = round(g("SQ_INSTS_LDS", 0) / g("SQ_INSTS_MFMA"), 3)
    if g("SQ_VALU_MFMA_BUSY_CYCLES") and g("SQ_VALU_MFMA_COEXEC_CYCLES") is not None:
        rec["coexec_frac"] = round(g("SQ_VALU_MFMA_COEXEC_CYCLES") / g("SQ_VALU_MFMA_BUSY_CYCLES"), 4)
    if g("SQ_WAVE_CYCLES"):
        for name, c in (("wait_frac", "SQ_WAIT_INST_ANY"), ("lds_wait_frac", "SQ_WAIT_INST_LDS"), ("valu_active_frac", "SQ_ACTIVE_INST_VALU"),
                        ("lds_active_frac", "SQ_ACTIVE_INST_LDS"), ("vmem_active_frac", "SQ_ACTIVE_INST_VMEM")):
            if g(c) is not None:
                rec[name] = round(g(c) / g("SQ_WAVE_CYCLES"), 4)
    out[k] = rec
json.dump(out, sys.stdout, indent=1)
