#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc output (counter_collection csv) per kernel: mean of each counter over dispatches."""
import csv, glob, sys, collections
root = sys.argv[1]
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not any(s in k for s in ("gemm", "attention", "layernorm", "preproc")):
            continue
        rows[(k[:100], r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in rows.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} n={len(v):3d} mean={sum(v)/len(v):.5g}")
