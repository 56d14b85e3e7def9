#!/usr/bin/env python3
"""Mean counter values per kernel from rocprofv3 --pmc csv output: python tools/pmc_any.py <dir> [kernel substring]"""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for k, cs in sorted(acc.items()):
    if sub in k:
        print(k[:90])
        for c, v in sorted(cs.items()):
            print(f"    {c:34s} {sum(v) / len(v):.4g}  (n={len(v)})")
