#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per-kernel mean of every collected counter."""
import csv, glob, json, os, sys
from collections import defaultdict
root = sys.argv[1]
out = {}
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    acc = defaultdict(lambda: defaultdict(list))
    with open(f) as fh:
        for row in csv.DictReader(fh):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in acc.items():
        if "gvi::" not in k:                      # the library's kernels (moments / chain / prep / epilogue), not torch's
            continue
        d = out.setdefault(k, {})
        for c, v in cs.items():
            d[c] = {"mean": sum(v) / len(v), "n": len(v)}
print(json.dumps(out, indent=1))
