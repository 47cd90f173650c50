#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: mean counter value per dispatch of the kernels of interest."""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "k_dygformer_fused"
acc = defaultdict(list)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if want in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k:36s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
