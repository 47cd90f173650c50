#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: mean counter value per dispatch of one kernel, grouped by grid size (a bench run launches the fused
kernel in several shapes: the timed launches are the most frequent large grid).  usage: pmc_summary.py DIR [kernel substring] [--json OUT]"""
import collections, csv, glob, json, os, sys
root = sys.argv[1]
args = [a for a in sys.argv[2:] if not a.startswith("--")]
want = args[0] if args else "k_dygformer_fused"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if want in r.get("Kernel_Name", ""):
            acc[int(r["Grid_Size"]) // int(r["Workgroup_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for wgs in sorted(acc):
    print(f"-- {want}: grid of {wgs} workgroups")
    for k in sorted(acc[wgs]):
        v = acc[wgs][k]
        print(f"{k:36s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
if "--json" in sys.argv:
    out = sys.argv[sys.argv.index("--json") + 1]
    json.dump({str(w): {k: sum(v) / len(v) for k, v in c.items()} for w, c in acc.items()}, open(out, "w"), indent=1)
