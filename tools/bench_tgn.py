#!/usr/bin/env python3
"""Secondary benchmark (BASELINE config 5): TGN on the MOOC-shaped synthetic graph.  The workload lives in bench.py (`bench_tgn`,
also run as part of `python bench.py` -> secondary.tgn); this is its stand-alone command line.  One JSON line."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=100); ap.add_argument("--warmup", type=int, default=60); ap.add_argument("--cpu-seconds", type=float, default=10.0)
ap.add_argument("--two-calls", action="store_true", help="negative call then positive call per step, as the reference issues them")
a = ap.parse_args()
print(json.dumps(bench.bench_tgn("cuda:0", steps=a.steps, warmup=a.warmup, cpu_budget_s=a.cpu_seconds, two_calls=a.two_calls)))
