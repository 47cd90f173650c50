#!/usr/bin/env python3
"""Training-step throughput of the DyGFormer path.  The workload lives in bench.py (`bench_train`, also run as part of
`python bench.py` -> secondary.train); this is its stand-alone command line.  One JSON line."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20); ap.add_argument("--warmup", type=int, default=8); ap.add_argument("--cpu-seconds", type=float, default=12.0)
ap.add_argument("--separate-calls", action="store_true", help="the reference's two calls per step instead of one pass over both")
a = ap.parse_args()
print(json.dumps(bench.bench_train("cuda:0", steps=a.steps, warmup=a.warmup, separate_calls=a.separate_calls, cpu_budget_s=a.cpu_seconds)))
