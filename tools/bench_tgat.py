#!/usr/bin/env python3
"""Secondary benchmark (BASELINE config 3): TGAT on the Reddit-shaped synthetic graph.  The workload lives in bench.py
(`bench_tgat`, also run as part of `python bench.py` -> secondary.tgat); this is its stand-alone command line.  One JSON line."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=192); ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--edges", type=int, default=672447); ap.add_argument("--cpu-seconds", type=float, default=20.0)
ap.add_argument("--fuse-steps", type=int, default=32)
ap.add_argument("--plain", action="store_true", help="the fused `recent` calls only (profiling): no one-step calls beyond the first, no `uniform` leg, no CPU leg")
a = ap.parse_args()
kw = dict(one_step_calls=1, uniform_steps=0, cpu_budget_s=0.0, larger_calls=()) if a.plain else dict(cpu_budget_s=a.cpu_seconds)
print(json.dumps(bench.bench_tgat("cuda:0", steps=a.steps, warmup=a.warmup, fuse_steps=a.fuse_steps, edges=a.edges, **kw)))
