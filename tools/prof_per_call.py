#!/usr/bin/env python3
"""The drop-in caller's rate alone (bench.py: stages.per_call), for rocprofv3: tools/prof_generic.sh percall tools/prof_per_call.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
wk = bench.DygformerWorkload("wikipedia", torch.device("cuda", 0))
bench._prime_gpu("cuda:0")
print(json.dumps(bench.per_call_stage(wk, n_batches=int(os.environ.get("NB", "60")))))
