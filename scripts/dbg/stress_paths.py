"""Repeat one training step of a workload from the same state many times (both MLP paths) and report every loss that differs from the
first one: a race shows up as an occasional outlier.   python scripts/dbg/stress_paths.py <workload> [repeats]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import phc_gnn_amd
from phc_gnn_amd import harness, ops
DEV = "cuda:0"
wl, reps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40
cnt = collections.Counter()
for i in range(reps):
    for fused in (True, False):
        model = harness.build_model(wl, DEV, seed=1).train()
        data = harness.to_device(harness.make_batch(wl, seed=5), DEV)
        ops.MLP_NORM_FUSED = fused
        loss = harness.train_step(model, data, None, harness.StepConfig(optimizer=False, regression=bool(harness.WORKLOADS[wl].get("regression"))))
        cnt[(fused, round(float(loss), 6))] += 1
print(wl, dict(cnt))
