"""Repeat one training step of a workload from the same state many times (both MLP paths) and report every loss that differs from the
first one: a race shows up as an occasional outlier.  Between the steps the caching allocator's free blocks are filled with a large
finite value (``pollute``), so that a kernel that reads memory nobody wrote in THIS step gives a different loss every time instead of
once in a few hundred runs.   python scripts/dbg/stress_paths.py <workload> [repeats] [pollute 0/1]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import phc_gnn_amd
from phc_gnn_amd import harness, ops
DEV = "cuda:0"
wl, reps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40
pollute = len(sys.argv) > 3 and sys.argv[3] == "1"
cnt = collections.Counter()


def dirty(seed):
    """Allocate and free blocks of many sizes filled with 1e3 + noise: what torch.empty hands out next is not zero."""
    g = torch.Generator(device=DEV).manual_seed(seed)
    keep = []
    for sz in (1 << 10, 1 << 12, 1 << 14, 1 << 16, 1 << 18, 1 << 20, 1 << 22, 3 << 20, 5 << 18, 7 << 14, 9 << 10):
        for _ in range(3):
            t = torch.empty(sz, device=DEV)
            t.uniform_(900.0, 1100.0, generator=g)
            keep.append(t)
    del keep


for i in range(reps):
    for fused in (True, False):
        model = harness.build_model(wl, DEV, seed=1).train()
        data = harness.to_device(harness.make_batch(wl, seed=5), DEV)
        if pollute:
            dirty(i)
        ops.MLP_NORM_FUSED = fused
        loss = harness.train_step(model, data, None, harness.StepConfig(optimizer=False, regression=bool(harness.WORKLOADS[wl].get("regression"))))
        cnt[(fused, round(float(loss), 6))] += 1
        del model, data
print(wl, "pollute" if pollute else "clean", dict(cnt))
