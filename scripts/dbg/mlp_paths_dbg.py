"""fused MLP + norm node vs the separate nodes on one workload: loss of both, and the loss with the H-stationary kernel off."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import phc_gnn_amd
from phc_gnn_amd import harness, ops, lib as L
DEV = "cuda:0"
wl = sys.argv[1]
lib = L.load()
def run(fused, hs=True):
    lib.phc_gemm_hs_set_enabled(1 if hs else 0)
    model = harness.build_model(wl, DEV, seed=1).train()
    data = harness.to_device(harness.make_batch(wl, seed=5), DEV)
    old = ops.MLP_NORM_FUSED
    ops.MLP_NORM_FUSED = fused
    try:
        loss = harness.train_step(model, data, None, harness.StepConfig(optimizer=False, regression=bool(harness.WORKLOADS[wl].get("regression"))))
    finally:
        ops.MLP_NORM_FUSED = old
    torch.cuda.synchronize()
    lib.phc_gemm_hs_set_enabled(1)
    return float(loss)
print(wl, "fused/hs", run(True), "separate/hs", run(False), "fused/nohs", run(True, False), "separate/nohs", run(False, False))
