import numpy as np, torch
from phc_gnn_amd import harness, ops, lib
DEV = torch.device("cuda:0")
for wl in ("molhiv_b32",):
  for hl in (True, False):
    ops.HEAD_LOSS = hl
    model = harness.build_model(wl, DEV, seed=3).train()
    gb = harness.make_batch(wl, seed=11)
    gb.y[::7] = np.nan
    data = harness.to_device(gb, DEV)
    keep = {}
    orig = ops.masked_bce_with_logits
    def wrapped(logits, y):
        keep["logits"] = logits.detach().clone(); keep["y"] = y.clone()
        out = orig(logits, y)
        keep["loss0"] = float(out.detach())
        return out
    harness.masked_bce_with_logits = wrapped
    cfg = harness.StepConfig(optimizer=False, weight_decay=1e-3)
    loss = harness.train_step(model, data, None, cfg)
    torch.cuda.synchronize()
    z, y = keep["logits"], keep["y"]
    m = ~torch.isnan(y)
    ref = torch.nn.functional.binary_cross_entropy_with_logits(z[m], y[m])
    from phc.hypercomplex.regularization import phm_weight_regularization
    print("reg (torch)", float(phm_weight_regularization(model, p=2)) * 1e-6)
    print(wl, "head_loss", hl, "loss", float(loss.detach()), "at return", keep["loss0"], "torch on same logits", float(ref), "logits sum", float(z.sum()))
    harness.masked_bce_with_logits = orig
