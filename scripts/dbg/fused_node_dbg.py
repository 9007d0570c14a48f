"""Fused MLP+norm node vs separate nodes: per-parameter gradient differences (debug)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import phc_gnn_amd  # noqa
import torch
from phc_gnn_amd import harness, ops
DEV = "cuda:0"
wl = sys.argv[1] if len(sys.argv) > 1 else "molhiv_b32"


def run(fused, seed=5):
    model = harness.build_model(wl, DEV, seed=1)
    model.train()
    data = harness.to_device(harness.make_batch(wl, seed=seed), DEV)
    old = ops.MLP_NORM_FUSED, ops.MLP_NORM_FUSED_MIN_WORK
    ops.MLP_NORM_FUSED, ops.MLP_NORM_FUSED_MIN_WORK = fused, 0
    try:
        loss = harness.train_step(model, data, None, harness.StepConfig(optimizer=False, regression=bool(harness.WORKLOADS[wl].get("regression"))))
    finally:
        ops.MLP_NORM_FUSED, ops.MLP_NORM_FUSED_MIN_WORK = old
    torch.cuda.synchronize()
    return float(loss.detach()), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}


l0, g0 = run(False)
for trial in range(3):
    l1, g1 = run(True)
    errs = sorted(((float((g1[k] - g0[k]).abs().max() / g0[k].abs().max().clamp_min(1e-30)), k) for k in g0 if float(g0[k].abs().max()) > 1e-8), reverse=True)
    print(f"trial {trial}: loss {l1:.7f} vs {l0:.7f}; worst:", "; ".join(f"{e:.1e} {k}" for e, k in errs[:6]))

if wl == "molhiv_b32":
    from oracle import phc_oracle as O
    model = harness.build_model(wl, DEV, seed=1)
    gb = harness.make_batch(wl, seed=5)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    names = [k for k, _ in model.named_parameters()]
    p = {k: v.clone() for k, v in sd.items()}
    for k in names:
        p[k].requires_grad_()
    lt = O.skip_connect_add_forward(p, torch.from_numpy(gb.x), torch.from_numpy(gb.edge_index), torch.from_numpy(gb.edge_attr), torch.from_numpy(gb.batch),
                                    phm_dim=4, num_layers=4, num_atom_feats=9, num_bond_feats=3, num_dn_hidden=2, mlp=True, training=True, num_graphs=gb.num_graphs)
    lo = O.bce_loss_masked(lt, torch.from_numpy(gb.y))
    gr = dict(zip(names, torch.autograd.grad(lo, [p[k] for k in names])))
    for tag, g in (("separate", g0), ("fused", g1)):
        errs = sorted(((float((g[k].cpu() - gr[k]).abs().max() / gr[k].abs().max().clamp_min(1e-30)), k) for k in g if float(gr[k].abs().max()) > 1e-7), reverse=True)
        print(f"{tag} vs oracle: worst:", "; ".join(f"{e:.1e} {k}" for e, k in errs[:5]))
    print("N =", gb.num_nodes, "E =", gb.num_edges)
