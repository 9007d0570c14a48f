import sys; sys.path.insert(0, "/root/repo")
import phc_gnn_amd, torch
from phc_gnn_amd import harness
DEV="cuda:0"
for wl in ("molhiv_b32","molhiv_b512_softmax","molhiv_b512_conv","molhiv_b512"):
    d = harness.to_device(harness.make_batch(wl, seed=33), DEV)
    m = harness.build_model(wl, DEV, seed=9); m.train()
    harness.train_step(m, d, None, harness.StepConfig(optimizer=False))
    g_def = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    for p in m.parameters(): p.grad = None
    harness.loss_fn(m(d), d).backward()
    g_plain = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    for p in m.parameters(): p.grad = None
    harness.loss_fn(m(d), d).backward()
    g_plain2 = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    top = max(float(v.abs().max()) for v in g_plain.values())
    def err(a,b): return sorted(((float((a[k]-b[k]).abs().max())/max(float(b[k].abs().max()),0.01*top), k) for k in a), reverse=True)[:3]
    print(wl, "top", top)
    print("  deferred vs plain:", err(g_def,g_plain))
    print("  plain vs plain   :", err(g_plain2,g_plain))
