"""Which path is right?  float64 CPU oracle vs the plain-autograd GPU path vs the deferred/prebuilt GPU path."""
import sys; sys.path.insert(0, "/root/repo")
import phc_gnn_amd, torch
from phc_gnn_amd import harness
from oracle import phc_oracle as O
DEV = "cuda:0"
wl = sys.argv[1] if len(sys.argv) > 1 else "molhiv_b512"
w = harness.WORKLOADS[wl]
gb = harness.make_batch(wl, seed=33)
d = harness.to_device(gb, DEV)
m = harness.build_model(wl, DEV, seed=9); m.train()
names = [k for k, _ in m.named_parameters()]
sd = {k: v.detach().cpu().double().clone() for k, v in m.state_dict().items()}
for k in names: sd[k].requires_grad_()
x, ei, ea, batch = (torch.from_numpy(a) for a in (gb.x, gb.edge_index, gb.edge_attr, gb.batch))
y = torch.from_numpy(gb.y).double()
logits = O.skip_connect_add_forward(sd, x, ei, ea, batch, phm_dim=w["phm_dim"], num_layers=w["layers"], num_atom_feats=x.shape[1],
                                    num_bond_feats=ea.shape[1], num_dn_hidden=len(w["d_units"]), mlp=w.get("mlp", True),
                                    sc_type=w["sc_type"], training=True, num_graphs=gb.num_graphs, msg_aggr=w.get("msg_aggr", "sum"))
loss = O.bce_loss_masked(logits, y); loss.backward()
ref = {k: sd[k].grad.float() for k in names if sd[k].grad is not None}
harness.train_step(m, d, None, harness.StepConfig(optimizer=False))
g_def = {k: p.grad.clone().cpu() for k, p in m.named_parameters() if p.grad is not None}
for p in m.parameters(): p.grad = None
lp = harness.loss_fn(m(d), d); lp.backward()
g_plain = {k: p.grad.clone().cpu() for k, p in m.named_parameters() if p.grad is not None}
top = max(float(v.abs().max()) for v in ref.values())
def err(a): return sorted(((float((a[k] - ref[k]).abs().max()) / max(float(ref[k].abs().max()), 0.01 * top), k) for k in a if k in ref), reverse=True)[:4]
print(wl, "loss64 %.8f  gpu %.8f" % (float(loss), float(lp)))
print("  deferred vs oracle64:", err(g_def))
print("  plain    vs oracle64:", err(g_plain))
