"""Per-layer check of the fused MLP+norm node's backward against torch formulas on device (debug)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import phc_gnn_amd  # noqa
import torch
from phc_gnn_amd import harness, ops
DEV = "cuda:0"
wl = sys.argv[1] if len(sys.argv) > 1 else "molhiv_b32"
orig = ops._MLPNormFn.backward
layer = [0]


def act_grad(act, z):
    if act == 1:
        return (z > 0).to(z.dtype)
    if act == 4:
        sg = torch.sigmoid(z)
        return sg * (1 + z * (1 - sg))
    return torch.ones_like(z)


def checked(ctx, dout):
    out = orig(ctx, dout)
    torch.cuda.synchronize()
    s, z1, y2, save_i, save_o, A1, W1, A2, W2 = ctx.saved_tensors
    meta = ctx.meta
    h = s.shape[1]
    n = s.shape[0]

    def bwd(dy, x, st, act):
        mean, inv, w, b = st[:h].double(), st[h:2 * h].double(), st[4 * h:5 * h].double(), st[5 * h:6 * h].double()
        xh = (x.double() - mean) * inv
        g = dy.double() * act_grad(act, xh * w + b)
        return w * inv * (g - g.mean(0) - xh * (g * xh).mean(0)), g, xh
    dy2, _, _ = bwd(dout, y2, save_o, meta["act_outer"])
    da = dy2 @ meta["pre2"][1].double()
    dz1, g_i, xh_i = bwd(da.float(), z1, save_i, meta["act_inner"])
    ds = dz1 @ meta["pre1"][1].double()
    e = float((out[0].double() - ds).abs().max() / ds.abs().max())
    rows = ((out[0].double() - ds).abs().amax(1) > 1e-4 * float(ds.abs().max())).nonzero().flatten()
    print(f"fused backward #{layer[0]}: N={n} ds err {e:.2e}; bad rows: {rows[:12].tolist()} ... ({rows.numel()} rows); dout contiguous={dout.is_contiguous()} stride={dout.stride()} ptr%16={dout.data_ptr() % 16}")
    layer[0] += 1
    return out


ops._MLPNormFn.backward = staticmethod(checked)
model = harness.build_model(wl, DEV, seed=1)
model.train()
data = harness.to_device(harness.make_batch(wl, seed=5), DEV)
ops.MLP_NORM_FUSED_MIN_WORK = 0
harness.train_step(model, data, None, harness.StepConfig(optimizer=False))
