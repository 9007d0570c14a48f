"""Dropout behind the norm's activation inside the engine pass (phc_bn_fwd / phc_bn_bwd with a Philox mask regenerated in
the backward): the structure the reference's phm_dropout guarantees (layers.py:31-55, its own test
test_ops_equal_quaternion.py:64-103: entries are 0 or x / (1 - p); `same=True` shares the mask between the hypercomplex
components), exact gradients for the mask that was drawn, a fresh mask per step under hipGraph replay."""
import pytest
import torch

from tests._util import TOL, assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _setup(n, h, N, seed):
    g = torch.Generator().manual_seed(seed)
    k = h // n
    x = torch.randn(N, h, generator=g) * 1.5 + 0.3
    res = torch.randn(N, h, generator=g)
    dy = torch.randn(N, h, generator=g)
    w = [torch.rand(k, generator=g) + 0.5 for _ in range(n)]
    b = [torch.randn(k, generator=g) * 0.1 for _ in range(n)]
    return x, res, dy, w, b, k


@pytest.mark.parametrize("same,h", [(False, 128), (True, 128), (False, 200)])      # 200 = 4 x 50: per-element masks need only h % 4 == 0
@pytest.mark.parametrize("act", ["swish", "relu"])
def test_fused_dropout_structure_and_gradients(same, h, act):
    from oracle import phc_oracle as O
    from phc_gnn_amd import ops
    n, N, p = 4, 1031, 0.3
    x, res, dy, w, b, k = _setup(n, h, N, 5)
    dev = torch.device(DEV)
    ops.rng_seed(dev, 1234)

    def run(dropout, act=act):
        xd, rd = x.to(DEV).requires_grad_(), res.to(DEV).requires_grad_()
        wd, bd = [t.to(DEV).requires_grad_() for t in w], [t.to(DEV).requires_grad_() for t in b]
        rm, rv = [torch.zeros(k, device=DEV) for _ in range(n)], [torch.ones(k, device=DEV) for _ in range(n)]
        out = ops.batch_norm_act(xd, wd, bd, rm, rv, None, True, 0.1, 1e-5, act, rd, None, dropout)
        grads = torch.autograd.grad(out, [xd, rd] + wd + bd, dy.to(DEV))
        return out.detach().cpu(), [g_.cpu() for g_ in grads]
    plain, _ = run(None)
    out, grads = run((p, same, 3))
    # the mask is a function of (seed, step, site, row, channel) only: read it off a run with the identity activation, where
    # the un-dropped value is non-zero everywhere
    probe, _ = run((p, same, 3), act="identity")
    dropped = (probe - res).abs() <= 1e-9
    mask = (~dropped).float()
    frac = float(dropped.sum()) / dropped.numel()
    assert abs(frac - p) < 0.01, frac
    y0 = plain - res                                   # act(norm(x))
    y1 = out - res                                     # the same after dropout
    assert_close(y1, y0 * mask / (1 - p), TOL, "every entry is either dropped or scaled by 1 / (1 - p)")
    if same:
        m3 = dropped.view(N, n, k)
        assert bool((m3 == m3[:, :1]).all()), "same=True: one mask per (row, channel), shared by the components"
    else:
        m3 = dropped.view(N, n, k)
        assert not bool((m3 == m3[:, :1]).all())
    # gradients for exactly this mask, through the oracle's norm
    xr, rr = x.clone().requires_grad_(), res.clone().requires_grad_()
    wr, br = [t.clone().requires_grad_() for t in w], [t.clone().requires_grad_() for t in b]
    rm, rv = [torch.zeros(k) for _ in range(n)], [torch.ones(k) for _ in range(n)]
    y = O.activation(act, O.batch_norm_naive(xr, n, wr, br, rm, rv, True)) * (mask / (1 - p)) + rr
    ref_g = torch.autograd.grad(y, [xr, rr] + wr + br, dy)
    assert_close(grads[0], ref_g[0], 3e-5, "dx")
    assert torch.equal(grads[1], dy)
    for a_, r_ in zip(grads[2:], ref_g[2:]):
        assert_close(a_, r_, 5e-5, "dw / db")
    # same (seed, step, site) -> same mask; another site or the next step -> another mask
    again, _ = run((p, same, 3))
    assert torch.equal(again, out)
    other, _ = run((p, same, 4))
    assert not torch.equal(other, out)
    ops.rng_advance(dev)
    nxt, _ = run((p, same, 3))
    assert not torch.equal(nxt, out)


def test_training_step_with_the_reference_default_dropout_under_graph_replay():
    """--dropout_mpnn 0.3,0.1,... / --dropout_dn 0.3,0.1 (train_hiv.py:99,110) through the captured step: finite losses, a
    different mask every replay (same batch, frozen weights => different losses), eval mode unaffected."""
    from phc.hypercomplex.undirectional.models import PHMSkipConnectAdd
    from phc_gnn_amd import graphed, harness
    from phc_gnn_amd.optim import ClipAdam
    w = dict(harness.WORKLOADS["molhiv_b32"])
    kw = harness.model_kwargs(w)
    kw.update(dropout_mpnn=[0.3, 0.1, 0.1, 0.1], dropout_dn=[0.3, 0.1], same_dropout=True)
    torch.manual_seed(0)
    model = PHMSkipConnectAdd(**kw).to(DEV).train()
    gb = harness.make_batch("molhiv_b32", seed=3)
    n_max, e_max = graphed.bucket_shape([gb])
    padded = graphed.padded_to_device(graphed.pad_batch(gb, n_max, e_max), DEV)
    cfg = harness.StepConfig(optimizer=False)
    step = graphed.GraphedTrainStep(model, ClipAdam(model.parameters(), lr=1e-3), cfg, padded, warmup=2)
    losses = [float(step.step(padded)) for _ in range(6)]
    assert all(l == l and abs(l) < 1e3 for l in losses), losses
    assert len(set(losses)) == len(losses), losses                     # a new mask every replay
    grads = [p.grad for p in model.parameters() if p.grad is not None]
    assert len(grads) > 100 and all(bool(torch.isfinite(g_).all()) for g_ in grads)
    model.eval()
    with torch.no_grad():
        a = model(step.data)
        b = model(step.data)
    assert torch.equal(a, b)


# ---- dropout INSIDE the fused MLP + norm node (round 5: the mask in the norm's apply pass and in the norm backward folded into the dx
# GEMM, HS_DR) against the separate nodes that layers with dropout used before: the masks are a function of (seed, step, site, row,
# column) only, so both paths draw the SAME masks and the whole step must agree ---------------------------------------------------
_DROP_WORKLOADS = {
    "molhiv_small_drop": dict(dataset="molhiv", batch_size=48, phm_dim=4, hidden=128, layers=3, d_units=[64, 32], target_dim=1, sc_type="first",
                              dropout_mpnn=[0.2, 0.1, 0.3], dropout_dn=[0.2, 0.1], activation="swish"),
    "molhiv_small_drop_same": dict(dataset="molhiv", batch_size=48, phm_dim=4, hidden=128, layers=2, d_units=[64, 32], target_dim=1, sc_type="first",
                                   dropout_mpnn=[0.25, 0.25], same_dropout=True, activation="swish"),
    "zinc_small_256_drop": dict(dataset="zinc", batch_size=64, phm_dim=8, hidden=256, layers=2, d_units=[64, 32], target_dim=1, sc_type="last",
                                regression=True, dropout_mpnn=[0.2, 0.2], activation="swish"),
    "hiv_pub_200_drop": dict(dataset="molhiv", batch_size=64, phm_dim=4, hidden=200, layers=2, d_units=[128, 32], target_dim=1, sc_type="first",
                             msg_aggr="softmax", dropout_mpnn=[0.2, 0.2], dropout_dn=[0.3, 0.1], activation="swish"),
    "zinc_pub_180_drop_relu": dict(dataset="zinc", batch_size=64, phm_dim=4, hidden=180, layers=2, d_units=[180, 80], target_dim=1, sc_type="last",
                                   regression=True, dropout_mpnn=[0.1, 0.1]),
}


def _drop_step(workload, fused, graph=False):
    from phc_gnn_amd import graphed, harness, lib as L, ops
    from phc_gnn_amd.optim import ClipAdam
    harness.WORKLOADS.setdefault(workload, _DROP_WORKLOADS[workload])
    keep = ops.MLP_NORM_DROPOUT
    ops.MLP_NORM_DROPOUT = fused
    try:
        dev = torch.device(DEV)
        model = harness.build_model(workload, DEV, seed=4).train()
        if _DROP_WORKLOADS[workload].get("same_dropout"):
            model.same_dropout = True
        gb = harness.make_batch(workload, seed=9)
        cfg = harness.StepConfig(regression=bool(_DROP_WORKLOADS[workload].get("regression")), optimizer=False, weight_decay=0.0)
        ops.rng_seed(dev, 777)
        if not graph:                                        # (the launch profiler's events cannot be recorded inside a capture)
            L.prof_enable(True)
            L.prof_reset()
        try:
            if graph:
                n_max, e_max = graphed.bucket_shape([gb])
                padded = graphed.padded_to_device(graphed.pad_batch(gb, n_max + 21, e_max + 33), DEV)
                sd = {k: v.clone() for k, v in model.state_dict().items()}
                step = graphed.GraphedTrainStep(model, ClipAdam(model.parameters(), lr=1e-3), cfg, padded, warmup=2)
                model.load_state_dict(sd)
                ops.rng_seed(dev, 777)
                loss = step.step(padded)
            else:
                loss = harness.train_step(model, harness.to_device(gb, DEV), None, cfg)
            torch.cuda.synchronize()
            names = {k for k, v in L.prof_read().items() if v["launches"] > 0} if not graph else set()
        finally:
            if not graph:
                L.prof_enable(False)
        return loss.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}, names
    finally:
        ops.MLP_NORM_DROPOUT = keep


@pytest.mark.parametrize("workload", list(_DROP_WORKLOADS))
def test_dropout_inside_the_fused_mlp_norm_node_equals_the_separate_nodes(workload):
    """Loss and every gradient of one training step (same seed: same masks) with the layers' dropout inside the fused node
    (ops.MLP_NORM_DROPOUT) and on the separate norm nodes; widths 128 / 256 / 200 / 180, `same` masks, softmax aggregation, ReLU."""
    on = _drop_step(workload, True)
    off = _drop_step(workload, False)
    assert abs(float(on[0]) - float(off[0])) <= 1e-5 * max(abs(float(off[0])), 1e-30), (float(on[0]), float(off[0]))
    relu = _DROP_WORKLOADS[workload].get("activation", "relu") == "relu"
    for k in off[1]:
        ref = off[1][k]
        if float(ref.abs().max()) < 1e-7:
            continue
        err = float((on[1][k] - ref).abs().max()) / max(float(ref.abs().max()), 1e-30)
        assert err <= (2e-3 if relu else 3e-5), (k, err)          # (ReLU: a gate at the kink may fall on the other side in another summation order)
    # the fused node really ran: the separate nodes' element-wise norm backward is gone from the layers (the head keeps its own)
    assert "bn_bwd_apply" in off[2]


def test_dropout_inside_the_fused_node_draws_a_new_mask_every_replay():
    """hipGraph replay: the step counter is a device value — two replays from the same state give different losses, the same seed and
    step the same loss (the fused node's three mask sites agree with each other, or the gradients above would not match)."""
    a = _drop_step("molhiv_small_drop", True, graph=True)
    b = _drop_step("molhiv_small_drop", True, graph=True)
    assert float(a[0]) == float(b[0])
    e = _drop_step("molhiv_small_drop", True, graph=False)       # the eager step from the same state, seed and step counter: the same masks
    assert abs(float(a[0]) - float(e[0])) <= 1e-5 * abs(float(e[0]))
