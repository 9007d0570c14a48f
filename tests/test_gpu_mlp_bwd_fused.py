"""The MLP + norm node's backward as ONE launch (phc_mlp_norm_bwd_fused, csrc/mlp_fused.hip) against the three launches it replaces
(phc_bn_bwd_partial_acc + 2 x phc_gemm_bwd_apply) and, through the whole step, against the oracle (autograd of
phc/hypercomplex/layers.py:349-355 + undirectional/models.py:208-215)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0") if torch.cuda.is_available() else None


def _step(workload, fused, monkeypatch, graph, seed=5):
    from phc_gnn_amd import graphed, harness, lib, ops
    from phc_gnn_amd.optim import ClipAdam
    monkeypatch.setattr(ops, "MLP_BWD_FUSED", fused)
    calls = []
    real = lib.load().phc_mlp_norm_bwd_fused
    model = harness.build_model(workload, DEV, seed=seed).train()
    w = harness.WORKLOADS[workload]
    cfg = harness.StepConfig(regression=bool(w.get("regression")), optimizer=False, weight_decay=1e-3)
    gb = harness.make_batch(workload, seed=13)
    if graph:
        n_max, e_max = graphed.bucket_shape([gb])
        padded = graphed.padded_to_device(graphed.pad_batch(gb, n_max, e_max), DEV)
        step = graphed.GraphedTrainStep(model, ClipAdam(model.parameters(), lr=1e-3), cfg, padded, warmup=2)
        with step.on_stream():
            step.step(padded)
            loss = step.step(padded).clone()
        torch.cuda.synchronize()
    else:
        data = harness.to_device(gb, DEV)
        loss = harness.train_step(model, data, None, cfg)
        torch.cuda.synchronize()
    lib.check_indices(DEV)
    return float(loss.detach()), {k: p.grad.detach().clone() for k, p in model.named_parameters()}


@pytest.mark.parametrize("workload", ["molhiv_b32", "molhiv_b512", "molhiv_b512_swish", "molhiv_b512_softmax"])
@pytest.mark.parametrize("graph", [False, True])
def test_fused_backward_equals_the_three_launches(workload, graph, monkeypatch):
    la, ga = _step(workload, True, monkeypatch, graph)
    lb, gb_ = _step(workload, False, monkeypatch, graph)
    assert la == lb                                           # (the forward is untouched)
    for k in gb_:
        ref = float(gb_[k].abs().max())
        if ref < 1e-7:                                        # (a bias in front of a batch-norm: its gradient is round-off around zero)
            continue
        err = float((ga[k] - gb_[k]).abs().max()) / ref
        assert err <= 1e-5, (k, err)


def test_fused_backward_is_bit_reproducible(monkeypatch):
    a = _step("molhiv_b512", True, monkeypatch, True)
    b = _step("molhiv_b512", True, monkeypatch, True)
    assert a[0] == b[0]
    for k in a[1]:
        assert torch.equal(a[1][k], b[1][k]), k


def test_fused_backward_runs():
    """The switch is on by default and the molecule shapes take the one-launch form."""
    from phc_gnn_amd import lib, ops
    l = lib.load()
    assert ops.MLP_BWD_FUSED
    assert l.phc_mlp_norm_bwd_fused_supported(12874, 128) == 1 and l.phc_mlp_norm_bwd_fused_supported(13312, 128) == 1
    assert l.phc_mlp_norm_bwd_fused_supported(12874, 256) == 0 and l.phc_mlp_norm_bwd_fused_supported(40000, 128) == 0


# ---- the dx of the sum aggregation ABOVE built inside the launch (phc_mlp_norm_bwd_desc.agg_src, ops.DxLink / PendingDx) ----
def _count_agg_launches(monkeypatch):
    """Wrap the entry point: how many launches carried an aggregation."""
    from phc_gnn_amd import lib
    h = lib.load()
    real = h.phc_mlp_norm_bwd_fused
    seen = {"agg": 0, "plain": 0}

    def wrapped(desc, stream):
        d = desc._obj if hasattr(desc, "_obj") else desc.contents
        seen["agg" if d.agg_src else "plain"] += 1
        return real(desc, stream)
    monkeypatch.setattr(h, "phc_mlp_norm_bwd_fused", wrapped)
    return seen


def _step_agg(workload, agg, monkeypatch, graph, seed=5):
    from phc_gnn_amd import ops
    monkeypatch.setattr(ops, "MLP_BWD_FUSED_AGG", agg)
    return _step(workload, True, monkeypatch, graph, seed)


@pytest.mark.parametrize("workload", ["molhiv_b32", "molhiv_b512", "molhiv_b512_swish", "molhiv_b512_last"])
@pytest.mark.parametrize("graph", [False, True])
def test_aggregation_dx_inside_the_backward_launch(workload, graph, monkeypatch):
    """sc_type "first" (the skip is the encoded atoms: no residual) and "last" (the skip's share is the residual)."""
    from phc_gnn_amd import harness
    if workload == "molhiv_b512_last":
        monkeypatch.setitem(harness.WORKLOADS, workload, dict(harness.WORKLOADS["molhiv_b512"], sc_type="last"))
    seen = _count_agg_launches(monkeypatch)
    la, ga = _step_agg(workload, True, monkeypatch, graph)
    assert seen["agg"] >= 3, seen                             # layers 0 .. 2 build the dx of the aggregation above them
    seen["agg"] = 0
    lb, gb_ = _step_agg(workload, False, monkeypatch, graph)
    assert seen["agg"] == 0
    assert la == lb
    for k in gb_:
        ref = float(gb_[k].abs().max())
        if ref < 1e-7:
            continue
        err = float((ga[k] - gb_[k]).abs().max()) / ref
        assert err <= 1e-5, (k, err)


def test_aggregation_dx_inside_the_backward_is_bit_reproducible(monkeypatch):
    a = _step_agg("molhiv_b512", True, monkeypatch, True)
    b = _step_agg("molhiv_b512", True, monkeypatch, True)
    assert a[0] == b[0]
    for k in a[1]:
        assert torch.equal(a[1][k], b[1][k]), k


@pytest.mark.parametrize("residual", [False, True])
@pytest.mark.parametrize("add_self", [False, True])
def test_aggregation_dx_rows_are_the_row_walkers(residual, add_self):
    """Op level: dx_out bit-identical to phc_aggregate_dx_rows (out-degrees 0 .. 16, an isolated tail of padding rows); dy2 / dz1 / ds
    of the launch equal the plain one-launch form's fed with that dx (the column sums meet in another order: round-off)."""
    import ctypes as C
    from phc_gnn_amd import lib as L, ops
    from phc_gnn_amd.graph import EdgeStructure
    lib = L.load()
    g = torch.Generator(device="cpu").manual_seed(3)
    N, h, nv = 1000, 128, 950
    # molecule-like edges plus a hub of out-degree 16 and rows of out-degree 5 .. 7
    src = torch.randint(0, nv, (2400,), generator=g)
    dst = (src + torch.randint(-6, 7, (2400,), generator=g)).clamp(0, nv - 1)
    src = src[src != 17]
    dst = (src + torch.randint(-6, 7, (src.numel(),), generator=g)).clamp(0, nv - 1)
    hub_s = torch.full((16,), 17, dtype=torch.long)          # (more than 16 out-edges: the row walker's cooperative form, another order)
    hub_d = torch.randint(0, nv, (16,), generator=g)
    ei = torch.stack([torch.cat([src, hub_s]), torch.cat([dst, hub_d])]).to(DEV)
    E = ei.shape[1]
    st = EdgeStructure(ei, N, N)
    bond = torch.zeros((E, 3), dtype=torch.long, device=DEV)
    vocab = (C.c_int32 * 3)(5, 6, 2)
    st.prepare(for_backward=True, codes_for=(bond, vocab, 3))
    assert st._ell_t is not None
    rowptr_t, col_t, _ = st.csc
    deg = (rowptr_t[1:] - rowptr_t[:-1])
    assert int(deg.max()) == 16 and int((deg > 4).sum()) > 20 and int((deg == 0).sum()) > 50
    gsrc = torch.randn((N, h), generator=g).to(DEV)
    gsrc[nv:] = 0
    res = torch.randn((N, h), generator=g).to(DEV) if residual else None
    dx_ref = torch.empty((N, h), device=DEV)
    L.check(lib.phc_aggregate_dx_rows(L.ptr(gsrc), L.ptr(rowptr_t), L.ptr(col_t), N, N, h, 1 if add_self else 0, L.ptr(res), L.ptr(dx_ref),
                                      L.stream_ptr(DEV)), "phc_aggregate_dx_rows")
    y2, z1 = torch.randn((N, h), generator=g).to(DEV), torch.randn((N, h), generator=g).to(DEV)

    def stat(x):
        m, v = x[:nv].mean(0), x[:nv].var(0, unbiased=False)
        inv = 1.0 / torch.sqrt(v + 1e-5)
        w, b = torch.rand(h, generator=g).to(DEV) + 0.5, torch.randn(h, generator=g).to(DEV) * 0.1
        return torch.cat([m, inv, inv * w, b - m * inv * w, w, b]).contiguous()
    save_o, save_i = stat(y2), stat(z1)
    H2, H1 = torch.randn((h, h), generator=g).to(DEV) * 0.1, torch.randn((h, h), generator=g).to(DEV) * 0.1
    k4 = lambda H: H.reshape(h // 4, 4, h).permute(0, 2, 1).contiguous()        # [k / 4][col][4]
    H2Tk4, H1Tk4 = k4(H2.t().contiguous()), k4(H1.t().contiguous())
    n_valid = torch.tensor([nv], dtype=torch.int32, device=DEV)

    def run(agg):
        fd = L.MlpNormBwdDesc()
        outs = [torch.empty((N, h), device=DEV) for _ in range(3)]
        dwb = [torch.empty(2 * h, device=DEV) for _ in range(2)]
        accs = [torch.zeros(lib.phc_stat_acc_words(h), dtype=torch.int64, device=DEV) for _ in range(2)]
        bar = torch.zeros(lib.phc_mlp_norm_bwd_fused_barrier_words(), dtype=torch.int32, device=DEV)
        dx = torch.empty((N, h), device=DEV)
        fd.y2, fd.z1, fd.save_o, fd.save_i, fd.H2Tk4, fd.H1Tk4 = (L.ptr(t) for t in (y2, z1, save_o, save_i, H2Tk4, H1Tk4))
        fd.dy2, fd.dz1, fd.ds, fd.dwb_o, fd.dwb_i = (L.ptr(t) for t in outs + dwb)
        fd.acc_o, fd.acc_i, fd.n_valid_dev, fd.N, fd.h, fd.act_o, fd.act_i = L.ptr(accs[0]), L.ptr(accs[1]), L.ptr(n_valid), N, h, 1, 1
        fd.barrier = L.ptr(bar)
        if agg:
            fd.agg_src, fd.agg_residual, fd.dx_out = L.ptr(gsrc), L.ptr(res), L.ptr(dx)
            fd.agg_rowptr_t, fd.agg_col_t, fd.agg_ell_t, fd.agg_add_self = L.ptr(rowptr_t), L.ptr(col_t), L.ptr(st._ell_t), 1 if add_self else 0
        else:
            fd.dout = L.ptr(dx_ref)
        L.check(lib.phc_mlp_norm_bwd_fused(C.byref(fd), L.stream_ptr(DEV)), "phc_mlp_norm_bwd_fused")
        torch.cuda.synchronize()
        L.check_indices(DEV)
        return dx, outs, dwb
    dx, outs, dwb = run(True)
    bad = (dx != dx_ref).any(1).nonzero().flatten()
    assert bad.numel() == 0, (bad[:8].tolist(), deg[bad[:8]].tolist())
    _, outs0, dwb0 = run(False)
    for a, b in zip(outs + dwb, outs0 + dwb0):
        ref = float(b.abs().max())
        assert float((a - b).abs().max()) <= 2e-6 * max(ref, 1.0), float((a - b).abs().max())


def test_parked_aggregation_dx_is_launched_when_the_consumer_cannot_fuse(monkeypatch):
    """The safety net: the aggregation parked its dx (the link said the node below would build it), but at backward time that node's
    one-launch form is not available — it must launch the stand-alone kernel itself (PendingDx.materialize), same results."""
    from phc_gnn_amd import ops
    la, ga = _step_agg("molhiv_b32", True, monkeypatch, False)[:2]
    real = ops._supported
    state = {"calls": 0, "materialized": 0}

    def supported(name, *args):
        if name == "phc_mlp_norm_bwd_fused_agg_supported":
            state["calls"] += 1
            return state["calls"] <= 4                       # the four forward-time queries (link.ready) say yes, the backward's say no
        return real(name, *args)
    real_mat = ops.PendingDx.materialize

    def mat(self):
        state["materialized"] += 1
        return real_mat(self)
    monkeypatch.setattr(ops, "_supported", supported)
    monkeypatch.setattr(ops.PendingDx, "materialize", mat)
    lb, gb_ = _step_agg("molhiv_b32", True, monkeypatch, False)[:2]
    assert state["materialized"] == 3, state
    assert la == lb
    for k in gb_:
        ref = float(gb_[k].abs().max())
        if ref >= 1e-7:
            assert float((ga[k] - gb_[k]).abs().max()) <= 1e-5 * ref, k


def test_aggregation_dx_inside_the_backward_under_autograd_grad(monkeypatch):
    """torch.autograd.grad (no AccumulateGrad nodes, no deferral): the link between the nodes works the same."""
    from phc_gnn_amd import harness, ops
    seen = _count_agg_launches(monkeypatch)

    def run(agg):
        monkeypatch.setattr(ops, "MLP_BWD_FUSED_AGG", agg)
        model = harness.build_model("molhiv_b32_swish", DEV, seed=9).train()
        data = harness.to_device(harness.make_batch("molhiv_b32_swish", seed=4), DEV)
        params = [p for p in model.parameters() if p.requires_grad]
        with ops.stat_accumulators(DEV), ops.prebuilt_H(ops.phm_pairs(model), ops.dense_weights(model), ops.gate_specs(model)):
            loss = model(data).square().mean()
            grads = torch.autograd.grad(loss, params, allow_unused=True)
        torch.cuda.synchronize()
        return float(loss.detach()), grads
    la, ga = run(True)
    assert seen["agg"] == 3, seen
    lb, gb_ = run(False)
    assert la == lb
    for a, b in zip(ga, gb_):
        if b is None:
            assert a is None
            continue
        ref = float(b.abs().max())
        if ref >= 1e-7:
            assert float((a - b).abs().max()) <= 1e-5 * ref
