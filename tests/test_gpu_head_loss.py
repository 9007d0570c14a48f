"""The training loss inside the prediction head's forward launch (phc_downstream_desc.loss_*, ABI 8; ops.expect_loss) against the
stand-alone loss launch and against the oracle's masked BCE (benchmarks/train_hiv.py:174-178) / L1 (train_zinc.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0") if torch.cuda.is_available() else None


def _step(workload, head_loss, monkeypatch, seed=3, graph=False):
    from phc_gnn_amd import graphed, harness, ops
    from phc_gnn_amd.optim import ClipAdam
    monkeypatch.setattr(ops, "HEAD_LOSS", head_loss)
    model = harness.build_model(workload, DEV, seed=seed).train()
    w = harness.WORKLOADS[workload]
    cfg = harness.StepConfig(regression=bool(w.get("regression")), optimizer=False, weight_decay=1e-3)
    gb = harness.make_batch(workload, seed=11)
    if w["target_dim"] == 1 and not w.get("regression"):
        gb.y[::7] = np.nan                                   # unlabeled entries (ogbg-molhiv has none; the mask must still be honoured)
    launches = []
    # (the loss launches go through _masked_loss_forward: count its calls; restored by monkeypatch at the end of the test, so a second
    # _step of the same test wraps the wrapper — each list only sees the calls made while it is the innermost one)
    inner = ops._masked_loss_forward
    active = {"on": True}
    def counted(kind, ctx, a, b):
        if active["on"]:
            launches.append(kind)
        return inner(kind, ctx, a, b)
    monkeypatch.setattr(ops, "_masked_loss_forward", counted)
    if graph:
        n_max, e_max = graphed.bucket_shape([gb])
        padded = graphed.padded_to_device(graphed.pad_batch(gb, n_max, e_max), DEV)
        step = graphed.GraphedTrainStep(model, ClipAdam(model.parameters(), lr=1e-3), cfg, padded, warmup=2)
        with step.on_stream():
            loss = step.step(padded).clone()
            loss = step.step(padded).clone()
        torch.cuda.synchronize()
    else:
        data = harness.to_device(gb, DEV)
        loss = harness.train_step(model, data, None, cfg)
        torch.cuda.synchronize()
    from phc_gnn_amd import lib
    lib.check_indices(DEV)
    active["on"] = False
    return float(loss.detach()), {k: p.grad.detach().clone() for k, p in model.named_parameters()}, launches, gb, model


@pytest.mark.parametrize("workload", ["molhiv_b32", "molhiv_b512", "zinc_b1024", "molhiv_pub_phc4_2x200_nodrop"])
@pytest.mark.parametrize("graph", [False, True])
def test_loss_inside_the_head_equals_the_loss_launch(workload, graph, monkeypatch):
    la, ga, launches_a, _, _ = _step(workload, True, monkeypatch, graph=graph)
    lb, gb_, launches_b, _, _ = _step(workload, False, monkeypatch, graph=graph)
    assert launches_a == [] and len(launches_b) >= 1          # the head took the loss; without the switch the loss launch runs
    assert abs(la - lb) <= 2e-6 * max(abs(lb), 1e-3), (la, lb)   # (another summation tree: 512 lanes instead of 1024)
    for k in gb_:
        err = float((ga[k] - gb_[k]).abs().max()) / max(float(gb_[k].abs().max()), 1e-30)
        assert err <= 2e-6, (k, err)


def test_head_loss_against_the_oracle(monkeypatch):
    from oracle import phc_oracle as O
    loss, grads, launches, gb, model = _step("molhiv_b32_swish", True, monkeypatch)
    assert launches == []
    p = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    names = [k for k, _ in model.named_parameters()]
    for k in names:
        p[k].requires_grad_()
    logits = O.skip_connect_add_forward(p, torch.from_numpy(gb.x), torch.from_numpy(gb.edge_index), torch.from_numpy(gb.edge_attr),
                                        torch.from_numpy(gb.batch), phm_dim=4, num_layers=4, num_atom_feats=9, num_bond_feats=3,
                                        num_dn_hidden=2, mlp=True, training=True, num_graphs=gb.num_graphs, act="swish")
    ref = O.bce_loss_masked(logits, torch.from_numpy(gb.y))
    ref = ref + 1e-3 * 1e-3 * O.phm_weight_regularization([p[k] for k in names if k.endswith(".W")], p=2)
    ref_grads = torch.autograd.grad(ref, [p[k] for k in names])
    assert abs(loss - float(ref)) <= 1e-5 * abs(float(ref))
    for k, g in zip(names, ref_grads):
        if float(g.abs().max()) > 1e-7:
            err = float((grads[k].cpu().double() - g.double()).abs().max()) / float(g.abs().max())
            assert err <= 3e-5, (k, err)


def test_loss_hint_is_ignored_when_the_labels_differ(monkeypatch):
    """``masked_bce_with_logits`` on OTHER labels than the hinted ones must not return the head's loss."""
    from phc_gnn_amd import harness, ops
    model = harness.build_model("molhiv_b32", DEV, seed=1).train()
    gb = harness.make_batch("molhiv_b32", seed=2)
    data = harness.to_device(gb, DEV)
    other = (1.0 - data.y).contiguous()
    with ops.stat_accumulators(DEV), ops.prebuilt_H(ops.phm_pairs(model), ops.dense_weights(model), ops.gate_specs(model)):
        with ops.expect_loss(data.y, "bce"):
            logits = model(data)
        assert getattr(logits, "_phc_loss", None) is not None
        a = ops.masked_bce_with_logits(logits, data.y)
        b = ops.masked_bce_with_logits(logits, other)
        c = ops._MaskedBCEFn.apply(logits.detach(), data.y)
    assert abs(float(a) - float(c)) <= 2e-6 * abs(float(c))
    assert abs(float(a) - float(b)) > 1e-3


# ---- forward AND backward of the head in ONE launch (phc_downstream_desc.fused_bwd, ops.HEAD_FUSED_BWD) -------------------------------
def _count_head_launches(monkeypatch):
    from phc_gnn_amd import lib
    h = lib.load()
    seen = {"fwd": 0, "fwd_fused": 0, "bwd": 0}
    real_f, real_b = h.phc_downstream_fwd, h.phc_downstream_bwd

    def fwd(desc, stream):
        d = desc._obj if hasattr(desc, "_obj") else desc.contents
        seen["fwd_fused" if d.fused_bwd else "fwd"] += 1
        return real_f(desc, stream)

    def bwd(desc, stream):
        seen["bwd"] += 1
        return real_b(desc, stream)
    monkeypatch.setattr(h, "phc_downstream_fwd", fwd)
    monkeypatch.setattr(h, "phc_downstream_bwd", bwd)
    return seen


@pytest.mark.parametrize("workload", ["molhiv_b32", "molhiv_b512", "zinc_b1024", "molhiv_pub_phc4_2x200_nodrop", "molhiv_pub_phc4_2x200"])
@pytest.mark.parametrize("graph", [False, True])
def test_head_backward_inside_the_forward_launch_is_bit_identical(workload, graph, monkeypatch):
    """BCE and L1 heads, 16- and 32-row workgroups, dropout behind the head's activations: the one launch == the two launches."""
    from phc_gnn_amd import ops
    seen = _count_head_launches(monkeypatch)
    monkeypatch.setattr(ops, "HEAD_FUSED_BWD", True)
    ops.rng_seed(DEV, 77)                                     # (the dropout workload: the same masks in both runs)
    la, ga, _, _, _ = _step(workload, True, monkeypatch, graph=graph)
    assert seen["fwd_fused"] >= 1 and seen["bwd"] == 0 and seen["fwd"] == 0, seen
    for k in seen:
        seen[k] = 0
    monkeypatch.setattr(ops, "HEAD_FUSED_BWD", False)
    ops.rng_seed(DEV, 77)
    lb, gb_, _, _, _ = _step(workload, True, monkeypatch, graph=graph)
    assert seen["fwd_fused"] == 0 and seen["bwd"] >= 1, seen
    assert la == lb
    for k in gb_:
        assert torch.equal(ga[k], gb_[k]), (k, float((ga[k] - gb_[k]).abs().max()))


def test_head_backward_falls_back_when_the_upstream_gradient_is_not_one(monkeypatch):
    """(2 loss).backward(): the gradient the head's node receives is not the forward launch's d loss / d logits — it launches
    phc_downstream_bwd with what it received, and the results are those of the two-launch path."""
    from phc_gnn_amd import harness, ops
    seen = _count_head_launches(monkeypatch)

    def run(fused):
        monkeypatch.setattr(ops, "HEAD_FUSED_BWD", fused)
        model = harness.build_model("molhiv_b32", DEV, seed=4).train()
        data = harness.to_device(harness.make_batch("molhiv_b32", seed=5), DEV)
        with ops.stat_accumulators(DEV), ops.prebuilt_H(ops.phm_pairs(model), ops.dense_weights(model), ops.gate_specs(model)):
            with ops.expect_loss(data.y, "bce"):
                logits = model(data)
            loss = ops.masked_bce_with_logits(logits, data.y)
            (2.0 * loss).backward()
        torch.cuda.synchronize()
        return float(loss), {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    la, ga = run(True)
    assert seen["fwd_fused"] == 1 and seen["bwd"] == 1, seen
    lb, gb_ = run(False)
    assert la == lb
    for k in gb_:
        assert torch.equal(ga[k], gb_[k]), k
