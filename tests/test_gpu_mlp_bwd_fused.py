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
