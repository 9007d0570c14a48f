"""The unchanged loop body (benchmarks/train_hiv.py:174-202: ``logits = model(data)``, boolean-mask BCE, ``loss.backward()``) on the
mirror model: a training-mode forward outside the step's contexts opens them itself and closes them behind its backward
(ops.auto_step_forward) — same gradients as harness.train_step, ``torch.autograd.grad`` still works, a forward
without a backward leaves nothing behind."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0") if torch.cuda.is_available() else None


def _ref_loop_loss(model, data):
    mask = ~torch.isnan(data.y)
    logits = model(data)
    return F.binary_cross_entropy_with_logits(input=logits[mask], target=data.y[mask])


def _fresh(workload="molhiv_b32_swish", seed=3):
    from phc_gnn_amd import harness
    model = harness.build_model(workload, DEV, seed=seed).train()
    data = harness.to_device(harness.make_batch(workload, seed=9), DEV)
    return model, data


def test_plain_backward_takes_the_batched_tail_and_matches_the_harness_step():
    from phc_gnn_amd import harness, ops
    model, data = _fresh()
    loss = _ref_loop_loss(model, data)
    assert ops.auto_step_open() and ops._PREBUILT.table        # the forward opened the step's block by itself
    assert not ops._DEFER.active                               # ... and nothing is deferred outside the forward and the backward pass
    loss.backward()
    assert not ops.step_context_active() and getattr(ops._AUTO, "stack", None) is None      # ... and the backward closed it
    got = {k: p.grad.clone() for k, p in model.named_parameters()}
    m2, d2 = _fresh()
    l2 = harness.train_step(m2, d2, None, harness.StepConfig(optimizer=False, weight_decay=0.0))
    assert abs(float(loss) - float(l2)) <= 2e-6 * abs(float(l2))
    for k, p in m2.named_parameters():
        assert got[k] is not None and p.grad is not None, k
        torch.testing.assert_close(got[k], p.grad, rtol=1e-6, atol=1e-9, msg=k)


def test_autograd_grad_returns_the_gradients_and_leaves_dot_grad_alone():
    from phc_gnn_amd import ops
    model, data = _fresh()
    loss = _ref_loop_loss(model, data)
    params = [p for p in model.parameters()]
    grads = torch.autograd.grad(loss, params, allow_unused=True, retain_graph=True)
    assert all(p.grad is None for p in params)
    assert not ops._DEFER.active
    ops.close_auto_step()
    m2, d2 = _fresh()
    l2 = _ref_loop_loss(m2, d2)
    l2.backward()
    for (k, p), g in zip(m2.named_parameters(), grads):
        if p.grad is None:
            continue
        assert g is not None, k
        torch.testing.assert_close(g, p.grad, rtol=1e-5, atol=1e-8, msg=k)


def test_forward_without_backward_then_a_real_step():
    from phc_gnn_amd import ops
    model, data = _fresh()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    with torch.enable_grad():
        _ = model(data)                                        # opened, never closed by a backward
    assert ops.step_context_active()
    losses = []
    for _ in range(3):                                         # the loop body, three steps: parameters change in between
        opt.zero_grad()
        loss = _ref_loop_loss(model, data)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=2.0)
        opt.step()
        losses.append(loss.item())
    assert not ops.step_context_active()
    assert losses[2] < losses[0]                               # it trains
    model.eval()
    with torch.no_grad():
        out = model(data)                                      # eval: no block is opened
    assert not ops.step_context_active() and torch.isfinite(out).all()


def test_weight_regulariser_function_on_rocm_weights():
    from phc.hypercomplex.regularization import phm_weight_regularization
    model, _ = _fresh()
    for p_norm in (1, 2):
        reg = phm_weight_regularization(model, p=p_norm, device=DEV)
        grads = torch.autograd.grad(1e-3 * reg, [m.W for _, m in model.named_modules() if getattr(m, "W", None) is not None])
        ref = sum(m.W.norm(p=p_norm, dim=0).mean() for _, m in model.named_modules() if getattr(m, "W", None) is not None)
        rgrads = torch.autograd.grad(1e-3 * ref, [m.W for _, m in model.named_modules() if getattr(m, "W", None) is not None])
        assert abs(float(reg) - float(ref)) <= 2e-6 * abs(float(ref))
        for a, b in zip(grads, rgrads):
            torch.testing.assert_close(a, b, rtol=2e-5, atol=1e-9)
