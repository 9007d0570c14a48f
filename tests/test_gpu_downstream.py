"""The downstream head as one engine launch each way (csrc/downstream.hip) against the CPU oracle's restatement of
PHMDownstreamNet.forward + RealTransformer (downstream.py:90-120, layers.py:399-414): logits, dx, every parameter
gradient, running statistics; padded batches (device-side valid-row count); repeated launches (the in-kernel rendezvous
counters are never reset); bitwise reproducibility."""
import pytest
import torch

from tests._util import TOL, assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _make(n, d_in, hidden, target, act, seed):
    from phc.hypercomplex.downstream import PHMDownstreamNet
    torch.manual_seed(seed)
    __import__("numpy").random.seed(seed)
    dn = PHMDownstreamNet(in_features=d_in, phm_dim=n, phm_rule=None, hidden_layers=list(hidden), out_features=target,
                          activation=act, bias=True, norm="naive-batch-norm", w_init="phm", c_init="standard", dropout=[0.0, 0.0])
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in dn.named_parameters():
            if name.endswith("phm_rule"):
                p.add_(0.05 * torch.randn(p.shape, generator=g))
            elif p.dim() == 1 and "norm" in name:
                p.copy_(0.5 + torch.rand(p.shape, generator=g) if name.endswith("weight") else 0.1 * torch.randn(p.shape, generator=g))
            elif name.endswith(".b") or name.endswith("affine.bias"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
    return dn.train()


def _oracle(dn, x, dy, n, act, rows=None):
    from oracle import phc_oracle as O
    p = {"q." + k: v.detach().cpu().clone() for k, v in dn.state_dict().items()}
    names = ["q." + k for k, _ in dn.named_parameters()]
    for k in names:
        p[k].requires_grad_()
    xr = x.clone().requires_grad_()
    xin = xr if rows is None else xr[:rows]
    y = O.downstream_net(xin, p, "q.", n, 2, act, True, True)
    grads = torch.autograd.grad(y, [xr] + [p[k] for k in names], dy if rows is None else dy[:rows], allow_unused=True)
    return y, grads[0], dict(zip([k[2:] for k in names], grads[1:])), p


def _run_fused(dn, x, dy, n_valid=None):
    from phc_gnn_amd import ops
    xd = x.to(DEV).requires_grad_()
    with ops.prebuilt_H(ops.phm_pairs(dn), ops.dense_weights(dn)):
        y = dn(xd, n_valid=n_valid)
        assert type(y.grad_fn).__name__ == "_DownstreamFnBackward", type(y.grad_fn).__name__      # the fused launch really ran
        y.backward(dy.to(DEV))
    return y.detach(), xd.grad


@pytest.mark.parametrize("n,d_in,hidden,target,act,B", [
    (4, 128, (128, 32), 1, "relu", 544), (4, 128, (128, 32), 1, "relu", 9),
    (4, 64, (64, 32), 2, "swish", 100), (8, 128, (64, 64), 1, "lrelu", 33),
    (4, 32, (128, 128), 3, "relu", 1025),
    # the 16-row kernels' wider range (round 4): configs[2]'s head (ZINC PHM-8, d_units 256,128), a 256-wide PHM-4 head, the
    # reference's own ZINC head (run_script_zinc_phm4.sh: 180 -> 180 -> 80: partial k-groups and column tiles), a 200-wide PHM-5
    # head, 32-row workgroups above 1280 rows, one partial workgroup, fewer rows than one MFMA block
    (8, 256, (256, 128), 1, "relu", 1024), (4, 256, (256, 128), 2, "swish", 777), (4, 180, (180, 80), 1, "relu", 1000),
    (5, 200, (200, 100), 4, "relu", 513), (4, 128, (128, 32), 1, "relu", 1500), (4, 128, (256, 16), 1, "elu", 2048),
    (4, 128, (128, 32), 1, "relu", 3), (4, 16, (16, 16), 1, "relu", 70)])
def test_fused_head_matches_oracle(n, d_in, hidden, target, act, B):
    from phc_gnn_amd import lib
    if not lib.load().phc_downstream_supported_k4(B, d_in, hidden[0], hidden[1], n * target, target):
        pytest.skip("shape outside the fused head")
    dn = _make(n, d_in, hidden, target, act, seed=7)
    g = torch.Generator().manual_seed(8)
    x, dy = torch.randn(B, d_in, generator=g), torch.randn(B, target, generator=g)
    ref_y, ref_dx, ref_g, ref_state = _oracle(dn, x, dy, n, act)
    dn.to(DEV)
    y, dx = _run_fused(dn, x, dy)
    tol = 3e-5
    assert_close(y, ref_y, tol, "logits")
    assert_close(dx, ref_dx, tol, "dx")
    checked = 0
    for k, p in dn.named_parameters():
        r = ref_g[k]
        if r is None or k in ("affine.0.b", "affine.1.b"):            # bias in front of a norm: zero gradient up to round-off
            assert p.grad is None or float(p.grad.abs().max()) < 1e-3, k
            continue
        assert_close(p.grad, r, tol, "grad " + k)
        checked += 1
    assert checked >= 12, checked
    for k, b in dn.named_buffers():
        if b.is_floating_point():
            assert_close(b, ref_state["q." + k], tol, k)
        else:
            assert int(b) == 1, k


@pytest.mark.parametrize("n,d_in,hidden,B,valid", [(4, 128, (128, 32), 544, 512), (8, 256, (256, 128), 1088, 1024),
                                                   (4, 180, (180, 80), 1200, 1111), (4, 128, (128, 32), 64, 17)])
def test_fused_head_padding_rows_are_inert_and_launches_repeat(n, d_in, hidden, B, valid):
    """Rows >= n_valid (padding graphs of a fixed-shape batch) stay out of the statistics and get zero gradient; ten
    back-to-back launches (generation / arrival counters run on) give bit-identical results."""
    from phc_gnn_amd import lib
    target, act = 1, "relu"
    if not lib.load().phc_downstream_supported_k4(B, d_in, hidden[0], hidden[1], n * target, target):
        pytest.skip("shape outside the fused head")
    g = torch.Generator().manual_seed(18)
    x, dy = torch.randn(B, d_in, generator=g), torch.randn(B, target, generator=g)
    dy[valid:] = 0.0                                                   # NaN-label rows: the loss hands back zero
    dn = _make(n, d_in, hidden, target, act, seed=17)
    ref_y, ref_dx, ref_g, _ = _oracle(dn, x, dy, n, act, rows=valid)
    sd = {k: v.clone() for k, v in dn.state_dict().items()}
    dn.to(DEV)
    nv = torch.tensor(valid, dtype=torch.int32, device=DEV)
    outs = []
    for it in range(10):
        dn.load_state_dict(sd)
        for p in dn.parameters():
            p.grad = None
        y, dx = _run_fused(dn, x, dy, n_valid=nv)
        outs.append((y.clone(), dx.clone(), {k: p.grad.clone() for k, p in dn.named_parameters() if p.grad is not None}))
    torch.cuda.synchronize()
    y, dx, grads = outs[0]
    assert_close(y[:valid], ref_y, 3e-5, "logits of the real rows")
    assert_close(dx[:valid], ref_dx[:valid], 3e-5, "dx")
    assert float(dx[valid:].abs().max()) == 0.0
    for k, r in ref_g.items():
        if r is not None and k not in ("affine.0.b", "affine.1.b"):
            assert_close(grads[k], r, 3e-5, "grad " + k)
    for y2, dx2, g2 in outs[1:]:
        assert torch.equal(y2, y) and torch.equal(dx2, dx)
        assert all(torch.equal(g2[k], grads[k]) for k in grads)


@pytest.mark.parametrize("n,d_in,hidden,act,B,p,same", [
    (4, 200, (128, 32), "relu", 544, (0.3, 0.1), False),       # the published ogbg-molhiv head (benchmarks/README.md: d_units 128, 32; dropout_dn 0.3, 0.1)
    (4, 180, (180, 80), "swish", 1056, (0.2, 0.1), False),     # run_script_zinc_phm4.sh's head at the graph-replay batch
    (4, 128, (128, 32), "swish", 100, (0.5, 0.25), True),      # masks shared by the components
    (8, 256, (256, 128), "relu", 1056, (0.2, 0.1), True),
    (4, 128, (128, 32), "swish", 1500, (0.0, 0.4), False),     # 32-row workgroups, dropout behind the second activation only
])
def test_fused_head_with_dropout_matches_the_per_layer_path(n, d_in, hidden, act, B, p, same):
    """phm_dropout behind the head's activations (downstream.py:104-107) inside the one-launch head: the masks are the stand-alone fused
    passes' (a function of seed, step, site, row and column group), so the same seed gives the same function — logits, dx and every
    parameter gradient against the layer-by-layer path with its `phc_bn_fwd` dropout."""
    from phc_gnn_amd import lib, ops
    if not lib.load().phc_downstream_dropout_supported(B, d_in, hidden[0], hidden[1], n * 1, 1):
        pytest.skip("shape outside the 16-row fused head")
    g = torch.Generator().manual_seed(38)
    x, dy = torch.randn(B, d_in, generator=g), torch.randn(B, 1, generator=g)
    res = []
    for fused in (True, False):
        dn = _make(n, d_in, hidden, 1, act, seed=37)
        dn.dropout, dn.same_dropout = list(p), same
        dn.to(DEV)
        keep = ops.HEAD_DROPOUT
        ops.HEAD_DROPOUT = fused
        try:
            ops.rng_seed(torch.device(DEV), 1234)
            xd = x.to(DEV).requires_grad_()
            with ops.prebuilt_H(ops.phm_pairs(dn), ops.dense_weights(dn)):
                y = dn(xd)
                assert (type(y.grad_fn).__name__ == "_DownstreamFnBackward") == fused, type(y.grad_fn).__name__
                y.backward(dy.to(DEV))
        finally:
            ops.HEAD_DROPOUT = keep
        res.append((y.detach(), xd.grad, {k: q.grad for k, q in dn.named_parameters() if q.grad is not None},
                    {k: b.clone() for k, b in dn.named_buffers() if b.is_floating_point()}))
    tol = 3e-5 if act != "relu" else 2e-4
    assert_close(res[0][0], res[1][0], tol, "logits")
    assert_close(res[0][1], res[1][1], tol, "dx")
    # the masks really dropped something: a p-fraction of the second layer's input is zero on both paths alike
    assert float((res[0][1] == 0).float().mean()) < 0.5
    for k in res[1][2]:
        if k not in ("affine.0.b", "affine.1.b"):
            assert_close(res[0][2][k], res[1][2][k], tol, "grad " + k)
    for k in res[1][3]:
        assert_close(res[0][3][k], res[1][3][k], tol, k)
