"""The downstream head at molpcba widths (256 -> 768 -> 256 -> 512 -> 128: BASELINE configs[3]; the published 7 x 512 model's
512 -> 768 -> 256 -> 512 -> 128) as ONE ``phc_head_chain`` launch each way (csrc/head_chain.hip) against the CPU oracle's restatement
of PHMDownstreamNet.forward + RealTransformer (downstream.py:90-120, layers.py:399-414): logits, dx, every parameter gradient,
running statistics; padded batches; repeated launches (generation / arrival counters run on) bit for bit."""
import pytest
import torch

from tests._util import assert_close
from tests.test_gpu_downstream import _make, _oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _supported(B, d_in, hidden, n, target, valid_max=0):
    import ctypes as C
    from phc_gnn_amd import lib
    d3 = n * target
    Ks, Ns = (C.c_int32 * 4)(d_in, hidden[0], hidden[1], d3), (C.c_int32 * 4)(hidden[0], hidden[1], d3, target)
    Kb, Nb = (C.c_int32 * 4)(target, d3, hidden[1], hidden[0]), (C.c_int32 * 4)(d3, hidden[1], hidden[0], d_in)
    L = lib.load()
    return bool(L.phc_head_chain_supported(B, valid_max, 4, Ks, Ns)) and bool(L.phc_head_chain_supported(B, valid_max, 4, Kb, Nb))


def _run_chain(dn, x, dy, n_valid=None, chain=True):
    from phc_gnn_amd import ops
    xd = x.to(DEV).requires_grad_()
    with ops.prebuilt_H(ops.phm_pairs(dn), ops.dense_weights(dn)):
        y = dn(xd, n_valid=n_valid)
        # the chain launch really ran (chain=False: a shape outside it must take the per-layer path, not the chain)
        assert (type(y.grad_fn).__name__ == "_HeadChainFnBackward") == chain, type(y.grad_fn).__name__
        y.backward(dy.to(DEV))
    return y.detach(), xd.grad


CASES = [
    (4, 256, (768, 256), 128, "relu", 1024),          # BASELINE configs[3]'s head at its per-GPU batch
    (4, 512, (768, 256), 128, "relu", 512),           # the published molpcba model (benchmarks/README.md: 7 x 512, d_units 768, 256)
    (4, 256, (768, 256), 128, "swish", 300),          # a partial last row block
    (4, 256, (768, 256), 128, "relu", 5),             # fewer rows than one MFMA block
    (8, 256, (512, 128), 16, "lrelu", 777),           # PHM-8, T = 16: a 16-column last stage leaves ONE column block of 512 columns —
                                                      # outside the chain kernel (HC_MAX_NC): the per-layer path must serve it
    (8, 256, (256, 128), 64, "lrelu", 777),           # PHM-8 inside the chain kernel: d3 = 512, T = 64 (four column blocks)
    (4, 128, (256, 64), 32, "elu", 130),              # d3 = 128, T = 32
    (2, 128, (768, 256), 128, "relu", 64),            # few row blocks: eight column blocks, 24 owned columns per workgroup
]


@pytest.mark.parametrize("n,d_in,hidden,target,act,B", CASES)
def test_head_chain_matches_oracle(n, d_in, hidden, target, act, B):
    chain = _supported(B, d_in, hidden, n, target)
    assert chain == ((n, target) != (8, 16)), "the chain kernel's range changed: update CASES"
    dn = _make(n, d_in, hidden, target, act, seed=7)
    g = torch.Generator().manual_seed(8)
    x, dy = torch.randn(B, d_in, generator=g), torch.randn(B, target, generator=g)
    ref_y, ref_dx, ref_g, ref_state = _oracle(dn, x, dy, n, act)
    dn.to(DEV)
    y, dx = _run_chain(dn, x, dy, chain=chain)
    tol = 3e-5
    assert_close(y, ref_y, tol, "logits")
    assert_close(dx, ref_dx, tol, "dx")
    checked = 0
    for k, p in dn.named_parameters():
        r = ref_g[k]
        if r is None or k in ("affine.0.b", "affine.1.b"):            # bias in front of a norm: zero gradient up to round-off
            assert p.grad is None or float(p.grad.abs().max()) < 2e-3, k
            continue
        assert_close(p.grad, r, tol, "grad " + k)
        checked += 1
    assert checked >= 12, checked
    for k, b in dn.named_buffers():
        if b.is_floating_point():
            assert_close(b, ref_state["q." + k], tol, k)
        else:
            assert int(b) == 1, k


@pytest.mark.parametrize("n,d_in,hidden,target,B,valid", [(4, 256, (768, 256), 128, 1024, 1000), (4, 256, (768, 256), 128, 560, 512),
                                                          (4, 128, (256, 64), 32, 64, 17),
                                                          # the graph-replay batch of configs[3]: 1024 graphs + 32 padding graphs = 66 row
                                                          # blocks x 4 column blocks on 256 CUs (the last two row blocks exit at once)
                                                          (4, 256, (768, 256), 128, 1056, 1024)])
def test_head_chain_padding_rows_are_inert_and_launches_repeat(n, d_in, hidden, target, B, valid):
    """Rows >= n_valid (padding graphs of a fixed-shape batch) stay out of the statistics and get zero gradient; ten back-to-back
    launches give bit-identical results."""
    act = "relu"
    if not _supported(B, d_in, hidden, n, target, valid_max=valid):
        pytest.skip("shape outside the chain kernel on this device")
    g = torch.Generator().manual_seed(18)
    x, dy = torch.randn(B, d_in, generator=g), torch.randn(B, target, generator=g)
    dy[valid:] = 0.0                                                   # NaN-label rows: the loss hands back zero
    dn = _make(n, d_in, hidden, target, act, seed=17)
    ref_y, ref_dx, ref_g, _ = _oracle(dn, x, dy, n, act, rows=valid)
    sd = {k: v.clone() for k, v in dn.state_dict().items()}
    dn.to(DEV)
    nv = torch.tensor(valid, dtype=torch.int32, device=DEV)
    nv._phc_max_valid = valid                      # what a fixed-shape batch promises (graphed.GraphedTrainStep)
    outs = []
    for it in range(10):
        dn.load_state_dict(sd)
        for p in dn.parameters():
            p.grad = None
        y, dx = _run_chain(dn, x, dy, n_valid=nv)
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


def test_head_chain_switch_falls_back_to_the_per_layer_path():
    """PHC_HEAD_CHAIN's module switch off: the same head runs layer by layer and agrees with the chain launch."""
    from phc_gnn_amd import ops
    n, d_in, hidden, target, act, B = 4, 256, (768, 256), 128, "relu", 256
    if not _supported(B, d_in, hidden, n, target):
        pytest.skip("shape outside the chain kernel on this device")
    g = torch.Generator().manual_seed(28)
    x, dy = torch.randn(B, d_in, generator=g), torch.randn(B, target, generator=g)
    res = []
    for on in (True, False):
        dn = _make(n, d_in, hidden, target, act, seed=27).to(DEV)
        keep = ops.HEAD_CHAIN
        ops.HEAD_CHAIN = on
        try:
            xd = x.to(DEV).requires_grad_()
            with ops.prebuilt_H(ops.phm_pairs(dn), ops.dense_weights(dn)):
                y = dn(xd)
                assert (type(y.grad_fn).__name__ == "_HeadChainFnBackward") == on
                y.backward(dy.to(DEV))
        finally:
            ops.HEAD_CHAIN = keep
        res.append((y.detach(), xd.grad, {k: p.grad for k, p in dn.named_parameters() if p.grad is not None}))
    assert_close(res[0][0], res[1][0], 2e-5, "logits")
    assert_close(res[0][1], res[1][1], 2e-5, "dx")
    for k in res[1][2]:
        if k not in ("affine.0.b", "affine.1.b"):
            assert_close(res[0][2][k], res[1][2][k], 2e-5, "grad " + k)


@pytest.mark.parametrize("n,d_in,hidden,target,act,B,p,same", [
    (4, 512, (768, 256), 128, "relu", 544, (0.3, 0.2), False),     # the published ogbg-molpcba head with its dropout, graph-replay batch
    (4, 256, (768, 256), 128, "swish", 300, (0.25, 0.5), True),    # masks shared by the components
])
def test_head_chain_with_dropout_matches_the_per_layer_path(n, d_in, hidden, target, act, B, p, same):
    """phm_dropout behind the head's activations (downstream.py:104-107) inside the chain launches: the masks are the stand-alone fused
    passes' (seed, step, site, row, column group), so the same seed gives the same function on both paths."""
    from phc_gnn_amd import ops
    if not _supported(B, d_in, hidden, n, target):
        pytest.skip("shape outside the chain kernel on this device")
    g = torch.Generator().manual_seed(48)
    x, dy = torch.randn(B, d_in, generator=g), torch.randn(B, target, generator=g)
    res = []
    for on in (True, False):
        dn = _make(n, d_in, hidden, target, act, seed=47)
        dn.dropout, dn.same_dropout = list(p), same
        dn.to(DEV)
        keep = ops.HEAD_CHAIN
        ops.HEAD_CHAIN = on
        try:
            ops.rng_seed(torch.device(DEV), 4321)
            xd = x.to(DEV).requires_grad_()
            with ops.prebuilt_H(ops.phm_pairs(dn), ops.dense_weights(dn)):
                y = dn(xd)
                assert (type(y.grad_fn).__name__ == "_HeadChainFnBackward") == on, type(y.grad_fn).__name__
                y.backward(dy.to(DEV))
        finally:
            ops.HEAD_CHAIN = keep
        res.append((y.detach(), xd.grad, {k: q.grad for k, q in dn.named_parameters() if q.grad is not None}))
    tol = 3e-5 if act != "relu" else 2e-4
    assert_close(res[0][0], res[1][0], tol, "logits")
    assert_close(res[0][1], res[1][1], tol, "dx")
    for k in res[1][2]:
        if k not in ("affine.0.b", "affine.1.b"):
            assert_close(res[0][2][k], res[1][2][k], tol, "grad " + k)
