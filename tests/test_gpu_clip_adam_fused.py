"""The one-launch clip + Adam (phc_clip_adam_fused, ABI 8) against the two launches (phc_clip_adam_step) and against
clip_grad_norm_ + torch.optim.Adam — the pair at benchmarks/train_hiv.py:198-201."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0") if torch.cuda.is_available() else None


def _params(seed, shapes):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(s, generator=g).to(DEV).requires_grad_() for s in shapes]


SHAPES = [(4, 32, 32), (4, 4, 4), (128,), (119, 32), (5000,), (1,), (13, 7), (4, 64, 64)]


@pytest.mark.parametrize("max_norm", [2.0, 0.0, 1e-3])
def test_fused_step_is_bit_identical_to_the_two_launches(max_norm, monkeypatch):
    from phc_gnn_amd import lib, optim
    lib.load()
    runs = {}
    for fused in (True, False):
        monkeypatch.setattr(optim, "FUSED_STEP", fused)
        ps = _params(3, SHAPES)
        opt = optim.ClipAdam(ps, lr=1e-2)
        assert (opt._fused_ws is not None) == fused
        norms = []
        for step in range(7):
            g = torch.Generator().manual_seed(100 + step)
            for i, p in enumerate(ps):
                # tensor 2 never has a gradient, tensor 5 only on even steps: untouched / per-tensor step counters
                p.grad = None if (i == 2 or (i == 5 and step % 2)) else (3.0 * torch.randn(p.shape, generator=g)).to(DEV)
            norms.append(float(opt.step(max_norm=max_norm)))
        torch.cuda.synchronize()
        lib.check_indices(DEV)
        runs[fused] = (norms, [p.detach().clone() for p in ps], opt.exp_avg_flat.clone(), opt.exp_avg_sq_flat.clone(), opt.step_count.clone(),
                       [None if p.grad is None else p.grad.clone() for p in ps])
    a, b = runs[True], runs[False]
    assert a[0] == b[0]
    for x, y in zip(a[1], b[1]):
        assert torch.equal(x, y)
    assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])
    for x, y in zip(a[5], b[5]):
        assert (x is None and y is None) or torch.equal(x, y)             # clip_grad_norm_ scales the gradients in place
    assert a[4].tolist() == [7, 7, 0, 7, 7, 4, 7, 7]


def test_fused_step_tracks_torch_adam():
    from phc_gnn_amd import lib, optim
    lib.load()
    ps = _params(5, SHAPES)
    qs = [p.detach().clone().requires_grad_() for p in ps]
    opt = optim.ClipAdam(ps, lr=1e-2)
    assert opt._fused_ws is not None
    ref = torch.optim.Adam(qs, lr=1e-2)
    for step in range(5):
        g = torch.Generator().manual_seed(200 + step)
        for p, q in zip(ps, qs):
            gr = (2.0 * torch.randn(p.shape, generator=g)).to(DEV)
            p.grad, q.grad = gr.clone(), gr.clone()
        opt.step(max_norm=2.0)
        torch.nn.utils.clip_grad_norm_(qs, max_norm=2.0)
        ref.step()
    for p, q in zip(ps, qs):
        torch.testing.assert_close(p, q, rtol=2e-5, atol=2e-6)


def test_fused_step_replays_inside_a_graph():
    """Captured once, replayed many times: the launch generation lives in the workspace, every replay finds its own tag."""
    from phc_gnn_amd import lib, optim
    lib.load()
    ps = _params(7, SHAPES)
    qs = [p.detach().clone().requires_grad_() for p in ps]
    grads = [torch.randn_like(p) for p in ps]
    for p, q, g in zip(ps, qs, grads):
        p.grad, q.grad = g.clone(), g.clone()
    o1, o2 = optim.ClipAdam(ps, lr=1e-2, write_back_clipped=False), optim.ClipAdam(qs, lr=1e-2, write_back_clipped=False)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        o1.step(max_norm=2.0)                                   # warm-up outside the capture
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            o1.step(max_norm=2.0)
        for _ in range(20):
            gr.replay()
    s.synchronize()
    for _ in range(21):                                          # the warm-up step + 20 replays (a capture executes nothing)
        o2.step(max_norm=2.0)
    torch.cuda.synchronize()
    lib.check_indices(DEV)
    for p, q in zip(ps, qs):
        assert torch.equal(p, q)
