"""The MLP + norm node's forward as ONE launch (phc_mlp_norm_fwd_fused, csrc/mlp_fused.hip) against the two launches it replaces
(phc_gemm_prebuilt_stats / phc_gemm_aggregate_stats + phc_gemm_prebuilt_xf_acc): layers.py:349-355 in training mode."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0") if torch.cuda.is_available() else None


def _step(workload, fused, monkeypatch, graph, seed=5, agg=True):
    from phc_gnn_amd import graphed, harness, lib, ops
    from phc_gnn_amd.optim import ClipAdam
    monkeypatch.setattr(ops, "MLP_FWD_FUSED", fused)
    monkeypatch.setattr(ops, "MLP_FWD_FUSED_AGG", agg)
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
    bufs = {k: v.detach().clone() for k, v in model.state_dict().items() if "running" in k or "num_batches" in k}
    return float(loss.detach()), {k: p.grad.detach().clone() for k, p in model.named_parameters()}, bufs


@pytest.mark.parametrize("workload", ["molhiv_b32", "molhiv_b512", "molhiv_b512_swish", "molhiv_b512_softmax"])
@pytest.mark.parametrize("graph", [False, True])
def test_fused_forward_equals_the_two_launches(workload, graph, monkeypatch):
    la, ga, ba = _step(workload, True, monkeypatch, graph)
    lb, gb_, bb = _step(workload, False, monkeypatch, graph)
    assert abs(la - lb) <= 2e-6 * abs(lb), (la, lb)
    for k in bb:                                              # running statistics, step counters
        if bb[k].dtype.is_floating_point:
            torch.testing.assert_close(ba[k], bb[k], rtol=2e-6, atol=3e-7)
        else:
            assert torch.equal(ba[k], bb[k]), k
    smooth = "swish" in workload
    for k in gb_:
        ref = float(gb_[k].abs().max())
        if ref < 1e-7:
            continue
        err = float((ga[k] - gb_[k]).abs().max()) / ref
        # (ReLU: a statistics total that moves in the last bit may flip a gate on the kink: a few elements of a few gradients)
        assert err <= (1e-5 if smooth else 2e-3), (k, err)


@pytest.mark.parametrize("workload", ["molhiv_b32", "molhiv_b512", "molhiv_b512_swish"])
@pytest.mark.parametrize("graph", [False, True])
def test_aggregation_inside_the_fused_forward_is_bit_identical_to_its_own_launch(workload, graph, monkeypatch):
    """Same edge order, same additions, same normalisation of the gathered rows: with the aggregation inside the launch the whole step
    — loss, every gradient, the running statistics — equals the step with the row walker as its own launch bit for bit."""
    la, ga, ba = _step(workload, True, monkeypatch, graph, agg=True)
    lb, gb_, bb = _step(workload, True, monkeypatch, graph, agg=False)
    assert la == lb
    for k in bb:
        assert torch.equal(ba[k], bb[k]), k
    for k in gb_:
        assert torch.equal(ga[k], gb_[k]), k


@pytest.mark.parametrize("workload", ["molhiv_b32_softmax", "molhiv_b512_softmax", "molhiv_b32_softmax_swish"])
@pytest.mark.parametrize("graph", [False, True])
def test_softmax_aggregation_inside_the_fused_forward_is_bit_identical_to_its_own_launch(workload, graph, monkeypatch):
    """The reference's default molhiv aggregation (messagepassing.py:211-214, 297-300) as the gather stage of the one-launch forward
    (mlp_fused.hip AGG = 2): the same online softmax in the same edge order, the same table-row sums — the whole step equals the step
    with phc_aggregate_softmax_norm_enc_fwd as its own launch bit for bit (loss, every gradient, the running statistics)."""
    from phc_gnn_amd import harness, lib, ops
    base = "molhiv_b512" if "b512" in workload else "molhiv_b32"
    extra = dict(msg_aggr="softmax", **({"activation": "swish"} if workload.endswith("swish") else {}))
    monkeypatch.setitem(harness.WORKLOADS, workload, dict(harness.WORKLOADS[base], **extra))
    h = lib.load()
    real = h.phc_mlp_norm_fwd_fused
    seen = {"softmax": 0, "other": 0}

    def wrapped(desc, stream):
        d = desc._obj if hasattr(desc, "_obj") else desc.contents
        seen["softmax" if d.softmax_beta else "other"] += 1
        return real(desc, stream)
    monkeypatch.setattr(h, "phc_mlp_norm_fwd_fused", wrapped)
    monkeypatch.setattr(ops, "SOFTMAX_INTO_MLP", True)
    la, ga, ba = _step(workload, True, monkeypatch, graph)
    assert seen["softmax"] >= 4 and seen["other"] == 0, seen  # every layer's walk ran inside its MLP launch
    seen["softmax"] = 0
    monkeypatch.setattr(ops, "SOFTMAX_INTO_MLP", False)
    lb, gb_, bb = _step(workload, True, monkeypatch, graph)
    assert seen["softmax"] == 0 and seen["other"] >= 4, seen
    assert la == lb
    for k in bb:
        assert torch.equal(ba[k], bb[k]), k
    for k in gb_:
        assert torch.equal(ga[k], gb_[k]), (k, float((ga[k] - gb_[k]).abs().max()))


def test_fused_forward_op_level():
    """z1, y2, save_stat and the accumulator totals of y2 against the two launches on the same operands (z1, y2 bit for bit)."""
    from phc_gnn_amd import lib as L, ops
    lib = L.load()
    N, h, n = 12874, 128, 4
    g = torch.Generator(device=DEV).manual_seed(0)
    A = [torch.randn(n, n, n, generator=g, device=DEV) for _ in range(2)]
    W = [0.1 * torch.randn(n, h // n, h // n, generator=g, device=DEV) for _ in range(2)]
    b = [0.1 * torch.randn(h, generator=g, device=DEV) for _ in range(2)]
    pre = ops.phm_build_operands([(A[0], W[0]), (A[1], W[1])])
    s = torch.randn(N, h, generator=g, device=DEV)
    nv = torch.tensor(N - 37, dtype=torch.int32, device=DEV)
    words = lib.phc_stat_acc_words(h)
    sp = L.stream_ptr(DEV)
    act = 1

    def norm_tensors():
        gg = torch.Generator(device=DEV).manual_seed(5)
        w = [0.5 + torch.rand(h // n, generator=gg, device=DEV) for _ in range(n)]
        bb = [0.1 * torch.randn(h // n, generator=gg, device=DEV) for _ in range(n)]
        rm = [torch.zeros(h // n, device=DEV) for _ in range(n)]
        rv = [torch.ones(h // n, device=DEV) for _ in range(n)]
        nbt = [torch.zeros((), dtype=torch.int64, device=DEV) for _ in range(n)]
        return w, bb, rm, rv, nbt

    outs = {}
    for fused in (True, False):
        w, bb, rm, rv, nbt = norm_tensors()
        z1, y2 = torch.empty(N, h, device=DEV), torch.empty(N, h, device=DEV)
        acc1, acc2 = torch.zeros(words, dtype=torch.int64, device=DEV), torch.zeros(words, dtype=torch.int64, device=DEV)
        save = torch.empty(6 * h, device=DEV)
        arrs = [ops._ptr_array(t) for t in (w, bb, rm, rv, nbt)]
        cast = lambda a: C.cast(a, C.c_void_p)
        if fused:
            bar = torch.zeros(lib.phc_mlp_norm_bwd_fused_barrier_words(), dtype=torch.int32, device=DEV)
            fd = L.MlpNormFwdDesc()
            fd.s, fd.H1k4, fd.b1, fd.H2k4, fd.b2 = L.ptr(s), L.ptr(pre[0][2]), L.ptr(b[0]), L.ptr(pre[1][2]), L.ptr(b[1])
            fd.z1, fd.y2, fd.acc1, fd.acc2 = L.ptr(z1), L.ptr(y2), L.ptr(acc1), L.ptr(acc2)
            fd.weight, fd.bias, fd.running_mean, fd.running_var, fd.num_batches_tracked = (cast(a) for a in arrs)
            fd.n_comp, fd.momentum, fd.eps, fd.save_stat, fd.act = n, 0.1, 1e-5, L.ptr(save), act
            fd.n_valid_dev, fd.N, fd.h, fd.barrier = L.ptr(nv), N, h, L.ptr(bar)
            L.check(lib.phc_mlp_norm_fwd_fused(C.byref(fd), sp), "fused")
        else:
            L.check(lib.phc_gemm_prebuilt_stats(L.ptr(s), L.ptr(pre[0][0]), L.ptr(pre[0][2]), L.ptr(b[0]), N, h, h, L.ptr(z1), L.ptr(nv),
                                                L.ptr(acc1), None, None, sp), "stats")
            L.check(lib.phc_gemm_prebuilt_xf_acc(L.ptr(z1), L.ptr(acc1), arrs[0], arrs[1], arrs[2], arrs[3], arrs[4], n, 0.1, 1e-5, L.ptr(save),
                                                 act, L.ptr(nv), L.ptr(pre[1][0]), L.ptr(pre[1][2]), L.ptr(b[1]), N, h, h, L.ptr(y2), L.ptr(acc2),
                                                 None, None, sp), "xf")
        torch.cuda.synchronize()
        tot = ops.stat_acc_read(acc2, h)
        outs[fused] = (z1, y2, save, tot, torch.cat(rm), torch.cat(rv), torch.stack(nbt))
    L.check_indices(DEV)
    a, bq = outs[True], outs[False]
    assert torch.equal(a[0], bq[0])                           # z1: same operands, same MFMA order
    torch.testing.assert_close(a[2], bq[2], rtol=1e-6, atol=1e-7)          # save_stat: totals agree to double round-off
    torch.testing.assert_close(a[1][:N - 37], bq[1][:N - 37], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(a[3], bq[3], rtol=1e-6, atol=1e-4)
    torch.testing.assert_close(a[4], bq[4], rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(a[5], bq[5], rtol=1e-6, atol=1e-7)
    assert torch.equal(a[6], bq[6])
