"""Batch-norm statistics through integer-atomic accumulators (include/phc_hip.h "ACCUMULATOR FORM", csrc/phc_common.h):
the training step inside ``ops.stat_accumulators`` against the same step on partial rows + merge launches, bit-level
reproducibility of the accumulated totals, and the raw entry points against a float64 reference."""
import numpy as np
import pytest
import torch

from tests._util import assert_close, relerr

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _step_grads(wl, use_acc, seed=5):
    from phc_gnn_amd import harness, ops
    model = harness.build_model(wl, DEV, seed=1)
    model.train()
    data = harness.to_device(harness.make_batch(wl, seed=seed), DEV)
    old = ops.STAT_ACC
    ops.STAT_ACC = use_acc
    try:
        loss = harness.train_step(model, data, None, harness.StepConfig(optimizer=False))
    finally:
        ops.STAT_ACC = old
    torch.cuda.synchronize()
    grads = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    stats = {k: v.clone() for k, v in model.state_dict().items() if "running_" in k}
    return float(loss), grads, stats


@pytest.mark.parametrize("wl", ["molhiv_b32", "molhiv_b512", "molhiv_b32_swish", "molhiv_b512_swish", "zinc_pub_phc5_4x200_nodrop_swish",
                                "zinc_pub_phc4_4x180_nodrop_swish", "molhiv_pub_phc4_2x200_nodrop_swish"])
def test_training_step_with_accumulators_equals_the_partial_row_path(wl):
    """Same loss, gradients and running statistics with and without the accumulators (only the summation order of the
    statistics differs: fp32 per workgroup, then exact integer totals instead of a fixed-order Chan merge)."""
    l1, g1, s1 = _step_grads(wl, True)
    l0, g0, s0 = _step_grads(wl, False)
    assert abs(l1 - l0) <= 2e-6 * max(1.0, abs(l0))
    assert g1.keys() == g0.keys() and len(g1) > 50
    # (a bias in front of a batch-norm has an exactly-zero gradient: what is left there is round-off noise, skipped by name)
    keep = [k for k in g0 if not (k.endswith(".b") and k not in ("pooling.linear.b", "downstream.affine.2.b"))
            and float(g0[k].abs().max()) > 0]
    worst, worst_key = max((relerr(g1[k], g0[k]), k) for k in keep)
    # fp32 noise through 12 batch-norms; with ReLU a pre-activation within 1e-7 of zero may flip (one flip: 3e-3 .. 1e-2 on the
    # tensors behind it; a wrong statistic shows as 1e-1) — the smooth twins are the tight check of the same pipelines
    assert worst < (5e-5 if wl.endswith("_swish") else 3e-2), (worst, worst_key)
    for k in s0:
        assert_close(s1[k], s0[k], 1e-5, k)


def test_accumulator_steps_are_bit_reproducible():
    """Integer atomics commute: two runs of the same step give identical bits although workgroups arrive in any order."""
    a = _step_grads("molhiv_b512", True)
    b = _step_grads("molhiv_b512", True)
    assert a[0] == b[0]
    assert all(torch.equal(a[1][k], b[1][k]) for k in a[1])
    assert all(torch.equal(a[2][k], b[2][k]) for k in a[2])


@pytest.mark.parametrize("N,h,scale", [(1000, 128, 1.0), (12874, 128, 30.0), (70001, 256, 1e-3), (33, 32, 1.0),
                                       (2500, 200, 1.0), (900, 180, 3.0), (640, 72, 1.0)])      # partial last 32-column block
def test_statistics_gemm_into_an_accumulator_matches_float64(N, h, scale):
    """phc_gemm_prebuilt_stats (accumulator form) -> phc_bn_fwd_from_partials (accumulator form): the normalised output and the
    saved statistics against float64, incl. a column with a large mean (cancellation in E[x^2] - E[x]^2 is done in double)."""
    from phc_gnn_amd import lib as L, ops
    import ctypes as C
    lib = L.load()
    g = torch.Generator().manual_seed(N)
    x = torch.randn(N, h, generator=g)
    M = torch.randn(h, h, generator=g) / h ** 0.5 * scale
    b = torch.randn(h, generator=g) * scale
    b[0] += 300.0 * scale                                    # |mean| / std ~ 300 in column 0
    xd, Md, bd = x.to(DEV), M.to(DEV).contiguous(), b.to(DEV)
    y = torch.empty(N, h, device=DEV)
    acc = torch.zeros(lib.phc_stat_acc_words(h), dtype=torch.int64, device=DEV)
    sp = L.stream_ptr(torch.device(DEV))
    L.check(lib.phc_gemm_prebuilt_stats(L.ptr(xd), L.ptr(Md), None, L.ptr(bd), N, h, h, L.ptr(y), None, L.ptr(acc), None, None, sp), "gemm")
    out = torch.empty_like(y)
    stat = torch.empty(6 * h, device=DEV)
    L.check(lib.phc_bn_fwd_from_partials(L.ptr(y), L.ptr(acc), None, 0, None, None, None, None, None, 4, None, N, h, 0.1, 1e-5, 0,
                                         None, L.ptr(out), L.ptr(stat), 0.0, 0, 0, None, sp), "bn")
    torch.cuda.synchronize()
    y64 = x.double() @ M.double() + b.double()
    mean = y64.mean(0)
    var = y64.var(0, unbiased=False)
    ref = (y64 - mean) / torch.sqrt(var + 1e-5)
    assert_close(stat[:h], mean.float(), 2e-6, "mean")
    assert_close(stat[h:2 * h], (1.0 / torch.sqrt(var + 1e-5)).float(), 2e-5, "invstd")
    # tolerance: y itself is fp32 — a column with |mean| / std ~ 300 keeps only ~4.5 digits of its deviations
    assert_close(out[:, 1:], ref[:, 1:].float(), 2e-5, "normalised output")
    assert_close(out[:, 0], ref[:, 0].float(), 5e-3, "normalised output, offset column")


@pytest.mark.parametrize("N,h", [(20001, 128), (70001, 256), (513, 128)])
def test_accumulator_read_back_and_row_sharded_norm_backward(N, h):
    """The two entry points the row-sharded norm adds to the accumulator protocol (dist._ShardedGINEFn), through the C ABI:
    phc_stat_acc_read turns a statistics epilogue's totals into the (sum, M2) pair phc_bn_moments produces (float64 reference),
    and phc_gemm_bwd_apply with phc_gemm_ba_desc.count = rows of ALL shards, run shard by shard on accumulators that were added
    as integers (what the all-reduce does), gives the rows of ONE launch over all rows — dz and dz . H^T to fp32 round-off."""
    from phc_gnn_amd import ops
    g = torch.Generator().manual_seed(N + h)
    n, k = 4, h // 4
    A = torch.randn(n, n, n, generator=g).to(DEV)
    W = (torch.randn(n, k, k, generator=g) * 0.1).to(DEV)
    b = torch.randn(h, generator=g).to(DEV)
    pre = ops.phm_build_operands([(A, W)])[0]
    x = torch.randn(N, h, generator=g).to(DEV)
    # forward statistics: epilogue totals -> (sum, M2)
    y, acc = ops.gemm_pre_stats(x, pre, b)
    mom = ops.stat_acc_read(acc, h, N).cpu().double()
    y64 = y.cpu().double()
    assert_close(mom[:h], y64.sum(0), 2e-6, "sum")
    assert_close(mom[h:], ((y64 - y64.mean(0)) ** 2).sum(0), 2e-5, "M2")
    tot = ops.stat_acc_read(acc, h, 0).cpu().double()
    assert_close(tot[h:], (y64 ** 2).sum(0), 2e-5, "sum of squares (count = 0: raw totals)")
    # backward: one launch over all rows against two "ranks" with added accumulators and count = N
    dy = torch.randn(N, h, generator=g).to(DEV)
    mean, var = y64.mean(0), y64.var(0, unbiased=False)
    w = (0.5 + torch.rand(h, generator=g)).double()
    bb = (0.1 * torch.randn(h, generator=g)).double()
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    stat = torch.cat([mean, invstd, invstd * w, bb - mean * invstd * w, w, bb]).float().to(DEV)
    for act in (1, 4):                                           # relu, swish (include/phc_hip.h PHC_ACT_*)
        acc_all = ops.bn_bwd_sums_acc(dy, y, stat, act)
        dz_all, ds_all = ops.gemm_pre_bwd_apply(dy, y, stat, acc_all, 0, act, pre)
        cut = N // 2 + 1
        parts = [(0, cut), (cut, N)]
        accs = [ops.bn_bwd_sums_acc(dy[lo:hi], y[lo:hi], stat, act) for lo, hi in parts]
        # integer words: exactly what an all-reduce(SUM) leaves on every rank, in any reduction order (the shards' own fp32
        # partial sums differ from the single launch's, so the totals agree to fp32 round-off, not bit for bit)
        summed = accs[0] + accs[1]
        assert torch.equal(summed, accs[1] + accs[0])
        assert_close(ops.stat_acc_read(summed, h, 0), ops.stat_acc_read(acc_all, h, 0), 2e-6, f"summed totals act {act}")
        for lo, hi in parts:
            dz, ds = ops.gemm_pre_bwd_apply(dy[lo:hi].contiguous(), y[lo:hi].contiguous(), stat, summed, N, act, pre)
            assert_close(dz, dz_all[lo:hi], 2e-6, f"dz rows {lo}:{hi} act {act}")
            assert_close(ds, ds_all[lo:hi], 2e-6, f"ds rows {lo}:{hi} act {act}")
        # and against the stand-alone passes (phc_bn_bwd_sums / phc_bn_bwd_apply) + plain dx GEMM
        dwb = ops.bn_bwd_sums(dy, y, stat, act)
        dz_ref = ops.bn_bwd_apply(dy, y, stat, dwb, N, act)
        assert_close(dz_all, dz_ref, 2e-5, f"dz vs stand-alone passes act {act}")
        assert_close(ds_all, ops.phm_linear_pre(dz_ref, pre, None, dx=True), 2e-5, f"ds vs stand-alone passes act {act}")
