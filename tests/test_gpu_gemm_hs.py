"""The H-stationary streaming GEMM (csrc/gemm_hs_device.h) through the C ABI: every phc_gemm_prebuilt* entry point with the
operand in k4 form against float64 — y = x.H + b (+ residual) (layers.py:198-219), the batch-norm statistics of the output
(norm.py:30-35) in accumulator form, the un-applied-norm operand transform (layers.py:349-355) with its statistics taken from
save_stat or straight from an accumulator, and the norm backward's column sums out of the dz GEMM — at the widths the kernel
serves (K, Nout in {128, 256}) and row counts around its 16-row steps, its ring depth and its persistent loop."""
import ctypes as C

import pytest
import torch

from tests._util import assert_close


pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROWS = [1, 15, 16, 17, 63, 333, 4097, 12874, 70001]


def _rows(N, K, Nout):
    """At K = Nout = 128 the library serves fewer than 32 k rows with the LDS-staged kernel (csrc/phm_linear.hip: hs_shape_ok):
    those cases run at 32 k + N rows so that the kernel under test is the H-stationary one at every width."""
    return N + 32768 if (K == 128 and Nout == 128) else N


def k4(M: torch.Tensor) -> torch.Tensor:
    """[K, N] -> k4 form [K/4][N][4]: float4 slot (k / 4) * N + j holds rows 4 (k / 4) .. +3 of column j."""
    K, N = M.shape
    return M.view(K // 4, 4, N).permute(0, 2, 1).contiguous()


def acc_totals(acc: torch.Tensor, h: int):
    """(qty 0, qty 1) column totals of a statistics accumulator [REP][2][hi | lo][h] (csrc/phc_common.h)."""
    from phc_gnn_amd import lib as L
    rep = L.load().phc_stat_acc_replicas()
    w = acc.cpu().view(rep, 2, 2, h).to(torch.float64)
    tot = (w[:, :, 0, :] * 2.0 ** -12 + w[:, :, 1, :] * 2.0 ** -62).sum(0)
    return tot[0], tot[1]


def _problem(N, K, Nout, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, K, generator=g)
    M = torch.randn(K, Nout, generator=g) / K ** 0.5 * scale
    b = torch.randn(Nout, generator=g) * scale
    return x, M, b


@pytest.mark.parametrize("K,Nout", [(128, 128), (256, 256), (128, 256), (256, 128), (180, 180), (200, 200)])
@pytest.mark.parametrize("N", ROWS)
def test_hs_plain_bias_residual(N, K, Nout):
    from phc_gnn_amd import lib as L
    lib = L.load()
    N = _rows(N, K, Nout)
    x, M, b = _problem(N, K, Nout, seed=N + K)
    res = torch.randn(N, Nout, generator=torch.Generator().manual_seed(7))
    xd, Md, M4, bd, rd = x.to(DEV), M.to(DEV), k4(M).to(DEV), b.to(DEV), res.to(DEV)
    sp = L.stream_ptr(torch.device(DEV))
    ref = x.double() @ M.double() + b.double()
    for use_b, use_r in ((True, False), (False, True), (True, True), (False, False)):
        y = torch.full((N, Nout), float("nan"), device=DEV)
        L.check(lib.phc_gemm_prebuilt(L.ptr(xd), L.ptr(Md), L.ptr(M4), L.ptr(bd) if use_b else None, L.ptr(rd) if use_r else None, N, K, Nout,
                                      L.ptr(y), 0, sp), "gemm")
        want = x.double() @ M.double() + (b.double() if use_b else 0) + (res.double() if use_r else 0)
        assert_close(y, want.float(), 2e-6, f"hs plain N={N} K={K} Nout={Nout} bias={use_b} residual={use_r}")
    # same numbers as the LDS-staged kernel up to the summation order
    y0, y1 = torch.empty(N, Nout, device=DEV), torch.empty(N, Nout, device=DEV)
    L.check(lib.phc_gemm_prebuilt(L.ptr(xd), L.ptr(Md), None, L.ptr(bd), None, N, K, Nout, L.ptr(y0), 0, sp), "gemm old")
    L.check(lib.phc_gemm_prebuilt(L.ptr(xd), L.ptr(Md), L.ptr(M4), L.ptr(bd), None, N, K, Nout, L.ptr(y1), 0, sp), "gemm hs")
    assert_close(y1, y0, 2e-6, "hs vs staged kernel")
    assert_close(y1, ref.float(), 2e-6)


@pytest.mark.parametrize("K,Nout", [(128, 128), (256, 256), (128, 256), (180, 180), (200, 200)])
@pytest.mark.parametrize("N,valid", [(16, 16), (17, 9), (333, 333), (12874, 12001), (70001, 70001)])
def test_hs_statistics_accumulator(N, valid, K, Nout):
    """Output statistics in accumulator form, incl. padding rows past n_valid and a column whose mean is 300 standard deviations."""
    from phc_gnn_amd import lib as L
    lib = L.load()
    if K == 128 and Nout == 128:
        N, valid = N + 32768, valid + 32768
    x, M, b = _problem(N, K, Nout, seed=N * 3 + K)
    b[0] += 300.0
    xd, Md, M4, bd = x.to(DEV), M.to(DEV), k4(M).to(DEV), b.to(DEV)
    nv = torch.tensor([valid], dtype=torch.int32, device=DEV)
    acc = torch.zeros(lib.phc_stat_acc_words(Nout), dtype=torch.int64, device=DEV)
    y = torch.empty(N, Nout, device=DEV)
    sp = L.stream_ptr(torch.device(DEV))
    L.check(lib.phc_gemm_prebuilt_stats(L.ptr(xd), L.ptr(Md), L.ptr(M4), L.ptr(bd), N, K, Nout, L.ptr(y), L.ptr(nv), L.ptr(acc), None, None, sp),
            "stats gemm")
    torch.cuda.synchronize()
    y64 = x.double() @ M.double() + b.double()
    assert_close(y, y64.float(), 2e-6, "y")
    S, Q = acc_totals(acc, Nout)
    mean = y64[:valid].mean(0)
    var = y64[:valid].var(0, unbiased=False)
    got_mean = S / valid
    got_var = (Q / valid - got_mean ** 2).clamp_min(0)
    assert_close(got_mean.float(), mean.float(), 2e-6, "mean from the accumulator")
    # y is fp32: the offset column keeps ~4.5 digits of its deviations
    assert_close(got_var[1:].float(), var[1:].float(), 3e-5, "variance from the accumulator")
    assert abs(float(got_var[0]) - float(var[0])) <= 5e-3 * float(var[0])
    # reproducible bits: integer atomics commute
    acc2 = torch.zeros_like(acc)
    L.check(lib.phc_gemm_prebuilt_stats(L.ptr(xd), L.ptr(Md), L.ptr(M4), L.ptr(bd), N, K, Nout, L.ptr(y), L.ptr(nv), L.ptr(acc2), None, None, sp),
            "stats gemm again")
    assert torch.equal(acc, acc2)


@pytest.mark.parametrize("act", ["relu", "swish", "identity"])
@pytest.mark.parametrize("K,Nout", [(128, 128), (256, 256), (256, 128), (180, 180), (200, 200)])
@pytest.mark.parametrize("N,valid", [(17, 17), (333, 301), (12874, 12874)])
def test_hs_operand_transform_from_save_stat(N, valid, K, Nout, act):
    from oracle import phc_oracle as O
    from phc_gnn_amd import lib as L
    lib = L.load()
    if K == 128 and Nout == 128:
        N, valid = N + 32768, valid + 32768
    x, M, b = _problem(N, K, Nout, seed=N + 11)
    g = torch.Generator().manual_seed(5)
    scale, shift = 0.5 + torch.rand(K, generator=g), torch.randn(K, generator=g)
    stat = torch.cat([torch.zeros(2 * K), scale, shift, torch.zeros(2 * K)])
    xd, Md, M4, bd, sd = x.to(DEV), M.to(DEV), k4(M).to(DEV), b.to(DEV), stat.to(DEV)
    nv = torch.tensor([valid], dtype=torch.int32, device=DEV)
    sp = L.stream_ptr(torch.device(DEV))
    a = O.activation(act, x.double() * scale.double() + shift.double())
    a[valid:] = 0
    y64 = a @ M.double() + b.double()
    for with_stats in (False, True):
        y = torch.empty(N, Nout, device=DEV)
        acc = torch.zeros(lib.phc_stat_acc_words(Nout), dtype=torch.int64, device=DEV)
        L.check(lib.phc_gemm_prebuilt_xf(L.ptr(xd), L.ptr(sd), L.ACT_IDS[act], L.ptr(nv), L.ptr(Md), L.ptr(M4), L.ptr(bd), N, K, Nout, L.ptr(y),
                                         L.ptr(acc) if with_stats else None, None, None, sp), "xf gemm")
        torch.cuda.synchronize()
        assert_close(y, y64.float(), 3e-6, f"xf y (stats={with_stats})")
        if with_stats:
            S, Q = acc_totals(acc, Nout)
            assert_close((S / valid).float(), y64[:valid].mean(0).float(), 3e-6, "xf: mean of the output")
            assert_close((Q / valid - (S / valid) ** 2).float(), y64[:valid].var(0, unbiased=False).float(), 5e-5, "xf: variance of the output")


@pytest.mark.parametrize("K,Nout,n_comp", [(128, 128, 4), (256, 256, 8), (256, 128, 4), (180, 180, 4), (200, 200, 5)])
@pytest.mark.parametrize("N,valid", [(333, 301), (12874, 12874)])
def test_hs_operand_transform_from_an_accumulator(N, valid, K, Nout, n_comp):
    """linear1 (statistics into an accumulator) -> [norm -> relu -> linear2] with the norm's scale / shift derived in the second
    GEMM's prologue (phc_gemm_prebuilt_xf_acc): output, save_stat and the running statistics against float64."""
    from phc_gnn_amd import lib as L
    lib = L.load()
    if K == 128 and Nout == 128:
        N, valid = N + 32768, valid + 32768
    g = torch.Generator().manual_seed(N + K)
    x0 = torch.randn(N, K, generator=g)
    M1 = torch.randn(K, K, generator=g) / K ** 0.5
    b1 = torch.randn(K, generator=g)
    M2 = torch.randn(K, Nout, generator=g) / K ** 0.5
    b2 = torch.randn(Nout, generator=g)
    k = K // n_comp
    w = [(0.5 + torch.rand(k, generator=g)).to(DEV) for _ in range(n_comp)]
    bb = [(0.1 * torch.randn(k, generator=g)).to(DEV) for _ in range(n_comp)]
    rm = [torch.zeros(k, device=DEV) for _ in range(n_comp)]
    rv = [torch.ones(k, device=DEV) for _ in range(n_comp)]
    nbt = [torch.zeros((), dtype=torch.int64, device=DEV) for _ in range(n_comp)]
    arr = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    nv = torch.tensor([valid], dtype=torch.int32, device=DEV)
    sp = L.stream_ptr(torch.device(DEV))
    xd = x0.to(DEV)
    z = torch.empty(N, K, device=DEV)
    acc1 = torch.zeros(lib.phc_stat_acc_words(K), dtype=torch.int64, device=DEV)
    M1d, M2d = M1.to(DEV), M2.to(DEV)
    M1k, M2k = k4(M1).to(DEV), k4(M2).to(DEV)
    b1d, b2d = b1.to(DEV), b2.to(DEV)
    L.check(lib.phc_gemm_prebuilt_stats(L.ptr(xd), L.ptr(M1d), L.ptr(M1k), L.ptr(b1d), N, K, K, L.ptr(z), L.ptr(nv), L.ptr(acc1), None, None, sp), "g1")
    y = torch.empty(N, Nout, device=DEV)
    save = torch.empty(6 * K, device=DEV)
    L.check(lib.phc_gemm_prebuilt_xf_acc(L.ptr(z), L.ptr(acc1), arr(w), arr(bb), arr(rm), arr(rv), arr(nbt), n_comp, 0.1, 1e-5, L.ptr(save),
                                         L.ACT_IDS["relu"], L.ptr(nv), L.ptr(M2d), L.ptr(M2k), L.ptr(b2d), N, K, Nout, L.ptr(y), None, None,
                                         None, sp), "g2")
    torch.cuda.synchronize()
    z64 = x0.double() @ M1.double() + b1.double()
    mean, var = z64[:valid].mean(0), z64[:valid].var(0, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    wcat, bcat = torch.cat([t.cpu() for t in w]).double(), torch.cat([t.cpu() for t in bb]).double()
    a = torch.relu((z64 - mean) * invstd * wcat + bcat)
    a[valid:] = 0
    y64 = a @ M2.double() + b2.double()
    assert_close(y, y64.float(), 1e-5, "y through the accumulator-derived norm")
    assert_close(save[:K], mean.float(), 3e-6, "save_stat mean")
    assert_close(save[K:2 * K], invstd.float(), 2e-5, "save_stat invstd")
    assert_close(torch.cat(rm), (0.1 * mean).float(), 3e-6, "running mean")
    unb = var * valid / (valid - 1)
    assert_close(torch.cat(rv), (0.9 + 0.1 * unb).float(), 2e-5, "running var")
    assert all(int(t) == 1 for t in nbt)


@pytest.mark.parametrize("act", ["relu", "swish"])
@pytest.mark.parametrize("K,Nout", [(128, 128), (256, 256), (128, 256), (180, 180), (200, 200)])
@pytest.mark.parametrize("N,valid", [(17, 11), (333, 333), (12874, 12001)])
def test_hs_backward_sums(N, valid, K, Nout, act):
    """dz = dy . H^T plus the norm backward's column sums (sum g, sum g x_hat), accumulator form."""
    from oracle import phc_oracle as O
    from phc_gnn_amd import lib as L
    lib = L.load()
    if K == 128 and Nout == 128:
        N, valid = N + 32768, valid + 32768
    g = torch.Generator().manual_seed(N + Nout)
    dy = torch.randn(N, K, generator=g)
    M = torch.randn(K, Nout, generator=g) / K ** 0.5
    bn_x = torch.randn(N, Nout, generator=g) * 2.0 + 0.3
    mean, invstd = torch.randn(Nout, generator=g) * 0.1 + 0.3, 0.4 + torch.rand(Nout, generator=g)
    w, b = 0.5 + torch.rand(Nout, generator=g), 0.2 * torch.randn(Nout, generator=g)
    stat = torch.cat([mean, invstd, invstd * w, b - mean * invstd * w, w, b])
    dyd, Md, M4, xd, sd = dy.to(DEV), M.to(DEV), k4(M).to(DEV), bn_x.to(DEV), stat.to(DEV)
    nv = torch.tensor([valid], dtype=torch.int32, device=DEV)
    acc = torch.zeros(lib.phc_stat_acc_words(Nout), dtype=torch.int64, device=DEV)
    dz = torch.empty(N, Nout, device=DEV)
    sp = L.stream_ptr(torch.device(DEV))
    L.check(lib.phc_gemm_prebuilt_bwsums(L.ptr(dyd), L.ptr(Md), L.ptr(M4), N, K, Nout, L.ptr(dz), L.ptr(xd), L.ptr(sd), L.ACT_IDS[act], L.ptr(nv),
                                         L.ptr(acc), None, sp), "bwsums")
    torch.cuda.synchronize()
    dz64 = dy.double() @ M.double()
    assert_close(dz, dz64.float(), 2e-6, "dz")
    xh = (bn_x.double() - mean.double()) * invstd.double()
    pre = (xh * w.double() + b.double()).requires_grad_()
    (O.activation(act, pre)).backward(dz64)
    gq = pre.grad[:valid]
    sg, sgx = acc_totals(acc, Nout)
    assert_close(sg.float(), gq.sum(0).float(), 2e-5, "sum g")
    assert_close(sgx.float(), (gq * xh[:valid]).sum(0).float(), 2e-5, "sum g x_hat")


def test_k4_operands_from_the_batched_builder():
    """phc_phm_build_H_batched writes H, H^T and their k4 forms: the k4 forms are bit-exact re-orderings."""
    from phc_gnn_amd import lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(3)
    for n, kin, kout in [(4, 32, 32), (8, 32, 32), (4, 64, 32), (2, 64, 128)]:
        A = torch.randn(n, n, n, generator=g).to(DEV)
        W = torch.randn(n, kin, kout, generator=g).to(DEV)
        fi, fo = n * kin, n * kout
        H, HT = torch.empty(fi, fo, device=DEV), torch.empty(fo, fi, device=DEV)
        Hk4, HTk4 = torch.empty(fi * fo, device=DEV), torch.empty(fi * fo, device=DEV)
        arr = (L.HJob * 1)()
        arr[0] = L.HJob(L.ptr(A), L.ptr(W), L.ptr(H), L.ptr(HT), n, kin, kout, 0, L.ptr(Hk4), L.ptr(HTk4))
        L.check(lib.phc_phm_build_H_batched(arr, 1, L.stream_ptr(torch.device(DEV))), "build")
        torch.cuda.synchronize()
        assert torch.equal(HT, H.t())
        assert torch.equal(Hk4.view(fi // 4, fo, 4), k4(H))
        assert torch.equal(HTk4.view(fo // 4, fi, 4), k4(HT.contiguous()))


@pytest.mark.parametrize("act,act_in", [("relu", "relu"), ("swish", "identity")])
@pytest.mark.parametrize("K,Nout", [(128, 128), (256, 256), (256, 128), (180, 180), (200, 200)])
@pytest.mark.parametrize("N,valid", [(17, 11), (333, 333), (12874, 12001)])
def test_hs_norm_backward_folded_into_the_dx_gemm(N, valid, K, Nout, act, act_in):
    """phc_bn_bwd_partial_acc -> phc_gemm_bwd_apply (with the second-stage sums): dz (the norm backward's dx), dwb, y = dz . M
    and the next norm's column sums against float64 autograd of  act(batch_norm(x))  followed by the linear map."""
    from oracle import phc_oracle as O
    from phc_gnn_amd import lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(N + K + Nout)
    x = torch.randn(N, K, generator=g) * 1.5 + 0.2                       # the norm's input
    dout = torch.randn(N, K, generator=g)
    M = torch.randn(K, Nout, generator=g) / K ** 0.5
    w, b = 0.5 + torch.rand(K, generator=g), 0.2 * torch.randn(K, generator=g)
    # float64 reference of the norm (training statistics over the valid rows) and its backward
    x64 = x.double().requires_grad_()
    mean, var = x64[:valid].mean(0), x64[:valid].var(0, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    xh = (x64 - mean) * invstd
    out = O.activation(act, xh * w.double() + b.double())
    w64, b64 = w.double().requires_grad_(), b.double().requires_grad_()
    out = O.activation(act, xh * w64 + b64)
    out[:valid].backward(dout.double()[:valid])
    dz64 = x64.grad.clone()
    dz64[valid:] = 0
    stat = torch.cat([mean.detach(), invstd.detach(), invstd.detach() * w.double(), b.double() - mean.detach() * invstd.detach() * w.double(),
                      w.double(), b.double()]).float()
    # second stage: y is the gradient of act_in(batch_norm(bw_x)) — its column sums
    bw_x = torch.randn(N, Nout, generator=g) + 0.1
    m2, i2 = 0.1 * torch.randn(Nout, generator=g), 0.5 + torch.rand(Nout, generator=g)
    w2, b2 = 0.5 + torch.rand(Nout, generator=g), 0.1 * torch.randn(Nout, generator=g)
    stat2 = torch.cat([m2, i2, i2 * w2, b2 - m2 * i2 * w2, w2, b2])
    y64 = dz64 @ M.double()
    xh2 = (bw_x.double() - m2.double()) * i2.double()
    pre2 = (xh2 * w2.double() + b2.double()).requires_grad_()
    O.activation(act_in, pre2).backward(y64)
    g2 = pre2.grad[:valid]

    sp = L.stream_ptr(torch.device(DEV))
    dev = lambda t: t.to(DEV).contiguous()
    xd, dd, M4, sd, bxd, s2d = dev(x), dev(dout), dev(k4(M)), dev(stat), dev(bw_x), dev(stat2)
    nv = torch.tensor([valid], dtype=torch.int32, device=DEV)
    acc = torch.zeros(lib.phc_stat_acc_words(K), dtype=torch.int64, device=DEV)
    acc2 = torch.zeros(lib.phc_stat_acc_words(Nout), dtype=torch.int64, device=DEV)
    L.check(lib.phc_bn_bwd_partial_acc(L.ptr(dd), L.ptr(xd), L.ptr(sd), L.ptr(nv), N, K, L.ACT_IDS[act], L.ptr(acc), sp), "partial")
    dz = torch.empty(N, K, device=DEV)
    dwb = torch.empty(2 * K, device=DEV)
    y = torch.empty(N, Nout, device=DEV)
    d = L.GemmBaDesc()
    d.dout, d.bn_x, d.bn_stat, d.sums_acc, d.n_valid_dev = L.ptr(dd), L.ptr(xd), L.ptr(sd), L.ptr(acc), L.ptr(nv)
    d.dz, d.dwb, d.Mk4, d.y = L.ptr(dz), L.ptr(dwb), L.ptr(M4), L.ptr(y)
    d.N, d.K, d.Nout, d.act = N, K, Nout, L.ACT_IDS[act]
    d.bw_act, d.bw_x, d.bw_stat, d.bw_sums_acc = L.ACT_IDS[act_in], L.ptr(bxd), L.ptr(s2d), L.ptr(acc2)
    L.check(lib.phc_gemm_bwd_apply(C.byref(d), sp), "bwd_apply")
    torch.cuda.synchronize()
    assert_close(dz, dz64.float(), 1e-5, "dz (norm backward applied inside the GEMM)")
    assert_close(dwb[:K], w64.grad.float(), 2e-5, "dweight")
    assert_close(dwb[K:], b64.grad.float(), 2e-5, "dbias")
    assert_close(y, y64.float(), 1e-5, "y = dz . M")
    sg, sgx = acc_totals(acc2, Nout)
    # (with the identity activation sum g is exactly zero in exact arithmetic — the rows of a batch-norm's dx sum to zero —
    # so the yardstick is the sum of magnitudes, not the result)
    scale = float(g2.abs().sum(0).max())
    assert float((sg - g2.sum(0)).abs().max()) <= 3e-6 * scale, "second stage: sum g"
    assert float((sgx - (g2 * xh2[:valid]).sum(0)).abs().max()) <= 3e-6 * float((g2 * xh2[:valid]).abs().sum(0).max()), "second stage: sum g x_hat"
    # and without the second stage: the same dz / y
    dz_b, y_b = torch.empty_like(dz), torch.empty_like(y)
    d.dz, d.y, d.dwb = L.ptr(dz_b), L.ptr(y_b), None
    d.bw_x, d.bw_stat, d.bw_sums_acc = None, None, None
    L.check(lib.phc_gemm_bwd_apply(C.byref(d), sp), "bwd_apply plain")
    torch.cuda.synchronize()
    assert torch.equal(dz_b, dz) and torch.equal(y_b, y)


@pytest.mark.parametrize("wl", ["molhiv_b32", "molhiv_b512", "molhiv_b512_swish", "zinc_b1024", "zinc_pub_phc4_4x180_nodrop", "zinc_pub_phc5_4x200_nodrop"])
def test_fused_mlp_norm_node_equals_the_separate_nodes(wl):
    """One training step with the PHMMLP + model norm of every layer as ONE autograd node (ops.mlp_norm_fused: both norm
    backwards folded into the dx GEMMs) against the same step on the separate nodes: loss, every gradient, running statistics."""
    from phc_gnn_amd import harness, ops
    from tests._util import relerr

    def run(fused):
        model = harness.build_model(wl, DEV, seed=1)
        model.train()
        data = harness.to_device(harness.make_batch(wl, seed=5), DEV)
        old = ops.MLP_NORM_FUSED, ops.MLP_NORM_FUSED_MIN_WORK
        ops.MLP_NORM_FUSED, ops.MLP_NORM_FUSED_MIN_WORK = fused, 0          # every size, also below the default threshold
        try:
            loss = harness.train_step(model, data, None, harness.StepConfig(optimizer=False, regression=bool(harness.WORKLOADS[wl].get("regression"))))
        finally:
            ops.MLP_NORM_FUSED, ops.MLP_NORM_FUSED_MIN_WORK = old
        torch.cuda.synchronize()
        return (float(loss), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None},
                {k: v.clone() for k, v in model.state_dict().items() if "running_" in k})

    l1, g1, s1 = run(True)
    l0, g0, s0 = run(False)
    assert abs(l1 - l0) <= 2e-6 * max(1.0, abs(l0))
    assert g1.keys() == g0.keys() and len(g1) > 50
    keep = [k for k in g0 if not (k.endswith(".b") and k not in ("pooling.linear.b", "downstream.affine.2.b")) and float(g0[k].abs().max()) > 0]
    worst, worst_key = max((relerr(g1[k], g0[k]), k) for k in keep)
    # fp32 noise through the batch-norms and ReLU kinks (cf. tests/test_gpu_stat_accumulators.py): a wrong formula shows as 1e-1.
    # At hidden 128 the fused node runs on the one-launch kernels (csrc/mlp_fused.hip, round 6), whose statistics are summed in double:
    # against the separate nodes' fp32 partials a few more elements sit on the other side of a ReLU kink (1e-2 on a norm bias's
    # gradient, seen); the smooth twin shows the formulas agree to round-off.
    bound = 3e-5 if "swish" in wl else (2e-2 if harness.WORKLOADS[wl]["hidden"] == 128 else 1e-3)
    assert worst < bound, (worst, worst_key)
    for k in s0:
        assert_close(s1[k], s0[k], 1e-5, k)
