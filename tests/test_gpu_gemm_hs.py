"""The H-stationary streaming GEMM (csrc/gemm_hs_device.h) through the C ABI: every phc_gemm_prebuilt* entry point with the
operand in k4 form against float64 — y = x.H + b (+ residual) (layers.py:198-219), the batch-norm statistics of the output
(norm.py:30-35) in accumulator form, the un-applied-norm operand transform (layers.py:349-355) with its statistics taken from
save_stat or straight from an accumulator, and the norm backward's column sums out of the dz GEMM — at the widths the kernel
serves (K, Nout in {128, 256}) and row counts around its 16-row steps, its ring depth and its persistent loop."""
import ctypes as C

import pytest
import torch

from tests._util import assert_close

import os

os.environ.setdefault("PHC_GEMM_HS_ALL", "1")      # every shape the kernel serves, also where the staged kernel is the default

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROWS = [1, 15, 16, 17, 63, 333, 4097, 12874, 70001]


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


@pytest.mark.parametrize("K,Nout", [(128, 128), (256, 256), (128, 256), (256, 128)])
@pytest.mark.parametrize("N", ROWS)
def test_hs_plain_bias_residual(N, K, Nout):
    from phc_gnn_amd import lib as L
    lib = L.load()
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


@pytest.mark.parametrize("K,Nout", [(128, 128), (256, 256), (128, 256)])
@pytest.mark.parametrize("N,valid", [(16, 16), (17, 9), (333, 333), (12874, 12001), (70001, 70001)])
def test_hs_statistics_accumulator(N, valid, K, Nout):
    """Output statistics in accumulator form, incl. padding rows past n_valid and a column whose mean is 300 standard deviations."""
    from phc_gnn_amd import lib as L
    lib = L.load()
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
@pytest.mark.parametrize("K,Nout", [(128, 128), (256, 256), (256, 128)])
@pytest.mark.parametrize("N,valid", [(17, 17), (333, 301), (12874, 12874)])
def test_hs_operand_transform_from_save_stat(N, valid, K, Nout, act):
    from oracle import phc_oracle as O
    from phc_gnn_amd import lib as L
    lib = L.load()
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


@pytest.mark.parametrize("K,Nout,n_comp", [(128, 128, 4), (256, 256, 8), (256, 128, 4)])
@pytest.mark.parametrize("N,valid", [(333, 301), (12874, 12874)])
def test_hs_operand_transform_from_an_accumulator(N, valid, K, Nout, n_comp):
    """linear1 (statistics into an accumulator) -> [norm -> relu -> linear2] with the norm's scale / shift derived in the second
    GEMM's prologue (phc_gemm_prebuilt_xf_acc): output, save_stat and the running statistics against float64."""
    from phc_gnn_amd import lib as L
    lib = L.load()
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
@pytest.mark.parametrize("K,Nout", [(128, 128), (256, 256), (128, 256)])
@pytest.mark.parametrize("N,valid", [(17, 11), (333, 333), (12874, 12001)])
def test_hs_backward_sums(N, valid, K, Nout, act):
    """dz = dy . H^T plus the norm backward's column sums (sum g, sum g x_hat), accumulator form."""
    from oracle import phc_oracle as O
    from phc_gnn_amd import lib as L
    lib = L.load()
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
