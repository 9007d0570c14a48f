"""In-kernel phase clocks of the H-stationary GEMM variants (phc_debug_hs_stamps): per variant at one shape, the median over
wavefronts of prologue / loop / per-step phase times.   python scripts/dbg/hs_phases.py [N K Nout]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import phc_gnn_amd  # noqa
import torch
from phc_gnn_amd import lib as L

DEV = torch.device("cuda:0")
lib = L.load()
N, K, Nout = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (23600, 256, 256)
g = torch.Generator().manual_seed(1)
x = torch.randn(N, K, generator=g).to(DEV)
xin = torch.randn(N, K, generator=g).to(DEV)
M = (torch.randn(K, Nout, generator=g) / K ** 0.5).to(DEV)
M4 = M.view(K // 4, 4, Nout).permute(0, 2, 1).contiguous()
b = torch.randn(Nout, generator=g).to(DEV)
bwx = torch.randn(N, Nout, generator=g).to(DEV)
stat = torch.cat([torch.zeros(K), torch.ones(K), torch.ones(K), torch.zeros(K), torch.ones(K), torch.zeros(K)]).to(DEV)
stat2 = torch.cat([torch.zeros(Nout), torch.ones(Nout), torch.ones(Nout), torch.zeros(Nout), torch.ones(Nout), torch.zeros(Nout)]).to(DEV)
y = torch.empty(N, Nout, device=DEV)
dz = torch.empty(N, K, device=DEV)
nv = torch.tensor([N], dtype=torch.int32, device=DEV)
sp = L.stream_ptr(DEV)
words = 12 * 16 * 4096
stamps = torch.zeros(words, dtype=torch.int64, device=DEV)


def acc(h):
    return torch.zeros(lib.phc_stat_acc_words(h), dtype=torch.int64, device=DEV)


def plain():
    L.check(lib.phc_gemm_prebuilt(L.ptr(x), L.ptr(M), L.ptr(M4), L.ptr(b), None, N, K, Nout, L.ptr(y), 0, sp), "plain")


def st():
    L.check(lib.phc_gemm_prebuilt_stats(L.ptr(x), L.ptr(M), L.ptr(M4), L.ptr(b), N, K, Nout, L.ptr(y), L.ptr(nv), L.ptr(A1), None, None, sp), "st")


def xf():
    L.check(lib.phc_gemm_prebuilt_xf(L.ptr(x), L.ptr(stat), 1, L.ptr(nv), L.ptr(M), L.ptr(M4), L.ptr(b), N, K, Nout, L.ptr(y), None, None, None, sp), "xf")


def xf_st():
    L.check(lib.phc_gemm_prebuilt_xf(L.ptr(x), L.ptr(stat), 1, L.ptr(nv), L.ptr(M), L.ptr(M4), L.ptr(b), N, K, Nout, L.ptr(y), L.ptr(A1), None, None, sp), "xf")


def bw():
    L.check(lib.phc_gemm_prebuilt_bwsums(L.ptr(x), L.ptr(M), L.ptr(M4), N, K, Nout, L.ptr(y), L.ptr(bwx), L.ptr(stat2), 1, L.ptr(nv), L.ptr(A1), None, sp), "bw")


def ba(second):
    d = L.GemmBaDesc()
    d.dout, d.bn_x, d.bn_stat, d.sums_acc, d.n_valid_dev = L.ptr(x), L.ptr(xin), L.ptr(stat), L.ptr(A2), L.ptr(nv)
    d.dz, d.dwb, d.Mk4, d.y = L.ptr(dz), None, L.ptr(M4), L.ptr(y)
    d.N, d.K, d.Nout, d.act = N, K, Nout, 1
    d.bw_act, d.bw_x, d.bw_stat, d.bw_sums_acc = (1, L.ptr(bwx), L.ptr(stat2), L.ptr(A1)) if second else (0, None, None, None)
    L.check(lib.phc_gemm_bwd_apply(C.byref(d), sp), "ba")


A1, A2 = acc(Nout), acc(K)
for name, fn in (("plain", plain), ("ST", st), ("XF", xf), ("XF|ST", xf_st), ("BW", bw), ("BA", lambda: ba(False)), ("BA|BW", lambda: ba(True))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    stamps.zero_()
    L.check(lib.phc_debug_hs_stamps(L.ptr(stamps)), "stamps")
    fn()
    torch.cuda.synchronize()
    L.check(lib.phc_debug_hs_stamps(None), "stamps off")
    w = stamps.cpu().view(-1, 12)
    w = w[w[:, 3] != 0].double()
    pro = (w[:, 4] - w[:, 3]) * 0.01
    loop = (w[:, 5] - w[:, 4]) * 0.01
    steps = w[:, 10].clamp_min(1)
    ph = w[:, 6:10] * 0.01 / steps[:, None]
    if w.numel() == 0:
        print(name, "no stamps (this shape ran on the staged kernel)", us)
        continue
    span = (w[:, 5].max() - w[:, 3].min()) * 0.01
    clk = ((w[:, 2] - w[:, 1]) / (w[:, 5] - w[:, 4]).clamp_min(1) * 0.1).median()
    d11 = w[:, 11].long()
    pd = [((d11 >> sh) & 0xfffff).double() * 0.01 for sh in (0, 20, 40)]
    if name.startswith("BA"):
        print(f"       prologue detail: accumulator words arrived med {pd[0][pd[0] > 0].median():5.2f} | constants in LDS + barrier {pd[1].median():5.2f} | first DMA image {pd[2].median():5.2f} us after entry")
    print(f"{name:6s} N={N} K={K} Nout={Nout}: {us:7.2f} us/launch | in-kernel span {span:6.2f} | prologue med {pro.median():5.2f} max {pro.max():5.2f} | loop med "
          f"{loop.median():6.2f} max {loop.max():6.2f} | per step: wait+barrier {ph[:, 0].median():5.2f}  issue+transform {ph[:, 1].median():5.2f}  "
          f"MFMA {ph[:, 2].median():5.2f}  epilogue {ph[:, 3].median():5.2f} | steps {int(steps.median())} | clock {clk:4.2f} GHz")
