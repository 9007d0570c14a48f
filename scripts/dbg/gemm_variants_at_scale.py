"""The GEMM variants that carry a batch-norm pass (statistics epilogue, normalising operand, backward column sums, folded norm
backward) against the plain launch and against the stand-alone norm kernels, at config-5 size (N rows x 128) — the numbers behind the
decision whether the sharded layer's four norm passes should move into its GEMMs.  HIP events around `reps` launches.
    python scripts/dbg/gemm_variants_at_scale.py [N] [K]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from phc_gnn_amd import ops, lib as L

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
n, k = 4, K // 4
A = torch.randn(n, n, n, device=dev, generator=g)
W = torch.randn(n, k, k, device=dev, generator=g) * 0.1
b = torch.randn(K, device=dev, generator=g)
H, HT, Hk4, HTk4 = ops.phm_build_operands([(A, W)])[0]
x = torch.randn(N, K, device=dev, generator=g)
dy = torch.randn(N, K, device=dev, generator=g)
y = torch.empty_like(x)
dz = torch.empty_like(x)
lib = L.load()
sp = L.stream_ptr(dev)
words = lib.phc_stat_acc_words(K)
acc = torch.zeros(words, dtype=torch.int64, device=dev)
acc2 = torch.zeros(words, dtype=torch.int64, device=dev)
stat = torch.zeros(6 * K, device=dev)
stat[K:2 * K] = 1.0; stat[2 * K:3 * K] = 1.0; stat[4 * K:5 * K] = 1.0     # mean 0 | invstd 1 | scale 1 | shift 0 | w 1 | b 0
ACT = 1                                                                   # relu


def timed(name, fn, reps=10, flops=2.0 * N * K * K, bytes_=None):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    extra = f"{flops / us / 1e6:7.1f} TFLOP/s" if flops else f"{bytes_ / us / 1e3:7.0f} GB/s"
    print(f"N={N} K={K}  {name:44s} {us:9.1f} us  {extra}", flush=True)
    return us


def plain():
    L.check(lib.phc_gemm_prebuilt(L.ptr(x), L.ptr(H), L.ptr(Hk4), L.ptr(b), None, N, K, K, L.ptr(y), 0, sp), "plain")


def st():
    L.check(lib.phc_gemm_prebuilt_stats(L.ptr(x), L.ptr(H), L.ptr(Hk4), L.ptr(b), N, K, K, L.ptr(y), None, L.ptr(acc), None, None, sp), "st")


def xf():
    L.check(lib.phc_gemm_prebuilt_xf(L.ptr(x), L.ptr(stat), ACT, None, L.ptr(H), L.ptr(Hk4), L.ptr(b), N, K, K, L.ptr(y), None, None, None, sp), "xf")


def xf_st():
    L.check(lib.phc_gemm_prebuilt_xf(L.ptr(x), L.ptr(stat), ACT, None, L.ptr(H), L.ptr(Hk4), L.ptr(b), N, K, K, L.ptr(y), L.ptr(acc), None, None, sp), "xf_st")


def bw():
    L.check(lib.phc_gemm_prebuilt_bwsums(L.ptr(dy), L.ptr(HT), L.ptr(HTk4), N, K, K, L.ptr(dz), L.ptr(x), L.ptr(stat), ACT, None, L.ptr(acc2), None, sp), "bw")


def ba(two_stage):
    d = L.GemmBaDesc()
    d.dout = L.ptr(dy); d.bn_x = L.ptr(x); d.bn_stat = L.ptr(stat); d.sums_acc = L.ptr(acc2); d.n_valid_dev = None
    d.dz = L.ptr(dz); d.dwb = None; d.Mk4 = L.ptr(HTk4); d.y = L.ptr(y); d.N = N; d.K = K; d.Nout = K; d.act = ACT; d.bw_act = ACT
    d.bw_x = L.ptr(x) if two_stage else None; d.bw_stat = L.ptr(stat) if two_stage else None
    d.bw_sums_acc = L.ptr(acc) if two_stage else None
    return lambda: L.check(lib.phc_gemm_bwd_apply(C.byref(d), sp), "ba")


t_plain = timed("gemm plain", plain)
timed("gemm + statistics epilogue (ST)", st)
timed("gemm, normalising operand (XF)", xf)
timed("gemm XF | ST", xf_st)
timed("dx gemm + backward column sums (BW)", bw)
timed("dx gemm with the norm backward folded (BA)", ba(False))
timed("dx gemm BA | BW", ba(True))

# the stand-alone passes the variants would replace (the sharded layer's backend calls, dist.EngineBackend)
row = 4.0 * N * K
w = torch.ones(K, device=dev)
ws = torch.empty(1 << 22, device=dev)
mom = torch.empty(3 * K, device=dev)
timed("bn moments pass (reads x)", lambda: ops.bn_moments(x), flops=0, bytes_=row)
timed("bn backward sums pass (reads dy, x)", lambda: ops.bn_bwd_sums(dy, x, stat, ACT), flops=0, bytes_=2 * row)
dwb = ops.bn_bwd_sums(dy, x, stat, ACT)
timed("bn backward apply pass (reads dy, x; writes dx)", lambda: ops.bn_bwd_apply(dy, x, stat, dwb, N, ACT), flops=0, bytes_=3 * row)
