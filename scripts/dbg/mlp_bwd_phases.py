"""In-kernel phase clocks of phc_mlp_norm_bwd_fused (csrc/mlp_fused.hip) next to the three launches it replaces, isolated launches:
    PYTHONPATH=. python scripts/dbg/mlp_bwd_phases.py [N]"""
import ctypes as C
import sys
import torch
from phc_gnn_amd import lib as L, ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 12874
h, n = 128, 4
dev = torch.device("cuda:0")
lib = L.load()
g = torch.Generator(device=dev).manual_seed(0)
A = [torch.randn(n, n, n, generator=g, device=dev) for _ in range(2)]
W = [0.1 * torch.randn(n, h // n, h // n, generator=g, device=dev) for _ in range(2)]
pre = ops.phm_build_operands([(A[0], W[0]), (A[1], W[1])])
dout, y2, z1 = (torch.randn(N, h, generator=g, device=dev) for _ in range(3))
def stat():
    mean, inv = torch.randn(h, generator=g, device=dev) * 0.1, 0.5 + torch.rand(h, generator=g, device=dev)
    w, b = 0.5 + torch.rand(h, generator=g, device=dev), 0.1 * torch.randn(h, generator=g, device=dev)
    return torch.cat([mean, inv, inv * w, b - mean * inv * w, w, b]).contiguous()
save_o, save_i = stat(), stat()
nv = torch.tensor(N, dtype=torch.int32, device=dev)
dy2, dz1, ds, da = (torch.empty(N, h, device=dev) for _ in range(4))
dwb_o, dwb_i = torch.empty(2 * h, device=dev), torch.empty(2 * h, device=dev)
words = lib.phc_stat_acc_words(h)
bar = torch.zeros(lib.phc_mlp_norm_bwd_fused_barrier_words(), dtype=torch.int32, device=dev)
stamps = torch.zeros(32, dtype=torch.int64, device=dev)
act = 1

def fused(acc_o, acc_i):
    sp = L.stream_ptr(dev)
    fd = L.MlpNormBwdDesc()
    fd.dout, fd.y2, fd.z1, fd.save_o, fd.save_i = L.ptr(dout), L.ptr(y2), L.ptr(z1), L.ptr(save_o), L.ptr(save_i)
    fd.H2Tk4, fd.H1Tk4 = L.ptr(pre[1][3]), L.ptr(pre[0][3])
    fd.dy2, fd.dz1, fd.ds, fd.dwb_o, fd.dwb_i = L.ptr(dy2), L.ptr(dz1), L.ptr(ds), L.ptr(dwb_o), L.ptr(dwb_i)
    fd.acc_o, fd.acc_i, fd.n_valid_dev = L.ptr(acc_o), L.ptr(acc_i), L.ptr(nv)
    fd.N, fd.h, fd.act_o, fd.act_i = N, h, act, act
    fd.barrier = L.ptr(bar)
    L.check(lib.phc_mlp_norm_bwd_fused(C.byref(fd), sp), "fused")

def separate(acc_o, acc_i):
    sp = L.stream_ptr(dev)
    L.check(lib.phc_bn_bwd_partial_acc(L.ptr(dout), L.ptr(y2), L.ptr(save_o), L.ptr(nv), N, h, act, L.ptr(acc_o), sp), "partial")
    d = L.GemmBaDesc()
    d.dout, d.bn_x, d.bn_stat, d.sums_acc, d.n_valid_dev = L.ptr(dout), L.ptr(y2), L.ptr(save_o), L.ptr(acc_o), L.ptr(nv)
    d.dz, d.dwb, d.Mk4, d.y = L.ptr(dy2), L.ptr(dwb_o), L.ptr(pre[1][3]), L.ptr(da)
    d.N, d.K, d.Nout, d.act = N, h, h, act
    d.bw_act, d.bw_x, d.bw_stat, d.bw_sums_acc = act, L.ptr(z1), L.ptr(save_i), L.ptr(acc_i)
    L.check(lib.phc_gemm_bwd_apply(C.byref(d), sp), "outer")
    d2 = L.GemmBaDesc()
    d2.dout, d2.bn_x, d2.bn_stat, d2.sums_acc, d2.n_valid_dev = L.ptr(da), L.ptr(z1), L.ptr(save_i), L.ptr(acc_i), L.ptr(nv)
    d2.dz, d2.dwb, d2.Mk4, d2.y = L.ptr(dz1), L.ptr(dwb_i), L.ptr(pre[0][3]), L.ptr(ds)
    d2.N, d2.K, d2.Nout, d2.act = N, h, h, act
    L.check(lib.phc_gemm_bwd_apply(C.byref(d2), sp), "inner")

R = 40
accs = [torch.zeros(2 * R + 8, words, dtype=torch.int64, device=dev) for _ in range(2)]
def timed(fn, name):
    for a in accs:
        a.zero_()
    for i in range(4):
        fn(accs[0][2 * R + i], accs[1][2 * R + i])
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(gr, stream=s):
            for i in range(R):
                fn(accs[0][i], accs[1][i])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    for a in accs:
        a.zero_()
    torch.cuda.synchronize()
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    print(f"{name}: {1e3 * e0.elapsed_time(e1) / R:.2f} us per backward (graph of {R} back-to-back, N={N})")

timed(separate, "three launches")
ref = (dy2.clone(), dz1.clone(), ds.clone(), dwb_o.clone(), dwb_i.clone())
timed(fused, "one launch   ")
for nm, a, b in zip(("dy2", "dz1", "ds", "dwb_o", "dwb_i"), ref, (dy2, dz1, ds, dwb_o, dwb_i)):
    print(f"  {nm}: max rel diff {float((a - b).abs().max() / a.abs().max()):.2e}")
lib.phc_debug_mlp_bwd_stamps(L.ptr(stamps))
for a in accs:
    a.zero_()
rows = []
for i in range(10):
    fused(accs[0][i], accs[1][i])
    torch.cuda.synchronize()
    rows.append(stamps.clone().cpu())
lib.phc_debug_mlp_bwd_stamps(None)
names = ["rows in LDS", "sums_o issued", "barrier 1", "P1 done (dz_o, MFMA, da)", "sums_i issued", "barrier 2", "end"]
for which, off in (("wg0", 0), ("wgLast", 16)):
    med = []
    for j in range(1, 8):
        v = sorted(float(r[off + j] - r[off + j - 1]) / 100.0 for r in rows)
        med.append(v[len(v) // 2])
    print(which, " | ".join(f"{nm} {m:.2f}" for nm, m in zip(names, med)), f"| total {sum(med):.2f} us")
L.check_indices(dev)
