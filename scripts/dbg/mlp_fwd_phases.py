"""In-kernel phase clocks of phc_mlp_norm_fwd_fused (csrc/mlp_fused.hip), isolated launches:  PYTHONPATH=. python scripts/dbg/mlp_fwd_phases.py [N]"""
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
b = [0.1 * torch.randn(h, generator=g, device=dev) for _ in range(2)]
pre = ops.phm_build_operands([(A[0], W[0]), (A[1], W[1])])
s = torch.randn(N, h, generator=g, device=dev)
nv = torch.tensor(N, dtype=torch.int32, device=dev)
words = lib.phc_stat_acc_words(h)
w = [0.5 + torch.rand(h // n, device=dev) for _ in range(n)]
bb = [0.1 * torch.randn(h // n, device=dev) for _ in range(n)]
rm = [torch.zeros(h // n, device=dev) for _ in range(n)]
rv = [torch.ones(h // n, device=dev) for _ in range(n)]
nbt = [torch.zeros((), dtype=torch.int64, device=dev) for _ in range(n)]
arrs = [ops._ptr_array(t) for t in (w, bb, rm, rv, nbt)]
z1, y2 = torch.empty(N, h, device=dev), torch.empty(N, h, device=dev)
save = torch.empty(6 * h, device=dev)
bar = torch.zeros(lib.phc_mlp_norm_bwd_fused_barrier_words(), dtype=torch.int32, device=dev)
stamps = torch.zeros(32, dtype=torch.int64, device=dev)
cast = lambda a: C.cast(a, C.c_void_p)

def fused(acc1, acc2):
    fd = L.MlpNormFwdDesc()
    fd.s, fd.H1k4, fd.b1, fd.H2k4, fd.b2 = L.ptr(s), L.ptr(pre[0][2]), L.ptr(b[0]), L.ptr(pre[1][2]), L.ptr(b[1])
    fd.z1, fd.y2, fd.acc1, fd.acc2 = L.ptr(z1), L.ptr(y2), L.ptr(acc1), L.ptr(acc2)
    fd.weight, fd.bias, fd.running_mean, fd.running_var, fd.num_batches_tracked = (cast(a) for a in arrs)
    fd.n_comp, fd.momentum, fd.eps, fd.save_stat, fd.act = n, 0.1, 1e-5, L.ptr(save), 1
    fd.n_valid_dev, fd.N, fd.h, fd.barrier = L.ptr(nv), N, h, L.ptr(bar)
    L.check(lib.phc_mlp_norm_fwd_fused(C.byref(fd), L.stream_ptr(dev)), "fused")

def separate(acc1, acc2):
    sp = L.stream_ptr(dev)
    L.check(lib.phc_gemm_prebuilt_stats(L.ptr(s), L.ptr(pre[0][0]), L.ptr(pre[0][2]), L.ptr(b[0]), N, h, h, L.ptr(z1), L.ptr(nv), L.ptr(acc1), None, None, sp), "stats")
    L.check(lib.phc_gemm_prebuilt_xf_acc(L.ptr(z1), L.ptr(acc1), arrs[0], arrs[1], arrs[2], arrs[3], arrs[4], n, 0.1, 1e-5, L.ptr(save), 1, L.ptr(nv),
                                         L.ptr(pre[1][0]), L.ptr(pre[1][2]), L.ptr(b[1]), N, h, h, L.ptr(y2), L.ptr(acc2), None, None, sp), "xf")

R = 40
accs = [torch.zeros(2 * R + 8, words, dtype=torch.int64, device=dev) for _ in range(2)]
def timed(fn, name):
    for a in accs:
        a.zero_()
    for i in range(4):
        fn(accs[0][2 * R + i], accs[1][2 * R + i])
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        with torch.cuda.graph(gr, stream=st):
            for i in range(R):
                fn(accs[0][i], accs[1][i])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for a in accs:
        a.zero_()
    torch.cuda.synchronize()
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    print(f"{name}: {1e3 * e0.elapsed_time(e1) / R:.2f} us per forward (graph of {R} back-to-back, N={N})")

timed(separate, "two launches (plain GEMM1 + XF GEMM2)")
timed(fused, "one launch")
lib.phc_debug_mlp_bwd_stamps(L.ptr(stamps))
for a in accs:
    a.zero_()
rows = []
for i in range(10):
    fused(accs[0][i], accs[1][i])
    torch.cuda.synchronize()
    rows.append(stamps.clone().cpu())
lib.phc_debug_mlp_bwd_stamps(None)
names = ["rows in LDS", "GEMM1 + z1 out", "stats1 issued", "barrier", "totals, XF, a in LDS", "GEMM2 + y2 out", "stats2 issued"]
for which, off in (("wg0", 0), ("wgLast", 16)):
    med = []
    for j in range(1, 8):
        v = sorted(float(r[off + j] - r[off + j - 1]) / 100.0 for r in rows)
        med.append(v[len(v) // 2])
    print(which, " | ".join(f"{nm} {m:.2f}" for nm, m in zip(names, med)), f"| total {sum(med):.2f} us")
L.check_indices(dev)
