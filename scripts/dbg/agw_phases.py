"""In-kernel phase clocks of the WIDE aggregation-inside-the-GEMM launch (gemm_hs_agw_kernel; phc_debug_hs_stamps): medians over
wavefronts of the prologue, the six per-step phases and the finale, beside the launches it replaces.
    python scripts/dbg/agw_phases.py [N E normed n k]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import phc_gnn_amd  # noqa
import torch
from phc_gnn_amd import lib as L, ops, synth
from phc_gnn_amd.graph import edge_structure

DEV = torch.device("cuda:0")
lib = L.load()
a = [int(v) for v in sys.argv[1:6]] if len(sys.argv) > 5 else (1024, 0, 1, 8, 32)
B, _, normed, n, k = a
h, F = n * k, 1
gb = synth.molecule_batch(B, seed=3, dataset="zinc", regression=True)        # a real molecule batch: degrees, locality
N, E = gb.num_nodes, gb.num_edges
vocab = [4]
g = torch.Generator().manual_seed(1)
ei = torch.from_numpy(gb.edge_index).to(DEV)
bond = torch.from_numpy(gb.edge_attr).to(DEV).long().view(E, -1)[:, :1].contiguous()
tables = [[torch.randn(v, k, generator=g).to(DEV) for v in vocab] for _ in range(n)]
ts = ops.TableSet(tables).vector_form(tables)
st = edge_structure(ei, N, N)
st.prepare(for_backward=False, codes_for=(bond, ts.vocab_c, ts.F))
rowptr, col, _ = st.csr
codes = st.packed_codes(bond, ts.vocab_c, ts.F)
nv = torch.tensor([N], dtype=torch.int32, device=DEV)
sp = L.stream_ptr(DEV)
M = (torch.randn(h, h, generator=g) / h ** 0.5).to(DEV)
M4 = M.view(h // 4, 4, h).permute(0, 2, 1).contiguous()
b = torch.randn(h, generator=g).to(DEV)
y_in = torch.randn(N, h, generator=g).to(DEV)
skip = torch.randn(N, h, generator=g).to(DEV)
words = lib.phc_stat_acc_words(h)
arr = lambda tsr: (C.c_void_p * len(tsr))(*[t.data_ptr() for t in tsr])
split = lambda t: [t[c * k:(c + 1) * k] for c in range(n)]
w, bb = split((0.5 + torch.rand(h, generator=g)).to(DEV)), split(torch.randn(h, generator=g).to(DEV))
rm, rv = split(torch.zeros(h, device=DEV)), split(torch.ones(h, device=DEV))
nbt = [torch.zeros((), dtype=torch.int64, device=DEV) for _ in range(n)]
norm_acc = torch.zeros(words, dtype=torch.int64, device=DEV)
L.check(lib.phc_gemm_prebuilt_stats(L.ptr(torch.randn(N, h, generator=g).to(DEV)), L.ptr(M), None, None, N, h, h, L.ptr(y_in), L.ptr(nv),
                                    L.ptr(norm_acc), None, None, sp), "producer")
save, x_out, s_out, y = torch.zeros(6 * h, device=DEV), torch.zeros(N, h, device=DEV), torch.empty(N, h, device=DEV), torch.empty(N, h, device=DEV)
acc = torch.zeros(words, dtype=torch.int64, device=DEV)
keep = (arr(w), arr(bb), arr(rm), arr(rv), arr(nbt))
d = L.GemmAgDesc()
d.x, d.codes, d.tables, d.vocab = L.ptr(y_in), L.ptr(codes), L.ptr(ts.ptrs), C.cast(ts.vocab_c, C.c_void_p)
d.rowptr, d.col, d.N, d.F, d.n, d.k, d.add_self, d.s_out = L.ptr(rowptr), L.ptr(col), N, F, ts.n, ts.k, 1, L.ptr(s_out)
if normed:
    d.norm_acc = L.ptr(norm_acc)
    d.weight, d.bias, d.running_mean, d.running_var, d.num_batches_tracked = (C.cast(a_, C.c_void_p) for a_ in keep)
    d.n_comp, d.norm_act, d.momentum, d.eps = n, L.ACT_IDS["relu"], 0.1, 1e-5
    d.skip, d.save_stat, d.x_out = L.ptr(skip), L.ptr(save), L.ptr(x_out)
d.Mk4, d.b, d.Nout, d.y, d.n_valid_dev, d.stat_acc = L.ptr(M4), L.ptr(b), h, L.ptr(y), L.ptr(nv), L.ptr(acc)


def fused():
    L.check(lib.phc_gemm_aggregate_stats(C.byref(d), sp), "phc_gemm_aggregate_stats")


def gemm_only():
    L.check(lib.phc_gemm_prebuilt_stats(L.ptr(s_out), L.ptr(M), L.ptr(M4), L.ptr(b), N, h, h, L.ptr(y), L.ptr(nv), L.ptr(acc), None, None, sp), "gemm")


def agg_only():
    if normed:
        L.check(lib.phc_aggregate_norm_encp_fwd(L.ptr(y_in), L.ptr(norm_acc), keep[0], keep[1], keep[2], keep[3], keep[4], n, 0.1, 1e-5,
                                                L.ACT_IDS["relu"], L.ptr(nv), L.ptr(skip), L.ptr(save), L.ptr(x_out), L.ptr(codes),
                                                L.ptr(ts.ptrs), ts.vocab_c, L.ptr(rowptr), L.ptr(col), N, F, ts.n, ts.k, 1, L.ptr(s_out), sp), "agg norm")
    else:
        L.check(lib.phc_aggregate_encp_fwd(L.ptr(y_in), L.ptr(codes), L.ptr(ts.ptrs), ts.vocab_c, L.ptr(rowptr), L.ptr(col), N, F, ts.n, ts.k, 0, 1,
                                           L.ptr(s_out), sp), "agg")


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


t_f, t_g, t_a = timeit(fused), timeit(gemm_only), timeit(agg_only)
stamps = torch.zeros(12 * 8 * 1024, dtype=torch.int64, device=DEV)
L.check(lib.phc_debug_hs_stamps(L.ptr(stamps)), "stamps")
fused()
torch.cuda.synchronize()
L.check(lib.phc_debug_hs_stamps(None), "stamps off")
wv = stamps.cpu().view(-1, 12)
wv = wv[wv[:, 0] != 0].double()
t0 = wv[:, 0].min()
steps = wv[:, 10]
us = lambda t: t * 0.01
names = ["image barrier", "MFMAs", "row A", "row B", "rotate + issue", "epilogue + stores"]
widx = torch.arange(stamps.numel() // 12).view(-1)[(stamps.cpu().view(-1, 12)[:, 0] != 0)] % 8          # wavefront index in its workgroup
for ns in sorted(set(steps.tolist())):
    for grp, gname in ((widx < 4, "wavefronts 0-3"), (widx >= 4, "wavefronts 4-7 (their SIMD partners)")):
        sel = wv[(steps == ns) & grp]
        if len(sel) == 0:
            continue
        print(f"steps={int(ns)} {gname} n={len(sel)}: chain -> first finish {us(sel[:, 11] - sel[:, 0]).median():5.2f} | "
              f"prologue {us(sel[:, 1] - sel[:, 0]).median():5.2f} | loop {us(sel[:, 2] - sel[:, 1]).median():5.2f} | finale {us(sel[:, 3] - sel[:, 2]).median():5.2f} | "
              f"end {us(sel[:, 3] - t0).median():5.2f} (max {us(sel[:, 3] - t0).max():5.2f})")
        print("    per step (median / max over wavefronts): " + " | ".join(f"{nm} {us(sel[:, 4 + q]).median() / ns:5.2f}/{us(sel[:, 4 + q]).max() / ns:5.2f}" for q, nm in enumerate(names)))
print(f"h={h} N={N} E={E} normed={normed}: fused {t_f:.2f} us | statistics GEMM alone {t_g:.2f} | aggregation (+ norm) alone {t_a:.2f}   (back to back, hot caches)")
