"""In-kernel phase clocks of the aggregation-inside-the-GEMM launch (phc_gemm_aggregate_stats with phc_debug_hs_stamps): medians over
wavefronts of the prologue (entry -> first barrier), the per-step phases and the statistics finale.
    python scripts/dbg/ag_phases.py [N E normed]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import phc_gnn_amd  # noqa
import torch
from phc_gnn_amd import lib as L, ops
from phc_gnn_amd.graph import edge_structure

DEV = torch.device("cuda:0")
lib = L.load()
N, E, normed = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (12874, 27650, 1)
n, k, h, F = 4, 32, 128, 3
vocab = [5, 6, 2]
g = torch.Generator().manual_seed(1)
ei = torch.stack([torch.randint(0, N, (E,), generator=g), torch.randint(0, N, (E,), generator=g)]).to(DEV)
bond = torch.stack([torch.randint(0, v, (E,), generator=g) for v in vocab], 1).to(DEV)
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
FLAT = os.environ.get("AG_FLAT", "1") != "0"            # the components' parameters back to back (as phc/hypercomplex/norm.py keeps them)
split = (lambda t: [t[c * k:(c + 1) * k] for c in range(n)]) if FLAT else (lambda t: [t[c * k:(c + 1) * k].clone() for c in range(n)])
w = split((0.5 + torch.rand(h, generator=g)).to(DEV))
bb = split(torch.randn(h, generator=g).to(DEV))
rm = split(torch.zeros(h, device=DEV))
rv = split(torch.ones(h, device=DEV))
nbt = [torch.zeros((), dtype=torch.int64, device=DEV) for _ in range(n)]
norm_acc = torch.zeros(words, dtype=torch.int64, device=DEV)
L.check(lib.phc_gemm_prebuilt_stats(L.ptr(torch.randn(N, h, generator=g).to(DEV)), L.ptr(M), None, None, N, h, h, L.ptr(y_in), L.ptr(nv),
                                    L.ptr(norm_acc), None, None, sp), "producer")
save, x_out, s_out, y = torch.zeros(6 * h, device=DEV), torch.zeros(N, h, device=DEV), torch.empty(N, h, device=DEV), torch.empty(N, h, device=DEV)
acc = torch.zeros(words, dtype=torch.int64, device=DEV)
keep = (arr(w), arr(bb), arr(rm), arr(rv), arr(nbt))
d = L.GemmAgDesc()
d.x, d.codes, d.tables, d.vocab = L.ptr(y_in), L.ptr(codes), L.ptr(ts.ptrs), C.cast(ts.vocab_c, C.c_void_p)
d.rowptr, d.col, d.N, d.F, d.n, d.k, d.add_self, d.s_out = L.ptr(rowptr), L.ptr(col), N, F, n, k, 1, L.ptr(s_out)
if normed:
    d.norm_acc = L.ptr(norm_acc)
    d.weight, d.bias, d.running_mean, d.running_var, d.num_batches_tracked = (C.cast(a_, C.c_void_p) for a_ in keep)
    d.n_comp, d.norm_act, d.momentum, d.eps = n, L.ACT_IDS["relu"], 0.1, 1e-5
    d.skip, d.save_stat, d.x_out = L.ptr(skip), L.ptr(save), L.ptr(x_out)
d.Mk4, d.b, d.Nout, d.y, d.n_valid_dev, d.stat_acc = L.ptr(M4), L.ptr(b), h, L.ptr(y), L.ptr(nv), L.ptr(acc)


def run():
    L.check(lib.phc_gemm_aggregate_stats(C.byref(d), sp), "phc_gemm_aggregate_stats")


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 20
stamps = torch.zeros(12 * 8 * 1024, dtype=torch.int64, device=DEV)
L.check(lib.phc_debug_hs_stamps(L.ptr(stamps)), "stamps")
run()
torch.cuda.synchronize()
L.check(lib.phc_debug_hs_stamps(None), "stamps off")
wv = stamps.cpu().view(-1, 12)
wv = wv[wv[:, 0] != 0].double()
t0 = wv[:, 0].min()
steps = wv[:, 7]
for ns in sorted(set(steps.tolist())):
    sel = wv[steps == ns]
    f = lambda col: (sel[:, col] * 0.01)
    print(f"steps={int(ns)} wavefronts={len(sel)}: entry at {((sel[:, 0] - t0) * 0.01).median():5.2f} (max {((sel[:, 0] - t0) * 0.01).max():5.2f}) | prologue {((sel[:, 1] - sel[:, 0]) * 0.01).median():5.2f} | "
          f"loop {((sel[:, 2] - sel[:, 1]) * 0.01).median():5.2f} = add-up {f(4).median():5.2f} + image barrier {f(5).median():5.2f} + MFMA/epilogue {f(6).median():5.2f} | "
          f"finale {((sel[:, 3] - sel[:, 2]) * 0.01).median():5.2f} | end at {((sel[:, 3] - t0) * 0.01).median():5.2f} (max {((sel[:, 3] - t0) * 0.01).max():5.2f})")
nw_ = wv.shape[0] // 8 * 8
per_wave = wv[:nw_].view(-1, 8, 12)
for wi in (0, 1, 2, 7):
    sel = per_wave[:, wi, :]
    g_ = lambda col: ((sel[:, col] - sel[:, 0]) * 0.01).median()
    print(f"  wave {wi}: since entry: sources / codes issued {g_(9):5.2f} | everything issued {g_(10):5.2f} | "
          f"all arrived {g_(11):5.2f} | barrier passed {g_(1):5.2f}")
print(f"N={N} E={E} normed={normed}: {us:.2f} us per launch (back to back)")
