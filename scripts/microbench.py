"""Back-to-back timing of the individual engine operators (kernel tuning aid).

    python scripts/microbench.py [mol|big|all]

Each op is launched `iters` times back to back between two HIP events, so the GPU
stays busy and clocks are up; reported: µs per call and achieved GB/s / TFLOP/s
against the algorithmic work formulas of DESIGN.md §4."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phc_gnn_amd import ops, synth, lib
from phc_gnn_amd.graph import EdgeStructure

dev = torch.device("cuda:0")


def timeit(fn, iters=50, warm=5):
    rc = fn()
    if isinstance(rc, int) and rc != 0:
        raise RuntimeError(f"engine call failed: rc={rc}: {lib.load().phc_last_error().decode()}")
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters          # µs


def report(name, us, flops=0.0, nbytes=0.0):
    s = f"{name:34s} {us:10.1f} us"
    if flops:
        s += f"  {flops / us / 1e6:8.2f} TFLOP/s ({100 * flops / us / 1e6 / 157.3:5.1f}% of fp32 MFMA)"
    if nbytes:
        s += f"  {nbytes / us / 1e3:8.1f} GB/s ({100 * nbytes / us / 1e3 / 8000:5.1f}% of 8 TB/s)"
    print(s, flush=True)


def run(tag, N, E, h=128, n=4, mol=False):
    print(f"--- {tag}: N={N} E={E} h={h} n={n}")
    g = torch.Generator(device="cpu").manual_seed(0)
    if mol:
        b = synth.molecule_batch(512, seed=0)
        ei = torch.from_numpy(b.edge_index).to(dev)
        N, E = b.num_nodes, b.num_edges
    else:
        ei = torch.randint(0, N, (2, E), device=dev)
    k = h // n
    x = torch.randn(N, h, device=dev)
    ea = torch.randn(E, h, device=dev)
    dy = torch.randn(N, h, device=dev)
    A = torch.randn(n, n, n, device=dev)
    W = torch.randn(n, k, k, device=dev) * 0.1
    bias = torch.randn(h, device=dev)
    l = lib.load()
    st = EdgeStructure(ei, N)
    t0 = time.perf_counter(); st.csr; st.csc; torch.cuda.synchronize()
    print(f"csr+csc build (first, incl. alloc)  {1e3 * (time.perf_counter() - t0):8.2f} ms")
    def rebuild():
        s2 = EdgeStructure(ei, N); s2.csr; s2.csc
    report("csr+csc build", timeit(rebuild, 10, 2), 0, 2 * E * 28.0)

    gemm_f = 2.0 * N * h * h
    gemm_b = 4.0 * N * 2 * h
    y = torch.empty(N, h, device=dev)
    sp = lib.stream_ptr(dev)
    report("phm_linear_fwd", timeit(lambda: l.phc_phm_linear_fwd(x.data_ptr(), A.data_ptr(), W.data_ptr(), bias.data_ptr(), None, N, n, k, k, y.data_ptr(), sp)), gemm_f, gemm_b)
    wsb = l.phc_phm_linear_bwd_ws_bytes(N, n, k, k)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    dx = torch.empty_like(x); dW = torch.empty_like(W); dA = torch.empty_like(A); db = torch.empty(h, device=dev)
    report("phm_linear_bwd (dx only)", timeit(lambda: l.phc_phm_linear_bwd(x.data_ptr(), A.data_ptr(), W.data_ptr(), dy.data_ptr(), N, n, k, k, dx.data_ptr(), None, None, None, None, 0, sp)), gemm_f, gemm_b)
    report("phm_linear_bwd (dW,dA,db only)", timeit(lambda: l.phc_phm_linear_bwd(x.data_ptr(), A.data_ptr(), W.data_ptr(), dy.data_ptr(), N, n, k, k, None, dW.data_ptr(), dA.data_ptr(), db.data_ptr(), ws.data_ptr(), wsb, sp)), gemm_f, gemm_b)

    rowptr, col, perm = st.csr
    rt, ct, pt = st.csc
    out = torch.empty(N, h, device=dev)
    agg_bytes = E * (8.0 * h + 4) + N * 8.0 * h + 4.0 * (N + 1)
    report("aggregate_fwd (ea, self)", timeit(lambda: l.phc_aggregate_fwd(x.data_ptr(), ea.data_ptr(), rowptr.data_ptr(), col.data_ptr(), perm.data_ptr(), N, h, 0, 1, out.data_ptr(), sp)), 0, agg_bytes)
    report("aggregate_fwd (no ea)", timeit(lambda: l.phc_aggregate_fwd(x.data_ptr(), None, rowptr.data_ptr(), col.data_ptr(), None, N, h, 0, 1, out.data_ptr(), sp)), 0, E * (4.0 * h + 4) + N * 8.0 * h + 4.0 * (N + 1))
    dea = torch.empty_like(ea)
    report("aggregate_bwd (dx, dea)", timeit(lambda: l.phc_aggregate_bwd(dy.data_ptr(), None, None, rt.data_ptr(), ct.data_ptr(), pt.data_ptr(), N, N, h, 0, 1, dx.data_ptr(), dea.data_ptr(), sp)), 0, E * (8.0 * h + 4) + N * 8.0 * h)
    report("aggregate_bwd (relu, dx, dea)", timeit(lambda: l.phc_aggregate_bwd(dy.data_ptr(), x.data_ptr(), ea.data_ptr(), rt.data_ptr(), ct.data_ptr(), pt.data_ptr(), N, N, h, 1, 1, dx.data_ptr(), dea.data_ptr(), sp)), 0, E * (12.0 * h + 4) + N * 12.0 * h)

    # encoders
    vocab_a, vocab_b = synth.OGB_ATOM_DIMS, synth.OGB_BOND_DIMS
    ta = [[torch.randn(v, k, device=dev) for v in vocab_a] for _ in range(n)]
    tb = [[torch.randn(v, k, device=dev) for v in vocab_b] for _ in range(n)]
    xa = torch.stack([torch.randint(0, v, (N,), device=dev) for v in vocab_a], 1)
    xb = torch.stack([torch.randint(0, v, (E,), device=dev) for v in vocab_b], 1)
    tsa, tsb = ops.TableSet(ta), ops.TableSet(tb)
    report("encode_fwd (atoms)", timeit(lambda: l.phc_encode_fwd(xa.data_ptr(), tsa.ptrs.data_ptr(), tsa.vocab_c, N, 9, n, k, out.data_ptr(), sp)), 0, N * (72.0 + 4 * h))
    fl = torch.empty(n * tsa.R * k, device=dev)
    wsb2 = l.phc_encode_bwd_ws_bytes(N, 9, tsa.vocab_c, n, k); ws2 = torch.empty(wsb2, dtype=torch.uint8, device=dev)
    report("encode_bwd (atoms)", timeit(lambda: l.phc_encode_bwd(dy.data_ptr(), xa.data_ptr(), N, 9, tsa.vocab_c, n, k, fl.data_ptr(), ws2.data_ptr(), wsb2, sp)), 0, N * (72.0 + 4 * h))
    enc_bytes = E * (4.0 * h + 8 + 24) + N * 8.0 * h + 4.0 * (N + 1)
    report("aggregate_enc_fwd", timeit(lambda: l.phc_aggregate_enc_fwd(x.data_ptr(), xb.data_ptr(), tsb.ptrs.data_ptr(), tsb.vocab_c, rowptr.data_ptr(), col.data_ptr(), perm.data_ptr(), N, 3, n, k, 0, 1, out.data_ptr(), sp)), 0, enc_bytes)
    codes = st.packed_codes(xb, tsb.vocab_c, 3)
    report("aggregate_encp_fwd (packed codes)", timeit(lambda: l.phc_aggregate_encp_fwd(x.data_ptr(), codes.data_ptr(), tsb.ptrs.data_ptr(), tsb.vocab_c, rowptr.data_ptr(), col.data_ptr(), N, 3, n, k, 0, 1, out.data_ptr(), sp)), 0, E * (4.0 * h + 8) + N * 8.0 * h + 4.0 * (N + 1))
    flb = torch.empty(n * tsb.R * k, device=dev)
    wsb3 = l.phc_aggregate_enc_bwd_ws_bytes(E, 3, tsb.vocab_c, n, k); ws3 = torch.empty(wsb3, dtype=torch.uint8, device=dev)
    report("aggregate_enc_bwd (dx only)", timeit(lambda: l.phc_aggregate_enc_bwd(dy.data_ptr(), x.data_ptr(), xb.data_ptr(), tsb.ptrs.data_ptr(), tsb.vocab_c, st._src.data_ptr(), st._dst.data_ptr(), E, rt.data_ptr(), ct.data_ptr(), pt.data_ptr(), N, N, 3, n, k, 0, 1, dx.data_ptr(), None, None, 0, sp)), 0, E * (4.0 * h + 4) + N * 8.0 * h)
    report("aggregate_enc_bwd (tables only)", timeit(lambda: l.phc_aggregate_enc_bwd(dy.data_ptr(), x.data_ptr(), xb.data_ptr(), tsb.ptrs.data_ptr(), tsb.vocab_c, st._src.data_ptr(), st._dst.data_ptr(), E, rt.data_ptr(), ct.data_ptr(), pt.data_ptr(), N, N, 3, n, k, 0, 1, None, flb.data_ptr(), ws3.data_ptr(), wsb3, sp)), 0, N * 4.0 * h + E * 28.0)
    tj = (lib.TableGradJob * 1)(lib.TableGradJob(dy.data_ptr(), None, None, flb.data_ptr()))
    wsb4 = l.phc_table_grads_counted_ws_bytes(N, 3, tsb.vocab_c, n, k, 1); ws4 = torch.empty(wsb4, dtype=torch.uint8, device=dev)
    report("table_grads_counted (bond tables, identity)", timeit(lambda: l.phc_table_grads_counted(tj, 1, rowptr.data_ptr(), codes.data_ptr(), N, E, tsb.vocab_c, 3, n, k, ws4.data_ptr(), wsb4, sp)), 0, N * 4.0 * h + E * 4.0 + 4.0 * (N + 1))

    # batch norm
    wl = [torch.rand(k, device=dev) + 0.5 for _ in range(n)]; bl = [torch.randn(k, device=dev) for _ in range(n)]
    rm = [torch.zeros(k, device=dev) for _ in range(n)]; rv = [torch.ones(k, device=dev) for _ in range(n)]
    xg = x.clone().requires_grad_()
    def bn_f():
        return ops.batch_norm_act(xg, wl, bl, rm, rv, None, True, 0.1, 1e-5, "relu", x)
    report("bn fwd (relu+residual, 3 kernels)", timeit(bn_f), 0, 4.0 * N * h * 4)
    o = bn_f()
    report("bn bwd (3 kernels)", timeit(lambda: torch.autograd.grad(o, [xg], dy, retain_graph=True)), 0, 4.0 * N * h * 5)


which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("mol", "all"):
    run("molhiv B=512", 0, 0, mol=True)
if which in ("big", "all"):
    run("random graph (config-5 shard scale)", 1_250_000, 6_250_000)
if which in ("huge",):
    run("random graph 10M/50M", 10_000_000, 50_000_000)
