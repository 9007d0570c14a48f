"""Round-4 launch-chain fusions of the training step, each against the launches it replaces on the SAME step (same model state, same
batch) — they reorder no floating-point operation, so every output is compared bit for bit:
  * the model-level norm's element-wise pass folded into the next layer's aggregation (phc_aggregate_norm_encp_fwd; models.py:208-215
    followed by messagepassing.py:132-146), sc_type "first" and "last", ReLU and swish, padded batches under hipGraph replay;
  * the skips' gradient shares added inside the first layer's aggregation backward (phc_aggregate_dx_rows_multi; models.py:230-234)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _step(workload, toggles, graph=False):
    from phc_gnn_amd import graphed, harness, ops
    from phc_gnn_amd.optim import ClipAdam
    keep = {k: getattr(ops, k) for k in toggles}
    try:
        for k, v in toggles.items():
            setattr(ops, k, v)
        model = harness.build_model(workload, DEV, seed=3).train()
        gb = harness.make_batch(workload, seed=11)
        cfg = harness.StepConfig(regression=bool(harness.WORKLOADS[workload].get("regression")), optimizer=False, weight_decay=1e-3)
        if graph:
            n_max, e_max = graphed.bucket_shape([gb])
            padded = graphed.padded_to_device(graphed.pad_batch(gb, n_max + 37, e_max + 91), DEV)
            sd = {k: v.clone() for k, v in model.state_dict().items()}
            step = graphed.GraphedTrainStep(model, ClipAdam(model.parameters(), lr=1e-3), cfg, padded, warmup=2)
            model.load_state_dict(sd)
            loss = step.step(padded)
        else:
            loss = harness.train_step(model, harness.to_device(gb, DEV), None, cfg)
        torch.cuda.synchronize()
        return (loss.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None},
                {k: b.clone() for k, b in model.named_buffers()})
    finally:
        for k, v in keep.items():
            setattr(ops, k, v)


def _same(a, b):
    assert torch.equal(a[0], b[0]), (float(a[0]), float(b[0]))
    assert a[1].keys() == b[1].keys()
    for k in a[1]:
        assert torch.equal(a[1][k], b[1][k]), ("grad", k, float((a[1][k] - b[1][k]).abs().max()))
    for k in a[2]:
        assert torch.equal(a[2][k], b[2][k]), ("buffer", k)


@pytest.mark.parametrize("workload", ["molhiv_b32", "molhiv_b32_swish", "zinc_small_last", "molhiv_b32_softmax", "molhiv_small_200_softmax_swish"])
@pytest.mark.parametrize("graph", [False, True])
def test_norm_folded_into_the_next_aggregation_is_bit_identical(workload, graph):
    from phc_gnn_amd import harness
    if workload == "zinc_small_last":                                     # sc_type "last": the skip IS the parked tensor
        harness.WORKLOADS.setdefault("zinc_small_last", dict(dataset="zinc", batch_size=48, phm_dim=4, hidden=64, layers=3, d_units=[64, 32],
                                                             target_dim=1, sc_type="last", regression=True))
    # round 5: the same fold inside the SOFTMAX aggregation (phc_aggregate_softmax_norm_enc_fwd; messagepassing.py:211-214, 297-300)
    harness.WORKLOADS.setdefault("molhiv_b32_softmax", dict(harness.WORKLOADS["molhiv_b32"], msg_aggr="softmax"))
    harness.WORKLOADS.setdefault("molhiv_small_200_softmax_swish", dict(dataset="molhiv", batch_size=40, phm_dim=4, hidden=200, layers=3,
                                                                        d_units=[128, 32], target_dim=1, sc_type="first", msg_aggr="softmax",
                                                                        activation="swish"))
    on = _step(workload, dict(NORM_INTO_AGGREGATE=True), graph)
    off = _step(workload, dict(NORM_INTO_AGGREGATE=False), graph)
    _same(on, off)


@pytest.mark.parametrize("workload", ["molhiv_b32", "molpcba_small", "molhiv_b32_softmax", "molhiv_small_200_softmax_swish"])
def test_skip_gradient_shares_meet_in_the_first_aggregation_backward(workload):
    """Same sums in a different order (slot order inside the dx kernel instead of the n-way sum kernel's): 1e-6 of the largest entry.
    Softmax aggregation: the shares ride in phc_aggregate_softmax_enc_bwd_res's walk."""
    from phc_gnn_amd import harness
    harness.WORKLOADS.setdefault("molhiv_b32_softmax", dict(harness.WORKLOADS["molhiv_b32"], msg_aggr="softmax"))
    harness.WORKLOADS.setdefault("molhiv_small_200_softmax_swish", dict(dataset="molhiv", batch_size=40, phm_dim=4, hidden=200, layers=3,
                                                                        d_units=[128, 32], target_dim=1, sc_type="first", msg_aggr="softmax",
                                                                        activation="swish"))
    if workload == "molpcba_small":                                       # five layers: the eight-slot instantiation
        harness.WORKLOADS.setdefault("molpcba_small", dict(dataset="molpcba", batch_size=24, phm_dim=4, hidden=64, layers=5, d_units=[64, 32],
                                                           target_dim=8, sc_type="first", nan_frac=0.5))
    # (round 6: the aggregation's dx inside the backward launch exists only with the hand-off — it regroups the norms' column sums;
    # off in both runs, so that the two differ by the hand-off alone)
    on = _step(workload, dict(SKIP_SUM_HANDOFF=True, MLP_BWD_FUSED_AGG=False))
    off = _step(workload, dict(SKIP_SUM_HANDOFF=False, MLP_BWD_FUSED_AGG=False))
    assert torch.allclose(on[0], off[0], rtol=0, atol=0)
    for k in on[1]:
        ref = off[1][k]
        assert float((on[1][k] - ref).abs().max()) <= 2e-6 * max(float(ref.abs().max()), 1e-30), k


@pytest.mark.parametrize("workload", ["molhiv_b32", "zinc_small_last", "zinc_pub_phc4_4x180_nodrop"])
def test_atom_encoder_inside_the_operand_build_launch_is_bit_identical(workload):
    """phc_phm_build_H_batched_ex: the atom encoder as passenger workgroups of the step's first launch (the same table rows added in
    the same order as phc_encode_fwd) — k % 4 != 0 models (flat table images) included."""
    from phc_gnn_amd import harness
    if workload == "zinc_small_last":
        harness.WORKLOADS.setdefault("zinc_small_last", dict(dataset="zinc", batch_size=48, phm_dim=4, hidden=64, layers=3, d_units=[64, 32],
                                                             target_dim=1, sc_type="last", regression=True))
    _same(_step(workload, dict(ENCODE_FRONT=True)), _step(workload, dict(ENCODE_FRONT=False)))


def test_atom_encoder_passenger_reports_bad_indices():
    """An atom feature outside its vocabulary is read as row 0 and raises at the next check point, as in the stand-alone launch."""
    from phc_gnn_amd import harness, lib
    model = harness.build_model("molhiv_b32", DEV, seed=3).train()
    data = harness.to_device(harness.make_batch("molhiv_b32", seed=11), DEV)
    data.x = data.x.clone()
    data.x[5, 0] = 100000
    harness.train_step(model, data, None, harness.StepConfig(optimizer=False))
    torch.cuda.synchronize()
    with pytest.raises(IndexError):
        lib.check_indices(torch.device(DEV))


# ---- the aggregation inside the layer's first GEMM (phc_gemm_aggregate_stats, csrc/gemm_hs_ag_device.h) ----------------------------
def _random_graph(N, E, hubs, seed):
    g = torch.Generator().manual_seed(seed)
    src = torch.randint(0, N, (E,), generator=g)
    dst = torch.randint(0, N, (E,), generator=g)
    if hubs:                                                              # rows with > 16 in-edges (the two-pass walk), one of 300
        dst[: E // 6] = N // 3
        dst[E // 6: E // 6 + 40] = 5
    dst[dst == N - 2] = 0                                                 # an isolated destination near the end
    return torch.stack([src, dst])


WIDTHS = [(4, 32), (8, 32), (4, 64), (4, 45), (5, 40)]                  # h = 128 | 256 (configs[2] PHM-8, configs[3] PHM-4) | 180, 200 (published)


def _gemm_aggregate_case(N, E, hubs, F, normed, act, with_skip, add_self, n, k, valid, tags=("separate", "fused")):
    """One phc_gemm_aggregate_stats launch and / or the launches it replaces on the same inputs; returns (results by tag, inputs)."""
    import ctypes as C
    from phc_gnn_amd import lib as L, ops
    from phc_gnn_amd.graph import edge_structure
    lib = L.load()
    h = n * k
    vocab = [5, 6, 2, 3][:F]
    if not lib.phc_gemm_aggregate_supported(N, h, h, F, sum(vocab)):
        pytest.skip("outside the fused form")
    g = torch.Generator().manual_seed(N + F + h)
    ei = _random_graph(N, E, hubs, seed=N).to(DEV)
    bond = torch.stack([torch.randint(0, v, (E,), generator=g) for v in vocab], 1).to(DEV)
    tables = [[torch.randn(v, k, generator=g).to(DEV) for v in vocab] for _ in range(n)]
    ts = ops.TableSet(tables).vector_form(tables)                         # (k % 4 != 0: one flat image, n = 1, k = h)
    st = edge_structure(ei, N, N)
    st.prepare(for_backward=False, codes_for=(bond, ts.vocab_c, ts.F))
    rowptr, col, _ = st.csr
    codes = st.packed_codes(bond, ts.vocab_c, ts.F)
    nv = torch.tensor([valid], dtype=torch.int32, device=DEV)
    sp = L.stream_ptr(torch.device(DEV))
    M = (torch.randn(h, h, generator=g) / h ** 0.5).to(DEV)
    M4 = M.view(h // 4, 4, h).permute(0, 2, 1).contiguous()
    b = torch.randn(h, generator=g).to(DEV)
    y_in = torch.randn(N, h, generator=g).to(DEV)
    skip = torch.randn(N, h, generator=g).to(DEV) if with_skip else None
    words = lib.phc_stat_acc_words(h)
    arr = lambda tsr: (C.c_void_p * len(tsr))(*[t.data_ptr() for t in tsr])
    w = [(0.5 + torch.rand(k, generator=g)).to(DEV) for _ in range(n)]
    bb = [torch.randn(k, generator=g).to(DEV) for _ in range(n)]
    norm_acc = torch.zeros(words, dtype=torch.int64, device=DEV)
    if normed:                                                            # the statistics a producing GEMM would have left behind
        L.check(lib.phc_gemm_prebuilt_stats(L.ptr(torch.randn(N, h, generator=g).to(DEV)), L.ptr(M), None, None, N, h, h, L.ptr(y_in), L.ptr(nv),
                                            L.ptr(norm_acc), None, None, sp), "producer")
    res = {}
    for tag in tags:
        rm = [torch.zeros(k, device=DEV) for _ in range(n)]
        rv = [torch.ones(k, device=DEV) for _ in range(n)]
        nbt = [torch.zeros((), dtype=torch.int64, device=DEV) for _ in range(n)]
        save = torch.zeros(6 * h, device=DEV)
        x_out = torch.zeros(N, h, device=DEV)
        s_out = torch.empty(N, h, device=DEV)
        y = torch.empty(N, h, device=DEV)
        acc = torch.zeros(words, dtype=torch.int64, device=DEV)
        if tag == "separate":
            if normed:
                L.check(lib.phc_aggregate_norm_encp_fwd(L.ptr(y_in), L.ptr(norm_acc), arr(w), arr(bb), arr(rm), arr(rv), arr(nbt), n, 0.1, 1e-5,
                                                        L.ACT_IDS[act], L.ptr(nv), L.ptr(skip), L.ptr(save), L.ptr(x_out), L.ptr(codes),
                                                        L.ptr(ts.ptrs), ts.vocab_c, L.ptr(rowptr), L.ptr(col), N, F, ts.n, ts.k, int(add_self),
                                                        L.ptr(s_out), sp), "aggregate_norm")
            else:
                L.check(lib.phc_aggregate_encp_fwd(L.ptr(y_in), L.ptr(codes), L.ptr(ts.ptrs), ts.vocab_c, L.ptr(rowptr), L.ptr(col), N, F, ts.n, ts.k, 0,
                                                   int(add_self), L.ptr(s_out), sp), "aggregate")
            L.check(lib.phc_gemm_prebuilt_stats(L.ptr(s_out), L.ptr(M), L.ptr(M4), L.ptr(b), N, h, h, L.ptr(y), L.ptr(nv), L.ptr(acc), None, None,
                                                sp), "gemm stats")
        else:
            d = L.GemmAgDesc()
            d.x, d.codes, d.tables, d.vocab = L.ptr(y_in), L.ptr(codes), L.ptr(ts.ptrs), C.cast(ts.vocab_c, C.c_void_p)
            d.rowptr, d.col, d.N, d.F, d.n, d.k, d.add_self, d.s_out = L.ptr(rowptr), L.ptr(col), N, F, ts.n, ts.k, int(add_self), L.ptr(s_out)
            keep = (arr(w), arr(bb), arr(rm), arr(rv), arr(nbt))
            if normed:
                d.norm_acc = L.ptr(norm_acc)
                d.weight, d.bias, d.running_mean, d.running_var, d.num_batches_tracked = (C.cast(a_, C.c_void_p) for a_ in keep)
                d.n_comp, d.norm_act, d.momentum, d.eps = n, L.ACT_IDS[act], 0.1, 1e-5
                d.skip, d.save_stat, d.x_out = L.ptr(skip), L.ptr(save), L.ptr(x_out)
            d.Mk4, d.b, d.Nout, d.y, d.n_valid_dev, d.stat_acc = L.ptr(M4), L.ptr(b), h, L.ptr(y), L.ptr(nv), L.ptr(acc)
            L.check(lib.phc_gemm_aggregate_stats(C.byref(d), sp), "phc_gemm_aggregate_stats")
        torch.cuda.synchronize()
        res[tag] = dict(s=s_out, y=y, acc=acc, x=x_out, save=save, rm=torch.cat(rm), rv=torch.cat(rv), nbt=torch.stack(nbt))
    return res, dict(ei=ei, bond=bond, tables=tables, M=M, b=b, y_in=y_in, skip=skip, w=w, bb=bb, h=h)


@pytest.mark.parametrize("n,k", WIDTHS)
@pytest.mark.parametrize("N,E,hubs", [(1000, 2100, False), (12874, 27650, False), (333, 1500, True), (16, 40, False), (25000, 52000, True)])
@pytest.mark.parametrize("F,normed,act,with_skip,add_self", [(3, False, "relu", False, True), (3, True, "relu", True, True),
                                                             (1, True, "swish", False, True), (2, True, "relu", True, False),
                                                             (4, False, "relu", False, False)])
def test_aggregation_inside_the_first_gemm_equals_the_separate_launches(N, E, hubs, F, normed, act, with_skip, add_self, n, k):
    """s, x, save_stat, running statistics, y and the statistics accumulator of ONE phc_gemm_aggregate_stats launch against
    phc_aggregate_encp_fwd / phc_aggregate_norm_encp_fwd followed by phc_gemm_prebuilt_stats: bit for bit (same walk, same MFMA
    order, same partial sums per workgroup) — hidden 128 (gemm_hs_ag_device.h) and the wide layers 256 / 180 / 200
    (gemm_hs_agw_device.h; round 5)."""
    res, _ = _gemm_aggregate_case(N, E, hubs, F, normed, act, with_skip, add_self, n, k, valid=N - 7 if N > 16 else N)
    for key in res["fused"]:
        a_, b_ = res["fused"][key], res["separate"][key]
        assert torch.equal(a_, b_), (key, float((a_.double() - b_.double()).abs().max()))


@pytest.mark.parametrize("n,k", WIDTHS)
@pytest.mark.parametrize("N,E,hubs", [(2000, 4300, False), (333, 1500, True)])
@pytest.mark.parametrize("F,normed,act,with_skip,add_self", [(3, False, "relu", False, True), (3, True, "relu", True, True),
                                                             (1, True, "swish", False, False)])
def test_aggregation_inside_the_first_gemm_against_the_oracle(N, E, hubs, F, normed, act, with_skip, add_self, n, k):
    """The fused launch DIRECTLY against the oracle (not only against the launches it replaces):
    x = act(norm(y)) + skip (norm.py:30-35, models.py:208-215), edge embedding = sum of the integer encoder's table rows
    (encoder.py), s = propagate_add(x, edge_index, e) (+ x) (messagepassing.py:132-146), z = s . H + b (layers.py:198-219), and the
    batch statistics of z that the launch leaves in its accumulator — float64 on the host, 1e-5 of each tensor's largest entry."""
    from oracle import phc_oracle as O
    from phc_gnn_amd import lib as L, ops
    res, inp = _gemm_aggregate_case(N, E, hubs, F, normed, act, with_skip, add_self, n, k, valid=N, tags=("fused",))
    r = res["fused"]
    h = inp["h"]
    y_in, ei = inp["y_in"].double().cpu(), inp["ei"].cpu()
    x = y_in
    if normed:
        w, bb = [t.double().cpu() for t in inp["w"]], [t.double().cpu() for t in inp["bb"]]
        rm, rv = [torch.zeros(k, dtype=torch.float64) for _ in range(n)], [torch.ones(k, dtype=torch.float64) for _ in range(n)]
        x = O.activation(act, O.batch_norm_naive(y_in, n, w, bb, rm, rv, True, 0.1, 1e-5))
        if with_skip:
            x = x + inp["skip"].double().cpu()
        assert_rel(r["x"], x, 1e-5, "x")
        assert_rel(r["rm"], torch.cat(rm), 1e-5, "running_mean")
        assert_rel(r["rv"], torch.cat(rv), 1e-5, "running_var")
    bond = inp["bond"].cpu()
    ea = torch.cat([sum(inp["tables"][c][f].double().cpu()[bond[:, f]] for f in range(F)) for c in range(n)], 1)    # component-major [E, h]
    s = O.propagate_add(x, ei, ea, "identity")
    if add_self:
        s = s + x
    assert_rel(r["s"], s, 1e-5, "s")
    z = s @ inp["M"].double().cpu() + inp["b"].double().cpu()
    assert_rel(r["y"], z, 1e-5, "z")
    mom = ops.stat_acc_read(r["acc"], h, N).double().cpu()                 # (sum | M2) of z as the next norm reads them
    assert_rel(mom[:h], z.sum(0), 1e-5, "column sums of z")
    assert_rel(mom[h:], ((z - z.mean(0)) ** 2).sum(0), 1e-4, "column M2 of z")


def assert_rel(got, ref, tol, what):
    got, ref = got.double().cpu(), ref.double().cpu()
    err = float((got - ref).abs().max()) / max(float(ref.abs().max()), 1e-30)
    assert err <= tol, (what, err)


_AG_EXTRA = {
    "zinc_small_last_128": dict(dataset="zinc", batch_size=48, phm_dim=4, hidden=128, layers=3, d_units=[64, 32], target_dim=1,
                                sc_type="last", regression=True),
    # the wide layers (gemm_hs_agw_device.h): configs[2]'s and configs[3]'s geometry, the published 180 / 200 widths (k = 45 / 40)
    "zinc_small_last_256": dict(dataset="zinc", batch_size=96, phm_dim=8, hidden=256, layers=3, d_units=[64, 32], target_dim=1,
                                sc_type="last", regression=True),
    "molpcba_small_256": dict(dataset="molpcba", batch_size=40, phm_dim=4, hidden=256, layers=3, d_units=[64, 32], target_dim=8,
                              sc_type="first", nan_frac=0.5),
    "zinc_small_last_180_swish": dict(dataset="zinc", batch_size=64, phm_dim=4, hidden=180, layers=3, d_units=[60, 32], target_dim=1,
                                      sc_type="last", regression=True, activation="swish"),
    "zinc_small_last_200": dict(dataset="zinc", batch_size=64, phm_dim=5, hidden=200, layers=2, d_units=[60, 30], target_dim=1,
                                sc_type="last", regression=True),
}


@pytest.mark.parametrize("workload", ["molhiv_b32", "molhiv_b32_swish", "molhiv_b512", "zinc_small_last_128", "zinc_small_last_256",
                                      "molpcba_small_256", "zinc_small_last_180_swish", "zinc_small_last_200", "zinc_b1024"])
@pytest.mark.parametrize("graph", [False, True])
def test_aggregation_inside_the_first_gemm_leaves_the_step_bit_identical(workload, graph):
    """The whole training step (loss, every gradient, every buffer) with ops.GEMM_AGGREGATE on and off; the fused launch must really
    have run (the launch profiler books it under its own name)."""
    from phc_gnn_amd import harness, lib as L
    if workload in _AG_EXTRA:
        harness.WORKLOADS.setdefault(workload, _AG_EXTRA[workload])
    on = _step(workload, dict(GEMM_AGGREGATE=True), graph)
    off = _step(workload, dict(GEMM_AGGREGATE=False), graph)
    _same(on, off)
    if not graph:
        L.prof_enable(True)
        L.prof_reset()
        try:
            _step(workload, dict(GEMM_AGGREGATE=True), False)
            names = {k for k, v in L.prof_read().items() if v["launches"] > 0}
        finally:
            L.prof_enable(False)
        assert "phm_gemm_ag_fwd" in names, sorted(names)


# ---- the read-out's forward as one launch (phc_readout_fused_fwd, csrc/readout.hip) ------------------------------------------------
@pytest.mark.parametrize("B,sizes", [(7, "small"), (544, "molecules"), (40, "big")])
@pytest.mark.parametrize("normed,act,with_skip", [(False, "relu", False), (True, "relu", True), (True, "swish", False)])
@pytest.mark.parametrize("n,k", [(4, 32), (8, 32), (4, 64)])             # h = 128 (configs[1]), 256 (configs[2] PHM-8, configs[3] PHM-4)
def test_readout_in_one_launch_equals_the_three_launches(B, sizes, normed, act, with_skip, n, k):
    """x (bit for bit), gate and the per-graph sums (fp32 summation order: 2e-6 of the largest entry) of ONE phc_readout_fused_fwd
    launch against phc_bn_fwd_from_partials -> phc_gemm_prebuilt -> phc_gated_pool_fwd_sorted; empty graphs, graphs of 1 ... 300 rows,
    padding rows behind *n_valid."""
    import ctypes as C
    from phc_gnn_amd import lib as L
    lib = L.load()
    h, kg = n * k, k
    g = torch.Generator().manual_seed(B)
    hi = dict(small=5, molecules=45, big=300)[sizes]
    cnt = torch.randint(0, hi, (B,), generator=g)
    cnt[B // 2] = 0                                                       # an empty graph in the middle
    cnt[0] = max(int(cnt[0]), 1)
    batch = torch.repeat_interleave(torch.arange(B), cnt).to(DEV)
    N = int(batch.numel())
    if not lib.phc_readout_fused_supported(N, B, h, n, kg):
        pytest.skip("outside the fused form")
    valid = max(N - 5, 1)
    nv = torch.tensor([valid], dtype=torch.int32, device=DEV)
    sp = L.stream_ptr(torch.device(DEV))
    M = (torch.randn(h, kg, generator=g) / h ** 0.5).to(DEV)
    c = torch.randn(kg, generator=g).to(DEV)
    y_in = torch.randn(N, h, generator=g).to(DEV)
    skip = torch.randn(N, h, generator=g).to(DEV) if with_skip else None
    words = lib.phc_stat_acc_words(h)
    arr = lambda tsr: (C.c_void_p * len(tsr))(*[t.data_ptr() for t in tsr])
    split = lambda t: [t[cc * k:(cc + 1) * k] for cc in range(n)]
    w, bb = split((0.5 + torch.rand(h, generator=g)).to(DEV)), split(torch.randn(h, generator=g).to(DEV))
    norm_acc = torch.zeros(words, dtype=torch.int64, device=DEV)
    if normed:
        P = (torch.randn(h, h, generator=g) / h ** 0.5).to(DEV)
        L.check(lib.phc_gemm_prebuilt_stats(L.ptr(torch.randn(N, h, generator=g).to(DEV)), L.ptr(P), None, None, N, h, h, L.ptr(y_in), L.ptr(nv),
                                            L.ptr(norm_acc), None, None, sp), "producer")
    res = {}
    for tag in ("separate", "fused"):
        rm, rv = split(torch.zeros(h, device=DEV)), split(torch.ones(h, device=DEV))
        nbt = [torch.zeros((), dtype=torch.int64, device=DEV) for _ in range(n)]
        save = torch.zeros(6 * h, device=DEV)
        x_out = torch.zeros(N, h, device=DEV)
        gate = torch.zeros(N, kg, device=DEV)
        out = torch.full((B, h), 7.0, device=DEV)
        keep = (arr(w), arr(bb), arr(rm), arr(rv), arr(nbt))
        if tag == "separate":
            x = y_in
            if normed:
                L.check(lib.phc_bn_fwd_from_partials(L.ptr(y_in), L.ptr(norm_acc), None, 0, keep[0], keep[1], keep[2], keep[3], keep[4], n, L.ptr(nv),
                                                     N, h, 0.1, 1e-5, L.ACT_IDS[act], L.ptr(skip), L.ptr(x_out), L.ptr(save), 0.0, 0, 0, None, sp),
                        "bn_fwd_from_partials")
                x = x_out
            L.check(lib.phc_gemm_prebuilt(L.ptr(x), L.ptr(M), None, L.ptr(c), None, N, h, kg, L.ptr(gate), 0, sp), "gate gemm")
            L.check(lib.phc_gated_pool_fwd_sorted(L.ptr(x), L.ptr(gate), L.ptr(batch), N, B, h, n, L.ptr(out), sp), "gated_pool")
        else:
            d = L.ReadoutDesc()
            d.x, d.batch_sorted_i64, d.N, d.num_graphs, d.h, d.n_comp, d.kg = L.ptr(y_in), L.ptr(batch), N, B, h, n, kg
            d.M, d.c, d.gate, d.out, d.n_valid_dev = L.ptr(M), L.ptr(c), L.ptr(gate), L.ptr(out), L.ptr(nv)
            if normed:
                d.norm_acc = L.ptr(norm_acc)
                d.weight, d.bias, d.running_mean, d.running_var, d.num_batches_tracked = (C.cast(a_, C.c_void_p) for a_ in keep)
                d.norm_n_comp, d.norm_act, d.momentum, d.eps = n, L.ACT_IDS[act], 0.1, 1e-5
                d.skip, d.save_stat, d.x_out = L.ptr(skip), L.ptr(save), L.ptr(x_out)
            L.check(lib.phc_readout_fused_fwd(C.byref(d), sp), "phc_readout_fused_fwd")
        torch.cuda.synchronize()
        res[tag] = dict(x=x_out, save=save, rm=torch.cat(rm), rv=torch.cat(rv), nbt=torch.stack(nbt), gate=gate, out=out)
    for key in ("x", "save", "rm", "rv", "nbt"):
        assert torch.equal(res["fused"][key], res["separate"][key]), key
    for key in ("gate", "out"):
        a_, b_ = res["fused"][key].double(), res["separate"][key].double()
        assert float((a_ - b_).abs().max()) <= 2e-6 * max(float(b_.abs().max()), 1e-30), (key, float((a_ - b_).abs().max()), float(b_.abs().max()))


@pytest.mark.parametrize("B,sizes", [(7, "small"), (300, "molecules"), (40, "big")])
@pytest.mark.parametrize("normed,act,with_skip", [(False, "relu", False), (True, "relu", True), (True, "swish", False)])
@pytest.mark.parametrize("n,k", [(4, 32), (8, 32), (4, 64)])
def test_readout_in_one_launch_against_the_oracle(B, sizes, normed, act, with_skip, n, k):
    """phc_readout_fused_fwd DIRECTLY against the oracle: x = act(norm(y)) + skip (norm.py:30-35,
    models.py:208-215), out = soft_attention_pool(x, batch; PHMLinear h -> h, Linear h -> k) (pooling.py:57-66, layers.py:399-414) —
    the launch takes the composed gate operand M = H . Wg^T and bias c = b . Wg^T + bg, built here in float64 from the layer's own
    parameters; float64 oracle on the host, 1e-5 of each tensor's largest entry."""
    import ctypes as C
    from oracle import phc_oracle as O
    from phc_gnn_amd import lib as L
    lib = L.load()
    h, kg = n * k, k
    g = torch.Generator().manual_seed(B + h)
    hi = dict(small=5, molecules=45, big=300)[sizes]
    cnt = torch.randint(0, hi, (B,), generator=g)
    cnt[B // 2] = 0
    cnt[0] = max(int(cnt[0]), 1)
    batch_h = torch.repeat_interleave(torch.arange(B), cnt)
    N = int(batch_h.numel())
    if not lib.phc_readout_fused_supported(N, B, h, n, kg):
        pytest.skip("outside the fused form")
    batch = batch_h.to(DEV)
    nv = torch.tensor([N], dtype=torch.int32, device=DEV)
    sp = L.stream_ptr(torch.device(DEV))
    # the gate's two maps: PHMLinear(h -> h) with rule A [n, n, n], weights W [n, k, k], bias b [h]; Linear(h -> kg)
    A = torch.randn(n, n, n, generator=g, dtype=torch.float64) * 0.5
    W = torch.randn(n, k, k, generator=g, dtype=torch.float64) / k ** 0.5
    b = torch.randn(h, generator=g, dtype=torch.float64) * 0.1
    lin_w = torch.randn(kg, h, generator=g, dtype=torch.float64) / h ** 0.5
    lin_b = torch.randn(kg, generator=g, dtype=torch.float64) * 0.1
    H = O.kron_sum_H(A, W)                                                # [h, h]: y = x . H + b
    M = (H @ lin_w.t()).float().to(DEV).contiguous()
    c = (b @ lin_w.t() + lin_b).float().to(DEV)
    y_in = torch.randn(N, h, generator=g).to(DEV)
    skip = torch.randn(N, h, generator=g).to(DEV) if with_skip else None
    words = lib.phc_stat_acc_words(h)
    arr = lambda tsr: (C.c_void_p * len(tsr))(*[t.data_ptr() for t in tsr])
    split = lambda t: [t[cc * k:(cc + 1) * k] for cc in range(n)]
    w_h, b_h = 0.5 + torch.rand(h, generator=g), torch.randn(h, generator=g)
    w, bb = split(w_h.to(DEV)), split(b_h.to(DEV))
    norm_acc = torch.zeros(words, dtype=torch.int64, device=DEV)
    if normed:
        P = (torch.randn(h, h, generator=g) / h ** 0.5).to(DEV)
        L.check(lib.phc_gemm_prebuilt_stats(L.ptr(torch.randn(N, h, generator=g).to(DEV)), L.ptr(P), None, None, N, h, h, L.ptr(y_in), L.ptr(nv),
                                            L.ptr(norm_acc), None, None, sp), "producer")
    rm, rv = split(torch.zeros(h, device=DEV)), split(torch.ones(h, device=DEV))
    nbt = [torch.zeros((), dtype=torch.int64, device=DEV) for _ in range(n)]
    save = torch.zeros(6 * h, device=DEV)
    x_out = torch.zeros(N, h, device=DEV)
    gate = torch.zeros(N, kg, device=DEV)
    out = torch.full((B, h), 7.0, device=DEV)
    keep = (arr(w), arr(bb), arr(rm), arr(rv), arr(nbt))
    d = L.ReadoutDesc()
    d.x, d.batch_sorted_i64, d.N, d.num_graphs, d.h, d.n_comp, d.kg = L.ptr(y_in), L.ptr(batch), N, B, h, n, kg
    d.M, d.c, d.gate, d.out, d.n_valid_dev = L.ptr(M), L.ptr(c), L.ptr(gate), L.ptr(out), L.ptr(nv)
    if normed:
        d.norm_acc = L.ptr(norm_acc)
        d.weight, d.bias, d.running_mean, d.running_var, d.num_batches_tracked = (C.cast(a_, C.c_void_p) for a_ in keep)
        d.norm_n_comp, d.norm_act, d.momentum, d.eps = n, L.ACT_IDS[act], 0.1, 1e-5
        d.skip, d.save_stat, d.x_out = L.ptr(skip), L.ptr(save), L.ptr(x_out)
    L.check(lib.phc_readout_fused_fwd(C.byref(d), sp), "phc_readout_fused_fwd")
    torch.cuda.synchronize()
    x = y_in.double().cpu()
    if normed:
        rmh, rvh = [torch.zeros(k, dtype=torch.float64) for _ in range(n)], [torch.ones(k, dtype=torch.float64) for _ in range(n)]
        x = O.activation(act, O.batch_norm_naive(x, n, split(w_h.double()), split(b_h.double()), rmh, rvh, True, 0.1, 1e-5))
        if with_skip:
            x = x + skip.double().cpu()
        assert_rel(x_out, x, 1e-5, "x")
        assert_rel(torch.cat(rm), torch.cat(rmh), 1e-5, "running_mean")
        assert_rel(torch.cat(rv), torch.cat(rvh), 1e-5, "running_var")
    ref = O.soft_attention_pool(x, batch_h, A, W, b, lin_w, lin_b, n, num_graphs=B)
    pre = O.phm_linear(x, A, W, b) @ lin_w.t() + lin_b                    # the pre-sigmoid gate the launch leaves behind
    assert_rel(gate, pre, 1e-5, "gate (pre-sigmoid)")
    assert_rel(out, ref, 1e-5, "per-graph sums")


@pytest.mark.parametrize("workload", ["molhiv_b32_swish", "molhiv_b512_swish", "molhiv_b32", "zinc_small_256_swish", "molpcba_small_256_swish"])
@pytest.mark.parametrize("graph", [False, True])
def test_readout_in_one_launch_leaves_the_step_unchanged(workload, graph):
    """Loss and every gradient with ops.READOUT_FUSED on and off (a different fp32 summation order in the gate and in the per-graph
    sums: 2e-5 of each tensor's largest entry on the smooth models; the ReLU model's loss only — a gate at the kink may flip)."""
    from phc_gnn_amd import harness
    harness.WORKLOADS.setdefault("zinc_small_256_swish", dict(dataset="zinc", batch_size=48, phm_dim=8, hidden=256, layers=2, d_units=[64, 32],
                                                              target_dim=1, sc_type="last", regression=True, activation="swish"))
    harness.WORKLOADS.setdefault("molpcba_small_256_swish", dict(dataset="molpcba", batch_size=40, phm_dim=4, hidden=256, layers=2, d_units=[64, 32],
                                                                 target_dim=8, sc_type="first", nan_frac=0.5, activation="swish"))
    on = _step(workload, dict(READOUT_FUSED=True), graph)
    off = _step(workload, dict(READOUT_FUSED=False), graph)
    assert abs(float(on[0]) - float(off[0])) <= 2e-6 * abs(float(off[0])), (float(on[0]), float(off[0]))
    if "swish" in workload:
        for k in off[1]:
            ref = off[1][k]
            if float(ref.abs().max()) < 1e-7:                         # a bias in front of a batch-norm: zero up to round-off noise
                continue
            assert float((on[1][k] - ref).abs().max()) <= 2e-5 * max(float(ref.abs().max()), 1e-30), k
    for k in off[2]:
        ref = off[2][k].double()
        assert float((on[2][k].double() - ref).abs().max()) <= 1e-6 * max(float(ref.abs().max()), 1e-30), ("buffer", k)


def test_flat_table_image_is_not_taken_for_refreshed_because_of_a_recycled_python_id():
    """Regression (round 4): the per-step set of already-refreshed flat table images was keyed by id().  An image marked at the step's
    start and garbage-collected during the step leaves its id() to the next Python object — a new model's image that inherited it was
    skipped and embedded everything as zero for one step (seen as a loss 3-4 % off once in a few hundred runs of
    tests/test_gpu_gemm_hs.py::test_fused_mlp_norm_node_equals_the_separate_nodes[zinc_pub_phc4_4x180_nodrop])."""
    from phc_gnn_amd import ops
    g = torch.Generator().manual_seed(3)
    n, k, vocab = 4, 45, [5, 6, 2]
    tables = [[torch.randn(v, k, generator=g).to(DEV) for v in vocab] for _ in range(n)]
    fs = ops.TableSet(tables).vector_form(tables)
    assert isinstance(fs, ops.FlatTableSet)
    fs.image.zero_()
    keep = getattr(ops._FLAT_STEP, "packed", None)
    ops._FLAT_STEP.packed = {id(fs)}                                       # a dead object's id that this image happened to inherit
    try:
        fs.refresh_image()
    finally:
        ops._FLAT_STEP.packed = keep
    torch.cuda.synchronize()
    want = torch.cat([torch.cat([tables[c][f] for c in range(n)], 1) for f in range(len(vocab))], 0)
    assert torch.equal(fs.image, want)


@pytest.mark.parametrize("edges", [0, 3])
def test_a_batch_with_no_or_hardly_any_edges_takes_the_same_step_on_every_path(edges):
    """Edge case of the reference's own tests (empty inputs): a collated batch whose graphs have no bonds at all (or three in total).
    The round-4 fusions step aside (or run on empty rows); loss and gradients equal the unfused step's."""
    import dataclasses
    from phc_gnn_amd import harness, ops
    wl = "molhiv_b32_swish"
    gb = harness.make_batch(wl, seed=4)
    gb = dataclasses.replace(gb, edge_index=gb.edge_index[:, :edges].copy(), edge_attr=gb.edge_attr[:edges].copy())
    cfg = harness.StepConfig(optimizer=False, weight_decay=1e-3)
    res = []
    for on in (True, False):
        keep = {k: getattr(ops, k) for k in ("GEMM_AGGREGATE", "READOUT_FUSED", "NORM_INTO_AGGREGATE")}
        try:
            for k in keep:
                setattr(ops, k, on)
            model = harness.build_model(wl, DEV, seed=2).train()
            loss = harness.train_step(model, harness.to_device(gb, DEV), None, cfg)
            torch.cuda.synchronize()
            res.append((float(loss.detach()), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}))
        finally:
            for k, v in keep.items():
                setattr(ops, k, v)
    (l1, g1), (l0, g0) = res
    assert l1 == l1 and abs(l1 - l0) <= 2e-6 * abs(l0), (l1, l0)
    for k in g0:
        ref_ = g0[k]
        if float(ref_.abs().max()) < 1e-7:
            continue
        assert float((g1[k] - ref_).abs().max()) <= 2e-5 * float(ref_.abs().max()), k
    if edges == 0:
        return                          # (the reference itself cannot run this batch: models.py:238-240 reshapes an empty tensor with -1)
    # ... and it is the reference's step: the CPU oracle on the same state and batch (forward loss, no regulariser)
    from oracle import phc_oracle as O
    from tests.test_gpu_fullwidth import _oracle_kwargs
    model = harness.build_model(wl, "cpu", seed=2).train()
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    kw = harness.model_kwargs(harness.WORKLOADS[wl])
    x, ei, ea, bt, y = (torch.from_numpy(a_) for a_ in (gb.x, gb.edge_index, gb.edge_attr, gb.batch, gb.y))
    logits = O.skip_connect_add_forward(sd, x, ei, ea, bt, training=True, num_graphs=gb.num_graphs, **_oracle_kwargs(kw, x, ea))
    ref = float(O.bce_loss_masked(logits, y))
    mdl = harness.build_model(wl, DEV, seed=2).train()
    l_plain = float(harness.train_step(mdl, harness.to_device(gb, DEV), None, harness.StepConfig(optimizer=False, weight_decay=0.0)).detach())
    assert abs(l_plain - ref) <= 1e-5 * abs(ref), (l_plain, ref)


# ---- unfinished tensors stay inside: with forward hooks registered the model takes the reference's module-by-module sequence ---------
@pytest.mark.parametrize("workload", ["molhiv_b32", "molhiv_b32_swish"])
def test_forward_hooks_see_finished_activations(workload):
    """deferred norm / aggregate / gate state travels on tensors whose storage is filled by a LATER launch.  A forward hook
    on convs[i] / norms[i] / pooling must never see such a tensor: with hooks the model runs ref models.py:200-217 through every
    module's __call__.  Checked: the hooks fire, what they see equals the oracle's intermediate tensors (not uninitialised memory), and
    the hooked run's logits equal the fast path's."""
    from oracle import phc_oracle as O
    from phc_gnn_amd import harness
    wl = harness.WORKLOADS.get(workload)
    if wl is None:
        pytest.skip(f"no workload {workload}")
    model = harness.build_model(workload, DEV, seed=5).train()
    gb = harness.make_batch(workload, seed=2)
    data = harness.to_device(gb, DEV)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    fast = model(data).detach().clone()
    model.load_state_dict(sd)

    seen = {}
    hooks = []
    for i, (conv, norm) in enumerate(zip(model.convs, model.norms)):
        hooks.append(conv.register_forward_hook(lambda m, a, out, i=i: seen.__setitem__(("conv", i), out.detach().clone())))
        hooks.append(norm.register_forward_hook(lambda m, a, out, i=i: seen.__setitem__(("norm", i), out.detach().clone())))
    hooks.append(model.pooling.register_forward_hook(lambda m, a, out: seen.__setitem__(("pool", 0), out.detach().clone())))
    try:
        hooked = model(data).detach().clone()
    finally:
        for hk in hooks:
            hk.remove()
    torch.cuda.synchronize()
    L_ = len(model.convs)
    assert set(seen) == {("conv", i) for i in range(L_)} | {("norm", i) for i in range(L_)} | {("pool", 0)}
    for k, v in seen.items():
        assert torch.isfinite(v).all(), k
    scale = max(float(fast.abs().max()), 1e-30)
    assert float((hooked - fast).abs().max()) <= 1e-5 * scale

    # the oracle's own intermediates: norm_i's output, re-derived layer by layer from what the conv hook saw
    for i, norm in enumerate(model.norms):
        y = seen[("conv", i)].double().cpu()
        n, k = model.phm_dim, y.shape[1] // model.phm_dim
        ref = torch.empty_like(y)
        for c, bn in enumerate(norm.bn.bn if hasattr(norm.bn, "bn") else norm.bn):
            blk = y[:, c * k:(c + 1) * k]
            mean, var = blk.mean(0), blk.var(0, unbiased=False)
            ref[:, c * k:(c + 1) * k] = (blk - mean) / torch.sqrt(var + bn.eps) * bn.weight.detach().double().cpu() + bn.bias.detach().double().cpu()
        got = seen[("norm", i)].double().cpu()
        assert float((got - ref).abs().max()) <= 2e-5 * max(float(ref.abs().max()), 1e-30), i
    # after removing the hooks the fast path is back (deferred state allowed again) and agrees
    model.load_state_dict(sd)
    again = model(data).detach()
    assert torch.equal(again, fast)


def test_forward_hook_on_a_head_layer_fires():
    """the one-launch head never calls its layers — with a forward hook on one of them the head takes the per-layer
    sequence (PHMLinear.__call__), the hook sees that layer's output, and the logits stay the fast path's."""
    from phc_gnn_amd import harness
    model = harness.build_model("molhiv_b32_swish", DEV, seed=5).train()
    data = harness.to_device(harness.make_batch("molhiv_b32_swish", seed=2), DEV)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    fast = model(data).detach().clone()
    model.load_state_dict(sd)
    seen = []
    hk = model.downstream.affine[1].register_forward_hook(lambda m, a, out: seen.append(out.detach().clone()))
    try:
        hooked = model(data).detach().clone()
    finally:
        hk.remove()
    torch.cuda.synchronize()
    assert len(seen) == 1 and torch.isfinite(seen[0]).all() and seen[0].shape[0] == fast.shape[0]
    assert float((hooked - fast).abs().max()) <= 1e-5 * max(float(fast.abs().max()), 1e-30)


@pytest.mark.parametrize("n,k", [(4, 32), (4, 50), (8, 32)])
@pytest.mark.parametrize("N,E,hubs", [(1000, 2100, False), (333, 1500, True), (12874, 27650, False)])
@pytest.mark.parametrize("F,act,with_skip,add_self", [(3, "relu", True, True), (1, "swish", False, True), (2, "relu", True, False)])
def test_norm_folded_into_the_softmax_aggregation_equals_the_two_launches(N, E, hubs, F, act, with_skip, add_self, n, k):
    """phc_aggregate_softmax_norm_enc_fwd against phc_bn_fwd_from_partials followed by phc_aggregate_softmax_enc_fwd: x, save_stat,
    running statistics, the aggregate, its self-loop form and the log-sum-exp bit for bit (same walk, same online softmax order)."""
    import ctypes as C
    from phc_gnn_amd import lib as L, ops
    from phc_gnn_amd.graph import edge_structure
    lib = L.load()
    h = n * k
    vocab = [5, 6, 2, 3][:F]
    g = torch.Generator().manual_seed(N + F + h)
    ei = _random_graph(N, E, hubs, seed=N).to(DEV)
    bond = torch.stack([torch.randint(0, v, (E,), generator=g) for v in vocab], 1).to(DEV)
    tables = [[torch.randn(v, k, generator=g).to(DEV) for v in vocab] for _ in range(n)]
    ts = ops.TableSet(tables).vector_form(tables)
    if not lib.phc_aggregate_softmax_norm_supported(N, h, F, sum(vocab)):
        pytest.skip("outside the fused form")
    st = edge_structure(ei, N, N)
    st.prepare(for_backward=False, codes_for=(bond, ts.vocab_c, ts.F))
    rowptr, col, _ = st.csr
    codes = st.packed_codes(bond, ts.vocab_c, ts.F)
    valid = N - 7
    nv = torch.tensor([valid], dtype=torch.int32, device=DEV)
    sp = L.stream_ptr(torch.device(DEV))
    beta = torch.tensor([1.3], device=DEV)
    P = (torch.randn(h, h, generator=g) / h ** 0.5).to(DEV)
    y_in = torch.empty(N, h, device=DEV)
    skip = torch.randn(N, h, generator=g).to(DEV) if with_skip else None
    words = lib.phc_stat_acc_words(h)
    arr = lambda tsr: (C.c_void_p * len(tsr))(*[t.data_ptr() for t in tsr])
    w = [(0.5 + torch.rand(k, generator=g)).to(DEV) for _ in range(n)]
    bb = [torch.randn(k, generator=g).to(DEV) for _ in range(n)]
    norm_acc = torch.zeros(words, dtype=torch.int64, device=DEV)
    L.check(lib.phc_gemm_prebuilt_stats(L.ptr(torch.randn(N, h, generator=g).to(DEV)), L.ptr(P), None, None, N, h, h, L.ptr(y_in), L.ptr(nv),
                                        L.ptr(norm_acc), None, None, sp), "producer")
    res = {}
    for tag in ("separate", "fused"):
        rm = [torch.zeros(k, device=DEV) for _ in range(n)]
        rv = [torch.ones(k, device=DEV) for _ in range(n)]
        nbt = [torch.zeros((), dtype=torch.int64, device=DEV) for _ in range(n)]
        save = torch.zeros(6 * h, device=DEV)
        x_out = torch.zeros(N, h, device=DEV)
        out, lse = torch.empty(N, h, device=DEV), torch.empty(N, h, device=DEV)
        out_self = torch.empty(N, h, device=DEV) if add_self else None
        keep = (arr(w), arr(bb), arr(rm), arr(rv), arr(nbt))
        if tag == "separate":
            L.check(lib.phc_bn_fwd_from_partials(L.ptr(y_in), L.ptr(norm_acc), None, 0, keep[0], keep[1], keep[2], keep[3], keep[4], n, L.ptr(nv),
                                                 N, h, 0.1, 1e-5, L.ACT_IDS[act], L.ptr(skip), L.ptr(x_out), L.ptr(save), 0.0, 0, 0, None, sp),
                    "bn_fwd_from_partials")
            L.check(lib.phc_aggregate_softmax_enc_fwd(L.ptr(x_out), L.ptr(codes), L.ptr(ts.ptrs), ts.vocab_c, ts.F, ts.n, ts.k, L.ptr(rowptr),
                                                      L.ptr(col), L.ptr(beta), N, 0, L.ptr(out), L.ptr(out_self), L.ptr(lse), sp), "softmax")
        else:
            L.check(lib.phc_aggregate_softmax_norm_enc_fwd(L.ptr(y_in), L.ptr(norm_acc), keep[0], keep[1], keep[2], keep[3], keep[4], n, 0.1, 1e-5,
                                                           L.ACT_IDS[act], L.ptr(nv), L.ptr(skip), L.ptr(save), L.ptr(x_out), L.ptr(codes),
                                                           L.ptr(ts.ptrs), ts.vocab_c, ts.F, ts.n, ts.k, L.ptr(rowptr), L.ptr(col), L.ptr(beta), N,
                                                           L.ptr(out), L.ptr(out_self), L.ptr(lse), sp), "softmax norm")
        torch.cuda.synchronize()
        res[tag] = dict(x=x_out, save=save, rm=torch.cat(rm), rv=torch.cat(rv), nbt=torch.stack(nbt), out=out, lse=lse)
        if add_self:
            res[tag]["out_self"] = out_self
    for key in res["fused"]:
        a_, b_ = res["fused"][key], res["separate"][key]
        assert torch.equal(a_, b_), (key, float((a_.double() - b_.double()).abs().max()))


# ---- round 5: even shares in the step tail's dH family (GradJobDev.pair: the last splits of several tiles in one workgroup) ----
@pytest.mark.parametrize("workload", ["molhiv_b512", "molpcba_b1024"])
def test_paired_last_splits_of_the_tail_match_whole_split_counts(workload, tmp_path):
    """The planner's switch is read once per process: two child processes run the same eager step (scripts/dbg/tail_pair_step.py) with
    PHC_DH_PAIR=1 (the default; its plan must really pair: `pair=` > 1 in the plan dump) and 0.  Same loss bit for bit (the forward does
    not involve the tail); every gradient within summation-order round-off (the paired plan cuts the rows into other splits)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for pair in ("1", "0"):
        out = str(tmp_path / f"step_{pair}.pt")
        env = dict(os.environ, PHC_DH_PAIR=pair, PHC_DEBUG_PLANS="1")
        p = subprocess.run([sys.executable, os.path.join(root, "scripts", "dbg", "tail_pair_step.py"), workload, out], env=env, cwd=root,
                           capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        pairs = [int(l.rsplit("pair=", 1)[1]) for l in p.stderr.splitlines() if "[phc plan]" in l and "form=3" in l]
        assert pairs, "no plan dump"
        assert (max(pairs) > 1) == (pair == "1"), (pair, sorted(set(pairs)))
        res[pair] = torch.load(out)
    a, b = res["1"], res["0"]
    assert torch.equal(a["loss"], b["loss"])
    assert a["grads"].keys() == b["grads"].keys()
    for k in a["grads"]:
        ref = float(b["grads"][k].abs().max())
        err = float((a["grads"][k] - b["grads"][k]).abs().max())
        assert err <= 1e-5 * ref + 1e-7, (k, err, ref)
