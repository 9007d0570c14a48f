"""GPU parity tests of the C-ABI operators against the golden vectors and the
CPU oracle.  Tolerance: north_star's 1e-5 relative fp32, measured per tensor as
max|Δ| / max|ref| (tests/_util.py)."""
import numpy as np
import pytest
import torch

from tests._util import TOL, assert_close, case, load

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from phc_gnn_amd import ops as _ops
    return _ops


def test_library_reports_gfx950():
    import ctypes as C
    from phc_gnn_amd import lib
    l = lib.load()
    cu, lds = C.c_int(), C.c_int()
    arch = C.create_string_buffer(64)
    lib.check(l.phc_device_info(C.byref(cu), C.byref(lds), arch, 64), "phc_device_info")
    assert arch.value.decode().startswith("gfx950"), arch.value
    assert cu.value == 256


def test_build_H_matches_golden(ops):
    z = load("g2_kronecker_rules")
    for i in range(int(z["num_kron"])):
        A = torch.from_numpy(z[f"k{i}::A"]).to(DEV)
        W = torch.from_numpy(z[f"k{i}::B"]).to(DEV)
        assert_close(ops.phm_build_H(A, W), z[f"k{i}::H"], 1e-6, f"H case {i}")
    W = torch.from_numpy(z["quat::W"]).to(DEV)
    A = torch.from_numpy(z["rule::4"]).to(DEV)
    # known answer: standard quaternion rule => H == RealP(W0..W3) exactly (signed copies of W)
    assert torch.equal(ops.phm_build_H(A, W).cpu(), torch.from_numpy(z["quat::H"]))


def test_phm_linear_golden_fwd_bwd(ops):
    z = load("g1_phm_linear")
    for ci in range(int(z["num_cases"])):
        c = case(z, ci)
        x = c["x"].to(DEV).requires_grad_()
        A = c["A"].to(DEV).requires_grad_()
        W = c["W"].to(DEV).requires_grad_()
        b = c["b"].to(DEV).requires_grad_() if "b" in c else None
        y = ops.phm_linear(x, A, W, b)
        assert_close(y, c["y"], TOL, f"case {ci} y")
        grads = torch.autograd.grad(y, [x, W, A] + ([b] if b is not None else []), c["dy"].to(DEV))
        for name, g in zip(("dx", "dW", "dA", "db"), grads):
            assert_close(g, c[name], TOL, f"case {ci} {name} meta={c['meta'].tolist()}")


def test_phm_linear_residual_and_partial_grads(ops):
    from oracle import phc_oracle as O
    g = torch.Generator().manual_seed(3)
    n, kin, kout, N = 4, 32, 32, 300
    x = torch.randn(N, n * kin, generator=g)
    A = torch.randn(n, n, n, generator=g)
    W = torch.randn(n, kin, kout, generator=g) * 0.1
    res = torch.randn(N, n * kout, generator=g)
    ref = O.phm_linear(x, A, W, None) + res
    xd, Ad, Wd, rd = (t.to(DEV).requires_grad_() for t in (x, A, W, res))
    Ad.requires_grad_(False)                      # learn_phm=False: no dA requested
    y = ops.phm_linear(xd, Ad, Wd, None, rd)
    assert_close(y, ref, TOL)
    dy = torch.randn(N, n * kout, generator=g)
    gx, gW, gr = torch.autograd.grad(y, [xd, Wd, rd], dy.to(DEV))
    ex = O.phm_linear_backward(x, A, W, dy)
    assert_close(gx, ex[0], TOL)
    assert_close(gW, ex[1], TOL)
    assert torch.equal(gr.cpu(), dy)


def test_phm_linear_is_deterministic(ops):
    g = torch.Generator().manual_seed(4)
    x = torch.randn(5000, 128, generator=g).to(DEV)
    A = torch.randn(4, 4, 4, generator=g).to(DEV).requires_grad_()
    W = torch.randn(4, 32, 32, generator=g).to(DEV).requires_grad_()
    outs = []
    for _ in range(3):
        y = ops.phm_linear(x, A, W, None)
        gA, gW = torch.autograd.grad(y.square().sum(), [A, W])
        outs.append((y.detach().clone(), gA.clone(), gW.clone()))
    for o in outs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(o, outs[0]))


def test_deferred_param_grads_match_oracle_and_per_layer_path(ops):
    """phc_phm_param_grads (all layers' dW/dA/db in three batched launches, filed by deferred_param_grads())
    against the oracle and against the per-layer backward: ragged shapes, n = 1, scalar (non-float4) widths,
    a layer without bias, a frozen phm_rule, more jobs than one launch holds, a parameter shared by two layers
    and accumulation into an existing .grad."""
    from oracle import phc_oracle as O
    g = torch.Generator().manual_seed(11)
    shapes = [(4, 32, 32, 1500), (4, 32, 32, 70), (2, 16, 24, 333), (1, 40, 8, 90), (3, 5, 7, 129), (8, 16, 16, 64),
              (4, 32, 8, 1), (4, 64, 64, 700), (8, 32, 48, 333)] + \
             [(4, 8, 8, 40 + 3 * i) for i in range(14)]          # 23 jobs > 16 per launch; two of them on 128 x 128 tiles
    layers = []
    for li, (n, kin, kout, N) in enumerate(shapes):
        layers.append(dict(x=torch.randn(N, n * kin, generator=g), A=torch.randn(n, n, n, generator=g),
                           W=torch.randn(n, kin, kout, generator=g) * 0.2,
                           b=None if li == 2 else torch.randn(n * kout, generator=g),
                           dy=torch.randn(N, n * kout, generator=g), learn_A=li != 1))

    def run(deferred, shared=False):
        prm = []
        for l in layers:
            A = l["A"].to(DEV).requires_grad_(l["learn_A"])
            W = l["W"].to(DEV).requires_grad_()
            b = l["b"].to(DEV).requires_grad_() if l["b"] is not None else None
            prm.append((A, W, b))
        prm[0][1].grad = torch.full_like(prm[0][1], 0.5)                         # existing .grad: accumulate
        xs = [l["x"].to(DEV).requires_grad_() for l in layers]
        ctx = ops.deferred_param_grads() if deferred else __import__("contextlib").nullcontext()
        with ctx:
            total = 0.0
            for l, x, (A, W, b) in zip(layers, xs, prm):
                total = total + (ops.phm_linear(x, A, W, b) * l["dy"].to(DEV)).sum()
            if shared:                                                             # layer 0's parameters used twice
                total = total + (ops.phm_linear(xs[0] * 2.0, *prm[0]) * layers[0]["dy"].to(DEV)).sum()
            total.backward()
        return prm, xs

    prm_d, xs_d = run(True)
    prm_i, xs_i = run(False)
    for li, (l, pd, pi) in enumerate(zip(layers, prm_d, prm_i)):
        dx, dW, dA, db = O.phm_linear_backward(l["x"], l["A"], l["W"], l["dy"])
        assert_close(xs_d[li].grad, dx, TOL, f"layer {li} dx")
        assert_close(pd[1].grad, dW + (0.5 if li == 0 else 0.0), TOL, f"layer {li} dW")
        if l["learn_A"]:
            assert_close(pd[0].grad, dA, TOL, f"layer {li} dA")
        else:
            assert pd[0].grad is None
        if l["b"] is not None:
            assert_close(pd[2].grad, db, TOL, f"layer {li} db")
        for a, b_ in zip(pd, pi):                                                  # batched == per-layer path
            if a is not None and a.grad is not None:
                assert_close(a.grad, b_.grad, 2e-6, f"layer {li} batched vs per-layer")
    prm_s, _ = run(True, shared=True)
    dx, dW, dA, db = O.phm_linear_backward(layers[0]["x"], layers[0]["A"], layers[0]["W"], layers[0]["dy"])
    assert_close(prm_s[0][1].grad, 3.0 * dW + 0.5, TOL, "shared layer dW")         # x and 2x: dW + 2 dW
    assert_close(prm_s[0][2].grad, 2.0 * db, TOL, "shared layer db")
    # outside the block nothing is deferred; a backward that runs after the block exited takes the per-layer path
    A, W, b = (t.detach().clone().requires_grad_() for t in prm_d[0])
    with ops.deferred_param_grads():
        y = ops.phm_linear(layers[0]["x"].to(DEV), A, W, b)
    y.backward(layers[0]["dy"].to(DEV))
    assert_close(W.grad, dW, TOL, "late backward dW")


@pytest.mark.parametrize("n,kin,kout,N", [(4, 50, 50, 3000), (4, 45, 45, 1200), (5, 40, 40, 2500), (4, 45, 20, 900), (5, 40, 26, 700),
                                          (4, 50, 32, 1100), (4, 32, 50, 640)])
def test_param_grads_on_partial_register_tiles(ops, n, kin, kout, N):
    """Widths of the reference's published models (200 = 4 x 50 = 5 x 40, 180 = 4 x 45, heads 80 / 130 / 128): dH = x^T dy on the
    register-resident tiles whose last tile is partial, per-layer (phc_phm_linear_bwd) and batched (phc_phm_param_grads) — dW, dA,
    db and dx against the oracle."""
    from oracle import phc_oracle as O
    g = torch.Generator().manual_seed(n * 1000 + kin + kout)
    x = torch.randn(N, n * kin, generator=g)
    A = torch.randn(n, n, n, generator=g)
    W = torch.randn(n, kin, kout, generator=g) * 0.2
    b = torch.randn(n * kout, generator=g)
    dy = torch.randn(N, n * kout, generator=g)
    dx, dW, dA, db = O.phm_linear_backward(x, A, W, dy)
    for deferred in (False, True):
        xd = x.to(DEV).requires_grad_()
        Ad, Wd, bd = (t.to(DEV).requires_grad_() for t in (A, W, b))
        ctx = ops.deferred_param_grads() if deferred else __import__("contextlib").nullcontext()
        with ctx:
            (ops.phm_linear(xd, Ad, Wd, bd) * dy.to(DEV)).sum().backward()
        what = "batched" if deferred else "per-layer"
        assert_close(xd.grad, dx, TOL, f"{what} dx")
        assert_close(Wd.grad, dW, TOL, f"{what} dW")
        assert_close(Ad.grad, dA, TOL, f"{what} dA")
        assert_close(bd.grad, db, TOL, f"{what} db")


def _ref_sorted(key, other, num_keys):
    order = np.argsort(key, kind="stable")
    rowptr = np.zeros(num_keys + 1, dtype=np.int64)
    np.add.at(rowptr, key + 1, 1)
    return np.cumsum(rowptr), (other[order] if other is not None else order), order


@pytest.mark.parametrize("kind", ["mol", "ragged", "random_big", "empty", "random_huge", "skewed_huge", "hubs_big", "hubs_huge"])
def test_csr_build_is_the_stable_sort(kind):
    from phc_gnn_amd import synth
    from phc_gnn_amd.graph import EdgeStructure
    if kind == "mol":
        b = synth.molecule_batch(64, seed=3, isolated_frac=0.05)
        ei, n = b.edge_index, b.num_nodes
    elif kind == "ragged":
        ei, n = synth.ragged_edges(500, seed=1, hub_degree=700), 500
    elif kind == "random_big":
        ei, n = synth.random_edges(200_000, 1_000_000, seed=2), 200_000
    elif kind == "random_huge":                       # the two-level (bucketed) build: >= 64 K keys and >= 1 M items
        ei, n = synth.random_edges(300_000, 2_000_000, seed=5), 300_000
    elif kind == "skewed_huge":                       # one key range 500 x denser than the rest, a 40 K-item hub row, empty tail rows
        n = 400_123
        rng = np.random.default_rng(6)
        dst = np.concatenate([rng.integers(0, 1000, 600_000), rng.integers(0, 300_000, 900_000), np.full(40_000, 77)])
        src = rng.integers(0, n, dst.shape[0])
        p = rng.permutation(dst.shape[0])
        ei = np.stack([src[p], dst[p]]).astype(np.int64)
    elif kind in ("hubs_big", "hubs_huge"):
        # power-law-like rows: hubs of 3 000 / 8 192 / 8 193 / 20 000 (/ 150 000) items next to short rows — the workgroup-wide
        # bitonic sort of rows longer than 2 048 items (LDS up to 8 192, in place through L2 beyond), in the plain build
        # ("hubs_big": < 1 M items) and in the two-level one ("hubs_huge")
        n = 50_021 if kind == "hubs_big" else 300_017
        rng = np.random.default_rng(8)
        hubs = [(11, 3000), (n - 1, 8192), (n // 2, 8193), (4097, 20_000)] + ([(123_456, 150_000)] if kind == "hubs_huge" else [])
        dst = np.concatenate([rng.integers(0, n, 400_000 if kind == "hubs_big" else 1_200_000)] + [np.full(d, k) for k, d in hubs])
        src = rng.integers(0, n, dst.shape[0])
        src[-1000:] = 7                                    # and a hub on the source side of the backward view
        p = rng.permutation(dst.shape[0])
        ei = np.stack([src[p], dst[p]]).astype(np.int64)
    else:
        ei, n = np.zeros((2, 0), dtype=np.int64), 17
    st = EdgeStructure(torch.from_numpy(ei).to(DEV), n)
    for (rowptr, col, perm), key, other in ((st.csr, ei[1], ei[0]), (st.csc, ei[0], ei[1])):
        r_ref, c_ref, p_ref = _ref_sorted(key, other, n)
        assert np.array_equal(rowptr.cpu().numpy(), r_ref)
        assert np.array_equal(perm.cpu().numpy(), p_ref)
        assert np.array_equal(col.cpu().numpy(), c_ref)


ACTS = ["identity", "relu", "lrelu", "elu", "swish", "selu"]


@pytest.mark.parametrize("h", [128, 256, 64, 32, 200, 512, 6])
@pytest.mark.parametrize("graph", ["mol", "ragged"])
def test_aggregate_matches_oracle(ops, h, graph):
    from oracle import phc_oracle as O
    from phc_gnn_amd import synth
    g = torch.Generator().manual_seed(h)
    if graph == "mol":
        b = synth.molecule_batch(40, seed=7, isolated_frac=0.1)
        ei, n = torch.from_numpy(b.edge_index), b.num_nodes
    else:
        ei, n = torch.from_numpy(synth.ragged_edges(300, seed=8, hub_degree=400)), 300
    E = ei.shape[1]
    x = torch.randn(n, h, generator=g)
    ea = torch.randn(E, h, generator=g)
    dy = torch.randn(n, h, generator=g)
    for act in (ACTS if h in (128, 6) else ["identity", "relu"]):
        for add_self in (True, False):
            xr, er = x.clone().requires_grad_(), ea.clone().requires_grad_()
            ref = O.propagate_add(xr, ei, er, act)
            if add_self:
                ref = ref + xr
            rgx, rge = torch.autograd.grad(ref, [xr, er], dy)
            xd, ed = x.to(DEV).requires_grad_(), ea.to(DEV).requires_grad_()
            out = ops.aggregate_sum(xd, ei.to(DEV), ed, act, add_self)
            assert_close(out, ref, TOL, f"fwd {act} self={add_self}")
            gx, ge = torch.autograd.grad(out, [xd, ed], dy.to(DEV))
            assert_close(gx, rgx, TOL, f"dx {act} self={add_self}")
            assert_close(ge, rge, TOL, f"dea {act} self={add_self}")


def test_aggregate_without_edge_attr_and_reproducible(ops):
    from oracle import phc_oracle as O
    from phc_gnn_amd import synth
    ei = torch.from_numpy(synth.random_edges(20_000, 100_000, seed=5))
    g = torch.Generator().manual_seed(1)
    x = torch.randn(20_000, 128, generator=g)
    ref = O.scatter_sum_rows(x.index_select(0, ei[0]), ei[1], 20_000)
    a = ops.aggregate_sum(x.to(DEV), ei.to(DEV), None)
    b = ops.aggregate_sum(x.to(DEV), ei.to(DEV).clone(), None)
    assert_close(a, ref, TOL)
    assert torch.equal(a, b)


def test_global_add_pool(ops):
    from oracle import phc_oracle as O
    from phc_gnn_amd import synth
    b = synth.molecule_batch(50, seed=11)
    batch = torch.from_numpy(b.batch)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(b.num_nodes, 128, generator=g)
    dy = torch.randn(50, 128, generator=g)
    xd = x.to(DEV).requires_grad_()
    out = ops.global_add_pool(xd, batch.to(DEV), 50)
    assert_close(out, O.global_add_pool(x, batch, 50), TOL)
    (gx,) = torch.autograd.grad(out, [xd], dy.to(DEV))
    assert torch.equal(gx.cpu(), dy.index_select(0, batch))
    # unsorted segment ids (the reference test_qtensor_scatter.py:42-49 uses random ids)
    idx = torch.randint(0, 128, (2048,), generator=g)
    xx = torch.randn(2048, 64, generator=g)
    assert_close(ops.global_add_pool(xx.to(DEV), idx.to(DEV), 128), O.global_add_pool(xx, idx, 128), TOL)


def test_cpu_tensors_are_rejected(ops):
    from phc_gnn_amd.lib import PhcError
    with pytest.raises(PhcError):
        ops.phm_linear(torch.randn(4, 8), torch.randn(2, 2, 2), torch.randn(2, 4, 4), None)


# ---- fused integer encoders --------------------------------------------------
def _make_tables(n, vocab, k, gen):
    return [[torch.randn(v, k, generator=gen) for v in vocab] for _ in range(n)]


def _oracle_encode(idx, tables):
    from oracle import phc_oracle as O
    n = len(tables)
    p = {f"encoders.{c}.embeddings.{f}.weight": t for c, comp in enumerate(tables) for f, t in enumerate(comp)}
    return O.phm_encoder(idx, p, "", n, len(tables[0])).reshape(idx.shape[0], -1)


@pytest.mark.parametrize("n,k,vocab_a,vocab_b", [(4, 50, [119, 4, 12, 12, 10, 6, 6, 2, 2], [5, 6, 2]), (4, 45, [28], [4]), (5, 26, [28], [4])])
@pytest.mark.parametrize("flat", [True, False])
def test_encoders_with_component_size_not_a_multiple_of_four(ops, n, k, vocab_a, vocab_b, flat):
    """The reference's published widths (200 = 4 x 50, 180 = 4 x 45; a 130-wide head layer = 5 x 26): the encoders run on the flat
    one-component image of their tables (ops.FlatTableSet: float4 rows whatever k is) — atom encoder, sum aggregation and softmax
    aggregation with the bond embedding read in place, outputs and EVERY table gradient against the oracle; ``flat=False``: the same
    through the scalar fall-back kernels."""
    from oracle import phc_oracle as O
    from phc_gnn_amd import synth
    g = torch.Generator().manual_seed(n * 100 + k)
    old = ops.FLAT_TABLES
    ops.FLAT_TABLES = flat
    try:
        N, h = 700, n * k
        ei = torch.from_numpy(synth.ragged_edges(N, seed=9, hub_degree=60))
        E = ei.shape[1]
        idx = torch.stack([torch.randint(0, v, (N,), generator=g) for v in vocab_a], 1)
        bond = torch.stack([torch.randint(0, v, (E,), generator=g) for v in vocab_b], 1)
        ta, tb = _make_tables(n, vocab_a, k, g), _make_tables(n, vocab_b, k, g)
        dy = torch.randn(N, h, generator=g)
        # oracle
        ra = [[t.clone().requires_grad_() for t in comp] for comp in ta]
        rb = [[t.clone().requires_grad_() for t in comp] for comp in tb]
        beta_r = torch.tensor(1.3, requires_grad=True)
        x_r = _oracle_encode(idx, ra)
        ea_r = _oracle_encode(bond, rb)
        s_r = O.propagate_add(x_r, ei, ea_r, "identity") + x_r
        m_r = O.softmax_aggregate(x_r, ei, ea_r, beta_r, "identity")
        flat_r = [t for comp in ra for t in comp] + [t for comp in rb for t in comp]
        # engine
        da = [[t.to(DEV).requires_grad_() for t in comp] for comp in ta]
        db = [[t.to(DEV).requires_grad_() for t in comp] for comp in tb]
        beta_d = torch.tensor(1.3, device=DEV, requires_grad=True)
        tsa, tsb = ops.TableSet(da), ops.TableSet(db)
        x_d = ops.encode(idx.to(DEV), tsa, da)
        assert_close(x_d, x_r, TOL, "encode")
        m_d = ops.aggregate_softmax_encoded(x_d, ei.to(DEV), bond.to(DEV), tsb, db, beta_d, "identity") if ((flat and h % 4 == 0) or k % 4 == 0) else None
        outs_d, outs_r = [], []
        if s_r is not None:
            s_d = ops.aggregate_sum_encoded(x_d, ei.to(DEV), bond.to(DEV), tsb, db, "identity", add_self_loops=True)
            assert_close(s_d, s_r, 2e-5, "sum aggregation")
            outs_d.append(s_d); outs_r.append(s_r)
        if m_d is not None:
            assert_close(m_d, m_r, 2e-5, "softmax aggregation")
            outs_d.append(m_d); outs_r.append(m_r)
        loss_d = sum((o * dy.to(DEV)).sum() for o in outs_d)
        loss_r = sum((o * dy).sum() for o in outs_r)
        flat_d = [t for comp in da for t in comp] + [t for comp in db for t in comp]
        gd = torch.autograd.grad(loss_d, flat_d + ([beta_d] if m_d is not None else []))
        gr = torch.autograd.grad(loss_r, flat_r + ([beta_r] if m_d is not None else []))
        for i, (a, r) in enumerate(zip(gd, gr)):
            assert_close(a, r, 3e-5, f"gradient {i}")
    finally:
        ops.FLAT_TABLES = old


@pytest.mark.parametrize("n,k,vocab", [(4, 32, [119, 4, 12, 12, 10, 6, 6, 2, 2]), (8, 32, [28]), (2, 6, [5, 3]),
                                       (4, 128, [119, 4, 12, 12, 10, 6, 6, 2, 2])])
def test_encode_fwd_bwd(ops, n, k, vocab):
    g = torch.Generator().manual_seed(n * 100 + k)
    N = 3000
    idx = torch.stack([torch.randint(0, v, (N,), generator=g) for v in vocab], 1)
    tables = _make_tables(n, vocab, k, g)
    ref_t = [[t.clone().requires_grad_() for t in comp] for comp in tables]
    ref = _oracle_encode(idx, ref_t)
    dy = torch.randn(N, n * k, generator=g)
    ref_g = torch.autograd.grad(ref, [t for comp in ref_t for t in comp], dy)
    dev_t = [[t.to(DEV).requires_grad_() for t in comp] for comp in tables]
    ts = ops.TableSet(dev_t)
    out = ops.encode(idx.to(DEV), ts, dev_t)
    assert_close(out, ref, TOL, "encode fwd")
    got = torch.autograd.grad(out, [t for comp in dev_t for t in comp], dy.to(DEV))
    for a, b in zip(got, ref_g):
        assert_close(a, b, TOL, "table grad")
    got2 = torch.autograd.grad(ops.encode(idx.to(DEV), ts, dev_t), [t for comp in dev_t for t in comp], dy.to(DEV))
    assert all(torch.equal(a, b) for a, b in zip(got, got2)), "table gradients must be bitwise reproducible"


@pytest.mark.parametrize("act", ["identity", "relu", "swish"])
@pytest.mark.parametrize("n,k,vocab", [(4, 32, [5, 6, 2]), (8, 32, [4]), (4, 64, [5, 6, 2])])
def test_aggregate_encoded_matches_oracle(ops, act, n, k, vocab):
    from oracle import phc_oracle as O
    from phc_gnn_amd import synth
    g = torch.Generator().manual_seed(17)
    b = synth.molecule_batch(48, seed=21, isolated_frac=0.05)
    ei, N = torch.from_numpy(b.edge_index), b.num_nodes
    E, h = ei.shape[1], n * k
    bond = torch.stack([torch.randint(0, v, (E,), generator=g) for v in vocab], 1)
    tables = _make_tables(n, vocab, k, g)
    x = torch.randn(N, h, generator=g)
    dy = torch.randn(N, h, generator=g)
    for add_self in (True, False):
        ref_t = [[t.clone().requires_grad_() for t in comp] for comp in tables]
        xr = x.clone().requires_grad_()
        ref = O.propagate_add(xr, ei, _oracle_encode(bond, ref_t), act)
        if add_self:
            ref = ref + xr
        rg = torch.autograd.grad(ref, [xr] + [t for comp in ref_t for t in comp], dy)
        dev_t = [[t.to(DEV).requires_grad_() for t in comp] for comp in tables]
        xd = x.to(DEV).requires_grad_()
        ts = ops.TableSet(dev_t)
        assert ts.fusable_into_aggregate
        out = ops.aggregate_sum_encoded(xd, ei.to(DEV), bond.to(DEV), ts, dev_t, act, add_self)
        assert_close(out, ref, TOL, f"fused fwd {act}")
        gg = torch.autograd.grad(out, [xd] + [t for comp in dev_t for t in comp], dy.to(DEV))
        assert_close(gg[0], rg[0], TOL, f"fused dx {act}")
        for a, r in zip(gg[1:], rg[1:]):
            assert_close(a, r, 2e-5, f"fused table grad {act}")


def test_aggregate_encoded_large_graph_takes_the_lds_table_kernel(ops):
    """Above 32 768 destination rows phc_aggregate_encp_fwd stages the tables in LDS (persistent workgroups) instead of
    the one-wavefront-per-row direct form the molecule-sized cases above run: same result."""
    from oracle import phc_oracle as O
    from phc_gnn_amd import synth
    g = torch.Generator().manual_seed(15)
    n, k, vocab, N, E = 4, 8, [5, 6, 2], 40000, 120000
    ei = torch.from_numpy(synth.random_edges(N, E, seed=6))
    bond = torch.stack([torch.randint(0, v, (E,), generator=g) for v in vocab], 1)
    tables = _make_tables(n, vocab, k, g)
    x = torch.randn(N, n * k, generator=g)
    dev_t = [[t.to(DEV) for t in comp] for comp in tables]
    ts = ops.TableSet(dev_t)
    assert ts.fusable_into_aggregate
    out = ops.aggregate_sum_encoded(x.to(DEV), ei.to(DEV), bond.to(DEV), ts, dev_t, "relu", True)
    ref = O.propagate_add(x, ei, _oracle_encode(bond, tables), "relu") + x
    assert_close(out, ref, TOL)


def test_aggregate_encoded_falls_back_when_tables_exceed_lds(ops):
    from oracle import phc_oracle as O
    from phc_gnn_amd import synth
    g = torch.Generator().manual_seed(5)
    n, k, vocab = 4, 64, [40, 30]            # 70 rows * 256 cols * 4 B = 70 KiB > 48 KiB
    ei = torch.from_numpy(synth.ragged_edges(200, seed=3, hub_degree=100))
    E = ei.shape[1]
    bond = torch.stack([torch.randint(0, v, (E,), generator=g) for v in vocab], 1)
    tables = _make_tables(n, vocab, k, g)
    x = torch.randn(200, n * k, generator=g)
    dev_t = [[t.to(DEV) for t in comp] for comp in tables]
    ts = ops.TableSet(dev_t)
    assert not ts.fusable_into_aggregate
    out = ops.aggregate_sum_encoded(x.to(DEV), ei.to(DEV), bond.to(DEV), ts, dev_t, "relu", True)
    ref = O.propagate_add(x, ei, _oracle_encode(bond, tables), "relu") + x
    assert_close(out, ref, TOL)


# ---- fused batch-norm -----------------------------------------------------------
@pytest.mark.parametrize("n,h,N", [(4, 128, 3000), (8, 256, 777), (3, 30, 50), (4, 64, 1), (5, 200, 129), (4, 128, 70000)])
@pytest.mark.parametrize("act", ["identity", "relu", "swish"])
def test_batch_norm_act_matches_oracle(ops, n, h, N, act):
    from oracle import phc_oracle as O
    if N == 1:
        pytest.skip("BatchNorm1d refuses a single row in training mode (so does the reference)")
    g = torch.Generator().manual_seed(n * h + N)
    k = h // n
    x = torch.randn(N, h, generator=g) * 2 + 0.5
    res = torch.randn(N, h, generator=g)
    dy = torch.randn(N, h, generator=g)
    w = [torch.rand(k, generator=g) + 0.5 for _ in range(n)]
    b = [torch.randn(k, generator=g) * 0.1 for _ in range(n)]
    rm = [torch.randn(k, generator=g) * 0.1 for _ in range(n)]
    rv = [torch.rand(k, generator=g) + 0.5 for _ in range(n)]
    # oracle
    xr, rr = x.clone().requires_grad_(), res.clone().requires_grad_()
    wr, br = [t.clone().requires_grad_() for t in w], [t.clone().requires_grad_() for t in b]
    rm_r, rv_r = [t.clone() for t in rm], [t.clone() for t in rv]
    ref = O.activation(act, O.batch_norm_naive(xr, n, wr, br, rm_r, rv_r, True)) + rr
    rg = torch.autograd.grad(ref, [xr, rr] + wr + br, dy)
    # engine
    xd, rd = x.to(DEV).requires_grad_(), res.to(DEV).requires_grad_()
    wd, bd = [t.to(DEV).requires_grad_() for t in w], [t.to(DEV).requires_grad_() for t in b]
    rm_d, rv_d = [t.to(DEV) for t in rm], [t.to(DEV) for t in rv]
    nbt = [torch.zeros((), dtype=torch.int64, device=DEV) for _ in range(n)]
    out = ops.batch_norm_act(xd, wd, bd, rm_d, rv_d, nbt, True, 0.1, 1e-5, act, rd)
    assert_close(out, ref, TOL, "bn fwd")
    gg = torch.autograd.grad(out, [xd, rd] + wd + bd, dy.to(DEV))
    assert_close(gg[0], rg[0], 2e-5, "bn dx")
    assert torch.equal(gg[1].cpu(), dy)
    for a, r in zip(gg[2:], rg[2:]):
        assert_close(a, r, 3e-5, "bn dw/db")
    for a, r in zip(rm_d + rv_d, rm_r + rv_r):
        assert_close(a, r, TOL, "running stats")
    assert all(int(t) == 1 for t in nbt)
    out2 = ops.batch_norm_act(xd, wd, bd, [t.clone() for t in rm_d], [t.clone() for t in rv_d], None, True, 0.1, 1e-5, act, rd)
    assert torch.equal(out2, out)                            # a second call agrees bit for bit (fixed merge order)
    # eval mode uses the running statistics
    ev = ops.batch_norm_act(x.to(DEV), wd, bd, rm_d, rv_d, None, False, 0.1, 1e-5, act, None)
    ref_ev = O.activation(act, O.batch_norm_naive(x, n, w, b, rm_r, rv_r, False))
    assert_close(ev, ref_ev, TOL, "bn eval")


def test_batch_norm_padding_rows_are_inert(ops):
    """rows >= n_valid: outside the statistics, zero output, zero gradient."""
    from oracle import phc_oracle as O
    g = torch.Generator().manual_seed(9)
    n, h, N, valid = 4, 128, 1000, 730
    k = h // n
    x = torch.randn(N, h, generator=g)
    dy = torch.randn(N, h, generator=g)
    w = [torch.rand(k, generator=g) + 0.5 for _ in range(n)]
    b = [torch.randn(k, generator=g) * 0.1 for _ in range(n)]
    rm = [torch.zeros(k) for _ in range(n)]
    rv = [torch.ones(k) for _ in range(n)]
    xr = x[:valid].clone().requires_grad_()
    wr, br = [t.clone().requires_grad_() for t in w], [t.clone().requires_grad_() for t in b]
    ref = torch.relu(O.batch_norm_naive(xr, n, wr, br, [t.clone() for t in rm], [t.clone() for t in rv], True))
    rg = torch.autograd.grad(ref, [xr] + wr + br, dy[:valid])
    xd = x.to(DEV).requires_grad_()
    wd, bd = [t.to(DEV).requires_grad_() for t in w], [t.to(DEV).requires_grad_() for t in b]
    nv = torch.tensor(valid, dtype=torch.int32, device=DEV)
    out = ops.batch_norm_act(xd, wd, bd, [t.to(DEV) for t in rm], [t.to(DEV) for t in rv], None, True, 0.1, 1e-5, "relu",
                             None, nv)
    assert_close(out[:valid], ref, TOL)
    assert float(out[valid:].abs().max()) == 0.0
    gg = torch.autograd.grad(out, [xd] + wd + bd, dy.to(DEV))
    assert_close(gg[0][:valid], rg[0], 2e-5)
    assert float(gg[0][valid:].abs().max()) == 0.0
    for a, r in zip(gg[1:], rg[1:]):
        assert_close(a, r, 3e-5)


def test_sorted_segment_fast_path_equals_sorting_path(ops):
    from phc_gnn_amd import synth
    b = synth.molecule_batch(70, seed=13)
    batch = torch.from_numpy(b.batch).to(DEV)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(b.num_nodes, 128, generator=g).to(DEV).requires_grad_()
    dy = torch.randn(72, 128, generator=g).to(DEV)                       # two trailing empty graphs
    a = ops.global_add_pool(x, batch, 72, is_sorted=True)
    c = ops.global_add_pool(x, batch.clone(), 72, is_sorted=False)
    assert torch.equal(a, c)
    assert float(a[70:].abs().max()) == 0.0
    ga, = torch.autograd.grad(a, [x], dy)
    gc_, = torch.autograd.grad(c, [x], dy)
    assert torch.equal(ga, gc_)


def test_dense_linear_on_phm_kernels(ops):
    g = torch.Generator().manual_seed(6)
    lin = torch.nn.Linear(128, 32)
    x = torch.randn(3001, 128, generator=g)
    dy = torch.randn(3001, 32, generator=g)
    xr = x.clone().requires_grad_()
    ref = lin(xr)
    rg = torch.autograd.grad(ref, [xr, lin.weight, lin.bias], dy)
    lind = torch.nn.Linear(128, 32).to(DEV)
    lind.load_state_dict(lin.state_dict())
    xd = x.to(DEV).requires_grad_()
    out = ops.dense_linear(xd, lind.weight, lind.bias)
    assert_close(out, ref, TOL)
    gg = torch.autograd.grad(out, [xd, lind.weight, lind.bias], dy.to(DEV))
    for a, r in zip(gg, rg):
        assert_close(a, r, TOL)


# ---- softmax aggregation ----------------------------------------------------------
@pytest.mark.parametrize("act", ["identity", "relu", "swish"])
@pytest.mark.parametrize("h", [128, 32, 200])
def test_aggregate_softmax_matches_oracle(ops, act, h):
    from oracle import phc_oracle as O
    from phc_gnn_amd import synth
    g = torch.Generator().manual_seed(h + len(act))
    ei = torch.from_numpy(synth.ragged_edges(150, seed=4, hub_degree=200))
    E, N = ei.shape[1], 150
    x = torch.randn(N, h, generator=g)
    ea = torch.randn(E, h, generator=g)
    dy = torch.randn(N, h, generator=g)
    for beta0, add_self in ((1.0, False), (0.3, True), (2.5, False), (2.5, True)):
        xr, er, br = x.clone().requires_grad_(), ea.clone().requires_grad_(), torch.tensor(beta0, requires_grad=True)
        ref = O.softmax_aggregate(xr, ei, er, br, act)
        if add_self:                                        # the conv's self loop, fused into the same launches
            ref = ref + xr
        rg = torch.autograd.grad(ref, [xr, er, br], dy)
        xd, ed = x.to(DEV).requires_grad_(), ea.to(DEV).requires_grad_()
        bd = torch.tensor(beta0, device=DEV, requires_grad=True)
        out = ops.aggregate_softmax(xd, ei.to(DEV), ed, bd, act, add_self=add_self)
        assert_close(out, ref, 2e-5, f"softmax fwd {act} beta={beta0}")
        gg = torch.autograd.grad(out, [xd, ed, bd], dy.to(DEV))
        assert_close(gg[0], rg[0], 3e-5, "softmax dx")
        assert_close(gg[1], rg[1], 3e-5, "softmax dea")
        assert_close(gg[2], rg[2], 1e-4, "softmax dbeta")


@pytest.mark.parametrize("act,add_self", [("identity", True), ("relu", False), ("swish", True)])
def test_aggregate_softmax_encoded_matches_oracle(ops, act, add_self):
    """Softmax aggregation with the bond encoder read in place (packed codes + tables inside the kernels) against the
    oracle's embedding + softmax_aggregate: output, dx, every table gradient and d/d beta; isolated nodes, a hub."""
    from oracle import phc_oracle as O
    from phc_gnn_amd import synth
    g = torch.Generator().manual_seed(23)
    n, k, vocab = 4, 16, [5, 6, 2]
    ei = torch.from_numpy(synth.ragged_edges(180, seed=6, hub_degree=90))
    E, N, h = ei.shape[1], 180, n * k
    bond = torch.stack([torch.randint(0, v, (E,), generator=g) for v in vocab], 1)
    tables = _make_tables(n, vocab, k, g)
    x = torch.randn(N, h, generator=g)
    dy = torch.randn(N, h, generator=g)
    ref_t = [[t.clone().requires_grad_() for t in comp] for comp in tables]
    xr, br = x.clone().requires_grad_(), torch.tensor(1.7, requires_grad=True)
    ref = O.softmax_aggregate(xr, ei, _oracle_encode(bond, ref_t), br, act)
    if add_self:
        ref = ref + xr
    rg = torch.autograd.grad(ref, [xr, br] + [t for comp in ref_t for t in comp], dy)
    dev_t = [[t.to(DEV).requires_grad_() for t in comp] for comp in tables]
    xd, bd = x.to(DEV).requires_grad_(), torch.tensor(1.7, device=DEV, requires_grad=True)
    ts = ops.TableSet(dev_t)
    out = ops.aggregate_softmax_encoded(xd, ei.to(DEV), bond.to(DEV), ts, dev_t, bd, act, add_self=add_self)
    assert_close(out, ref, 2e-5, "softmax encoded fwd")
    gg = torch.autograd.grad(out, [xd, bd] + [t for comp in dev_t for t in comp], dy.to(DEV))
    assert_close(gg[0], rg[0], 3e-5, "dx")
    assert_close(gg[1], rg[1], 1e-4, "dbeta")
    for a, r in zip(gg[2:], rg[2:]):
        assert_close(a, r, 3e-5, "table grad")


@pytest.mark.parametrize("shape,frac_nan", [((1088, 1), 0.06), ((37, 12), 0.4), ((1100, 128), 0.7), ((5, 3), 0.0)])
def test_masked_l1_matches_torch(ops, shape, frac_nan):
    """NaN-masked mean L1 (benchmarks/train_zinc.py: torch.nn.L1Loss; padding graphs of a fixed-shape batch carry NaN targets) fwd +
    bwd against ``F.l1_loss(pred[mask], y[mask])``, incl. exact ties (gradient 0, torch's sign(0)) and the multi-workgroup path."""
    g = torch.Generator().manual_seed(31)
    z = (torch.randn(shape, generator=g) * 3)
    y = torch.randn(shape, generator=g)
    y[torch.rand(shape, generator=g) < frac_nan] = float("nan")
    z.view(-1)[:3] = torch.nan_to_num(y.view(-1)[:3])                 # ties where the target is labeled
    z.requires_grad_()
    mask = ~torch.isnan(y)
    ref = torch.nn.functional.l1_loss(z[mask], y[mask])
    ref.backward()
    zd = z.detach().to(DEV).requires_grad_()
    loss = ops.masked_l1(zd, y.to(DEV))
    loss.backward()
    assert_close(loss, ref, TOL, "masked L1")
    assert_close(zd.grad, z.grad, TOL, "d masked L1")
    w = torch.tensor(0.37, device=DEV)                                # a non-unit upstream gradient takes the backward launch
    zd2 = z.detach().to(DEV).requires_grad_()
    (ops.masked_l1(zd2, y.to(DEV)) * w).backward()
    assert_close(zd2.grad, z.grad * 0.37, TOL, "d masked L1, scaled")


@pytest.mark.parametrize("shape,frac_nan", [((544, 1), 0.06), ((37, 12), 0.4), ((1100, 128), 0.7), ((5, 3), 0.0)])
def test_masked_bce_matches_oracle(ops, shape, frac_nan):
    """NaN-masked BCE-with-logits (train_hiv.py:174-178) fwd + bwd; (1100,128) exercises the multi-workgroup path."""
    from oracle import phc_oracle as O
    g = torch.Generator().manual_seed(21)
    z = (torch.randn(shape, generator=g) * 3).requires_grad_()
    y = (torch.rand(shape, generator=g) > 0.5).float()
    y[torch.rand(shape, generator=g) < frac_nan] = float("nan")
    ref = O.bce_loss_masked(z, y)
    ref.backward()
    zd = z.detach().to(DEV).requires_grad_()
    loss = ops.masked_bce_with_logits(zd, y.to(DEV))
    (loss * 1.7).backward()
    assert_close(loss, ref, 2e-6, "loss")
    assert_close(zd.grad, 1.7 * z.grad, TOL, "dlogits")
    assert torch.equal(zd.grad.cpu() == 0, torch.isnan(y) | (z.grad == 0))      # unlabeled entries get exactly zero
    # backward seeded with the resident 1 (harness.backward_seed): for <= 4096 entries the gradient was written by the
    # forward launch and the backward launches nothing; same values either way
    from phc_gnn_amd import harness
    z2 = z.detach().to(DEV).requires_grad_()
    loss2 = ops.masked_bce_with_logits(z2, y.to(DEV))
    loss2.backward(harness.backward_seed(loss2))
    assert_close(z2.grad, z.grad, TOL, "dlogits (unit seed)")


@pytest.mark.parametrize("n,k,is_sorted", [(4, 32, True), (2, 5, True), (1, 70, True), (4, 32, False)])
def test_gated_segment_sum_matches_oracle(ops, n, k, is_sorted):
    """Soft-attention read-out core (pooling.py:57-66): x * sigmoid(gate) broadcast over components, summed per
    graph; ragged graphs incl. empty ones."""
    g = torch.Generator().manual_seed(22)
    sizes = [0, 3, 1, 40, 0, 17, 2, 0]
    batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    N, h = batch.numel(), n * k
    if not is_sorted:
        batch = batch[torch.randperm(N, generator=g)]
    x = torch.randn(N, h, generator=g).requires_grad_()
    gate = torch.randn(N, k, generator=g).requires_grad_()
    gated = (x.view(N, n, k) * torch.sigmoid(gate).unsqueeze(1)).reshape(N, h)
    ref = torch.zeros(len(sizes), h).index_add(0, batch, gated)
    dout = torch.randn(len(sizes), h, generator=g)
    ref.backward(dout)
    xd, gd = x.detach().to(DEV).requires_grad_(), gate.detach().to(DEV).requires_grad_()
    out = ops.gated_segment_sum(xd, gd, batch.to(DEV), n, num_graphs=len(sizes), is_sorted=is_sorted)
    out.backward(dout.to(DEV))
    assert_close(out, ref, TOL, "pooled")
    assert_close(xd.grad, x.grad, TOL, "dx")
    assert_close(gd.grad, gate.grad, TOL, "dgate")


@pytest.mark.parametrize("n,kin,kout,N", [(4, 32, 32, 1000), (4, 32, 16, 77), (2, 16, 64, 300), (1, 128, 32, 500), (8, 16, 8, 64),
                                          (4, 64, 64, 1000), (8, 32, 32, 40000)])
def test_prebuilt_operand_matrices_give_identical_results(ops, n, kin, kout, N):
    """Inside prebuilt_H() the forward and dx GEMMs read H / H^T built once by phc_phm_build_H_batched; the
    results must equal the in-kernel assembly bit for bit (same fma order) where both run the LDS-staged kernel, and to
    2e-6 where the prebuilt path takes the H-stationary kernel (widths 128 / 256: another summation order); H itself the golden formula."""
    from oracle import phc_oracle as O
    g = torch.Generator().manual_seed(31)
    x = torch.randn(N, n * kin, generator=g).to(DEV)
    A = torch.randn(n, n, n, generator=g).to(DEV).requires_grad_()
    W = (torch.randn(n, kin, kout, generator=g) * 0.2).to(DEV).requires_grad_()
    b = torch.randn(n * kout, generator=g).to(DEV).requires_grad_()
    dy = torch.randn(N, n * kout, generator=g).to(DEV)

    def run(pre):
        xr = x.clone().requires_grad_()
        for t in (A, W, b):
            t.grad = None
        ctx = ops.prebuilt_H([(A, W)]) if pre else __import__("contextlib").nullcontext()
        with ctx, ops.deferred_param_grads():
            y = ops.phm_linear(xr, A, W, b)
            y.backward(dy)
        return y.detach(), xr.grad, W.grad.clone(), A.grad.clone(), b.grad.clone()

    ref, got = run(False), run(True)
    hs = n * kin in (128, 256) and n * kout in (128, 256)     # csrc/phm_linear.hip: hs_shape_ok
    for name, a, r in zip(("y", "dx", "dW", "dA", "db"), got, ref):
        if hs and name in ("y", "dx"):
            assert_close(a, r, 2e-6, name)
        else:
            assert torch.equal(a, r), name
    assert_close(got[0], O.phm_linear(x.cpu(), A.detach().cpu(), W.detach().cpu(), b.detach().cpu()), TOL, "y vs oracle")
    # a weight update inside the block invalidates the entry (version check): the layer falls back to assembling H
    with ops.prebuilt_H([(A, W)]):
        with torch.no_grad():
            W.mul_(2.0)
        y2 = ops.phm_linear(x, A, W, b)
    assert_close(y2, O.phm_linear(x.cpu(), A.detach().cpu(), W.detach().cpu(), b.detach().cpu()), TOL, "stale entry is not used")


@pytest.mark.parametrize("n,kin,kout,N,valid", [(4, 32, 32, 13824, 12874), (4, 32, 32, 500, None), (2, 16, 48, 33, 20),
                                               (4, 32, 16, 70000, 69000), (1, 64, 128, 3000, None)])
def test_bn_statistics_from_the_gemm_epilogue(ops, n, kin, kout, N, valid):
    """A PHM layer asked for batch-norm statistics (BNStatsRequest) on a prebuilt operand emits (sum, M2) partials
    from the GEMM's registers; batch_norm_act on its output must agree with the oracle and with the two-pass path,
    forward and backward, including padding rows (n_valid) and running statistics."""
    from oracle import phc_oracle as O
    g = torch.Generator().manual_seed(41)
    h = n * kout
    k = kout
    x = torch.randn(N, n * kin, generator=g)
    x[::7] *= 30.0                                                       # a few dominant rows: no E[x^2]-E[x]^2 shortcuts
    A = torch.randn(n, n, n, generator=g)
    W = torch.randn(n, kin, kout, generator=g) * 0.2
    b = torch.randn(h, generator=g)
    w = [torch.rand(k, generator=g) + 0.5 for _ in range(n)]
    bb = [torch.randn(k, generator=g) * 0.1 for _ in range(n)]
    dy = torch.randn(N, h, generator=g)
    nv = N if valid is None else valid
    n_valid = None if valid is None else torch.tensor(valid, dtype=torch.int32, device=DEV)

    def run(fused):
        xd = x.to(DEV).requires_grad_()
        Ad, Wd, bd = (t.to(DEV).requires_grad_() for t in (A, W, b))
        wd, bbd = [t.to(DEV).requires_grad_() for t in w], [t.to(DEV).requires_grad_() for t in bb]
        rm, rv = [torch.zeros(k, device=DEV) for _ in range(n)], [torch.ones(k, device=DEV) for _ in range(n)]
        with ops.prebuilt_H([(Ad, Wd)]):
            req = ops.BNStatsRequest(n_valid) if fused else None
            y = ops.phm_linear(xd, Ad, Wd, bd, bn_stats=req)
            assert (getattr(y, "_phc_bn", None) is not None) == fused
            out = ops.batch_norm_act(y, wd, bbd, rm, rv, None, True, 0.1, 1e-5, "relu", None, n_valid)
            out.backward(dy.to(DEV))
        return [out.detach(), xd.grad, Wd.grad, bd.grad] + [t.grad for t in wd + bbd] + rm + rv

    fused, plain = run(True), run(False)
    for i, (a, r) in enumerate(zip(fused, plain)):
        if i == 3:        # a bias in front of a batch-norm has an exactly-zero gradient: both paths return round-off noise
            assert float(a.abs().max()) < 1e-3 * float(fused[2].abs().max()) + 1e-4
            continue
        assert_close(a, r, 2e-5, f"fused vs two-pass, tensor {i}")
    # oracle on the valid rows
    xr, Ar, Wr, br = (t.clone().requires_grad_() for t in (x[:nv], A, W, b))
    wr, bbr = [t.clone().requires_grad_() for t in w], [t.clone().requires_grad_() for t in bb]
    rm_r, rv_r = [torch.zeros(k) for _ in range(n)], [torch.ones(k) for _ in range(n)]
    ref = torch.relu(O.batch_norm_naive(O.phm_linear(xr, Ar, Wr, br), n, wr, bbr, rm_r, rv_r, True))
    ref.backward(dy[:nv])
    assert_close(fused[0][:nv], ref, 2e-5, "out vs oracle")
    assert float(fused[0][nv:].abs().max()) == 0.0 if nv < N else True
    assert_close(fused[1][:nv], xr.grad, 5e-5, "dx vs oracle")
    assert_close(fused[2], Wr.grad, 5e-5, "dW vs oracle")
    for a, r in zip(fused[4 + 2 * n:], rm_r + rv_r):
        assert_close(a, r, 2e-5, "running stats vs oracle")


@pytest.mark.parametrize("msg_act", ["relu", "identity"])
def test_deferred_bond_table_gradients_match_the_per_layer_path(ops, msg_act):
    """Inside deferred_param_grads() the bond-table gradients of several encoded aggregations over one edge list
    are produced at the end — phc_table_grads_batched (per edge) for a non-linear message function,
    phc_table_grads_counted (per destination node, in-edge row counts) for the identity; they must equal the per-layer
    per-edge gradients (different slab / summation order -> fp32 tolerance) and accumulate into .grad.  300 nodes and
    1500 random edges: most nodes have several in-edges naming the same table row (counts > 1)."""
    from phc_gnn_amd import synth
    g = torch.Generator().manual_seed(51)
    n, k, vocab, N, E = 4, 8, [5, 6, 2], 300, 1500
    ei = torch.from_numpy(synth.random_edges(N, E, seed=5)).to(DEV)
    bond = torch.stack([torch.randint(0, v, (E,), generator=g) for v in vocab], dim=1).to(DEV)
    layers = []
    for li in range(3):
        tabs = [[(torch.randn(v, k, generator=g) * 0.5).to(DEV).requires_grad_() for v in vocab] for _ in range(n)]
        layers.append((tabs, torch.randn(N, n * k, generator=g).to(DEV), torch.randn(N, n * k, generator=g).to(DEV)))

    def run(deferred):
        owners = []
        for tabs, _, _ in layers:
            for comp in tabs:
                for t in comp:
                    t.grad = None
        layers[0][0][0][0].grad = torch.full_like(layers[0][0][0][0], 0.25)
        ctx = ops.deferred_param_grads() if deferred else __import__("contextlib").nullcontext()
        with ctx:
            total = 0.0
            for tabs, x, dy in layers:
                owner = torch.nn.Module()
                owners.append(owner)
                ts = ops.tableset_for(owner, tabs)
                out = ops.aggregate_sum_encoded(x.clone().requires_grad_(), ei, bond, ts, tabs, msg_act, True)
                total = total + (out * dy).sum()
            total.backward()
        return [[t.grad.clone() for comp in tabs for t in comp] for tabs, _, _ in layers]

    a, b = run(True), run(False)
    for li, (ga, gb) in enumerate(zip(a, b)):
        for ti, (x1, x2) in enumerate(zip(ga, gb)):
            assert_close(x1, x2, 2e-6 if msg_act == "relu" else 5e-6, f"layer {li} table {ti}")


@pytest.mark.parametrize("k", [8, 16, 32])
def test_deferred_encoder_gradients_of_several_encoders_over_one_index_tensor(ops, k):
    """Inside deferred_param_grads() the table gradients of the encoders that were applied to ONE index tensor (the
    bond encoders of all layers when the [E, h] embedding is materialised) come from phc_encode_bwd_batched, the largest
    remaining job joins the step tail; all must equal the immediate per-encoder gradients bit for bit (same body, same
    plan is NOT guaranteed -> fp32 tolerance) and accumulate into .grad.  k = 16 / 32: widths of 64 / 128 columns, where the
    batched launch takes the register-resident form (enc_direct_batched_kernel)."""
    g = torch.Generator().manual_seed(91)
    n, vocab, E = 4, [5, 6, 2], 2100
    idx = torch.stack([torch.randint(0, v, (E,), generator=g) for v in vocab], dim=1).to(DEV)
    encs = []
    for li in range(4):
        tabs = [[(torch.randn(v, k, generator=g) * 0.5).to(DEV).requires_grad_() for v in vocab] for _ in range(n)]
        encs.append((tabs, torch.randn(E, n * k, generator=g).to(DEV), torch.nn.Module()))
    vocab_a, N = [7, 3, 4, 9], 333
    idx_a = torch.stack([torch.randint(0, v, (N,), generator=g) for v in vocab_a], dim=1).to(DEV)
    tabs_a = [[(torch.randn(v, k, generator=g) * 0.5).to(DEV).requires_grad_() for v in vocab_a] for _ in range(n)]
    dy_a, owner_a = torch.randn(N, n * k, generator=g).to(DEV), torch.nn.Module()

    def run(deferred):
        every = [t for tabs, _, _ in encs for comp in tabs for t in comp] + [t for comp in tabs_a for t in comp]
        for t in every:
            t.grad = None
        encs[1][0][0][0].grad = torch.full_like(encs[1][0][0][0], 0.5)
        ctx = ops.deferred_param_grads() if deferred else __import__("contextlib").nullcontext()
        with ctx:
            total = (ops.encode(idx_a, ops.tableset_for(owner_a, tabs_a), tabs_a) * dy_a).sum()
            for tabs, dy, owner in encs:
                total = total + (ops.encode(idx, ops.tableset_for(owner, tabs), tabs) * dy).sum()
            total.backward()
        return [t.grad.clone() for t in every]

    a, b = run(True), run(False)
    for i, (x1, x2) in enumerate(zip(a, b)):
        assert_close(x1, x2, 5e-6, f"table {i}")


def test_counted_bond_table_gradients_match_the_oracle(ops):
    """phc_table_grads_counted against autograd through the CPU oracle's embedding sum + scatter (N1 row), with
    isolated destination nodes, multi-edges and a node count that is not a multiple of the 64-node block."""
    from oracle import phc_oracle as O
    from phc_gnn_amd import synth
    g = torch.Generator().manual_seed(77)
    n, k, vocab, N, E = 4, 16, [5, 6, 2], 1000, 2600
    ei_np = synth.random_edges(N - 150, E, seed=8)                      # the last 150 nodes receive nothing
    ei = torch.from_numpy(ei_np)
    bond = torch.stack([torch.randint(0, v, (E,), generator=g) for v in vocab], dim=1)
    tabs_r = [[(torch.randn(v, k, generator=g) * 0.5).requires_grad_() for v in vocab] for _ in range(n)]
    x = torch.randn(N, n * k, generator=g)
    dy = torch.randn(N, n * k, generator=g)
    ea = torch.cat([O.integer_encoder(bond, comp) for comp in tabs_r], dim=1)
    (O.propagate_add(x, ei, ea) * dy).sum().backward()
    tabs = [[t.detach().clone().to(DEV).requires_grad_() for t in comp] for comp in tabs_r]
    owner = torch.nn.Module()
    with ops.deferred_param_grads():
        ts = ops.tableset_for(owner, tabs)
        out = ops.aggregate_sum_encoded(x.to(DEV).requires_grad_(), ei.to(DEV), bond.to(DEV), ts, tabs, "identity", False)
        (out * dy.to(DEV)).sum().backward()
    for c in range(n):
        for f in range(len(vocab)):
            assert_close(tabs[c][f].grad, tabs_r[c][f].grad, 1e-5, f"table {c},{f}")


@pytest.mark.parametrize("seed,iso", [(1, 0.0), (2, 0.1), (3, 0.5)])
def test_remove_isolated_nodes_matches_the_reference_transform(seed, iso):
    """Device version of the per-batch transform (benchmarks/utils.py:36-46): restated with numpy here — nodes that
    appear in no edge are dropped, the rest keep their order, edges are relabelled, x / batch rows follow."""
    from phc_gnn_amd import synth
    from phc_gnn_amd.graph import remove_isolated_nodes
    b = synth.molecule_batch(40, seed=seed, isolated_frac=iso)
    N = b.num_nodes
    keep = np.zeros(N, dtype=bool)
    keep[b.edge_index[0]] = True
    keep[b.edge_index[1]] = True
    newid = np.cumsum(keep) - 1
    x, ei, bt = (torch.from_numpy(a).to(DEV) for a in (b.x, b.edge_index, b.batch))
    x2, ei2, b2, kept = remove_isolated_nodes(x, ei, bt)
    assert torch.equal(x2.cpu(), torch.from_numpy(b.x[keep]))
    assert torch.equal(b2.cpu(), torch.from_numpy(b.batch[keep]))
    assert torch.equal(ei2.cpu(), torch.from_numpy(newid[b.edge_index]))
    assert torch.equal(kept.cpu(), torch.from_numpy(np.nonzero(keep)[0]))
    if iso > 0:
        assert int(keep.sum()) < N
    # fixed-shape form: full-size buffers + device count
    x3, ei3, b3, kept3, n_kept = remove_isolated_nodes(x, ei, bt, sync=False)
    k = int(n_kept)
    assert k == int(keep.sum()) and x3.shape[0] == N and torch.equal(x3[:k], x2) and torch.equal(ei3, ei2)


@pytest.mark.parametrize("kind", ["molecules", "molecules_1024", "rows_32767", "hub", "ragged"])
def test_csr_pair_build_equals_two_single_builds(kind):
    """phc_csr_build_pair (both sorted views in four launches on a resident workspace) == two phc_csr_build calls, bit for bit;
    rebuilt in place twice (the workspace must come back clean); up to 32 767 rows per side (the 1024-graph batches of the
    256-wide configurations: 23-27 k nodes)."""
    from phc_gnn_amd import synth
    from phc_gnn_amd.graph import EdgeStructure
    if kind == "molecules":
        b = synth.molecule_batch(64, seed=9, isolated_frac=0.05)
        ei, n = b.edge_index, b.num_nodes
    elif kind == "molecules_1024":
        b = synth.molecule_batch(1024, seed=10, dataset="zinc")
        ei, n = b.edge_index, b.num_nodes
        assert 16384 < n <= 32767
    elif kind == "rows_32767":
        n = 32767
        ei = synth.random_edges(n, 90_000, seed=12)
        ei[1, :200] = n - 1                                  # the last row is a small hub, row 0 may be empty
    elif kind == "hub":
        n = 300
        rng = np.random.default_rng(3)
        ei = np.stack([rng.integers(0, n, 2000), np.concatenate([np.full(700, 5), rng.integers(0, n, 1300)])]).astype(np.int64)
    else:
        ei, n = synth.ragged_edges(500, seed=4), 500
    t = torch.from_numpy(np.ascontiguousarray(ei)).to(DEV)
    a = EdgeStructure(t, n).prepare(for_backward=True)
    csr_a, csc_a = a.csr, a.csc
    with torch.no_grad():
        b2 = EdgeStructure(t, n)
        csr_b, csc_b = b2.csr, b2.csc
    for x, y in zip(csr_a + csc_a, csr_b + csc_b):
        assert torch.equal(x, y)
    assert a._pair_ws is not None
    for _ in range(2):                                       # in-place rebuilds on the same (resident) workspace
        for v in csr_a + csc_a:
            v.fill_(-7)
        a.rebuild()
        for x, y in zip(a.csr + a.csc, csr_b + csc_b):
            assert torch.equal(x, y)


def test_csr_pair_build_source_sorted_codes_equal_the_pack_launch():
    """The packed bond codes in SOURCE-sorted order out of the pair build (phc_csr_build_pair_codes_resident_ell_src) == phc_pack_edge_codes
    through perm_t in a launch of its own; refreshed by an in-place rebuild."""
    import ctypes as C
    from phc_gnn_amd import synth
    from phc_gnn_amd.graph import EdgeStructure
    b = synth.molecule_batch(64, seed=9)
    t = torch.from_numpy(np.ascontiguousarray(b.edge_index)).to(DEV)
    bond = torch.from_numpy(np.ascontiguousarray(b.edge_attr)).to(DEV)
    vocab = (C.c_int32 * 3)(5, 6, 2)
    a = EdgeStructure(t, b.num_nodes).prepare(for_backward=True, codes_for=(bond, vocab, 3), codes_src=True)
    assert a._pair_packed_src
    got = a.packed_codes_src(bond, vocab, 3)
    ref = torch.empty_like(got)
    a._pack_codes(bond, vocab, 3, ref, by_source=True)
    assert torch.equal(got, ref)
    got.fill_(-1)
    a.rebuild()
    assert torch.equal(a.packed_codes_src(bond, vocab, 3), ref)


def test_phm_linear_backward_large_n_uses_the_wide_tile(ops):
    """N >= 64k rows: dH = x^T.dy runs on 128 x 128 tiles (per-layer path); results vs the oracle."""
    from oracle import phc_oracle as O
    g = torch.Generator().manual_seed(61)
    n, kin, kout, N = 4, 32, 32, 70001
    x = torch.randn(N, n * kin, generator=g)
    A = torch.randn(n, n, n, generator=g)
    W = torch.randn(n, kin, kout, generator=g) * 0.1
    dy = torch.randn(N, n * kout, generator=g)
    xd, Ad, Wd = (t.to(DEV).requires_grad_() for t in (x, A, W))
    bd = torch.zeros(n * kout, device=DEV, requires_grad=True)
    y = ops.phm_linear(xd, Ad, Wd, bd)
    gx, gW, gA, gb = torch.autograd.grad(y, [xd, Wd, Ad, bd], dy.to(DEV))
    ex = O.phm_linear_backward(x.double(), A.double(), W.double(), dy.double())
    assert_close(gx, ex[0].float(), TOL, "dx")
    assert_close(gW, ex[1].float(), TOL, "dW")
    assert_close(gA, ex[2].float(), TOL, "dA")
    assert_close(gb, ex[3].float(), TOL, "db")


@pytest.mark.parametrize("max_norm,steps", [(2.0, 6), (0.0, 3), (1e-3, 3)])
def test_clip_adam_matches_torch_clip_grad_norm_and_adam(max_norm, steps):
    """phc_clip_adam_step (three launches) == clip_grad_norm_ + torch.optim.Adam (train_hiv.py:198-201,285) over
    several steps: parameters, clipped gradients and the reported norm; a parameter without gradient is skipped."""
    from phc_gnn_amd.optim import ClipAdam
    g = torch.Generator().manual_seed(71)
    shapes = [(4, 32, 32), (4, 4, 4), (128,), (119, 32), (5000,), (1,), (3, 7)] + [(13,)] * 40
    base = [torch.randn(s, generator=g) for s in shapes]
    pa = [b.clone().to(DEV).requires_grad_() for b in base]
    pb = [b.clone().to(DEV).requires_grad_() for b in base]
    oa = ClipAdam(pa, lr=1e-2, max_norm=max_norm)
    ob = torch.optim.Adam(pb, lr=1e-2)
    for it in range(steps):
        grads = [torch.randn(s, generator=g) * (5.0 if it % 2 else 0.05) for s in shapes]
        for i, (x, y, gr) in enumerate(zip(pa, pb, grads)):
            if i == 2 and it == 1:                        # no gradient for this tensor in this step
                x.grad = y.grad = None
                continue
            x.grad, y.grad = gr.to(DEV), gr.to(DEV)
        norm_a = oa.step()
        norm_b = torch.nn.utils.clip_grad_norm_(pb, max_norm) if max_norm > 0 else \
            torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(p.grad) for p in pb if p.grad is not None]))
        ob.step()
        assert_close(norm_a, norm_b, 2e-6, f"norm step {it}")
        for i, (x, y) in enumerate(zip(pa, pb)):
            assert_close(x, y, 2e-6, f"param {i} step {it}")
            if x.grad is not None:
                assert_close(x.grad, y.grad, 2e-6, f"clipped grad {i} step {it}")


@pytest.mark.parametrize("fi,fo,N", [(128, 32, 1000), (64, 64, 33), (256, 128, 500)])
def test_dense_linear_on_prebuilt_operand_with_deferred_gradients(ops, fi, fo, N):
    """nn.Linear weight [out, in] served from the step's batched operand build (H = W^T; H^T is W itself), weight and
    bias gradients from the batched parameter-gradient launch (written in the [out, in] layout)."""
    g = torch.Generator().manual_seed(81)
    lin = torch.nn.Linear(fi, fo).to(DEV)
    x = torch.randn(N, fi, generator=g).to(DEV)
    dy = torch.randn(N, fo, generator=g).to(DEV)
    xr = x.clone().requires_grad_()
    ref = torch.nn.functional.linear(xr.double(), lin.weight.double(), lin.bias.double())
    gx, gw, gb = torch.autograd.grad(ref, [xr, lin.weight, lin.bias], dy.double())
    xd = x.clone().requires_grad_()
    lin.weight.grad = lin.bias.grad = None
    with ops.prebuilt_H([], [lin.weight]), ops.deferred_param_grads():
        y = ops.dense_linear(xd, lin.weight, lin.bias)
        assert y.grad_fn.__class__.__name__.startswith("_DenseLinearFn")
        y.backward(dy)
    assert_close(y, ref.float(), TOL, "y")
    assert_close(xd.grad, gx.float(), TOL, "dx")
    assert_close(lin.weight.grad, gw.float(), TOL, "dweight")
    assert_close(lin.bias.grad, gb.float(), TOL, "dbias")


@pytest.mark.parametrize("fi,fo,N,bias", [(4, 1, 544, True), (8, 2, 1, True), (3, 5, 3000, False), (16, 4, 70, True)])
def test_small_linear_matches_torch(ops, fi, fo, N, bias):
    """The tiny read-out head (RealTransformer 'linear', layers.py:399-414) as one kernel each way."""
    g = torch.Generator().manual_seed(91)
    x = torch.randn(N, fi, generator=g)
    w = torch.randn(fo, fi, generator=g)
    b = torch.randn(fo, generator=g) if bias else None
    dy = torch.randn(N, fo, generator=g)
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    br = b.clone().requires_grad_() if bias else None
    ref = torch.nn.functional.linear(xr, wr, br)
    rg = torch.autograd.grad(ref, [xr, wr] + ([br] if bias else []), dy)
    xd, wd = x.to(DEV).requires_grad_(), w.to(DEV).requires_grad_()
    bd = b.to(DEV).requires_grad_() if bias else None
    y = ops.small_linear(xd, wd, bd)
    gg = torch.autograd.grad(y, [xd, wd] + ([bd] if bias else []), dy.to(DEV))
    assert_close(y, ref, TOL, "y")
    for name, a, r in zip(("dx", "dw", "db"), gg, rg):
        assert_close(a, r, TOL, name)


def test_fanout_sums_gradients_in_one_pass(ops):
    g = torch.Generator().manual_seed(95)
    x = torch.randn(1000, 128, generator=g).to(DEV).requires_grad_()
    ws = [torch.randn(1000, 128, generator=g).to(DEV) for _ in range(5)]
    outs = ops.fanout(x, 5)
    total = sum((o * w).sum() for o, w in zip(outs[:4], ws))          # the fifth alias is unused: its gradient is None
    total.backward()
    ref = ws[0] + ws[1] + ws[2] + ws[3]
    assert torch.equal(x.grad, ((ws[0] + ws[1]) + ws[2]) + ws[3]) or float((x.grad - ref).abs().max()) < 1e-6
    with torch.no_grad():
        assert ops.fanout(x, 3)[0] is x


@pytest.mark.parametrize("n,k,N,valid,act,deferred", [(4, 32, 13824, 12874, "relu", True), (4, 32, 500, None, "swish", True),
                                                      (2, 64, 333, 300, "relu", False), (8, 32, 700, None, "identity", True),
                                                      # 128-column / 8-wave configs; a smooth activation: at 9 M pre-activations
                                                      # some sit within an ulp of the ReLU kink and the two fp32 paths gate them
                                                      # differently (one flipped gate = one dx row off by 1e-1)
                                                      (4, 32, 70001, 69000, "swish", True),
                                                      (4, 64, 66000, None, "elu", True)])         # K = 256, 64-column / 8-wave
@pytest.mark.parametrize("bwd_sums", ["in_gemm", "separate_pass"])
def test_bn_act_linear_without_materialising_the_normalised_activation(ops, n, k, N, valid, act, deferred, bwd_sums, monkeypatch):
    """linear1 (statistics from its epilogue) -> [batch-norm -> act -> linear2 folded into linear2's operand staging]
    against the three separate ops: output, every gradient, running statistics, padding rows."""
    monkeypatch.setattr(ops, "BN_BWD_SUMS_IN_GEMM", bwd_sums == "in_gemm")
    g = torch.Generator().manual_seed(101)
    h = n * k
    x = torch.randn(N, h, generator=g)
    x[::11] *= 20.0
    dy = torch.randn(N, h, generator=g)
    base = dict(A1=torch.randn(n, n, n, generator=g), W1=torch.randn(n, k, k, generator=g) * 0.2, b1=torch.randn(h, generator=g),
                A2=torch.randn(n, n, n, generator=g), W2=torch.randn(n, k, k, generator=g) * 0.2, b2=torch.randn(h, generator=g))
    w = [torch.rand(k, generator=g) + 0.5 for _ in range(n)]
    bb = [torch.randn(k, generator=g) * 0.1 for _ in range(n)]
    n_valid = None if valid is None else torch.tensor(valid, dtype=torch.int32, device=DEV)

    def run(fused):
        P = {kk: v.to(DEV).requires_grad_() for kk, v in base.items()}
        wd, bd = [t.to(DEV).requires_grad_() for t in w], [t.to(DEV).requires_grad_() for t in bb]
        rm, rv = [torch.zeros(k, device=DEV) for _ in range(n)], [torch.ones(k, device=DEV) for _ in range(n)]
        nbt = [torch.zeros((), dtype=torch.int64, device=DEV) for _ in range(n)]
        xd = x.to(DEV).requires_grad_()
        ctx = ops.deferred_param_grads() if deferred else __import__("contextlib").nullcontext()
        with ops.prebuilt_H([(P["A1"], P["W1"]), (P["A2"], P["W2"])]), ctx:
            y1 = ops.phm_linear(xd, P["A1"], P["W1"], P["b1"], bn_stats=ops.BNStatsRequest(n_valid))
            y2 = ops.bn_act_linear(y1, wd, bd, rm, rv, nbt, 0.1, 1e-5, act, n_valid, P["A2"], P["W2"], P["b2"]) if fused else None
            assert (y2 is not None) == fused
            if y2 is None:
                z = ops.batch_norm_act(y1, wd, bd, rm, rv, nbt, True, 0.1, 1e-5, act, None, n_valid)
                y2 = ops.phm_linear(z, P["A2"], P["W2"], P["b2"])
            y2.backward(dy.to(DEV))
        return [y2.detach(), xd.grad] + [P[kk].grad for kk in ("W1", "A1", "W2", "A2", "b2")] + [t.grad for t in wd + bd] + rm + rv

    a, b = run(True), run(False)
    names = ["y2", "dx", "dW1", "dA1", "dW2", "dA2", "db2"] + [f"bn{i}" for i in range(4 * n)]
    nv = N if valid is None else valid
    for name, u, v in zip(names, a, b):
        if name == "y2":
            u, v = u[:nv], v[:nv]          # padding rows: both are b2-only rows, compare the valid part explicitly
        assert_close(u, v, 3e-5, name)


@pytest.mark.parametrize("N,k,vocab_b,n_layers", [(777, 16, [12, 9, 7], 5), (1203, 32, [5, 6, 2], 2), (64, 16, [3], 1),
                                                  # round 6: widths that are no multiple of 64 (the published 180 / 200-wide models: 64-column
                                                  # tiles with a partial last one), a width below one tile
                                                  (501, 45, [5, 6, 2], 2), (333, 50, [12, 9, 7], 5), (200, 10, [3], 1)])
def test_step_tail_register_resident_table_gradients(ops, N, k, vocab_b, n_layers):
    """phc_step_tail with the encoder tables in their register-resident forms (atom_direct_body / bond_direct_body): widths of 64
    and 128 columns, 28 bond-table rows (32-column count operand) and 13 (16 columns), five layers (a full group of four and a
    short one), 180 atom-table rows (two M groups), node counts that are no multiple of anything — against the immediate
    per-layer gradients (LDS tiles).  The PHM jobs beside them: h -> h, and two dense maps of an even and an odd output width
    (h -> 50, h -> 45: the register-resident dH tiles with two columns / one column of dy per lane, the gate maps of the published
    models) where the input width allows the form."""
    from phc_gnn_amd import synth
    g = torch.Generator().manual_seed(N)
    n = 4
    h = n * k
    E = 3 * N
    ei = torch.from_numpy(synth.random_edges(N - N // 7, E, seed=N)).to(DEV)          # the last nodes receive nothing
    bond = torch.stack([torch.randint(0, v, (E,), generator=g) for v in vocab_b], dim=1).to(DEV)
    vocab_a = [40, 50, 60, 30]
    idx_a = torch.stack([torch.randint(0, v, (N,), generator=g) for v in vocab_a], dim=1).to(DEV)
    tabs_a = [[(torch.randn(v, k, generator=g) * 0.5).to(DEV).requires_grad_() for v in vocab_a] for _ in range(n)]
    layers = []
    for li in range(n_layers):
        tabs = [[(torch.randn(v, k, generator=g) * 0.5).to(DEV).requires_grad_() for v in vocab_b] for _ in range(n)]
        layers.append((tabs, torch.randn(N, h, generator=g).to(DEV)))
    A = torch.randn(n, n, n, generator=g).to(DEV).requires_grad_()
    W = (torch.randn(n, k, k, generator=g) * 0.2).to(DEV).requires_grad_()
    dy_lin = torch.randn(N, h, generator=g).to(DEV)
    one = torch.ones(1, 1, 1, device=DEV)
    dense = [((torch.randn(1, h, o, generator=g) * 0.1).to(DEV).requires_grad_(), torch.randn(o, generator=g).to(DEV).requires_grad_(),
              torch.randn(N, o, generator=g).to(DEV)) for o in (50, 45)]
    owners = [torch.nn.Module() for _ in range(n_layers + 1)]

    def run(deferred):
        every = [t for comp in tabs_a for t in comp] + [t for tabs, _ in layers for comp in tabs for t in comp] + [A, W] + \
                [t for Wd, bd, _ in dense for t in (Wd, bd)]
        for t in every:
            t.grad = None
        ctx = ops.deferred_param_grads() if deferred else __import__("contextlib").nullcontext()
        with ctx:
            x0 = ops.encode(idx_a, ops.tableset_for(owners[-1], tabs_a), tabs_a)
            total = (ops.phm_linear(x0, A, W, None) * dy_lin).sum()                # a PHM job: the tail has something to batch
            for Wd, bd, dyd in dense:
                total = total + (ops.phm_linear(x0, one, Wd, bd) * dyd).sum()
            x = x0
            for (tabs, dy), owner in zip(layers, owners):
                out = ops.aggregate_sum_encoded(x, ei, bond, ops.tableset_for(owner, tabs), tabs, "identity", True)
                total = total + (out * dy).sum()
            total.backward()
        return [t.grad.clone() for t in every]

    a, b = run(True), run(False)
    assert len(a) == len(b) and len(a) > 2
    for i, (x1, x2) in enumerate(zip(a, b)):
        assert_close(x1, x2, 5e-6, f"gradient {i}")


@pytest.mark.parametrize("n,k,N,bias", [(4, 32, 13824, True), (8, 32, 2001, True), (4, 64, 777, False), (5, 40, 501, True)])
def test_gate_as_one_linear_map_equals_the_two_maps(ops, n, k, N, bias):
    """ops.gate_linear (the read-out's Linear(PHMLinear(x)) as x . (H . Wg^T) + (b . Wg^T + bg), pooling.py:57-66) against the two
    maps applied one after the other in float64: the gate, dx (with a parked share of dx added inside the GEMM) and the gradients of
    all five parameters — on the plain path, inside prebuilt_H (operands from the step's build launch) and with deferred gradients."""
    h = n * k
    g = torch.Generator().manual_seed(n * 1000 + k)
    x = torch.randn(N, h, generator=g)
    A = torch.randn(n, n, n, generator=g)
    W = torch.randn(n, k, k, generator=g) * 0.2
    b = torch.randn(h, generator=g) if bias else None
    Wg = torch.randn(k, h, generator=g) * 0.1
    bg = torch.randn(k, generator=g) if bias else None
    dg = torch.randn(N, k, generator=g)
    extra = torch.randn(N, h, generator=g)
    assert ops.gate_composable(W, Wg)
    # float64 reference: the two maps
    xr, Ar, Wr, Wgr = (t.double().requires_grad_() for t in (x, A, W, Wg))
    br = b.double().requires_grad_() if bias else None
    bgr = bg.double().requires_grad_() if bias else None
    H = torch.einsum("iac,ikp->akcp", Ar, Wr).reshape(h, h)
    y1 = xr @ H + (br if bias else 0.0)
    ref = y1 @ Wgr.t() + (bgr if bias else 0.0)
    ins = [xr, Ar, Wr, Wgr] + ([br, bgr] if bias else [])
    gr = torch.autograd.grad(ref, ins, dg.double())

    def run(mode):
        xd = x.to(DEV).requires_grad_()
        ps = [t.to(DEV).requires_grad_() if t is not None else None for t in (A, W, b, Wg, bg)]
        hand = ops.DxHandoff()
        ctxs = []
        if mode != "plain":
            ctxs.append(ops.prebuilt_H([(ps[0], ps[1])], gates=[tuple(ps)]))
        if mode == "deferred":
            ctxs.append(ops.deferred_param_grads())
        import contextlib
        with contextlib.ExitStack() as st:
            for c in ctxs:
                st.enter_context(c)
            gate = ops.gate_linear(xd, *ps, dx_handoff=hand)
            assert hand.accepted
            hand.tensor = extra.to(DEV)
            gate.backward(dg.to(DEV))
        torch.cuda.synchronize()
        return gate.detach(), xd.grad, [p.grad if p is not None else None for p in ps]

    for mode in ("plain", "prebuilt", "deferred"):
        gate, dx, pg = run(mode)
        assert_close(gate, ref.detach().float(), 2e-6, f"{mode}: gate")
        assert_close(dx, (gr[0] + extra.double()).float(), 2e-6, f"{mode}: dx + parked share")
        names = ["dA", "dW", "db", "dWg", "dbg"]
        refs = [gr[1], gr[2], gr[4] if bias else None, gr[3], gr[5] if bias else None]
        for nm, got, want in zip(names, pg, refs):
            if want is None:
                assert got is None, (mode, nm)
                continue
            assert_close(got, want.float(), 1e-5, f"{mode}: {nm}")
