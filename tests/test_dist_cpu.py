"""world_size-2 tests of the multi-GPU logic on CPU under gloo.

The collectives, the partition and the halo plan are the product code (phc_gnn_amd/dist.py); the local
compute steps are replaced by a stand-in backend built on the CPU oracle (test-only: the product's default
backend is the HIP engine and has no CPU path)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import phc_oracle as O
from phc_gnn_amd import synth
from phc_gnn_amd.dist import (FlatGradSync, ShardedAggregate, ShardedGINEStack, ShardedPHMGINEConv, build_halo_plan,
                              partition_bounds)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class OracleBackend:
    """Same interface as dist.EngineBackend, computed with torch CPU ops / the oracle."""

    class St:
        def __init__(self, ei, ns, nd):
            self.ei, self.num_src, self.num_dst, self.num_edges = ei, ns, nd, ei.shape[1]

    def structure(self, edge_index, num_src, num_dst):
        return self.St(edge_index, num_src, num_dst)

    def segments(self, index, num_rows):
        return (index, num_rows)

    def aggregate_fwd(self, x, ea, st, act, add_self, accumulate_into=None):
        msg = x.index_select(0, st.ei[0])
        if ea is not None:
            msg = msg + ea
        out = O.scatter_sum_rows(O.activation(act, msg), st.ei[1], st.num_dst)
        out = out + x[:st.num_dst] if add_self else out
        return out if accumulate_into is None else accumulate_into + out

    def aggregate_bwd(self, dout, x, ea, st, act, add_self, need_dea, extra=None):
        with torch.enable_grad():
            xr = x.detach().clone().requires_grad_()
            er = ea.detach().clone().requires_grad_() if ea is not None else None
            out = self.aggregate_fwd(xr, er, st, act, add_self)
            grads = torch.autograd.grad(out, [xr] + ([er] if er is not None else []), dout, allow_unused=True)
        dx = grads[0] if grads[0] is not None else torch.zeros_like(x)
        if extra is not None:                                     # another consumer's share of d x (the stack's skip: dist.NormLink)
            dx = dx + extra
        return dx, (grads[1] if (need_dea and er is not None) else None)

    def gather_rows(self, x, rows):
        return x.index_select(0, rows)

    def segment_sum_into(self, vals, seg, into):
        return into + O.scatter_sum_rows(vals, seg[0], seg[1])

    # ---- PHM linear pieces ----
    def linear_fwd(self, x, A, W, b):
        with torch.no_grad():
            return O.phm_linear(x, A, W, b)

    def linear_dx(self, dy, A, W):
        with torch.no_grad():
            return dy @ O.kron_sum_H(A, W).t()

    def linear_param_grads(self, x, dy, A, W, has_bias):
        with torch.enable_grad():
            Ar, Wr = A.detach().clone().requires_grad_(), W.detach().clone().requires_grad_()
            br = torch.zeros(W.shape[0] * W.shape[2], requires_grad=True) if has_bias else None
            y = O.phm_linear(x.detach(), Ar, Wr, br)
            g = torch.autograd.grad(y, [Ar, Wr] + ([br] if has_bias else []), dy)
        return g[0], g[1], (g[2] if has_bias else None)

    # ---- row-sharded batch-norm phases (restated from the formulas in include/phc_hip.h) ----
    def bn_moments(self, x):
        mean = x.mean(0) if x.shape[0] else torch.zeros(x.shape[1])
        return torch.cat([x.sum(0), ((x - mean) ** 2).sum(0)])

    def bn_fwd_from_moments(self, x, moments_all, counts_all, n_total, norm, act, residual=None):
        h = x.shape[1]
        sums, m2s = moments_all[:, :h], moments_all[:, h:]
        mean = sums.sum(0) / n_total
        M2 = torch.zeros(h)
        for r in range(moments_all.shape[0]):
            nr = float(counts_all[r])
            if nr > 0:
                M2 = M2 + m2s[r] + nr * (sums[r] / nr - mean) ** 2
        var = M2 / n_total
        invstd = 1.0 / torch.sqrt(var + norm.eps)
        k = h // len(norm.bn)
        w = torch.cat([m.weight.detach() for m in norm.bn]) if norm.affine else torch.ones(h)
        b = torch.cat([m.bias.detach() for m in norm.bn]) if norm.affine else torch.zeros(h)
        with torch.no_grad():
            for c, m in enumerate(norm.bn):
                m.running_mean.mul_(1 - norm.momentum).add_(norm.momentum * mean[c * k:(c + 1) * k])
                m.running_var.mul_(1 - norm.momentum).add_(norm.momentum * (M2 / (n_total - 1))[c * k:(c + 1) * k])
                m.num_batches_tracked += 1
        zhat = (x - mean) * invstd
        out = O.activation(act, zhat * w + b)
        return (out if residual is None else out + residual), torch.cat([mean, invstd, w, b])

    def bn_eval(self, x, norm, act, residual=None):
        """norm.py:30-35 in eval mode: the running statistics, one BatchNorm1d per component."""
        mods = list(norm.bn)
        out = O.batch_norm_naive(x, len(mods), [m.weight for m in mods], [m.bias for m in mods], [m.running_mean for m in mods],
                                 [m.running_var for m in mods], False, norm.momentum, norm.eps)
        out = O.activation(act, out)
        return out if residual is None else out + residual

    def _g(self, dy, x, save, act):
        mean, invstd, w, b = save.view(4, -1)
        xhat = (x - mean) * invstd
        with torch.enable_grad():
            pre = (xhat * w + b).detach().requires_grad_()
            out = O.activation(act, pre)
        return torch.autograd.grad(out, pre, dy)[0], xhat

    def bn_bwd_sums(self, dy, x, save, act):
        g, xhat = self._g(dy, x, save, act)
        return torch.cat([(g * xhat).sum(0), g.sum(0)])

    def bn_bwd_apply(self, dy, x, save, dwb_global, n_total, act):
        g, xhat = self._g(dy, x, save, act)
        h = x.shape[1]
        _, invstd, w, _ = save.view(4, -1)
        return w * invstd * (g - dwb_global[h:] / n_total - xhat * dwb_global[:h] / n_total)


def _graph(N, E, world, kind, seed):
    """Edge list [2, E'] of the multi-rank tests.  kind "uniform": iid edges.  kind "holes" (what an 8-rank run meets on real
    partitions and a uniform toy graph never shows): the LAST rank's nodes have local in-edges only (a rank with no halo at
    all), ranks 0 and 1 exchange nothing in either direction (empty peer segments inside the all-to-all), and from 4 ranks up
    rank 2's nodes have no in-edge whatsoever (empty local AND halo structures; it still serves rows to the others).
    kind "island": the last rank exchanges NOTHING in either direction — its send and receive splits coincide (all zero), the
    case in which a rank-local key of the exchange's start-up check would take the backward exchange for already verified while
    its peers run the (collective) check."""
    ei = torch.from_numpy(synth.random_edges(N, E, seed=seed))
    if kind == "uniform":
        return ei
    b = torch.tensor(partition_bounds(N, world)[1:])
    own_s, own_d = torch.bucketize(ei[0], b, right=True), torch.bucketize(ei[1], b, right=True)
    if kind == "island":
        return ei[:, ~((own_s != own_d) & ((own_s == world - 1) | (own_d == world - 1)))].contiguous()
    drop = (own_d == world - 1) & (own_s != world - 1)
    drop |= ((own_s == 0) & (own_d == 1)) | ((own_s == 1) & (own_d == 0))
    if world >= 4:
        drop |= own_d == 2
    return ei[:, ~drop].contiguous()


def _worker_sharded(rank, world, port, act, tmp, max_pair_bytes=None, kind="uniform"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if max_pair_bytes is not None:                          # force the chunked exchange (several all-to-alls per halo pull)
            import phc_gnn_amd.dist as D
            D.MAX_PAIR_BYTES = max_pair_bytes
        N, h = 203, 16
        ei = _graph(N, 1500, world, kind, seed=3)
        E = ei.shape[1]
        g = torch.Generator().manual_seed(5)
        x = torch.randn(N, h, generator=g)
        ea = torch.randn(E, h, generator=g)
        dy = torch.randn(N, h, generator=g)
        plan = build_halo_plan(ei, N)
        if kind == "island" and rank == world - 1:
            assert plan.recv_splits == plan.send_splits == [0] * world
        if kind == "holes":
            assert (plan.n_halo == 0) == (rank == world - 1 or (world >= 4 and rank == 2)), (rank, plan.n_halo)
            if rank in (0, 1):
                assert plan.recv_splits[1 - rank] == 0 and plan.send_splits[1 - rank] == 0
            if world >= 4 and rank == 2:
                assert plan.loc_edge_index.shape[1] == 0 and plan.halo_edge_index.shape[1] == 0 and sum(plan.send_splits) > 0
        if max_pair_bytes is not None:
            chunks = -(-(plan.max_pair_rows * h * 4) // max_pair_bytes)
            assert chunks > 1, (plan.max_pair_rows, max_pair_bytes)           # the case really cuts its exchanges
        bounds = partition_bounds(N, world)
        assert (plan.lo, plan.hi) == (bounds[rank], bounds[rank + 1])
        # plan sanity: halo ids are exactly the remote sources of my in-edges
        mine = (ei[1] >= plan.lo) & (ei[1] < plan.hi)
        rem = mine & ((ei[0] < plan.lo) | (ei[0] >= plan.hi))
        assert torch.equal(plan.halo_ids, torch.unique(ei[0][rem]))
        assert plan.loc_edge_index.shape[1] + plan.halo_edge_index.shape[1] == int(mine.sum())
        assert sum(plan.recv_splits) == plan.n_halo and sum(plan.send_splits) == plan.send_rows.numel()

        agg = ShardedAggregate(plan, backend=OracleBackend())
        x_own = x[plan.lo:plan.hi].clone().requires_grad_()
        ea_loc, ea_halo = agg.split_edge_attr(ea)
        ea_loc, ea_halo = ea_loc.requires_grad_(), ea_halo.requires_grad_()
        out = agg(x_own, ea_loc, ea_halo, act, True)
        gx, gl, gh = torch.autograd.grad(out, [x_own, ea_loc, ea_halo], dy[plan.lo:plan.hi])
        torch.save(dict(out=out.detach(), gx=gx, gl=gl, gh=gh, loc_ids=plan.loc_edge_ids, halo_ids=plan.halo_edge_ids,
                        lo=plan.lo, hi=plan.hi), os.path.join(tmp, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,act,max_pair_bytes,kind", [
    (2, "identity", None, "uniform"), (2, "relu", None, "uniform"), (2, "relu", 5000, "uniform"),
    (3, "relu", 700, "uniform"),            # odd world, three chunks per exchange
    (4, "identity", None, "holes"),
    (8, "relu", 300, "holes"),              # the driver's 8-rank shape: chunked, empty peer segments, ranks without halo / edges
    (3, "relu", None, "island"),            # a rank whose send and receive splits coincide: the start-up check stays collective
])
def test_sharded_aggregate_equals_single_process(tmp_path, world, act, max_pair_bytes, kind):
    """``max_pair_bytes``: every exchange is cut into row chunks (dist.MAX_PAIR_BYTES, the guard against > 1 GiB messages),
    forward and reverse, with the plan's global pair maximum deciding the cut on all ranks alike.  Every first exchange of a
    shape also runs the start-up pattern check (dist._verify_exchange)."""
    mp.spawn(_worker_sharded, args=(world, _free_port(), act, str(tmp_path), max_pair_bytes, kind), nprocs=world, join=True)
    N, h = 203, 16
    ei = _graph(N, 1500, world, kind, seed=3)
    E = ei.shape[1]
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, h, generator=g).requires_grad_()
    ea = torch.randn(E, h, generator=g).requires_grad_()
    dy = torch.randn(N, h, generator=g)
    ref = O.propagate_add(x, ei, ea, act) + x
    rgx, rge = torch.autograd.grad(ref, [x, ea], dy)
    out, gx, ge = torch.zeros(N, h), torch.zeros(N, h), torch.zeros(E, h)
    for r in range(world):
        d = torch.load(os.path.join(tmp_path, f"r{r}.pt"))
        out[d["lo"]:d["hi"]] = d["out"]
        gx[d["lo"]:d["hi"]] = d["gx"]
        ge[d["loc_ids"]] = d["gl"]
        ge[d["halo_ids"]] = d["gh"]
    assert torch.allclose(out, ref, rtol=1e-5, atol=1e-5)
    assert torch.allclose(gx, rgx, rtol=1e-5, atol=1e-5)
    assert torch.allclose(ge, rge, rtol=1e-5, atol=1e-5)


def _worker_gradsync(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        params = [torch.nn.Parameter(torch.randn(3, 4)), torch.nn.Parameter(torch.randn(5)), torch.nn.Parameter(torch.randn(2, 2, 2))]
        for i, p in enumerate(params):
            p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
        params[1].grad = None                                   # a parameter that got no gradient this step
        FlatGradSync(params)()
        torch.save([p.grad.clone() for p in params], os.path.join(tmp, f"g{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_flat_gradient_bucket_allreduce(tmp_path):
    world = 2
    mp.spawn(_worker_gradsync, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g0, g1 = torch.load(os.path.join(tmp_path, "g0.pt")), torch.load(os.path.join(tmp_path, "g1.pt"))
    for i, (a, b) in enumerate(zip(g0, g1)):
        assert torch.equal(a, b)
        expect = 0.0 if i == 1 else (1 + 2) / 2 * (i + 1)
        assert torch.allclose(a, torch.full_like(a, expect))


def test_partition_bounds_cover_everything():
    for n, w in [(10, 3), (7, 8), (10_000_000, 8), (0, 2)]:
        b = partition_bounds(n, w)
        assert b[0] == 0 and b[-1] == n and len(b) == w + 1
        sizes = np.diff(b)
        assert sizes.min() >= 0 and sizes.max() - sizes.min() <= 1


# ---- the sharded message-passing LAYER (aggregate -> +x -> linear1 -> norm with global statistics -> act -> linear2) ----
LAYER_CASES = {"relu_norm": dict(norm="naive-batch-norm", msg="relu", act="relu"),
               "identity_norm": dict(norm="naive-batch-norm", msg="identity", act="swish"),
               "no_norm": dict(norm=None, msg="identity", act="relu")}


def _make_conv(case, n, h, seed):
    from phc.hypercomplex.undirectional.messagepassing import PHMGINEConv
    c = LAYER_CASES[case]
    torch.manual_seed(seed)
    np.random.seed(seed)
    conv = PHMGINEConv(h, h, phm_dim=n, phm_rule=None, norm=c["norm"], activation=c["act"], msg_encoder=c["msg"])
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in conv.named_parameters():
            if name.endswith("phm_rule"):
                p.add_(0.05 * torch.randn(p.shape, generator=g))
            elif p.dim() == 1:
                p.copy_(0.5 + torch.rand(p.shape, generator=g) if name.endswith("weight") else 0.1 * torch.randn(p.shape, generator=g))
    return conv.train()


def _layer_problem():
    N, E, n, h = 157, 1100, 4, 16
    ei = torch.from_numpy(synth.random_edges(N, E, seed=13))
    g = torch.Generator().manual_seed(15)
    return N, E, n, h, ei, torch.randn(N, h, generator=g), torch.randn(E, h, generator=g), torch.randn(N, h, generator=g)


def _worker_layer(rank, world, port, case, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        N, E, n, h, ei, x, ea, dy = _layer_problem()
        conv = _make_conv(case, n, h, seed=21)                         # replicated parameters
        plan = build_halo_plan(ei, N)
        agg = ShardedAggregate(plan, backend=OracleBackend())
        layer = ShardedPHMGINEConv(conv, agg, N).train()
        x_own = x[plan.lo:plan.hi].clone().requires_grad_()
        ea_loc, ea_halo = agg.split_edge_attr(ea)
        ea_loc, ea_halo = ea_loc.requires_grad_(), ea_halo.requires_grad_()
        y = layer(x_own, ea_loc, ea_halo)
        y.backward(dy[plan.lo:plan.hi])
        local = {k: p.grad.clone() for k, p in conv.named_parameters()}
        sync = FlatGradSync(list(conv.parameters()), average=False)      # partial sums over own rows -> the gradient
        sync()
        torch.save(dict(y=y.detach(), gx=x_own.grad, gl=ea_loc.grad, gh=ea_halo.grad, loc_ids=plan.loc_edge_ids,
                        halo_ids=plan.halo_edge_ids, lo=plan.lo, hi=plan.hi, local=local,
                        grads={k: p.grad.clone() for k, p in conv.named_parameters()},
                        buffers={k: b.clone() for k, b in conv.named_buffers()}), os.path.join(tmp, f"l{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", list(LAYER_CASES))
def test_sharded_gine_layer_equals_single_process(tmp_path, case):
    """Two shards of one graph through ShardedPHMGINEConv == the unsharded layer (oracle restatement of
    messagepassing.py:132-146 + layers.py:349-355): output rows, dx, d edge_attr, every parameter gradient after the
    flat SUM all-reduce, and the batch-norm running statistics (identical on both ranks: global batch statistics)."""
    world = 2
    mp.spawn(_worker_layer, args=(world, _free_port(), case, str(tmp_path)), nprocs=world, join=True)
    N, E, n, h, ei, x, ea, dy = _layer_problem()
    c = LAYER_CASES[case]
    conv = _make_conv(case, n, h, seed=21)
    p = {"q." + k: v.detach().clone() for k, v in conv.state_dict().items()}
    names = ["q." + k for k, _ in conv.named_parameters()]
    for k in names:
        p[k].requires_grad_()
    xr, er = x.clone().requires_grad_(), ea.clone().requires_grad_()
    ref = O.phm_gine_conv(xr, ei, er, p, "q.transform.", n, c["act"], c["norm"] is not None, True, True, c["msg"], None)
    grads = torch.autograd.grad(ref, [xr, er] + [p[k] for k in names], dy, allow_unused=True)
    ref_param = {k[2:]: g for k, g in zip(names, grads[2:])}
    shards = [torch.load(os.path.join(tmp_path, f"l{r}.pt")) for r in range(world)]
    y, gx, ge = torch.zeros(N, h), torch.zeros(N, h), torch.zeros(E, h)
    for d in shards:
        y[d["lo"]:d["hi"]] = d["y"]
        gx[d["lo"]:d["hi"]] = d["gx"]
        ge[d["loc_ids"]] = d["gl"]
        ge[d["halo_ids"]] = d["gh"]

    def close(a, b, what, tol=2e-5):
        err = float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)
        assert err <= tol, (what, err)
    close(y, ref, "y")
    close(gx, grads[0], "dx")
    close(ge, grads[1], "d edge_attr")
    for k, g in ref_param.items():
        if g is None or float(g.abs().max()) < 1e-5:           # bias in front of the norm: zero gradient up to round-off
            continue
        close(shards[0]["grads"][k], g, "grad " + k)
        assert torch.equal(shards[0]["grads"][k], shards[1]["grads"][k]), k
        # each rank really held a PARTIAL sum before the all-reduce
        close(shards[0]["local"][k] + shards[1]["local"][k], g, "partial sums " + k)
    for k, b in shards[0]["buffers"].items():
        assert torch.equal(b, shards[1]["buffers"][k]), k          # same global statistics on every rank
        if b.is_floating_point():
            close(b, p["q." + k], "buffer " + k)
        else:
            assert int(b) == 1, k                                  # num_batches_tracked: one training step


# ---- two stacked layers with the model-level norm + skip between them (ShardedGINEStack) -------------------------------
def _make_stack(n, h, seed):
    from phc.hypercomplex.norm import PHMNorm
    convs = [_make_conv("relu_norm", n, h, seed + i) for i in range(2)]
    g = torch.Generator().manual_seed(seed + 50)
    norms = []
    for i in range(2):
        nm = PHMNorm(num_features=h, phm_dim=n, type="naive-batch-norm")
        with torch.no_grad():
            for name, p in nm.named_parameters():
                p.copy_(0.5 + torch.rand(p.shape, generator=g) if name.endswith("weight") else 0.1 * torch.randn(p.shape, generator=g))
        norms.append(nm.train())
    return convs, norms


def _stack_problem(world, kind):
    N, E, n, h, ei, x, ea, dy = _layer_problem()
    if kind != "uniform":
        ei = _graph(N, E, world, kind, seed=13)
        ea = ea[:ei.shape[1]].contiguous()
    return N, ei.shape[1], n, h, ei, x, ea, dy


def _worker_stack(rank, world, port, tmp, kind="uniform", max_pair_bytes=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if max_pair_bytes is not None:
            import phc_gnn_amd.dist as D
            D.MAX_PAIR_BYTES = max_pair_bytes
        N, E, n, h, ei, x, ea, dy = _stack_problem(world, kind)
        convs, norms = _make_stack(n, h, seed=31)
        plan = build_halo_plan(ei, N)
        stack = ShardedGINEStack(convs, norms, plan, N, activation="relu", backend=OracleBackend()).train()
        x_own = x[plan.lo:plan.hi].clone().requires_grad_()
        ea_loc, ea_halo = stack.agg.split_edge_attr(ea)
        # ONE edge-attribute tensor feeds both layers and wants its gradient: the stack gathers it once from the sum of the layers'
        # aggregate gradients (dist.EdgeGradSink)
        ea_loc = ea_loc.clone().requires_grad_()
        ea_halo = ea_halo.clone().requires_grad_()
        y = stack(x_own, ea_loc, ea_halo)
        y.backward(dy[plan.lo:plan.hi])
        params = [p for m in convs + norms for p in m.parameters()]
        FlatGradSync(params, average=False)()
        zero = torch.zeros(0, h)
        torch.save(dict(y=y.detach(), gx=x_own.grad, lo=plan.lo, hi=plan.hi, grads=[p.grad.clone() for p in params],
                        buffers=[b.clone() for m in convs + norms for b in m.buffers()],
                        dea_loc=ea_loc.grad if ea_loc.grad is not None else zero, loc_ids=plan.loc_edge_ids.clone(),
                        dea_halo=ea_halo.grad if ea_halo.grad is not None else zero, halo_ids=plan.halo_edge_ids.clone()),
                   os.path.join(tmp, f"s{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,kind,max_pair_bytes", [(2, "uniform", None), (3, "uniform", 900), (4, "holes", None), (8, "holes", 400)])
def test_sharded_stack_with_model_level_norm_equals_single_process(tmp_path, world, kind, max_pair_bytes):
    """conv -> norm(global statistics) -> relu -> + skip, twice, on 2 / 3 / 4 / 8 shards == the oracle on the whole graph:
    outputs, dx, every parameter gradient (convs and norms), running statistics — incl. shards without halo rows, without any
    edge, empty peer segments and chunked exchanges ("holes", max_pair_bytes)."""
    mp.spawn(_worker_stack, args=(world, _free_port(), str(tmp_path), kind, max_pair_bytes), nprocs=world, join=True)
    N, E, n, h, ei, x, ea, dy = _stack_problem(world, kind)
    convs, norms = _make_stack(n, h, seed=31)
    p = {}
    for i, (c, nm) in enumerate(zip(convs, norms)):
        p.update({f"c{i}." + k: v.detach().clone() for k, v in c.state_dict().items()})
        p.update({f"n{i}." + k: v.detach().clone() for k, v in nm.state_dict().items()})
    names = [f"c{i}." + k for i, c in enumerate(convs) for k, _ in c.named_parameters()] + \
            [f"n{i}." + k for i, nm in enumerate(norms) for k, _ in nm.named_parameters()]
    order = [f"c{i}." + k for i, c in enumerate(convs) for k, _ in c.named_parameters()]
    order += [f"n{i}." + k for i, nm in enumerate(norms) for k, _ in nm.named_parameters()]
    for k in names:
        p[k].requires_grad_()
    xr = x.clone().requires_grad_()
    ear = ea.clone().requires_grad_()
    hcur = xr
    for i in range(2):
        y = O.phm_gine_conv(hcur, ei, ear, p, f"c{i}.transform.", n, "relu", True, True, True, "relu", None)
        hcur = O.activation("relu", O._norm_from_state(y, p, f"n{i}.", n, True)) + hcur
    grads = torch.autograd.grad(hcur, [xr, ear] + [p[k] for k in order], dy, allow_unused=True)
    dea_ref, grads = grads[1], (grads[0],) + tuple(grads[2:])
    shards = [torch.load(os.path.join(tmp_path, f"s{r}.pt")) for r in range(world)]
    yy, gx, dea = torch.zeros(N, h), torch.zeros(N, h), torch.zeros(ei.shape[1], h)
    seen = torch.zeros(ei.shape[1], dtype=torch.int64)
    for d in shards:
        yy[d["lo"]:d["hi"]] = d["y"]
        gx[d["lo"]:d["hi"]] = d["gx"]
        for ids, g in ((d["loc_ids"], d["dea_loc"]), (d["halo_ids"], d["dea_halo"])):
            if ids.numel():
                dea[ids] = g
                seen[ids] += 1
    assert bool((seen == 1).all())                                    # every edge belongs to exactly one rank's local or halo set

    def close(a, b, what, tol=3e-5):
        err = float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)
        assert err <= tol, (what, err)
    close(yy, hcur.detach(), "y")
    close(gx, grads[0], "dx")
    close(dea, dea_ref, "d edge_attr (one gather from the sum of both layers' aggregate gradients)")
    for k, got, ref in zip(order, shards[0]["grads"], grads[1:]):
        if ref is None or k.endswith("linear1.b") or k.endswith("linear2.b"):     # biases in front of a norm: zero up to round-off
            continue
        close(got, ref, "grad " + k)
    for other in shards[1:]:
        for a, b in zip(shards[0]["buffers"], other["buffers"]):
            assert torch.equal(a, b)
        for a, b in zip(shards[0]["grads"], other["grads"]):
            assert torch.equal(a, b)                                  # one summed gradient on every rank


# ---- eval mode (validation between epochs): running statistics, the halo exchange is the only collective ------------------------
def _worker_stack_eval(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        N, E, n, h, ei, x, ea, dy = _stack_problem(world, "uniform")
        convs, norms = _make_stack(n, h, seed=31)
        _set_running_stats(convs, norms)
        plan = build_halo_plan(ei, N)
        stack = ShardedGINEStack(convs, norms, plan, N, activation="relu", backend=OracleBackend()).eval()
        before = [b.clone() for m in convs + norms for b in m.buffers()]
        ea_loc, ea_halo = stack.agg.split_edge_attr(ea.detach())
        y = stack(x[plan.lo:plan.hi].detach().clone(), ea_loc, ea_halo)
        after = [b.clone() for m in convs + norms for b in m.buffers()]
        assert all(torch.equal(a, b) for a, b in zip(before, after))     # eval leaves the running statistics alone
        assert all(m.training is False for m in stack.modules())
        # forward only: with PHC_EVAL_GRAD_STRICT=1 an input that requires grad is refused instead of answered with a detached tensor
        # (by default a warning) — before any collective of the call, so every rank raises alike
        os.environ["PHC_EVAL_GRAD_STRICT"] = "1"
        try:
            stack(x[plan.lo:plan.hi].detach().clone().requires_grad_(), ea_loc, ea_halo)
            raised = False
        except NotImplementedError:
            raised = True
        finally:
            os.environ.pop("PHC_EVAL_GRAD_STRICT")
        assert raised
        torch.save(dict(y=y, lo=plan.lo, hi=plan.hi), os.path.join(tmp, f"e{rank}.pt"))
    finally:
        dist.destroy_process_group()


def _set_running_stats(convs, norms):
    g = torch.Generator().manual_seed(77)
    for m in convs + norms:
        for name, b in m.named_buffers():
            if name.endswith("running_mean"):
                b.copy_(0.2 * torch.randn(b.shape, generator=g))
            elif name.endswith("running_var"):
                b.copy_(0.5 + torch.rand(b.shape, generator=g))


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_stack_eval_mode_uses_running_statistics(tmp_path, world):
    """``ShardedGINEStack.eval()`` (validation between epochs) == the oracle's eval-mode stack on the whole graph
    (norm.py:30-35 with the running statistics; models.py:208-215), buffers untouched."""
    mp.spawn(_worker_stack_eval, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    N, E, n, h, ei, x, ea, dy = _stack_problem(world, "uniform")
    convs, norms = _make_stack(n, h, seed=31)
    _set_running_stats(convs, norms)
    p = {}
    for i, (c, nm) in enumerate(zip(convs, norms)):
        p.update({f"c{i}." + k: v.detach().clone() for k, v in c.state_dict().items()})
        p.update({f"n{i}." + k: v.detach().clone() for k, v in nm.state_dict().items()})
    hcur = x
    with torch.no_grad():
        for i in range(2):
            y = O.phm_gine_conv(hcur, ei, ea, p, f"c{i}.transform.", n, "relu", True, False, True, "relu", None)
            hcur = O.activation("relu", O._norm_from_state(y, p, f"n{i}.", n, False)) + hcur
    yy = torch.zeros(N, h)
    for r in range(world):
        d = torch.load(os.path.join(tmp_path, f"e{r}.pt"))
        yy[d["lo"]:d["hi"]] = d["y"]
    err = float((yy - hcur).abs().max()) / max(float(hcur.abs().max()), 1e-30)
    assert err <= 1e-5, err


# ---- the exchange's start-up check: a transport that delivers a wrong row fails loudly, on every rank -----------------------
def _worker_corrupt(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import phc_gnn_amd.dist as D
        real = dist.all_to_all_single

        def flaky(output, input, *a, **kw):                      # noqa: A002 (torch's own parameter names)
            w = real(output, input, *a, **kw)
            if w is not None:
                w.wait()
            if rank == 1 and output.dim() == 2 and output.is_floating_point() and output.shape[0] > 3:
                output[3, 0] += 1.0                              # one wrong value in one received row, on one rank
            return None
        N, h = 203, 16
        ei = _graph(N, 1500, world, "uniform", seed=3)
        plan = build_halo_plan(ei, N)
        agg = ShardedAggregate(plan, backend=OracleBackend())
        D.dist.all_to_all_single = flaky
        x = torch.randn(N, h, generator=torch.Generator().manual_seed(5))
        msg = "no error"
        try:
            agg(x[plan.lo:plan.hi].clone(), None, None, "identity", True)
        except RuntimeError as exc:
            msg = str(exc)
        finally:
            D.dist.all_to_all_single = real
        open(os.path.join(tmp, f"c{rank}.txt"), "w").write(msg)
    finally:
        dist.destroy_process_group()


def test_exchange_self_check_catches_a_corrupted_row_on_every_rank(tmp_path):
    world = 2
    mp.spawn(_worker_corrupt, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    msgs = [open(os.path.join(tmp_path, f"c{r}.txt")).read() for r in range(world)]
    assert all("halo exchange self check failed" in m for m in msgs), msgs
    assert "rank 1 received 1 wrong rows" in msgs[1] and "rank 0 received 0 wrong rows" in msgs[0], msgs


def _worker_stack_reentrant(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        N, E, n, h, ei, x, ea, dy = _stack_problem(world, "uniform")
        convs, norms = _make_stack(n, h, seed=31)
        plan = build_halo_plan(ei, N)
        stack = ShardedGINEStack(convs, norms, plan, N, activation="relu", backend=OracleBackend()).train()
        x_own = x[plan.lo:plan.hi].clone().requires_grad_()
        ea_loc, ea_halo = stack.agg.split_edge_attr(ea)
        ea_loc = ea_loc.clone().requires_grad_()
        ea_halo = ea_halo.clone().requires_grad_()
        y = stack(x_own, ea_loc, ea_halo)
        dyo = dy[plan.lo:plan.hi]
        # (1) the edge gradient ALONE (autograd.grad, graph retained), (2) a full backward over the retained graph, (3) a second
        # full backward: .grad accumulates — every one of them must see the whole edge gradient
        only = torch.autograd.grad(y, [ea_loc, ea_halo], dyo, retain_graph=True, allow_unused=True)
        y.backward(dyo, retain_graph=True)
        first = [t.grad.clone() if t.grad is not None else None for t in (ea_loc, ea_halo, x_own)]
        y.backward(dyo)
        second = [t.grad.clone() if t.grad is not None else None for t in (ea_loc, ea_halo, x_own)]
        torch.save(dict(only=only, first=first, second=second), os.path.join(tmp, f"re{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_edge_grad_sink_is_reentrant(tmp_path):
    """ADVICE r5 (dist.EdgeGradSink): a second backward over a retained graph, and a backward that asks for the edge gradient alone,
    get the same edge-attribute gradient as the first full backward (the sink used to count down once and then return nothing)."""
    world = 2
    mp.spawn(_worker_stack_reentrant, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        d = torch.load(os.path.join(tmp_path, f"re{r}.pt"))
        for i in range(2):                                   # local / halo edge sets
            if d["first"][i] is None:
                assert d["only"][i] is None and d["second"][i] is None
                continue
            assert float(d["first"][i].abs().max()) > 0
            torch.testing.assert_close(d["only"][i], d["first"][i], rtol=1e-6, atol=1e-6)
            torch.testing.assert_close(d["second"][i], 2 * d["first"][i], rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(d["second"][2], 2 * d["first"][2], rtol=1e-6, atol=1e-6)
