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
from phc_gnn_amd.dist import FlatGradSync, ShardedAggregate, build_halo_plan, partition_bounds


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

    def aggregate_fwd(self, x, ea, st, act, add_self):
        msg = x.index_select(0, st.ei[0])
        if ea is not None:
            msg = msg + ea
        out = O.scatter_sum_rows(O.activation(act, msg), st.ei[1], st.num_dst)
        return out + x[:st.num_dst] if add_self else out

    def aggregate_bwd(self, dout, x, ea, st, act, add_self, need_dea):
        with torch.enable_grad():
            xr = x.detach().clone().requires_grad_()
            er = ea.detach().clone().requires_grad_() if ea is not None else None
            out = self.aggregate_fwd(xr, er, st, act, add_self)
            grads = torch.autograd.grad(out, [xr] + ([er] if er is not None else []), dout, allow_unused=True)
        dx = grads[0] if grads[0] is not None else torch.zeros_like(x)
        return dx, (grads[1] if (need_dea and er is not None) else None)

    def gather_rows(self, x, rows):
        return x.index_select(0, rows)

    def segment_sum(self, vals, idx, n):
        return O.scatter_sum_rows(vals, idx, n)


def _worker_sharded(rank, world, port, act, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        N, E, h = 203, 1500, 16
        ei = torch.from_numpy(synth.random_edges(N, E, seed=3))
        g = torch.Generator().manual_seed(5)
        x = torch.randn(N, h, generator=g)
        ea = torch.randn(E, h, generator=g)
        dy = torch.randn(N, h, generator=g)
        plan = build_halo_plan(ei, N)
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


@pytest.mark.parametrize("act", ["identity", "relu"])
def test_sharded_aggregate_equals_single_process(tmp_path, act):
    world = 2
    mp.spawn(_worker_sharded, args=(world, _free_port(), act, str(tmp_path)), nprocs=world, join=True)
    N, E, h = 203, 1500, 16
    ei = torch.from_numpy(synth.random_edges(N, E, seed=3))
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
