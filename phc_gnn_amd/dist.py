"""Multi-GPU operation of the message-passing path.  One process per GPU,
``torch.distributed`` (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference has no distributed code at all; both modes are new design whose only
contract is "same numbers as one GPU":

1. Mini-batches of small graphs (BASELINE configs 1-4): graphs never share edges, so every rank
   trains on its own batches and the only exchange is ONE flat-bucket all-reduce of the gradients
   (``FlatGradSync``).  ~0.3 MB at config 2: latency-bound, a single call.

2. One large graph (BASELINE config 5): destination-range sharding.  Rank g owns the contiguous node
   range [lo_g, hi_g) and every in-edge of those nodes.  Source rows owned by other ranks ("halo") are
   pulled once per layer with ONE all-to-all(v) of feature rows — point-to-point pairs, so each of the
   7 xGMI links of a GPU carries its own peer's rows concurrently — issued asynchronously and overlapped
   with the aggregation over the purely local edges; the halo edges are aggregated after the rows land.
   Backward mirrors it: gradient rows of halo sources travel back with the reverse all-to-all while the
   local backward runs, and are summed into the owners' rows in a fixed order (deterministic).

The compute steps are the HIP engine's (``phc_gnn_amd.ops``).  ``ShardedAggregate`` takes a ``backend``
object so that the exchange logic can be exercised on CPU under gloo with a stand-in backend; the
default backend has no CPU path.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist


# ---------------------------------------------------------------------------------------------
# 1. data-parallel graph batches: one flat gradient bucket
# ---------------------------------------------------------------------------------------------
class FlatGradSync:
    """All-reduce (mean) of every parameter gradient through ONE contiguous fp32 bucket."""

    def __init__(self, params: Sequence[torch.nn.Parameter], group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.world = dist.get_world_size(group)
        self.sizes = [p.numel() for p in self.params]
        self.flat: Optional[torch.Tensor] = None

    def pack(self) -> torch.Tensor:
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in self.params]
        if self.flat is None:
            self.flat = torch.empty(sum(self.sizes), dtype=torch.float32, device=grads[0].device)
        torch.cat([g.reshape(-1) for g in grads], out=self.flat)
        return self.flat

    def reduce(self) -> None:
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)

    def unpack(self) -> None:
        chunks = self.flat.split(self.sizes)
        with torch.no_grad():
            for p, c in zip(self.params, chunks):
                if p.grad is None:
                    p.grad = torch.empty_like(p)
            torch._foreach_copy_([p.grad for p in self.params], [c.view_as(p) for p, c in zip(self.params, chunks)])
            torch._foreach_div_([p.grad for p in self.params], float(self.world))

    def __call__(self, _model=None) -> None:
        self.pack()
        self.reduce()
        self.unpack()


# ---------------------------------------------------------------------------------------------
# 2. one large graph: destination-range shards + halo exchange
# ---------------------------------------------------------------------------------------------
def partition_bounds(num_nodes: int, world: int) -> List[int]:
    """world+1 boundaries of contiguous, near-equal node ranges."""
    base, rem = divmod(num_nodes, world)
    bounds = [0]
    for r in range(world):
        bounds.append(bounds[-1] + base + (1 if r < rem else 0))
    return bounds


@dataclass
class HaloPlan:
    """Everything rank-local about one sharded graph (built once per graph, collectively)."""
    rank: int
    world: int
    lo: int
    hi: int
    n_own: int
    n_halo: int
    halo_ids: torch.Tensor            # [n_halo] global ids of the remote sources, grouped by owner rank
    recv_splits: List[int]            # halo rows received from each rank
    send_rows: torch.Tensor           # [sum(send_splits)] LOCAL row ids to send, grouped by destination rank
    send_splits: List[int]
    loc_edge_index: torch.Tensor      # [2, E_loc]  src, dst both in own-local numbering
    loc_edge_ids: torch.Tensor        # [E_loc]     positions in the caller's edge list (for edge_attr)
    halo_edge_index: torch.Tensor     # [2, E_halo] src in halo-local numbering, dst in own-local numbering
    halo_edge_ids: torch.Tensor       # [E_halo]
    group: Optional[object] = None


def build_halo_plan(edge_index: torch.Tensor, num_nodes: int, group=None) -> HaloPlan:
    """``edge_index`` [2, E] int64 (row 0 = source, row 1 = target) must contain every in-edge of the nodes
    this rank owns (the full edge list works).  Collective: every rank of ``group`` must call it."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    bounds = partition_bounds(num_nodes, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    dev = edge_index.device
    src, dst = edge_index[0], edge_index[1]
    mine = ((dst >= lo) & (dst < hi)).nonzero(as_tuple=False).flatten()
    s, d = src[mine], dst[mine] - lo
    remote = (s < lo) | (s >= hi)
    loc_pos, halo_pos = (~remote).nonzero(as_tuple=False).flatten(), remote.nonzero(as_tuple=False).flatten()
    halo_ids = torch.unique(s[halo_pos])                                # sorted => grouped by owner (ranges are contiguous)
    bnd = torch.tensor(bounds[1:], device=dev, dtype=torch.int64)
    owner = torch.bucketize(halo_ids, bnd, right=True)
    need = torch.bincount(owner, minlength=world).to(torch.int64)

    # who wants what from me: exchange the counts, then the global ids
    asked = torch.empty_like(need)
    dist.all_to_all_single(asked, need, group=group)
    need_l, asked_l = need.tolist(), asked.tolist()
    req = torch.empty(int(sum(asked_l)), dtype=torch.int64, device=dev)
    dist.all_to_all_single(req, halo_ids.contiguous(), output_split_sizes=asked_l, input_split_sizes=need_l, group=group)
    if req.numel() and (int(req.min()) < lo or int(req.max()) >= hi):
        raise RuntimeError("halo plan: received a request for a row this rank does not own")

    loc_ei = torch.stack([s[loc_pos] - lo, d[loc_pos]], 0)
    halo_src = torch.searchsorted(halo_ids, s[halo_pos])
    halo_ei = torch.stack([halo_src, d[halo_pos]], 0)
    return HaloPlan(rank=rank, world=world, lo=lo, hi=hi, n_own=hi - lo, n_halo=int(halo_ids.numel()),
                    halo_ids=halo_ids, recv_splits=need_l, send_rows=(req - lo).contiguous(), send_splits=asked_l,
                    loc_edge_index=loc_ei.contiguous(), loc_edge_ids=mine[loc_pos], halo_edge_index=halo_ei.contiguous(),
                    halo_edge_ids=mine[halo_pos], group=group)


class EngineBackend:
    """The production compute backend: HIP engine kernels (no CPU path)."""

    def __init__(self):
        from . import ops as _ops
        from .graph import EdgeStructure, SegmentStructure
        self._ops, self._ES, self._SS = _ops, EdgeStructure, SegmentStructure

    def structure(self, edge_index, num_src, num_dst):
        return self._ES(edge_index, num_src, num_dst)

    def aggregate_fwd(self, x, ea, st, act, add_self):
        return self._ops.aggregate_fwd_raw(x, ea, st, act, add_self)

    def aggregate_bwd(self, dout, x, ea, st, act, add_self, need_dea):
        return self._ops.aggregate_bwd_raw(dout, x, ea, st, act, add_self, need_dea)

    def gather_rows(self, x, rows):
        return x.index_select(0, rows)

    def segment_sum(self, rows_values, rows_index, num_rows):
        """out[r] = sum of rows_values[j] with rows_index[j] == r, fixed order."""
        return self._ops.global_add_pool(rows_values, self._SS(rows_index, num_rows))


class _ShardedAggregateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_own, ea_loc, ea_halo, agg: "ShardedAggregate", act: str, add_self: bool):
        plan, be = agg.plan, agg.backend
        h = x_own.shape[1]
        x_own = x_own.contiguous()
        send = be.gather_rows(x_own, plan.send_rows)                               # staging for the all-to-all
        recv = torch.empty((plan.n_halo, h), dtype=x_own.dtype, device=x_own.device)
        work = dist.all_to_all_single(recv, send, output_split_sizes=plan.recv_splits,
                                      input_split_sizes=plan.send_splits, group=plan.group, async_op=True)
        out = be.aggregate_fwd(x_own, ea_loc, agg.st_loc, act, add_self)            # overlaps the exchange
        work.wait()
        if plan.halo_edge_index.shape[1] > 0:
            out = out + be.aggregate_fwd(recv, ea_halo, agg.st_halo, act, False)
        ctx.agg, ctx.act, ctx.add_self = agg, act, add_self
        ctx.save_for_backward(x_own, recv, ea_loc if ea_loc is not None else x_own.new_empty(0),
                              ea_halo if ea_halo is not None else x_own.new_empty(0))
        ctx.has_ea = (ea_loc is not None, ea_halo is not None)
        return out

    @staticmethod
    def backward(ctx, dout):
        agg = ctx.agg
        plan, be = agg.plan, agg.backend
        x_own, recv, ea_loc, ea_halo = ctx.saved_tensors
        ea_loc = ea_loc if ctx.has_ea[0] else None
        ea_halo = ea_halo if ctx.has_ea[1] else None
        dout = dout.contiguous()
        h = dout.shape[1]
        need_dea = (ctx.needs_input_grad[1] and ea_loc is not None, ctx.needs_input_grad[2] and ea_halo is not None)
        # gradient of the halo rows first: it has to travel
        if plan.n_halo > 0 and plan.halo_edge_index.shape[1] > 0:
            d_halo, dea_halo = be.aggregate_bwd(dout, recv, ea_halo, agg.st_halo, ctx.act, False, need_dea[1])
        else:
            d_halo = dout.new_zeros((plan.n_halo, h))
            dea_halo = dout.new_zeros((0, h)) if need_dea[1] else None
        back = torch.empty((int(plan.send_rows.numel()), h), dtype=dout.dtype, device=dout.device)
        work = dist.all_to_all_single(back, d_halo.contiguous(), output_split_sizes=plan.send_splits,
                                      input_split_sizes=plan.recv_splits, group=plan.group, async_op=True)
        d_own, dea_loc = be.aggregate_bwd(dout, x_own, ea_loc, agg.st_loc, ctx.act, ctx.add_self, need_dea[0])
        work.wait()
        if back.shape[0] > 0:
            d_own = d_own + be.segment_sum(back, plan.send_rows, plan.n_own)       # fixed order: deterministic
        return d_own, dea_loc, dea_halo, None, None, None


class ShardedAggregate:
    """out[i] = sum_{j->i} phi(x[j] + e_ji) (+ x[i]) for the nodes this rank owns, with the remote x[j]
    fetched by one all-to-all per call (reference op: messagepassing.py:57-60,72-74 on a single device)."""

    def __init__(self, plan: HaloPlan, backend=None):
        self.plan = plan
        self.backend = backend if backend is not None else EngineBackend()
        self.st_loc = self.backend.structure(plan.loc_edge_index, plan.n_own, plan.n_own)
        self.st_halo = self.backend.structure(plan.halo_edge_index, max(plan.n_halo, 1), plan.n_own)

    def split_edge_attr(self, edge_attr: torch.Tensor):
        """(rows of the local edges, rows of the halo edges) of a tensor aligned with the caller's edge list."""
        return edge_attr.index_select(0, self.plan.loc_edge_ids), edge_attr.index_select(0, self.plan.halo_edge_ids)

    def __call__(self, x_own: torch.Tensor, ea_loc: Optional[torch.Tensor], ea_halo: Optional[torch.Tensor],
                 msg_encoder: str = "identity", add_self_loops: bool = True) -> torch.Tensor:
        return _ShardedAggregateFn.apply(x_own, ea_loc, ea_halo, self, msg_encoder, add_self_loops)
