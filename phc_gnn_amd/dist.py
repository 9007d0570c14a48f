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

import os
import warnings
from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist


# ---------------------------------------------------------------------------------------------
# 1. data-parallel graph batches: one flat gradient bucket
# ---------------------------------------------------------------------------------------------
class FlatGradSync:
    """All-reduce (mean) of every parameter gradient through ONE contiguous fp32 bucket.

    With the engine's optimizer (``optimizer=ClipAdam`` over the same parameters, on the GPU) the bucket is the optimizer's own
    flat gradient buffer: ONE engine launch packs every ``.grad`` into it with the 1 / world of the mean folded in
    (``phc_pack_grads``), the all-reduce runs in place, and the parameters' ``.grad`` become views of the bucket — the
    clip + Adam launches read the reduced values where the collective left them: no unpack pass, no separate division.
    Without it (CPU tensors under gloo, ``torch.optim`` optimizers): ``torch.cat`` / ``_foreach_copy_`` / ``_foreach_div_``."""

    def __init__(self, params: Sequence[torch.nn.Parameter], group=None, average: bool = True, optimizer=None):
        """``average=True``: data-parallel replicas (mean of the per-replica gradients); ``average=False``: shards of
        ONE batch (config 5: every rank holds the partial sums over its own rows; the gradient is their plain sum)."""
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.world = dist.get_world_size(group)
        self.average = bool(average)
        self.sizes = [p.numel() for p in self.params]
        self.flat: Optional[torch.Tensor] = None
        self.opt = None
        if optimizer is not None and hasattr(optimizer, "pack_grads") and all(p.is_cuda for p in self.params) and \
                len(optimizer.params) == len(self.params) and all(a is b for a, b in zip(optimizer.params, self.params)):
            self.opt = optimizer

    def pack(self) -> torch.Tensor:
        if self.opt is not None:
            self.flat = self.opt.pack_grads(1.0 / self.world if self.average else 1.0)
            return self.flat
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in self.params]
        if self.flat is None:
            self.flat = torch.empty(sum(self.sizes), dtype=torch.float32, device=grads[0].device)
        torch.cat([g.reshape(-1) for g in grads], out=self.flat)
        return self.flat

    def reduce(self) -> None:
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)

    def unpack(self) -> None:
        if self.opt is not None:
            self.opt.adopt_bucket_grads()                       # views of the bucket: nothing to launch
            return
        chunks = self.flat.split(self.sizes)
        with torch.no_grad():
            for p, c in zip(self.params, chunks):
                if p.grad is None:
                    p.grad = torch.empty_like(p)
            torch._foreach_copy_([p.grad for p in self.params], [c.view_as(p) for p, c in zip(self.params, chunks)])
            if self.average:
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
    max_pair_rows: int = 0            # largest row count any PAIR of ranks exchanges (same value on every rank; 0 = unknown)
    serial: int = -1                  # number of this plan, AGREED by every rank of the group (key of the exchange's start-up check)


_next_plan_serial = 0


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

    # (host reads above: once per graph; the plan is reused by every layer and every step)
    # ... and one plan number every rank agrees on (ranks may have built different numbers of plans in other groups before)
    global _next_plan_serial
    biggest = torch.tensor([max(need_l + asked_l + [0]), _next_plan_serial], dtype=torch.int64, device=dev)
    dist.all_reduce(biggest, op=dist.ReduceOp.MAX, group=group)         # every rank must cut its exchanges alike (_exchange_rows)
    serial = int(biggest[1].item())
    _next_plan_serial = serial + 1
    loc_ei = torch.stack([s[loc_pos] - lo, d[loc_pos]], 0)
    halo_src = torch.searchsorted(halo_ids, s[halo_pos])
    halo_ei = torch.stack([halo_src, d[halo_pos]], 0)
    return HaloPlan(rank=rank, world=world, lo=lo, hi=hi, n_own=hi - lo, n_halo=int(halo_ids.numel()),
                    halo_ids=halo_ids, recv_splits=need_l, send_rows=(req - lo).contiguous(), send_splits=asked_l,
                    loc_edge_index=loc_ei.contiguous(), loc_edge_ids=mine[loc_pos], halo_edge_index=halo_ei.contiguous(),
                    halo_edge_ids=mine[halo_pos], group=group, max_pair_rows=int(biggest[0].item()), serial=serial)


class _CommLane:
    """The exchange's own HIP stream.  ``fork()`` makes it wait for everything queued on the compute stream so far and
    returns a context in which staging kernels and the collective are issued; ``join()`` makes the compute stream wait
    for the lane.  Work issued on the compute stream between the two overlaps the exchange.  On CPU (gloo tests) both
    are no-ops and the collective's own async handle provides the overlap."""

    def __init__(self, device: torch.device):
        self.device = device
        self.stream = torch.cuda.Stream(device=device) if device.type == "cuda" else None

    def fork(self):
        import contextlib
        if self.stream is None:
            return contextlib.nullcontext()
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        return torch.cuda.stream(self.stream)

    def join(self) -> None:
        if self.stream is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.stream)

    def keep(self, *tensors) -> None:
        """Tensors allocated on the compute stream but consumed on the lane (or vice versa): tell the caching allocator."""
        if self.stream is not None:
            cur = torch.cuda.current_stream(self.device)
            for t in tensors:
                if t is not None and t.is_cuda:
                    t.record_stream(self.stream)
                    t.record_stream(cur)


class EngineBackend:
    """The production compute backend: HIP engine kernels (no CPU path)."""

    def __init__(self):
        from . import ops as _ops
        from .graph import EdgeStructure, SegmentStructure
        self._ops, self._ES, self._SS = _ops, EdgeStructure, SegmentStructure

    def structure(self, edge_index, num_src, num_dst):
        return self._ES(edge_index, num_src, num_dst)

    def segments(self, index, num_rows):
        return self._SS(index, num_rows)

    def aggregate_fwd(self, x, ea, st, act, add_self, accumulate_into=None):
        return self._ops.aggregate_fwd_raw(x, ea, st, act, add_self, accumulate_into)

    def aggregate_bwd(self, dout, x, ea, st, act, add_self, need_dea, extra=None):
        """(dx, dea | None); ``extra``: another consumer's share of the gradient of x, added to dx (inside the walk where the
        row walker serves the call: identity messages, no edge gradient)."""
        if extra is not None and not need_dea and act == "identity":
            dx = self._ops.aggregate_dx_rows_raw(dout, st, add_self, extra)
            if dx is not None:
                return dx, None
        dx, dea = self._ops.aggregate_bwd_raw(dout, x, ea, st, act, add_self, need_dea)
        if extra is not None:
            dx += extra
        return dx, dea

    def gather_rows(self, x, rows):
        return self._ops.gather_rows(x, rows)                   # phc_gather_rows_i64

    def segment_sum_into(self, rows_values, seg, into):
        """into[r] += sum of rows_values[j] with index[j] == r, fixed order, accumulated inside the kernel."""
        return self._ops.aggregate_fwd_raw(rows_values, None, seg_as_edges(seg), 0, False, accumulate_into=into)

    # ---- PHM linear pieces (explicit forward / dx / parameter-gradient launches) ----
    def linear_fwd(self, x, A, W, b):
        with torch.no_grad():
            return self._ops.phm_linear(x, A, W, b)

    def linear_dx(self, dy, A, W):
        return self._ops.phm_linear_dx_raw(dy, A, W)

    # operand matrices built once per layer call (one launch for both linears): the H-stationary GEMM's k4 forms included
    def build_operands(self, pairs):
        ok = all((W.shape[0] * W.shape[1]) % 4 == 0 and (W.shape[0] * W.shape[2]) % 4 == 0 and W.shape[0] * W.shape[1] <= 256
                 and W.shape[0] * W.shape[2] <= 256 for _, W in pairs)
        return self._ops.phm_build_operands(pairs) if ok else None

    def linear_fwd_pre(self, x, pre, b):
        return self._ops.phm_linear_pre(x, pre, b)

    def linear_dx_pre(self, dy, pre):
        return self._ops.phm_linear_pre(dy, pre, None, dx=True)

    def linear_param_grads(self, x, dy, A, W, has_bias):
        return self._ops.phm_linear_param_grads_raw(x, dy, A, W, has_bias)

    # ---- row-sharded batch-norm phases ----
    def bn_moments(self, x):
        return self._ops.bn_moments(x)

    def bn_fwd_from_moments(self, x, moments_all, counts_all, n_total, norm, act, residual=None):
        mods = list(norm.bn)
        return self._ops.bn_fwd_from_moments(
            x, moments_all, counts_all, n_total, [m.weight for m in mods] if norm.affine else None,
            [m.bias for m in mods] if norm.affine else None, [m.running_mean for m in mods], [m.running_var for m in mods],
            [m.num_batches_tracked for m in mods], norm._momentum(), norm.eps,       # momentum=None is refused, as on one device
            self._ops._act_id(act), residual)

    def bn_eval(self, x, norm, act, residual=None):
        """act(norm(x)) (+ residual) with the RUNNING statistics (norm.py:30-35, eval mode): rows are independent — no exchange."""
        was = norm.training
        norm.train(False)
        try:
            with torch.no_grad():
                return norm.forward_fused(x, activation=act, residual=residual)
        finally:
            norm.train(was)

    def bn_bwd_sums(self, dy, x, save_stat, act):
        return self._ops.bn_bwd_sums(dy, x, save_stat, self._ops._act_id(act))

    def bn_bwd_apply(self, dy, x, save_stat, dwb_global, n_total, act):
        return self._ops.bn_bwd_apply(dy, x, save_stat, dwb_global, n_total, self._ops._act_id(act))

    # ---- the same norm folded into the GEMMs around it (engine only; the layer asks `norm_fusable` first) ----
    FUSED_NORM = __import__("os").environ.get("PHC_SHARDED_NORM_FUSED", "1") != "0"

    def norm_fusable(self, pre):
        """linear1 [in -> h], norm over h, linear2 [h -> out]: the folded forms need in and h in the H-stationary GEMM's range."""
        if not self.FUSED_NORM or pre is None:
            return False
        k_in, h = pre[0][0].shape
        return self._ops.sharded_norm_fusable(h, k_in) and pre[1][0].shape[0] == h

    def linear_fwd_stats(self, x, pre, b):
        """(z, moments [2h]): the GEMM's statistics epilogue instead of a pass over z."""
        z, acc = self._ops.gemm_pre_stats(x, pre, b)
        return z, self._ops.stat_acc_read(acc, z.shape[1], max(z.shape[0], 1))

    def bn_finalize_from_moments(self, moments_all, counts_all, n_total, norm, h):
        mods = list(norm.bn)
        return self._ops.bn_finalize_from_moments(
            moments_all, counts_all, n_total, [m.weight for m in mods] if norm.affine else None,
            [m.bias for m in mods] if norm.affine else None, [m.running_mean for m in mods], [m.running_var for m in mods],
            [m.num_batches_tracked for m in mods], norm._momentum(), norm.eps, h)

    def linear_fwd_xf(self, z, save_stat, act, pre, b):
        return self._ops.gemm_pre_xf(z, save_stat, self._ops._act_id(act), pre, b)

    def linear_fwd_xf_stats(self, z, save_stat, act, pre, b):
        """(y, moments [2h] of y): the normalising GEMM with the statistics epilogue for the norm behind the layer."""
        y, acc = self._ops.gemm_pre_xf(z, save_stat, self._ops._act_id(act), pre, b, want_stats=True)
        return y, self._ops.stat_acc_read(acc, y.shape[1], max(y.shape[0], 1))

    def bn_bwd_sums_acc(self, dy, x, save_stat, act):
        return self._ops.bn_bwd_sums_acc(dy, x, save_stat, self._ops._act_id(act))

    def acc_totals(self, acc, h):
        return self._ops.stat_acc_read(acc, h, 0)

    def linear_dx_bwd_apply(self, dy, x, save_stat, sums_acc, n_total, act, pre):
        return self._ops.gemm_pre_bwd_apply(dy, x, save_stat, sums_acc, n_total, self._ops._act_id(act), pre)

    def linear_param_grads_xf(self, x, save_stat, act, dy, A, W, has_bias):
        return self._ops.phm_param_grads_xf(x, save_stat, self._ops._act_id(act), dy, A, W, has_bias)


def seg_as_edges(seg):
    """A SegmentStructure (rows grouped by an index) read as the CSR the aggregation kernel walks."""
    class _View:
        pass
    v = _View()
    v.csr = seg.csr
    v.num_dst = seg.num_segments
    v.num_edges = int(seg.batch.numel())
    return v


# One message between a pair of ranks stays below this: a single 2.5 GB self-exchange came back corrupted from RCCL 2.26 /
# ROCm 7 (found by tests/test_gpu_config5_fullsize.py: 4.97 M rows x 512 B in ONE all-to-all on one rank), so larger exchanges
# are cut into row chunks.  At BASELINE config 5 on 8 ranks a pair exchanges ~0.3 GB: one call, as before.
MAX_PAIR_BYTES = 1 << 30


def _chunk_bounds(n: int, chunks: int):
    return [(n * k) // chunks for k in range(chunks + 1)]


# ---- start-up check of the exchange ------------------------------------------------------------------------------------
# The first exchange of every (lane, plan, direction, width) is preceded by ONE exchange of the same shape whose payload every
# receiver can predict: row j of the segment rank s sends to rank d holds (s * A + d * B + 97 j + column) mod 65521 — exact in
# fp32.  It travels the very path the real rows take (same splits, same chunk cut, same backend), so a transport that delivers
# wrong rows at this size — as one > 2 GiB RCCL all-to-all did — fails LOUDLY at start-up on every rank (the mismatch count is
# all-reduced, so no rank is left waiting in a collective) instead of training on silently corrupted halo rows.
# The check is itself collective (an all-to-all and an all-reduce), so whether it runs must be decided IDENTICALLY on every rank:
# its key is (HaloPlan.serial — agreed by an all-reduce when the plan is built —, direction, width, dtype, chunk size), never a
# rank's own split lists (a rank whose send and receive splits coincide would take the backward exchange for already verified
# while its peers run the check).  PHC_EXCHANGE_SELFCHECK=0 skips it.
EXCHANGE_SELFCHECK = __import__("os").environ.get("PHC_EXCHANGE_SELFCHECK", "1") != "0"
_PAT_A, _PAT_B, _PAT_M = 1315423911, 2654435761, 65521


def _pattern_rows(splits, h: int, me: int, me_is_sender: bool, device, dtype) -> torch.Tensor:
    """Rows of the check payload, grouped by peer like the splits: what `me` sends (me_is_sender) or must receive."""
    parts = []
    col = torch.arange(h, dtype=torch.int32, device=device)
    for peer, n in enumerate(splits):
        if n == 0:
            continue
        s, d = (me, peer) if me_is_sender else (peer, me)
        base = (s * _PAT_A + d * _PAT_B) % _PAT_M               # (host integers; everything on the device stays below 2^31)
        j = torch.arange(n, dtype=torch.int32, device=device) % _PAT_M
        parts.append((((97 * j) % _PAT_M + base)[:, None] + col[None, :]) % _PAT_M)
    if not parts:
        return torch.zeros((0, h), dtype=dtype, device=device)
    return torch.cat(parts).to(dtype)


def _verify_exchange(be, lane: "_CommLane", h: int, dtype, device, out_splits, in_splits, group, max_pair_rows: int) -> None:
    me = dist.get_rank(group)
    send = _pattern_rows(in_splits, h, me, True, device, dtype)
    want = _pattern_rows(out_splits, h, me, False, device, dtype)
    got, work = _exchange_rows(be, lane, send, None, int(sum(out_splits)), out_splits, in_splits, group, max_pair_rows, _check=False)
    _finish(lane, work)
    bad = (got != want).any(dim=1).sum().to(torch.int64).reshape(1) if want.numel() else torch.zeros(1, dtype=torch.int64, device=device)
    first = int((got != want).any(dim=1).nonzero()[0]) if int(bad) else -1
    total = bad.clone()
    dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
    if int(total):
        raise RuntimeError(f"halo exchange self check failed: rank {me} received {int(bad)} wrong rows of {int(sum(out_splits))} "
                           f"(first at row {first}; {int(total)} wrong rows over all ranks; splits in={list(in_splits)} "
                           f"out={list(out_splits)}, {h} columns, MAX_PAIR_BYTES={MAX_PAIR_BYTES}): the collective backend "
                           "delivers corrupted data at this message size")


def _exchange_rows(be, lane: _CommLane, rows_src: torch.Tensor, gather_index: Optional[torch.Tensor], out_rows: int,
                   out_splits, in_splits, group, max_pair_rows: int = 0, _check: bool = True, check_key=None):
    """Start one all-to-all(v) of feature rows on the comm lane: (optionally) gather the rows to send with the engine's
    row-gather kernel, then the collective.  Returns (recv buffer, handle); ``_finish`` joins.
    ``max_pair_rows``: the plan's global pair maximum (HaloPlan.max_pair_rows) — decides, identically on every rank, into how
    many row chunks the exchange is cut (0: this rank's own splits decide; only valid for a group of one).
    ``check_key``: (HaloPlan.serial, direction) — identical on every rank of the group; None = no start-up check."""
    h = rows_src.shape[1]
    if _check and EXCHANGE_SELFCHECK and check_key is not None and dist.is_initialized():
        key = (tuple(check_key), h, rows_src.dtype, MAX_PAIR_BYTES)
        seen = lane.__dict__.setdefault("_verified", set())
        if key not in seen:                                     # (every part of the key is the same on every rank)
            seen.add(key)
            _verify_exchange(be, lane, h, rows_src.dtype, rows_src.device, out_splits, in_splits, group, max_pair_rows)
    recv = torch.empty((out_rows, h), dtype=rows_src.dtype, device=rows_src.device)
    biggest = max_pair_rows if max_pair_rows > 0 else max(list(out_splits) + list(in_splits) + [0])
    chunks = max(1, -(-(biggest * h * rows_src.element_size()) // MAX_PAIR_BYTES))
    with lane.fork():
        if chunks == 1:
            send = be.gather_rows(rows_src, gather_index) if gather_index is not None else rows_src.contiguous()
            work = dist.all_to_all_single(recv, send, output_split_sizes=out_splits, input_split_sizes=in_splits, group=group,
                                          async_op=True)
            if lane.stream is not None:
                work.wait()                                     # stream-ordered on the LANE only; the host does not block
        else:
            # chunk k carries rows [b[k], b[k+1]) of EVERY pair's segment (same cut on both sides of a pair)
            in_off = [0]
            for n in in_splits:
                in_off.append(in_off[-1] + n)
            out_off = [0]
            for n in out_splits:
                out_off.append(out_off[-1] + n)
            in_b = [_chunk_bounds(n, chunks) for n in in_splits]
            out_b = [_chunk_bounds(n, chunks) for n in out_splits]
            src_all = rows_src if gather_index is not None else rows_src.contiguous()
            send = src_all
            work = None
            for k in range(chunks):
                isz = [b[k + 1] - b[k] for b in in_b]
                osz = [b[k + 1] - b[k] for b in out_b]
                if gather_index is not None:
                    idx = torch.cat([gather_index[in_off[r] + in_b[r][k]:in_off[r] + in_b[r][k + 1]] for r in range(len(isz))])
                    part = be.gather_rows(rows_src, idx)
                else:
                    part = torch.cat([src_all[in_off[r] + in_b[r][k]:in_off[r] + in_b[r][k + 1]] for r in range(len(isz))])
                got = torch.empty((sum(osz), h), dtype=rows_src.dtype, device=rows_src.device)
                wk = dist.all_to_all_single(got, part, output_split_sizes=osz, input_split_sizes=isz, group=group, async_op=True)
                wk.wait()                                       # stream-ordered on a lane; blocking only under gloo (CPU tests)
                o = 0
                for r, n in enumerate(osz):
                    recv[out_off[r] + out_b[r][k]:out_off[r] + out_b[r][k] + n].copy_(got[o:o + n])
                    o += n
                lane.keep(part, got)
    lane.keep(recv, send, rows_src)
    return recv, (work if lane.stream is None else None)


def _finish(lane: _CommLane, work) -> None:
    if work is not None:
        work.wait()
    lane.join()


class _ShardedAggregateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_own, ea_loc, ea_halo, agg: "ShardedAggregate", act: str, add_self: bool):
        plan, be, lane = agg.plan, agg.backend, agg.lane
        x_own = x_own.contiguous()
        recv, work = _exchange_rows(be, lane, x_own, plan.send_rows, plan.n_halo, plan.recv_splits, plan.send_splits, plan.group,
                                    plan.max_pair_rows, check_key=(plan.serial, "fwd"))
        out = be.aggregate_fwd(x_own, ea_loc, agg.st_loc, act, add_self)            # overlaps the exchange
        _finish(lane, work)
        if plan.halo_edge_index.shape[1] > 0:
            out = be.aggregate_fwd(recv, ea_halo, agg.st_halo, act, False, accumulate_into=out)   # summed inside the kernel
        ctx.agg, ctx.act, ctx.add_self = agg, act, add_self
        ctx.save_for_backward(x_own, recv, ea_loc if ea_loc is not None else x_own.new_empty(0),
                              ea_halo if ea_halo is not None else x_own.new_empty(0))
        ctx.has_ea = (ea_loc is not None, ea_halo is not None)
        return out

    @staticmethod
    def backward(ctx, dout):
        agg = ctx.agg
        x_own, recv, ea_loc, ea_halo = ctx.saved_tensors
        ea_loc = ea_loc if ctx.has_ea[0] else None
        ea_halo = ea_halo if ctx.has_ea[1] else None
        need_dea = (ctx.needs_input_grad[1] and ea_loc is not None, ctx.needs_input_grad[2] and ea_halo is not None)
        d_own, dea_loc, dea_halo = agg.backward_raw(dout.contiguous(), x_own, recv, ea_loc, ea_halo, ctx.act, ctx.add_self, need_dea)
        return d_own, dea_loc, dea_halo, None, None, None


class ShardedAggregate:
    """out[i] = sum_{j->i} phi(x[j] + e_ji) (+ x[i]) for the nodes this rank owns, with the remote x[j]
    fetched by one all-to-all per call (reference op: messagepassing.py:57-60,72-74 on a single device).

    The exchange runs on its own HIP stream (``_CommLane``): send rows are staged there with the engine's row gather
    (``phc_gather_rows_i64``), the local edges are aggregated meanwhile, and the halo edges are ADDED onto that result
    inside the aggregation kernel when the rows have landed (``add_self`` bit 2 of ``phc_aggregate_fwd``).  Backward:
    halo edges first, their gradient rows travel back while the local edges' backward (and whatever the caller queues
    through ``overlap``) runs; returned rows are summed into the owners' rows in a fixed order inside the same kernel."""

    def __init__(self, plan: HaloPlan, backend=None):
        self.plan = plan
        self.backend = backend if backend is not None else EngineBackend()
        self.lane = _CommLane(plan.loc_edge_index.device)
        self.st_loc = self.backend.structure(plan.loc_edge_index, plan.n_own, plan.n_own)
        self.st_halo = self.backend.structure(plan.halo_edge_index, max(plan.n_halo, 1), plan.n_own)
        self.seg_back = self.backend.segments(plan.send_rows, plan.n_own) if plan.send_rows.numel() else None

    def split_edge_attr(self, edge_attr: torch.Tensor):
        """(rows of the local edges, rows of the halo edges) of a tensor aligned with the caller's edge list."""
        return edge_attr.index_select(0, self.plan.loc_edge_ids), edge_attr.index_select(0, self.plan.halo_edge_ids)

    def forward_raw(self, x_own, ea_loc, ea_halo, act: str, add_self: bool):
        """(out, recv) without autograd bookkeeping (the fused layer below keeps its own)."""
        plan, be = self.plan, self.backend
        recv, work = _exchange_rows(be, self.lane, x_own, plan.send_rows, plan.n_halo, plan.recv_splits, plan.send_splits, plan.group,
                                    plan.max_pair_rows, check_key=(plan.serial, "fwd"))
        out = be.aggregate_fwd(x_own, ea_loc, self.st_loc, act, add_self)
        _finish(self.lane, work)
        if plan.halo_edge_index.shape[1] > 0:
            out = be.aggregate_fwd(recv, ea_halo, self.st_halo, act, False, accumulate_into=out)
        return out, recv

    def backward_raw(self, dout, x_own, recv, ea_loc, ea_halo, act: str, add_self: bool, need_dea=(False, False), overlap=None, extra=None):
        """(d_own, dea_loc, dea_halo); ``overlap``: a callable queued on the compute stream while the gradient rows of
        the halo sources travel back (the layer's parameter-gradient GEMMs); ``extra``: another consumer's share of d x_own."""
        plan, be = self.plan, self.backend
        h = dout.shape[1]
        if plan.n_halo > 0 and plan.halo_edge_index.shape[1] > 0:          # gradient of the halo rows first: it has to travel
            d_halo, dea_halo = be.aggregate_bwd(dout, recv, ea_halo, self.st_halo, act, False, need_dea[1])
        else:
            d_halo = dout.new_zeros((plan.n_halo, h))
            dea_halo = dout.new_zeros((0, h)) if need_dea[1] else None
        back, work = _exchange_rows(be, self.lane, d_halo, None, int(plan.send_rows.numel()), plan.send_splits, plan.recv_splits,
                                    plan.group, plan.max_pair_rows, check_key=(plan.serial, "bwd"))
        d_own, dea_loc = be.aggregate_bwd(dout, x_own, ea_loc, self.st_loc, act, add_self, need_dea[0], extra=extra) if extra is not None \
            else be.aggregate_bwd(dout, x_own, ea_loc, self.st_loc, act, add_self, need_dea[0])
        extra = overlap() if overlap is not None else None
        _finish(self.lane, work)
        if self.seg_back is not None:
            d_own = be.segment_sum_into(back, self.seg_back, d_own)         # fixed order: deterministic
        return (d_own, dea_loc, dea_halo) if overlap is None else (d_own, dea_loc, dea_halo, extra)

    def __call__(self, x_own: torch.Tensor, ea_loc: Optional[torch.Tensor], ea_halo: Optional[torch.Tensor],
                 msg_encoder: str = "identity", add_self_loops: bool = True) -> torch.Tensor:
        return _ShardedAggregateFn.apply(x_own, ea_loc, ea_halo, self, msg_encoder, add_self_loops)


# ---------------------------------------------------------------------------------------------
# 3. the sharded message-passing LAYER: aggregate -> + x -> linear1 -> norm (global statistics) -> act -> linear2
# ---------------------------------------------------------------------------------------------
class NormLink:
    """Hand-over between a sharded conv and the model-level norm directly behind it (``ShardedGINEStack``), engine backend with
    the folded norm forms only.  Forward: the conv's second GEMM leaves the column statistics of its output here (``moments``:
    the norm skips its pass over y).  Backward: the norm takes its column sums, all-reduces them and PARKS (dout, y, save_stat,
    sums, activation) instead of applying them; the conv's backward runs the element-wise part inside its first dx GEMM
    (``phc_gemm_bwd_apply``).  The conv's output must feed that norm and nothing else."""
    __slots__ = ("fused", "moments", "parked", "dea_sink", "take_res", "res_share")

    def __init__(self):
        self.fused = False
        self.moments = None
        self.parked = None
        self.dea_sink = None
        self.take_res = False        # the norm's residual IS the conv's input (ShardedGINEStack): its gradient share (= the norm's dout)
        self.res_share = None        # is parked here and added inside the conv's aggregation backward instead of by autograd


_EVAL_GRAD_WARNED: set = set()


def _refuse_eval_grad(who: str, *tensors) -> None:
    """The eval-mode forwards of the sharded modules are forward-only (they run under ``no_grad``).  A caller with gradients enabled
    whose inputs require grad gets a DETACHED result: say so, once per module kind.  Not an error — a validation loop
    that forgot ``torch.no_grad()`` feeds tensors derived from the parameters and wants exactly this forward; set
    ``PHC_EVAL_GRAD_STRICT=1`` to make it one (saliency / frozen-norm fine-tuning must use training mode)."""
    if torch.is_grad_enabled() and any(t is not None and torch.is_tensor(t) and t.requires_grad for t in tensors):
        msg = (f"{who}: eval mode is forward-only on the sharded path — an input requires grad and the result is detached; "
               "call it under torch.no_grad(), or in training mode if the gradient is wanted")
        if os.environ.get("PHC_EVAL_GRAD_STRICT", "0") == "1":
            raise NotImplementedError(msg)
        if who not in _EVAL_GRAD_WARNED:
            _EVAL_GRAD_WARNED.add(who)
            warnings.warn(msg, RuntimeWarning, stacklevel=3)


class EdgeGradSink:
    """The gradient of ONE edge-attribute tensor read by all L layers of a ``ShardedGINEStack`` through identity message functions:
        d ea[e] = sum_l dagg_l[dst(e)] = (sum_l dagg_l)[dst(e)]
    (messagepassing.py:132-146: the message is x_j + edge_attr; dagg_l = the gradient of layer l's aggregate, [n_own, h]).  Every
    layer's backward adds its dagg here and skips its own [E, h] gradient; the LAST layer to run gathers the sum once.  Autograd
    would otherwise write L tensors of E x h and add them up pairwise — at config 5 (E = 50 M, h = 128: 25.6 GB each) that was
    48 ms of element-wise adds and 3 x 5 ms of edge-gradient writes in a 255 ms four-layer step."""

    def __init__(self, n_layers: int):
        self.n_layers = int(n_layers)
        self.joined = 0              # layers of this forward that hand their edge gradient to the sink
        self.adds = 0                # ... of which this many have run their backward since the last flush
        self.total = None

    def join(self) -> bool:
        """Called from a layer's forward.  True for the FIRST layer (forward order) that joins: it is the flusher.  Every layer's
        backward needs the gradient of its own output, which comes out of the backward of the layer behind it, so the first
        participant of the forward is the last one to run in ANY backward through the stack — a full one, a second one over a retained
        graph, or one rooted at an intermediate layer's output (then the sum holds the layers in front of that root, which is that
        root's edge gradient).  A count of expected calls would be wrong in the last two cases."""
        self.joined += 1
        return self.joined == 1

    def add(self, dagg: torch.Tensor, flush: bool) -> bool:
        """``flush``: this is the flusher's backward — ``self.total`` is complete; the caller gathers from it and calls ``take``."""
        if self.adds >= self.joined:
            raise RuntimeError("EdgeGradSink: more layer backwards than participating layers since the last flush "
                               "(a backward pass through the stack was abandoned half-way?)")
        if self.total is None:
            self.total = dagg                                   # (a fresh temporary of the calling backward: nobody else writes it)
        else:
            self.total.add_(dagg)
        self.adds += 1
        return flush

    def take(self) -> torch.Tensor:
        """The finished sum; the sink is empty again (the next backward over a retained graph starts from nothing)."""
        t, self.total, self.adds = self.total, None, 0
        return t


class _ShardedGINEFn(torch.autograd.Function):
    """PHMGINEConv.forward (messagepassing.py:132-146) + PHMMLP.forward (layers.py:349-355) on one destination-range
    shard, forward and backward written out so that every exchange has compute queued next to it:

      forward : halo rows travel            || aggregation over the local edges
                moments all-gather (2h+0)      (tiny, between linear1 and the normalisation)
      backward: sums all-reduce (2h)           (tiny, between the two dx GEMMs)
                halo gradient rows travel   || local edges' backward + BOTH parameter-gradient GEMMs (dH1, dH2)

    Parameter gradients come out as this rank's partial sums over its own rows (``FlatGradSync(average=False)`` adds
    them up); batch statistics, running statistics and dx are those of the unsharded layer."""

    @staticmethod
    def forward(ctx, layer: "ShardedPHMGINEConv", link: Optional[NormLink], x_own, ea_loc, ea_halo, A1, W1, b1, A2, W2, b2, *affine):
        agg, be, plan = layer.agg, layer.agg.backend, layer.agg.plan
        mlp = layer.conv.transform
        act_msg, act = layer.msg_encoder, mlp.activation_str
        x_own = x_own.contiguous()
        s, recv = agg.forward_raw(x_own, ea_loc, ea_halo, act_msg, layer.conv.add_self_loops)
        build = getattr(be, "build_operands", None)
        pre = build([(A1, W1), (A2, W2)]) if build is not None else None      # engine: H, H^T (+ k4 forms) of both linears
        ctx.pre = pre
        # the norm's passes inside the GEMMs around it: every rank decides alike (width and row counts are global knowledge)
        fused = bool(mlp.norm_flag and getattr(be, "norm_fusable", None) is not None and be.norm_fusable(pre) and layer.min_count >= 1)
        ctx.fused = fused
        if fused:
            norm = mlp.norm.bn
            z, mom = be.linear_fwd_stats(s, pre[0], b1)                       # statistics: the GEMM's epilogue
            mom_flat = torch.empty(plan.world * mom.numel(), dtype=mom.dtype, device=mom.device)
            dist.all_gather_into_tensor(mom_flat, mom, group=plan.group)      # 2h floats per rank
            save_stat = be.bn_finalize_from_moments(mom_flat.view(plan.world, mom.numel()), layer.counts_all, layer.n_total, norm, z.shape[1])
            if link is not None:                                              # + the statistics of y for the model's norm behind
                y, link.moments = be.linear_fwd_xf_stats(z, save_stat, act, pre[1], b2)
            else:
                y = be.linear_fwd_xf(z, save_stat, act, pre[1], b2)           # act(norm(z)) exists only in the GEMM's LDS
            a = z.new_empty(0)
        else:
            z = be.linear_fwd_pre(s, pre[0], b1) if pre is not None else be.linear_fwd(s, A1, W1, b1)
            if mlp.norm_flag:
                norm = mlp.norm.bn
                mom = be.bn_moments(z)
                mom_flat = torch.empty(plan.world * mom.numel(), dtype=mom.dtype, device=mom.device)
                dist.all_gather_into_tensor(mom_flat, mom, group=plan.group)  # 2h floats per rank
                mom_all = mom_flat.view(plan.world, mom.numel())
                a, save_stat = be.bn_fwd_from_moments(z, mom_all, layer.counts_all, layer.n_total, norm, act)
            else:
                save_stat = z.new_empty(0)
                a = layer.act_fn(z)
            y = be.linear_fwd_pre(a, pre[1], b2) if pre is not None else be.linear_fwd(a, A2, W2, b2)
        ctx.layer = layer
        sink = link.dea_sink if link is not None else None
        # (grad mode is off inside Function.forward: the stack made the sink only where gradients are wanted)
        ctx.dea_sink = sink if (sink is not None and act_msg == "identity"
                                and ((ea_loc is not None and ctx.needs_input_grad[3]) or (ea_halo is not None and ctx.needs_input_grad[4]))) else None
        # (a layer that does not join returns its own edge gradients, if any)
        ctx.sink_flush = ctx.dea_sink.join() if ctx.dea_sink is not None else False
        ctx.link = link if (link is not None and fused) else None
        ctx.res_link = link if (link is not None and link.take_res and ctx.needs_input_grad[2]) else None
        if link is not None:
            link.fused = fused
            if ctx.res_link is None:
                link.take_res = False                           # (nobody will add a parked share: the norm returns it to autograd)
        ctx.has_ea = (ea_loc is not None, ea_halo is not None)
        ctx.has_bias = (b1 is not None, b2 is not None)
        ctx.n_affine = len(affine)
        e0 = x_own.new_empty(0)
        ctx.save_for_backward(x_own, recv, ea_loc if ea_loc is not None else e0, ea_halo if ea_halo is not None else e0,
                              s, z, a, save_stat, A1, W1, A2, W2)
        return y

    @staticmethod
    def backward(ctx, dy):
        layer = ctx.layer
        agg, be, plan = layer.agg, layer.agg.backend, layer.agg.plan
        mlp = layer.conv.transform
        act_msg, act = layer.msg_encoder, mlp.activation_str
        x_own, recv, ea_loc, ea_halo, s, z, a, save_stat, A1, W1, A2, W2 = ctx.saved_tensors
        ea_loc = ea_loc if ctx.has_ea[0] else None
        ea_halo = ea_halo if ctx.has_ea[1] else None
        pre = ctx.pre
        link = ctx.link
        if link is not None and link.parked is not None:
            # the model's norm behind this conv left its backward un-applied: dy (its dx) is evaluated inside the dx GEMM
            dout, y_in, stat_o, acc_o, act_o = link.parked
            link.parked = None
            if dy.data_ptr() != dout.data_ptr():
                raise RuntimeError("NormLink: the conv's output was consumed by something besides its norm")
            dy, da = be.linear_dx_bwd_apply(dout, y_in, stat_o, acc_o, layer.n_total, act_o, pre[1])
        else:
            dy = dy.contiguous()
            da = be.linear_dx_pre(dy, pre[1]) if pre is not None else be.linear_dx(dy, A2, W2)
        if ctx.fused:
            h = z.shape[1]
            acc = be.bn_bwd_sums_acc(da, z, save_stat, act)                   # this rank's sum g | sum g x_hat, integer totals
            dwb = None
            if ctx.n_affine:
                tot = be.acc_totals(acc, h)                                   # (before the all-reduce: the partial of this rank)
                dwb = torch.cat([tot[h:], tot[:h]])                           # dweight | dbias
            dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=plan.group)      # integer words: the same totals on every rank
            dz, ds = be.linear_dx_bwd_apply(da, z, save_stat, acc, layer.n_total, act, pre[0])
        elif mlp.norm_flag:
            dwb = be.bn_bwd_sums(da, z, save_stat, act)                       # this rank's partial dweight | dbias
            dwb_all = dwb.clone()
            dist.all_reduce(dwb_all, op=dist.ReduceOp.SUM, group=plan.group)
            dz = be.bn_bwd_apply(da, z, save_stat, dwb_all, layer.n_total, act)
        else:
            dwb = None
            dz = layer.act_bwd(z, da)
        if not ctx.fused:
            ds = be.linear_dx_pre(dz, pre[0]) if pre is not None else be.linear_dx(dz, A1, W1)
        need_dea = (ctx.needs_input_grad[3] and ea_loc is not None, ctx.needs_input_grad[4] and ea_halo is not None)
        sink = ctx.dea_sink
        want_dea = need_dea
        if sink is not None:
            need_dea = (False, False)                                         # the stack's layers share ONE edge gradient (EdgeGradSink)

        def param_grads():                                                    # queued while the halo gradients travel
            g2 = (be.linear_param_grads_xf(z, save_stat, act, dy, A2, W2, ctx.has_bias[1]) if ctx.fused
                  else be.linear_param_grads(a, dy, A2, W2, ctx.has_bias[1]))
            g1 = be.linear_param_grads(s, dz, A1, W1, ctx.has_bias[0])
            return g1, g2

        share = None
        if ctx.res_link is not None:
            share, ctx.res_link.res_share = ctx.res_link.res_share, None     # the skip's share of d x_own, parked by the norm behind this conv
        d_own, dea_loc, dea_halo, (g1, g2) = agg.backward_raw(ds, x_own, recv, ea_loc, ea_halo, act_msg,
                                                              layer.conv.add_self_loops, need_dea, overlap=param_grads, extra=share)
        if sink is not None and sink.add(ds, ctx.sink_flush):
            # the last layer of the stack to run: d ea[e] = (sum of the layers' aggregate gradients)[dst(e)], one gather per edge set
            total = sink.take()
            if want_dea[0]:
                dea_loc = be.gather_rows(total, plan.loc_edge_index[1])
            if want_dea[1] and plan.halo_edge_index.shape[1] > 0:
                dea_halo = be.gather_rows(total, plan.halo_edge_index[1])
            elif want_dea[1]:
                dea_halo = ds.new_zeros((0, ds.shape[1]))
            del total
        aff = []
        if ctx.n_affine:
            n = ctx.n_affine // 2
            k = z.shape[1] // n
            aff = [dwb[c * k:(c + 1) * k] for c in range(n)] + [dwb[n * k + c * k:n * k + (c + 1) * k] for c in range(n)]
        return (None, None, d_own if ctx.needs_input_grad[2] else None, dea_loc, dea_halo, g1[0], g1[1], g1[2], g2[0], g2[1], g2[2], *aff)


class ShardedPHMGINEConv(torch.nn.Module):
    """One ``PHMGINEConv`` (phc/hypercomplex/undirectional/messagepassing.py:91-161) evaluated on a destination-range
    shard of one large graph (BASELINE config 5).  Wraps an ordinary conv module — same parameters, same state_dict —
    and a ``ShardedAggregate`` built from the graph's ``HaloPlan``; every rank holds a replica of the parameters.

        y_own = layer(x_own, ea_loc, ea_halo)        # rows [plan.lo, plan.hi) of the unsharded layer's output
    """

    def __init__(self, conv, agg: ShardedAggregate, num_nodes: int):
        super().__init__()
        from phc.quaternion.activations import get_module_activation
        self.conv = conv
        self.agg = agg
        plan = agg.plan
        self.msg_encoder = getattr(conv, "msg_encoder_str", "identity")
        self.n_total = int(num_nodes)
        bounds = partition_bounds(self.n_total, plan.world)
        dev = plan.loc_edge_index.device
        self.counts_all = torch.tensor([float(bounds[r + 1] - bounds[r]) for r in range(plan.world)], dtype=torch.float32, device=dev)
        self.min_count = min(bounds[r + 1] - bounds[r] for r in range(plan.world))
        self.act_fn = get_module_activation(conv.transform.activation_str)

    def act_bwd(self, z, da):
        with torch.enable_grad():
            zr = z.detach().requires_grad_()
            out = self.act_fn(zr)
        return torch.autograd.grad(out, zr, da)[0]

    def forward(self, x_own: torch.Tensor, ea_loc: Optional[torch.Tensor], ea_halo: Optional[torch.Tensor],
                link: Optional[NormLink] = None) -> torch.Tensor:
        mlp = self.conv.transform
        if not self.training:
            return self._forward_eval(x_own, ea_loc, ea_halo)
        affine = []
        if mlp.norm_flag:
            nb = mlp.norm.bn
            if not hasattr(nb, "bn") or not nb.track_running_stats:
                raise NotImplementedError("ShardedPHMGINEConv needs the naive-batch-norm flavour with running statistics")
            if nb.affine:
                affine = [m.weight for m in nb.bn] + [m.bias for m in nb.bn]
        l1, l2 = mlp.linear1, mlp.linear2
        return _ShardedGINEFn.apply(self, link, x_own, ea_loc, ea_halo, l1.phm_rule, l1.W, l1.b, l2.phm_rule, l2.W, l2.b, *affine)


    def _forward_eval(self, x_own, ea_loc, ea_halo):
        """Validation between epochs (train_hiv.py:204-230 on a shard): the halo exchange is the only collective — with the running
        statistics every row of linear1 -> norm -> act -> linear2 is independent of the other ranks' rows.  Forward only."""
        mlp, be = self.conv.transform, self.agg.backend
        _refuse_eval_grad("ShardedPHMGINEConv", x_own, ea_loc, ea_halo)
        with torch.no_grad():
            s, _ = self.agg.forward_raw(x_own.contiguous(), ea_loc, ea_halo, self.msg_encoder, self.conv.add_self_loops)
            l1, l2 = mlp.linear1, mlp.linear2
            z = be.linear_fwd(s, l1.phm_rule, l1.W, l1.b)
            a = be.bn_eval(z, mlp.norm.bn, mlp.activation_str) if mlp.norm_flag else self.act_fn(z)
            return be.linear_fwd(a, l2.phm_rule, l2.W, l2.b)


# ---------------------------------------------------------------------------------------------
# 4. the model-level norm of a sharded stack: act(norm(x)) (+ skip) with the batch statistics of ALL rows
# ---------------------------------------------------------------------------------------------
class _ShardedNormActFn(torch.autograd.Function):
    """``act(batch_norm(x)) + residual`` on the rows a rank owns, statistics over the rows of every rank (models.py:208-215
    on a destination-range shard): moments -> all-gather -> normalise; backward: sums -> all-reduce -> apply.  The affine
    gradients come out as this rank's partial sums, like every other parameter gradient of the sharded path."""

    @staticmethod
    def forward(ctx, x, residual, mod: "ShardedPHMNorm", link: Optional[NormLink], *affine):
        be, plan, norm = mod.backend, mod.plan, mod.norm.bn
        x = x.contiguous()
        linked = link is not None and link.fused and link.moments is not None
        mom = link.moments if linked else be.bn_moments(x)        # linked: the producing GEMM's epilogue already has them
        if linked:
            link.moments = None
        mom_flat = torch.empty(plan.world * mom.numel(), dtype=mom.dtype, device=mom.device)
        dist.all_gather_into_tensor(mom_flat, mom, group=plan.group)
        out, save_stat = be.bn_fwd_from_moments(x, mom_flat.view(plan.world, mom.numel()), mod.counts_all, mod.n_total, norm,
                                                mod.activation, residual)
        ctx.mod = mod
        ctx.link = link if linked else None
        ctx.res_link = link if (link is not None and link.take_res and residual is not None) else None
        ctx.has_res = residual is not None
        ctx.n_affine = len(affine)
        ctx.save_for_backward(x, save_stat)
        return out

    @staticmethod
    def backward(ctx, dout):
        mod = ctx.mod
        be, plan = mod.backend, mod.plan
        x, save_stat = ctx.saved_tensors
        dout = dout.contiguous()
        if ctx.link is not None:
            # column sums as integer totals, all-reduced exactly; the element-wise part runs inside the conv's dx GEMM (NormLink)
            h = x.shape[1]
            acc = be.bn_bwd_sums_acc(dout, x, save_stat, mod.activation)
            dwb = None
            if ctx.n_affine:
                tot = be.acc_totals(acc, h)
                dwb = torch.cat([tot[h:], tot[:h]])                           # dweight | dbias, this rank's partial
            dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=plan.group)
            ctx.link.parked = (dout, x, save_stat, acc, mod.activation)
            dx = dout                                                         # placeholder: the conv's backward takes the parked operands
        else:
            dwb = be.bn_bwd_sums(dout, x, save_stat, mod.activation)
            dwb_all = dwb.clone()
            dist.all_reduce(dwb_all, op=dist.ReduceOp.SUM, group=plan.group)
            dx = be.bn_bwd_apply(dout, x, save_stat, dwb_all, mod.n_total, mod.activation)
        aff = []
        if ctx.n_affine:
            n = ctx.n_affine // 2
            h = x.shape[1]
            k = h // n
            aff = [dwb[c * k:(c + 1) * k] for c in range(n)] + [dwb[h + c * k:h + (c + 1) * k] for c in range(n)]
        dres = dout if ctx.has_res else None
        if ctx.res_link is not None and ctx.needs_input_grad[1]:
            ctx.res_link.res_share = dout                                     # added inside the conv's aggregation backward (NormLink)
            dres = None
        return (dx, dres, None, None, *aff)


class ShardedPHMNorm(torch.nn.Module):
    """A ``PHMNorm`` ("naive-batch-norm") evaluated on a shard with global batch statistics, activation and skip add fused
    (what PHMSkipConnectAdd.compute_hidden_layer_embedding does after every conv, models.py:208-215)."""

    def __init__(self, norm, plan: HaloPlan, num_nodes: int, activation: str = "relu", backend=None):
        super().__init__()
        self.norm = norm
        self.plan = plan
        self.backend = backend if backend is not None else EngineBackend()
        self.activation = activation
        self.n_total = int(num_nodes)
        bounds = partition_bounds(self.n_total, plan.world)
        dev = plan.loc_edge_index.device
        self.counts_all = torch.tensor([float(bounds[r + 1] - bounds[r]) for r in range(plan.world)], dtype=torch.float32, device=dev)

    def forward(self, x_own: torch.Tensor, residual: Optional[torch.Tensor] = None, link: Optional[NormLink] = None) -> torch.Tensor:
        nb = self.norm.bn
        if not hasattr(nb, "bn") or not nb.track_running_stats:
            raise NotImplementedError("ShardedPHMNorm: naive-batch-norm with running statistics only")
        if not self.training:                                   # eval: running statistics, no exchange (forward only)
            _refuse_eval_grad("ShardedPHMNorm", x_own, residual)
            return self.backend.bn_eval(x_own.contiguous(), nb, self.activation, residual)
        affine = ([m.weight for m in nb.bn] + [m.bias for m in nb.bn]) if nb.affine else []
        return _ShardedNormActFn.apply(x_own, residual, self, link, *affine)


EDGE_GRAD_SINK = __import__("os").environ.get("PHC_EDGE_GRAD_SINK", "1") != "0"
RESIDUAL_SHARE = __import__("os").environ.get("PHC_STACK_RESIDUAL_SHARE", "1") != "0"


class ShardedGINEStack(torch.nn.Module):
    """L message-passing layers of PHMSkipConnectAdd on one sharded graph: x <- act(norm(conv(x, edges))) + skip per layer
    (models.py:200-217 with sc_type "last"), every conv a ShardedPHMGINEConv and every norm a ShardedPHMNorm over ONE halo plan
    (built once per graph, shared by all layers and steps).  ``convs`` / ``norms``: the ordinary modules (replicated)."""

    def __init__(self, convs, norms, plan: HaloPlan, num_nodes: int, activation: str = "relu", backend=None):
        super().__init__()
        agg = ShardedAggregate(plan, backend=backend)
        self.agg = agg
        self.layers = torch.nn.ModuleList([ShardedPHMGINEConv(c, agg, num_nodes) for c in convs])
        self.norms = torch.nn.ModuleList([ShardedPHMNorm(nm, plan, num_nodes, activation, backend=agg.backend) for nm in norms])

    def forward(self, x_own, ea_loc, ea_halo):
        h = x_own
        # one edge-attribute tensor feeds every layer: its gradient is gathered once from the sum of the layers' aggregate gradients
        sink = EdgeGradSink(len(self.layers)) if (EDGE_GRAD_SINK and torch.is_grad_enabled() and len(self.layers) > 1) else None
        for layer, norm in zip(self.layers, self.norms):
            link = NormLink()                   # (does nothing unless the engine's folded norm forms serve the layer)
            link.dea_sink = sink
            link.take_res = RESIDUAL_SHARE and torch.is_grad_enabled()     # the norm's residual below is this conv's input
            h = norm(layer(h, ea_loc, ea_halo, link=link), residual=h, link=link)
        return h
