"""Differentiable operators of the PHC message-passing path, each a thin
``torch.autograd.Function`` over the C ABI of libphc_hip.so.

These are the engine-side counterparts of the reference's functional layer:
``matvec_product_new`` (phc/hypercomplex/layers.py:198-219), PyG
``propagate(aggr='add')`` as used by PHMConv / PHMGINEConv
(phc/hypercomplex/undirectional/messagepassing.py:55-74,132-146) and
``global_add_pool`` (phc/hypercomplex/pooling.py:14).
All inputs must be fp32 tensors on a ROCm device; there is no CPU path.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import lib as L
from .graph import EdgeStructure, SegmentStructure, edge_structure, segment_structure


def _act_id(name: str) -> int:
    try:
        return L.ACT_IDS[name.lower()]
    except KeyError:
        raise L.PhcError(f"unknown message activation '{name}' (known: {sorted(L.ACT_IDS)})") from None


# ---------------------------------------------------------------------------
# PHM linear
# ---------------------------------------------------------------------------
class _PHMLinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, A, W, b, residual):
        lib = L.load()
        dev = L.require_device(x, A, W, b, residual)
        n, kin, kout = W.shape
        if tuple(A.shape) != (n, n, n):
            raise L.PhcError(f"phm_rule must be [{n},{n},{n}], got {tuple(A.shape)}")
        if x.shape[-1] != n * kin:
            raise L.PhcError(f"x has {x.shape[-1]} features, PHM weight expects {n * kin}")
        x2 = L.f32c(x.reshape(-1, n * kin))
        A_, W_, b_ = L.f32c(A), L.f32c(W), L.f32c(b)
        res2 = L.f32c(residual.reshape(-1, n * kout)) if residual is not None else None
        N = x2.shape[0]
        y = torch.empty((N, n * kout), dtype=torch.float32, device=dev)
        L.check(lib.phc_phm_linear_fwd(L.ptr(x2), L.ptr(A_), L.ptr(W_), L.ptr(b_), L.ptr(res2), N, n, kin, kout,
                                       L.ptr(y), L.stream_ptr(dev)), "phc_phm_linear_fwd")
        ctx.save_for_backward(x2, A_, W_)
        ctx.has_bias = b is not None
        ctx.has_res = residual is not None
        ctx.x_shape = x.shape
        ctx.res_shape = residual.shape if residual is not None else None
        return y.reshape(*x.shape[:-1], n * kout)

    @staticmethod
    def backward(ctx, dy):
        lib = L.load()
        x2, A, W = ctx.saved_tensors
        dev = x2.device
        n, kin, kout = W.shape
        N = x2.shape[0]
        dy2 = L.f32c(dy.reshape(-1, n * kout))
        need_x, need_A, need_W, need_b, need_res = ctx.needs_input_grad
        need_b = need_b and ctx.has_bias
        dx = torch.empty_like(x2) if need_x else None
        need_param = need_A or need_W or need_b
        dW = torch.empty_like(W) if need_param else None
        dA = torch.empty_like(A) if need_A else None
        db = torch.empty(n * kout, dtype=torch.float32, device=dev) if need_b else None
        ws_bytes = lib.phc_phm_linear_bwd_ws_bytes(N, n, kin, kout) if need_param else 0
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev) if need_param else None
        L.check(lib.phc_phm_linear_bwd(L.ptr(x2), L.ptr(A), L.ptr(W), L.ptr(dy2), N, n, kin, kout, L.ptr(dx), L.ptr(dW),
                                       L.ptr(dA), L.ptr(db), L.ptr(ws), ws_bytes, L.stream_ptr(dev)),
                "phc_phm_linear_bwd")
        return (dx.reshape(ctx.x_shape) if need_x else None, dA, dW if need_W else None, db,
                dy.reshape(ctx.res_shape) if (need_res and ctx.has_res) else None)


def phm_linear(x: torch.Tensor, phm_rule: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor] = None,
               residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = x · (Σ_i phm_rule[i] ⊗ W[i]) + bias (+ residual)."""
    return _PHMLinearFn.apply(x, phm_rule, W, bias, residual)


def phm_build_H(phm_rule: torch.Tensor, W: torch.Tensor) -> torch.Tensor:
    """Materialise H = Σ_i A_i ⊗ W_i  [n*kin, n*kout] (inspection / tests; not differentiable)."""
    lib = L.load()
    dev = L.require_device(phm_rule, W)
    n, kin, kout = W.shape
    H = torch.empty((n * kin, n * kout), dtype=torch.float32, device=dev)
    L.check(lib.phc_phm_build_H(L.ptr(L.f32c(phm_rule.detach())), L.ptr(L.f32c(W.detach())), n, kin, kout, L.ptr(H),
                                L.stream_ptr(dev)), "phc_phm_build_H")
    return H


# ---------------------------------------------------------------------------
# neighbour aggregation
# ---------------------------------------------------------------------------
class _AggregateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ea, st: EdgeStructure, act: int, add_self: bool):
        lib = L.load()
        dev = L.require_device(x, ea)
        x_ = L.f32c(x)
        ea_ = L.f32c(ea)
        h = x_.shape[1]
        if x_.shape[0] < st.num_src:
            raise L.PhcError(f"x has {x_.shape[0]} rows, edge structure expects {st.num_src} source rows")
        if ea_ is not None and tuple(ea_.shape) != (st.num_edges, h):
            raise L.PhcError(f"edge_attr must be [{st.num_edges}, {h}], got {tuple(ea_.shape)}")
        rowptr, col, perm = st.csr
        out = torch.empty((st.num_dst, h), dtype=torch.float32, device=dev)
        L.check(lib.phc_aggregate_fwd(L.ptr(x_), L.ptr(ea_), L.ptr(rowptr), L.ptr(col), L.ptr(perm), st.num_dst, h, act,
                                      1 if add_self else 0, L.ptr(out), L.stream_ptr(dev)), "phc_aggregate_fwd")
        ctx.st, ctx.act, ctx.add_self, ctx.has_ea = st, act, add_self, ea is not None
        if act != 0:
            ctx.save_for_backward(x_, ea_) if ea is not None else ctx.save_for_backward(x_)
        ctx.x_rows = x_.shape[0]
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = L.load()
        st: EdgeStructure = ctx.st
        dout_ = L.f32c(dout)
        dev = dout_.device
        h = dout_.shape[1]
        x_ = ea_ = None
        if ctx.act != 0:
            saved = ctx.saved_tensors
            x_ = saved[0]
            ea_ = saved[1] if ctx.has_ea else None
        need_x, need_ea = ctx.needs_input_grad[0], ctx.needs_input_grad[1] and ctx.has_ea
        rowptr_t, col_t, perm_t = st.csc
        dx = torch.empty((ctx.x_rows, h), dtype=torch.float32, device=dev)
        if ctx.x_rows > st.num_src:
            dx[st.num_src:].zero_()
        dea = torch.empty((st.num_edges, h), dtype=torch.float32, device=dev) if need_ea else None
        L.check(lib.phc_aggregate_bwd(L.ptr(dout_), L.ptr(x_), L.ptr(ea_), L.ptr(rowptr_t), L.ptr(col_t), L.ptr(perm_t),
                                      st.num_src, st.num_dst, h, ctx.act, 1 if ctx.add_self else 0, L.ptr(dx), L.ptr(dea),
                                      L.stream_ptr(dev)), "phc_aggregate_bwd")
        return (dx if need_x else None), dea, None, None, None


def aggregate_sum(x: torch.Tensor, edge_index, edge_attr: Optional[torch.Tensor], msg_encoder: str = "identity",
                  add_self_loops: bool = False, num_dst: Optional[int] = None) -> torch.Tensor:
    """out[i] = Σ_{e: edge_index[1,e]=i} φ(x[edge_index[0,e]] + edge_attr[e])  (+ x[i]).

    ``edge_index`` is the reference's int64 [2, E] tensor (row 0 = source,
    row 1 = target) or an ``EdgeStructure`` built from one."""
    st = edge_index if isinstance(edge_index, EdgeStructure) else edge_structure(edge_index, x.shape[0], num_dst)
    return _AggregateFn.apply(x, edge_attr, st, _act_id(msg_encoder), bool(add_self_loops))


# ---------------------------------------------------------------------------
# read-out: per-graph sum
# ---------------------------------------------------------------------------
class _SegmentSumFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, st: SegmentStructure):
        lib = L.load()
        dev = L.require_device(x)
        x_ = L.f32c(x)
        h = x_.shape[1]
        rowptr, col, _ = st.csr
        out = torch.empty((st.num_segments, h), dtype=torch.float32, device=dev)
        L.check(lib.phc_aggregate_fwd(L.ptr(x_), None, L.ptr(rowptr), L.ptr(col), None, st.num_segments, h, 0, 0,
                                      L.ptr(out), L.stream_ptr(dev)), "phc_aggregate_fwd(segment sum)")
        ctx.st = st
        ctx.n_rows = x_.shape[0]
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = L.load()
        st: SegmentStructure = ctx.st
        dout_ = L.f32c(dout)
        dev = dout_.device
        h = dout_.shape[1]
        dx = torch.empty((ctx.n_rows, h), dtype=torch.float32, device=dev)
        L.check(lib.phc_gather_rows_i64(L.ptr(dout_), L.ptr(st.batch), ctx.n_rows, h, L.ptr(dx), L.stream_ptr(dev)),
                "phc_gather_rows_i64")
        return dx, None


def global_add_pool(x: torch.Tensor, batch, num_graphs: Optional[int] = None) -> torch.Tensor:
    """out[g] = Σ_{i: batch[i]=g} x[i]  (torch_geometric.nn.global_add_pool semantics)."""
    st = batch if isinstance(batch, SegmentStructure) else segment_structure(batch, num_graphs)
    return _SegmentSumFn.apply(x, st)
