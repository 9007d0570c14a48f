"""BASELINE configs[4] at its REAL size on one GPU: 10 M nodes / 50 M directed edges, PHM-4, hidden 128 — one PHMGINEConv
layer forward + backward through the sharded operator (messagepassing.py:132-146 + layers.py:349-355 on a destination-range
shard, phc_gnn_amd/dist.py), with a loop-back halo plan so that half of the source rows travel through the exchange path.

At this size the edge-side tensors hold 3.2e9 elements (offsets beyond 2^31), which no smaller test reaches.  A CPU oracle
run is out of the question (25.6 GB edge tensor), so the checks are size-independent properties evaluated ON DEVICE with
plain torch ops (index_select / index_add_ / matmul), as bench.py's own check does:

  * 4096 sampled destination rows of the aggregate  ==  recomputed from their own in-edges (local + halo) + self loop
  * the first PHM GEMM on those rows                ==  rows @ H(A, W) + b   (float64 on the sample)
  * 4096 sampled source rows of dx                  ==  ds[row] + sum of ds[dst] over their out-edges
  * 4096 sampled rows of d edge_attr (both halves)  ==  ds[dst_e]            (identity message: bit-exact)
  * sampled rows of the layer output, and every parameter gradient, against plain torch autograd over the whole
    [10 M, 128] activation (torch's GEMMs / reductions in a different summation order: 1e-4 / 5e-4)
"""
import os
import socket

import pytest
import torch

from tests._util import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def rccl_single_rank():
    import torch.distributed as dist
    if dist.is_initialized():
        yield
        return
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    yield
    dist.destroy_process_group()


def _loopback_plan(ei, N, frac=0.5):
    """world_size 1, yet sources >= frac * N are treated as remote rows (see tests/test_gpu_model.py)."""
    from phc_gnn_amd.dist import HaloPlan
    cut = int(N * frac)
    src, dst = ei[0], ei[1]
    remote = src >= cut
    loc_pos, halo_pos = (~remote).nonzero().flatten(), remote.nonzero().flatten()
    halo_ids = torch.unique(src[halo_pos])
    halo_src = torch.searchsorted(halo_ids, src[halo_pos])
    k = int(halo_ids.numel())
    return HaloPlan(rank=0, world=1, lo=0, hi=N, n_own=N, n_halo=k, halo_ids=halo_ids, recv_splits=[k],
                    send_rows=halo_ids.clone(), send_splits=[k], loc_edge_index=torch.stack([src[loc_pos], dst[loc_pos]], 0).contiguous(),
                    loc_edge_ids=loc_pos, halo_edge_index=torch.stack([halo_src, dst[halo_pos]], 0).contiguous(),
                    halo_edge_ids=halo_pos, group=None)


def test_config5_full_size_layer_on_one_gpu(rccl_single_rank):
    from phc.hypercomplex.undirectional.messagepassing import PHMGINEConv
    from phc_gnn_amd import lib
    from phc_gnn_amd.dist import ShardedAggregate
    free, _total = torch.cuda.mem_get_info()
    if free < 150 * (1 << 30):
        pytest.skip(f"needs ~150 GB of free HBM, {free >> 30} GB available")
    N, E, n, h, S = 10_000_000, 50_000_000, 4, 128, 4096
    gen = torch.Generator(device=DEV).manual_seed(0)
    ei = torch.randint(0, N, (2, E), generator=gen, device=DEV, dtype=torch.int64)
    plan = _loopback_plan(ei, N)
    del ei
    e_loc, e_halo = plan.loc_edge_index.shape[1], plan.halo_edge_index.shape[1]
    assert e_loc + e_halo == E and min(e_loc, e_halo) * h > (1 << 31)          # both edge tensors in the > 2^31-element regime
    agg = ShardedAggregate(plan)
    torch.manual_seed(5)
    conv = PHMGINEConv(h, h, phm_dim=n, phm_rule=None, norm="naive-batch-norm", activation="swish", msg_encoder="identity")
    with torch.no_grad():
        for name, p in conv.named_parameters():
            if name.endswith("phm_rule"):
                p.add_(0.05 * torch.randn(p.shape))
            elif p.dim() == 1:
                p.copy_(0.5 + torch.rand(p.shape) if name.endswith("weight") else 0.1 * torch.randn(p.shape))
    conv.to(DEV).train()
    mlp = conv.transform
    x = torch.randn(N, h, generator=gen, device=DEV).requires_grad_()
    ea_loc = torch.randn(e_loc, h, generator=gen, device=DEV).requires_grad_()
    ea_halo = torch.randn(e_halo, h, generator=gen, device=DEV).requires_grad_()
    dy = torch.randn(N, h, generator=gen, device=DEV)

    s = agg(x, ea_loc, ea_halo, "identity", True)
    s.retain_grad()
    y = mlp(s)
    y.backward(dy)
    torch.cuda.synchronize()
    lib.check_indices(torch.device(DEV))
    ds = s.grad
    assert bool(torch.isfinite(y).all()) and bool(torch.isfinite(x.grad).all())

    with torch.no_grad():
        xd, sd = x.detach(), s.detach()
        halo_global = plan.halo_ids                                            # halo-local row -> global node id
        parts = ((plan.loc_edge_index, ea_loc.detach(), None), (plan.halo_edge_index, ea_halo.detach(), halo_global))

        # ---- (1) sampled destination rows of the aggregate ----
        gs = torch.Generator(device=DEV).manual_seed(7)
        dst_sample = torch.randperm(N, generator=gs, device=DEV)[:S].sort().values
        pos = torch.full((N,), -1, dtype=torch.int64, device=DEV)
        pos[dst_sample] = torch.arange(S, device=DEV)
        ref_rows = xd.index_select(0, dst_sample).double()
        for ei_part, ea_part, remap in parts:
            tgt = pos[ei_part[1]]
            sel = (tgt >= 0).nonzero().flatten()
            src_rows = ei_part[0][sel] if remap is None else remap[ei_part[0][sel]]
            ref_rows.index_add_(0, tgt[sel], (xd.index_select(0, src_rows) + ea_part.index_select(0, sel)).double())
        got_rows = sd.index_select(0, dst_sample)
        assert_close(got_rows, ref_rows.float(), 1e-5, "aggregate rows at 10 M / 50 M")

        # ---- (2) the first PHM GEMM on those rows ----
        l1 = mlp.linear1
        H1 = torch.einsum("iac,ikp->akcp", l1.phm_rule.double(), l1.W.double()).reshape(h, h)
        z_ref = got_rows.double() @ H1 + l1.b.double()
        z_got = l1(got_rows.contiguous())
        assert_close(z_got, z_ref.float(), 1e-5, "first PHM GEMM rows")

        # ---- (3) sampled source rows of dx ----
        src_sample = torch.randperm(N, generator=gs, device=DEV)[:S].sort().values
        pos.fill_(-1)
        pos[src_sample] = torch.arange(S, device=DEV)
        dx_ref = ds.index_select(0, src_sample).double()                     # the self loop
        for ei_part, _, remap in parts:
            gsrc = ei_part[0] if remap is None else remap[ei_part[0]]
            tgt = pos[gsrc]
            sel = (tgt >= 0).nonzero().flatten()
            dx_ref.index_add_(0, tgt[sel], ds.index_select(0, ei_part[1][sel]).double())
        assert_close(x.grad.index_select(0, src_sample), dx_ref.float(), 1e-5, "dx rows at 10 M / 50 M")

        # ---- (4) d edge_attr: identity message => a pure copy of ds[dst_e] (rows near the END of the tensors included) ----
        for (ei_part, _, _), g_ea, what in ((parts[0], ea_loc.grad, "local"), (parts[1], ea_halo.grad, "halo")):
            ne = ei_part.shape[1]
            e_sample = torch.cat([torch.randint(0, ne, (S - 64,), generator=gs, device=DEV), torch.arange(ne - 64, ne, device=DEV)])
            assert torch.equal(g_ea.index_select(0, e_sample), ds.index_select(0, ei_part[1][e_sample])), what

    # ---- (5) the MLP (linear1 -> batch-norm over 10 M rows -> swish -> linear2) and its parameter gradients vs torch autograd ----
    params = dict(mlp.named_parameters())
    leaves = {k: v.detach().clone().requires_grad_() for k, v in params.items()}

    def H_of(prefix):
        return torch.einsum("iac,ikp->akcp", leaves[prefix + ".phm_rule"], leaves[prefix + ".W"]).reshape(h, h)
    s_leaf = s.detach().clone().requires_grad_()
    z = s_leaf @ H_of("linear1") + leaves["linear1.b"]
    var, mean = torch.var_mean(z, dim=0, unbiased=False)
    w = torch.cat([leaves[f"norm.bn.bn.{c}.weight"] for c in range(n)])
    b = torch.cat([leaves[f"norm.bn.bn.{c}.bias"] for c in range(n)])
    zn = (z - mean) * torch.rsqrt(var + 1e-5) * w + b
    y_ref = (zn * torch.sigmoid(zn)) @ H_of("linear2") + leaves["linear2.b"]
    rows = torch.randperm(N, generator=torch.Generator(device=DEV).manual_seed(9), device=DEV)[:S]
    assert_close(y.detach().index_select(0, rows), y_ref.detach().index_select(0, rows), 1e-4, "layer output rows")
    y_ref.backward(dy)
    assert_close(ds.index_select(0, rows), s_leaf.grad.index_select(0, rows), 1e-4, "ds rows")
    for k, p in params.items():
        if k == "linear1.b":                                                   # in front of the norm: zero up to round-off
            continue
        assert_close(p.grad, leaves[k].grad, 5e-4, f"grad {k} over 10 M rows")
