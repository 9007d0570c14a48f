"""Localise a mismatch of the sharded aggregate at 10 M / 50 M with a loop-back halo plan (tests/test_gpu_config5_fullsize.py):
checks the exchange, the local-edge aggregate and the halo accumulate one by one against torch ops on device."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import phc_gnn_amd  # noqa
import torch
import torch.distributed as dist

from tests.test_gpu_config5_fullsize import _loopback_plan

DEV = "cuda:0"
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
from phc_gnn_amd.dist import ShardedAggregate, _exchange_rows, _finish

N, E, h, S = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, int(sys.argv[2]) if len(sys.argv) > 2 else 50_000_000, 128, 4096
gen = torch.Generator(device=DEV).manual_seed(0)
ei = torch.randint(0, N, (2, E), generator=gen, device=DEV, dtype=torch.int64)
plan = _loopback_plan(ei, N)
del ei
agg = ShardedAggregate(plan)
be = agg.backend
e_loc, e_halo = plan.loc_edge_index.shape[1], plan.halo_edge_index.shape[1]
print("n_halo", plan.n_halo, "e_loc", e_loc, "e_halo", e_halo)
x = torch.randn(N, h, generator=gen, device=DEV)
ea_loc = torch.randn(e_loc, h, generator=gen, device=DEV)
ea_halo = torch.randn(e_halo, h, generator=gen, device=DEV)


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


# (a) the exchange
recv, wk = _exchange_rows(be, agg.lane, x, plan.send_rows, plan.n_halo, plan.recv_splits, plan.send_splits, plan.group, plan.max_pair_rows)
_finish(agg.lane, wk)
torch.cuda.synchronize()
idx = torch.randint(0, plan.n_halo, (S,), device=DEV)
print("exchange rows equal:", bool(torch.equal(recv[idx], x[plan.halo_ids[idx]])), " last rows:", bool(torch.equal(recv[-64:], x[plan.halo_ids[-64:]])))
g = be.gather_rows(x, plan.send_rows)
print("gather_rows equal:", bool(torch.equal(g[idx], x[plan.halo_ids[idx]])), bool(torch.equal(g[-64:], x[plan.halo_ids[-64:]])))

gs = torch.Generator(device=DEV).manual_seed(7)
sample = torch.randperm(N, generator=gs, device=DEV)[:S].sort().values
pos = torch.full((N,), -1, dtype=torch.int64, device=DEV)
pos[sample] = torch.arange(S, device=DEV)


def ref_part(ei_part, ea_part, src_table, remap):
    out = torch.zeros(S, h, dtype=torch.float64, device=DEV)
    tgt = pos[ei_part[1]]
    sel = (tgt >= 0).nonzero().flatten()
    src_rows = ei_part[0][sel] if remap is None else remap[ei_part[0][sel]]
    out.index_add_(0, tgt[sel], (src_table.index_select(0, src_rows) + ea_part.index_select(0, sel)).double())
    return out


# (b) local edges only, with self loop
out_loc = be.aggregate_fwd(x, ea_loc, agg.st_loc, "identity", True)
torch.cuda.synchronize()
ref_loc = ref_part(plan.loc_edge_index, ea_loc, x, None) + x[sample].double()
print("local aggregate err:", rel(out_loc[sample], ref_loc))
# structure sanity
rowptr, col, perm = agg.st_loc.csr
print("st_loc rowptr[-1] == e_loc:", int(rowptr[-1]) == e_loc, " perm range:", int(perm.min()), int(perm.max()), " col range", int(col.min()), int(col.max()))
d = int(sample[17])
b0, b1 = int(rowptr[d]), int(rowptr[d + 1])
pe = perm[b0:b1].long()
print("row", d, "deg", b1 - b0, "dst of its perm entries all == d:", bool((plan.loc_edge_index[1][pe] == d).all()), "col == src:", bool((plan.loc_edge_index[0][pe] == col[b0:b1].long()).all()))
# (c) halo edges alone (no accumulate), then accumulate
out_h = be.aggregate_fwd(recv, ea_halo, agg.st_halo, "identity", False)
torch.cuda.synchronize()
ref_h = ref_part(plan.halo_edge_index, ea_halo, x, plan.halo_ids)
print("halo aggregate (no accumulate) err:", rel(out_h[sample], ref_h))
rowptr, col, perm = agg.st_halo.csr
print("st_halo rowptr[-1] == e_halo:", int(rowptr[-1]) == e_halo, " perm range:", int(perm.min()), int(perm.max()), " col range", int(col.min()), int(col.max()), "n_halo", plan.n_halo)
b0, b1 = int(rowptr[d]), int(rowptr[d + 1])
pe = perm[b0:b1].long()
print("row", d, "deg", b1 - b0, "dst ok:", bool((plan.halo_edge_index[1][pe] == d).all()), "col ok:", bool((plan.halo_edge_index[0][pe] == col[b0:b1].long()).all()))
out_acc = be.aggregate_fwd(recv, ea_halo, agg.st_halo, "identity", False, accumulate_into=out_loc.clone())
torch.cuda.synchronize()
print("accumulate err:", rel(out_acc[sample], ref_loc + ref_h))
dist.destroy_process_group()
