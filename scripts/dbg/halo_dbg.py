import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import phc_gnn_amd, torch, torch.distributed as dist
from phc_gnn_amd import synth, ops
from phc_gnn_amd.dist import ShardedAggregate, _exchange_rows, _finish
from tests.test_gpu_model import _loopback_plan
from oracle import phc_oracle as O
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
dev = torch.device("cuda:0")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
N, E, h = 20000, 110000, 128
ei = torch.from_numpy(synth.random_edges(N, E, seed=11))
g = torch.Generator().manual_seed(12)
x = torch.randn(N, h, generator=g); ea = torch.randn(E, h, generator=g)
plan = _loopback_plan(ei.to(dev), N)
agg = ShardedAggregate(plan)
xd = x.to(dev)
send = ops.gather_rows(xd, plan.send_rows)
print("gather ok", torch.equal(send.cpu(), x[plan.send_rows.cpu()]))
recv, wk = _exchange_rows(agg.backend, agg.lane, xd, plan.send_rows, plan.n_halo, plan.recv_splits, plan.send_splits, plan.group)
_finish(agg.lane, wk)
torch.cuda.synchronize()
print("exchange ok", torch.equal(recv.cpu(), x[plan.send_rows.cpu()]), float(recv.abs().sum()))
el, eh = agg.split_edge_attr(ea.to(dev))
for act in ("identity", "relu"):
    out_loc = agg.backend.aggregate_fwd(xd, el, agg.st_loc, act, True)
    ref_loc = O.propagate_add(x, plan.loc_edge_index.cpu(), el.cpu(), act) + x
    print(act, "local", float((out_loc.cpu() - ref_loc).abs().max()))
    halo_only = agg.backend.aggregate_fwd(recv, eh, agg.st_halo, act, False)
    hei = plan.halo_edge_index.cpu()
    msg = O.activation(act, x[plan.send_rows.cpu()].index_select(0, hei[0]) + eh.cpu())
    ref_h = O.scatter_sum_rows(msg, hei[1], N)
    print(act, "halo only", float((halo_only.cpu() - ref_h).abs().max()))
    acc = agg.backend.aggregate_fwd(recv, eh, agg.st_halo, act, False, accumulate_into=out_loc.clone())
    print(act, "accumulated", float((acc.cpu() - (ref_loc + ref_h)).abs().max()))
    full = O.propagate_add(x, ei, ea, act) + x
    print(act, "vs full", float((acc.cpu() - full).abs().max()))
dist.destroy_process_group()
