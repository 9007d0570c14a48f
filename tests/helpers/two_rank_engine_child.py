"""Child process of tests/test_gpu_two_rank_engine.py: ONE rank of a real multi-rank run of the sharded message-passing
stack (phc_gnn_amd.dist.ShardedGINEStack) with the HIP engine as compute backend.

The GPU box has one GPU and RCCL refuses two ranks on one device, so both ranks open ``cuda:0`` and the collectives of
phc_gnn_amd/dist.py travel through a gloo group, staged through host memory by the test-only shim below (the product's
collectives are RCCL on device memory; what this run covers is everything else of the product path: the HIP kernels on a
real 2-rank ``HaloPlan`` with unequal row counts, ``recv_splits != send_splits``, batch-norm moments merged across ranks,
partial parameter gradients summed by ``FlatGradSync(average=False)``).

    python two_rank_engine_child.py RANK WORLD PORT OUTDIR
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import phc_gnn_amd  # noqa: E402,F401
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as tdist  # noqa: E402


class _Done:
    def wait(self):
        return True


class HostStagedCollectives:
    """``torch.distributed`` as phc_gnn_amd.dist sees it, with CUDA tensors staged through the host around gloo."""

    def __init__(self, real):
        self._real = real

    def __getattr__(self, name):
        return getattr(self._real, name)

    def all_to_all_single(self, output, input, output_split_sizes=None, input_split_sizes=None, group=None, async_op=False):
        o = torch.empty(output.shape, dtype=output.dtype)
        self._real.all_to_all_single(o, input.contiguous().cpu(), output_split_sizes, input_split_sizes, group=group)
        output.copy_(o)
        return _Done() if async_op else None

    def all_gather_into_tensor(self, output, input, group=None, async_op=False):
        o = torch.empty(output.shape, dtype=output.dtype)
        self._real.all_gather_into_tensor(o, input.contiguous().cpu(), group=group)
        output.copy_(o)
        return _Done() if async_op else None

    def all_reduce(self, tensor, op=tdist.ReduceOp.SUM, group=None, async_op=False):
        t = tensor.cpu()
        self._real.all_reduce(t, op=op, group=group)
        tensor.copy_(t)
        return _Done() if async_op else None


# ---- the problem (shared with the parent test) ----------------------------------------------------------------------------
N_NODES, N_EDGES, PHM_DIM, HIDDEN, LAYERS = 20_001, 110_000, 4, 128, 2       # odd row count: the two shards differ in size


def problem():
    from phc_gnn_amd import synth
    ei = torch.from_numpy(synth.random_edges(N_NODES, N_EDGES, seed=41))
    # a skewed graph on top: every 7th edge points INTO the lower half, so the ranks need different numbers of halo rows
    ei[1, ::7] = ei[1, ::7] % (N_NODES // 2)
    g = torch.Generator().manual_seed(42)
    x = torch.randn(N_NODES, HIDDEN, generator=g)
    ea = torch.randn(N_EDGES, HIDDEN, generator=g)
    dy = torch.randn(N_NODES, HIDDEN, generator=g)
    return ei, x, ea, dy


def make_modules(activation: str):
    """The replicated parameters: LAYERS x (PHMGINEConv, PHMNorm), deterministic."""
    from phc.hypercomplex.norm import PHMNorm
    from phc.hypercomplex.undirectional.messagepassing import PHMGINEConv
    torch.manual_seed(7)
    np.random.seed(7)
    g = torch.Generator().manual_seed(43)
    convs, norms = [], []
    for _ in range(LAYERS):
        conv = PHMGINEConv(HIDDEN, HIDDEN, phm_dim=PHM_DIM, phm_rule=None, norm="naive-batch-norm", activation=activation,
                           msg_encoder="identity")
        nm = PHMNorm(num_features=HIDDEN, phm_dim=PHM_DIM, type="naive-batch-norm")
        with torch.no_grad():
            for mod in (conv, nm):
                for name, p in mod.named_parameters():
                    if name.endswith("phm_rule"):
                        p.add_(0.05 * torch.randn(p.shape, generator=g))
                    elif p.dim() == 1:
                        p.copy_(0.5 + torch.rand(p.shape, generator=g) if name.endswith("weight") else 0.1 * torch.randn(p.shape, generator=g))
        convs.append(conv.train())
        norms.append(nm.train())
    return convs, norms


def main():
    rank, world, port, outdir, activation = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import phc_gnn_amd.dist as D
        from phc_gnn_amd import lib
        D.dist = HostStagedCollectives(tdist)                      # test-only: see the module docstring
        dev = torch.device("cuda:0")
        lib.load()
        ei, x, ea, dy = problem()
        convs, norms = make_modules(activation)
        for m in convs + norms:
            m.to(dev)
        plan = D.build_halo_plan(ei.to(dev), N_NODES)
        stack = D.ShardedGINEStack(convs, norms, plan, N_NODES, activation=activation).train()
        assert isinstance(stack.agg.backend, D.EngineBackend)
        x_own = x[plan.lo:plan.hi].to(dev).requires_grad_()
        ea_loc, ea_halo = stack.agg.split_edge_attr(ea.to(dev))
        ea_loc, ea_halo = ea_loc.requires_grad_(), ea_halo.requires_grad_()
        y = stack(x_own, ea_loc, ea_halo)
        y.backward(dy[plan.lo:plan.hi].to(dev))
        params = [p for m in convs + norms for p in m.parameters()]
        local = [p.grad.detach().cpu().clone() for p in params]
        D.FlatGradSync(params, average=False)()
        torch.cuda.synchronize()
        lib.check_indices(dev)
        torch.save(dict(y=y.detach().cpu(), gx=x_own.grad.cpu(), gl=ea_loc.grad.cpu(), gh=ea_halo.grad.cpu(),
                        loc_ids=plan.loc_edge_ids.cpu(), halo_ids=plan.halo_edge_ids.cpu(), lo=plan.lo, hi=plan.hi,
                        n_halo=plan.n_halo, recv_splits=plan.recv_splits, send_splits=plan.send_splits, local=local,
                        grads=[p.grad.detach().cpu().clone() for p in params],
                        buffers=[b.detach().cpu().clone() for m in convs + norms for b in m.buffers()]),
                   os.path.join(outdir, f"rank{rank}.pt"))
        print(f"TWO_RANK_CHILD_OK rank={rank} own={plan.n_own} halo={plan.n_halo} recv={plan.recv_splits} send={plan.send_splits}")
    finally:
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
