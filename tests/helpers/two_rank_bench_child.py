"""Child process of tests/test_gpu_two_rank_bench.py: ONE rank of ``bench.py --gpus 2`` (the data-parallel training step the
driver's scaling run launches under torch.distributed.run), on a box with a single GPU.

RCCL refuses two ranks on one device, so both ranks open ``cuda:0`` (LOCAL_RANK=0) and every collective travels through a
gloo group, CUDA tensors staged through host memory by the test-only wrappers below.  What this run covers is everything
else of the product's multi-rank path: bench.py's control flow (rank 0's profiling pass while the others wait, the state
broadcast, the barrier-bracketed timing windows with their all-reduced stop flag), the two-graph step with the eager
gradient exchange in between (``phc_pack_grads`` -> all-reduce -> gradients adopted as views of the optimizer's bucket) and
the closing check that the replicas hold bit-identical parameters.

    python two_rank_bench_child.py RANK WORLD PORT [bench arguments...]
"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
rank, world, port = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)

import phc_gnn_amd  # noqa: E402,F401
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

_init, _all_reduce, _broadcast = dist.init_process_group, dist.all_reduce, dist.broadcast
_a2a, _agt = dist.all_to_all_single, dist.all_gather_into_tensor


class _Done:
    def wait(self):
        return True


def all_to_all_single(output, input, output_split_sizes=None, input_split_sizes=None, group=None, async_op=False):
    if not output.is_cuda:
        return _a2a(output, input, output_split_sizes, input_split_sizes, group=group, async_op=async_op)
    o = torch.empty(output.shape, dtype=output.dtype)
    _a2a(o, input.contiguous().cpu(), output_split_sizes, input_split_sizes, group=group)
    output.copy_(o)
    return _Done() if async_op else None


def all_gather_into_tensor(output, input, group=None, async_op=False):
    if not output.is_cuda:
        return _agt(output, input, group=group, async_op=async_op)
    o = torch.empty(output.shape, dtype=output.dtype)
    _agt(o, input.contiguous().cpu(), group=group)
    output.copy_(o)
    return _Done() if async_op else None



def init_process_group(backend=None, *a, device_id=None, **kw):          # "nccl" on cuda:<local rank>  ->  gloo
    return _init("gloo", *a, **kw)


def all_reduce(tensor, op=dist.ReduceOp.SUM, group=None, async_op=False):
    if not tensor.is_cuda:
        return _all_reduce(tensor, op=op, group=group, async_op=async_op)
    t = tensor.cpu()
    _all_reduce(t, op=op, group=group)
    tensor.copy_(t)
    return _Done() if async_op else None


def broadcast(tensor, src=0, group=None, async_op=False):
    if not tensor.is_cuda:
        return _broadcast(tensor, src=src, group=group, async_op=async_op)
    t = tensor.cpu()
    _broadcast(t, src=src, group=group)
    tensor.copy_(t)


dist.init_process_group, dist.all_reduce, dist.broadcast = init_process_group, all_reduce, broadcast
dist.all_to_all_single, dist.all_gather_into_tensor = all_to_all_single, all_gather_into_tensor
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[4:]
runpy.run_path(sys.argv[0], run_name="__main__")
print("TWO_RANK_BENCH_CHILD_OK", flush=True)
