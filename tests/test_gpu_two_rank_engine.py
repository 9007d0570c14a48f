"""A REAL two-rank run of the sharded message-passing stack with the HIP engine as compute backend (one
large graph): two processes, both on the box's single GPU, collectives through gloo staged via host memory (test-only shim
in tests/helpers/two_rank_engine_child.py; RCCL refuses two ranks on one device).  Odd node count and a skewed edge list,
so that the shards' row counts differ and ``recv_splits != send_splits``.

    two ranks (engine)  ==  one process (engine, the ordinary modules on the whole graph)  ==  the CPU oracle

on output rows, dx, d edge_attr, every parameter gradient (convs and model-level norms; partial sums per rank summed by
FlatGradSync(average=False)) and the running statistics.  Reference ops being sharded:
phc/hypercomplex/undirectional/messagepassing.py:132-146, phc/hypercomplex/norm.py:30-35, models.py:208-215."""
import os
import socket
import subprocess

# Two PROCESSES share one GPU here (a test-only stand-in for two ranks): the launches whose workgroups wait for each other inside the
# launch (the one-launch MLP / optimizer kernels of round 6) assume their grid is co-resident, which two processes interleaving on one
# device cannot promise — each could hold part of the CUs and wait for the rest (the poll is bounded: the run would end with
# PHC_ERR_RENDEZVOUS, not hang).  A rank that owns its GPU, as bench.py --gpus N launches them, is not in that position.
SHARED_GPU_ENV = {"PHC_MLP_BWD_FUSED": "0", "PHC_MLP_FWD_FUSED": "0", "PHC_CLIP_ADAM_FUSED": "0"}
import sys

import pytest
import torch

from tests._util import assert_close
from tests.helpers import two_rank_engine_child as C

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_two_ranks(tmp, activation):
    port = str(_free_port())
    child = os.path.join(HERE, "helpers", "two_rank_engine_child.py")
    procs = [subprocess.Popen([sys.executable, child, str(r), "2", port, str(tmp), activation], env=dict(__import__("os").environ, **SHARED_GPU_ENV), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and "TWO_RANK_CHILD_OK" in out, f"rank {r}:\n{out[-4000:]}"
    return [torch.load(os.path.join(tmp, f"rank{r}.pt")) for r in range(2)]


@pytest.mark.parametrize("activation", ["swish", "relu"])
def test_two_rank_engine_stack_equals_single_process_and_oracle(tmp_path, activation):
    from oracle import phc_oracle as O
    shards = _run_two_ranks(tmp_path, activation)
    N, E, n, h = C.N_NODES, C.N_EDGES, C.PHM_DIM, C.HIDDEN
    # the plan really is a two-rank one: unequal shards, asymmetric exchange
    assert shards[0]["hi"] - shards[0]["lo"] != shards[1]["hi"] - shards[1]["lo"]
    assert shards[0]["n_halo"] > 0 and shards[1]["n_halo"] > 0 and shards[0]["n_halo"] != shards[1]["n_halo"]
    for d in shards:
        assert d["recv_splits"] != d["send_splits"]
    assert shards[0]["recv_splits"][1] == shards[1]["send_splits"][0] and shards[1]["recv_splits"][0] == shards[0]["send_splits"][1]

    ei, x, ea, dy = C.problem()
    y2, gx2, ge2 = torch.zeros(N, h), torch.zeros(N, h), torch.zeros(E, h)
    for d in shards:
        y2[d["lo"]:d["hi"]] = d["y"]
        gx2[d["lo"]:d["hi"]] = d["gx"]
        ge2[d["loc_ids"]] = d["gl"]
        ge2[d["halo_ids"]] = d["gh"]

    # ---- one process, engine: the ordinary modules on the whole graph ----
    convs, norms = C.make_modules(activation)
    sd0 = [{k: v.detach().clone() for k, v in m.state_dict().items()} for m in convs + norms]
    for m in convs + norms:
        m.to(DEV)
    xd, ead, eid = x.to(DEV).requires_grad_(), ea.to(DEV).requires_grad_(), ei.to(DEV)
    hcur = xd
    for conv, nm in zip(convs, norms):
        hcur = nm.forward_fused(conv(hcur, eid, ead), activation, residual=hcur)
    hcur.backward(dy.to(DEV))
    torch.cuda.synchronize()
    params1 = [p for m in convs + norms for p in m.parameters()]
    names = [f"{'conv' if i < C.LAYERS else 'norm'}{i % C.LAYERS}.{k}" for i, m in enumerate(convs + norms) for k, _ in m.named_parameters()]
    buffers1 = [b for m in convs + norms for b in m.buffers()]

    # ---- the CPU oracle on the whole graph ----
    p = {}
    for i, sd in enumerate(sd0):
        p.update({f"m{i}." + k: v.clone() for k, v in sd.items()})
    order = [f"m{i}." + k for i, m in enumerate(convs + norms) for k, _ in m.named_parameters()]
    for k in order:
        p[k].requires_grad_()
    xr, er = x.clone().requires_grad_(), ea.clone().requires_grad_()
    href = xr
    for l in range(C.LAYERS):
        yl = O.phm_gine_conv(href, ei, er, p, f"m{l}.transform.", n, activation, True, True, True, "identity", None)
        href = O.activation(activation, O._norm_from_state(yl, p, f"m{C.LAYERS + l}.", n, True)) + href
    gref = torch.autograd.grad(href, [xr, er] + [p[k] for k in order], dy, allow_unused=True)
    # relu: the same oracle in float64 is the yardstick for the parameter gradients (tests/test_gpu_fullwidth.py): W = how far
    # the float32 oracle's own gradients sit from it (gate flips at the ReLU kink move whole tensors by 1e-3 .. 1e-2)
    param_allow = 3e-5
    if activation == "relu":
        p64 = {k: (v.detach().double() if v.is_floating_point() else v.clone()) for k, v in
               {f"m{i}." + k: v for i, sd in enumerate(sd0) for k, v in sd.items()}.items()}
        for k in order:
            p64[k].requires_grad_()
        h64 = x.double()
        for l in range(C.LAYERS):
            yl = O.phm_gine_conv(h64, ei, ea.double(), p64, f"m{l}.transform.", n, activation, True, True, True, "identity", None)
            h64 = O.activation(activation, O._norm_from_state(yl, p64, f"m{C.LAYERS + l}.", n, True)) + h64
        g64 = torch.autograd.grad(h64, [p64[k] for k in order], dy.double(), allow_unused=True)
        worst32 = max(float((a.double() - b).abs().max() / b.abs().max()) for a, b in zip(gref[2:], g64)
                      if b is not None and float(b.abs().max()) > 1e-7)
        param_allow = max(1e-5, 3.0 * worst32)
        print(f"[two ranks, relu] float32 oracle vs float64 oracle, worst parameter gradient: {worst32:.2e} -> allowance {param_allow:.2e}")

    # swish is the tight comparison (every tensor within 3e-5).  relu: a pre-activation within an ulp of the kink falls on
    # either side depending on the summation order, and one flipped gate changes that node's gradient row by O(1) of its size
    # (and, through the aggregation's backward, the rows of its neighbours): row-wise tensors are compared row by row — all but
    # a handful of rows within 3e-5 — and the parameter gradients (sums over all rows) within 3 x the float32 oracle's own
    # distance from the float64 oracle
    tight = activation == "swish"
    tol_y = 2e-5

    def rows_close(got, ref, what):
        if tight:
            return assert_close(got, ref, 3e-5, what)
        scale = float(ref.abs().max())
        bad = ((got - ref).abs().amax(dim=1) > 3e-5 * scale)
        # one flipped gate in the top layer reaches (1 + E / N) ** LAYERS rows of dx through the aggregations' backward (42 here);
        # with 10 M pre-activations per step and statistics that differ in the last bit between a GEMM epilogue's one pass and
        # the oracle's two, a handful of flips is the expectation: room for four
        per_flip = (1.0 + C.N_EDGES / C.N_NODES) ** C.LAYERS
        allowed = max(8, got.shape[0] // 500, int(4 * per_flip))
        assert int(bad.sum()) <= allowed, f"{what}: {int(bad.sum())} of {got.shape[0]} rows differ (allowed {allowed})"
    assert_close(y2, href.detach(), tol_y, "2 ranks vs oracle: y")
    assert_close(y2, hcur.detach().cpu(), tol_y, "2 ranks vs 1 process: y")
    rows_close(gx2, gref[0], "2 ranks vs oracle: dx")
    rows_close(gx2, xd.grad.cpu(), "2 ranks vs 1 process: dx")
    rows_close(ge2, gref[1], "2 ranks vs oracle: d edge_attr")
    rows_close(ge2, ead.grad.cpu(), "2 ranks vs 1 process: d edge_attr")
    checked = 0
    for i, (k, g_or) in enumerate(zip(names, gref[2:])):
        g2 = shards[0]["grads"][i]
        assert torch.equal(g2, shards[1]["grads"][i]), k               # every rank holds the same summed gradient
        if g_or is None or k.endswith("linear1.b") or k.endswith("linear2.b"):
            # a bias in front of a batch-norm: exactly zero in exact arithmetic, round-off noise in fp32
            continue
        assert_close(g2, g_or, param_allow, f"2 ranks vs oracle: grad {k}")
        assert_close(g2, params1[i].grad.cpu(), param_allow, f"2 ranks vs 1 process: grad {k}")
        # each rank really contributed a PARTIAL sum (and neither part is the whole)
        part = shards[0]["local"][i] + shards[1]["local"][i]
        assert_close(part, g2, 1e-6, f"partial sums {k}")
        assert float((shards[0]["local"][i] - g2).abs().max()) > 0, k
        checked += 1
    assert checked >= 20, checked
    for i, (b2a, b2b, b1) in enumerate(zip(shards[0]["buffers"], shards[1]["buffers"], buffers1)):
        assert torch.equal(b2a, b2b), i                                 # global statistics: identical on both ranks
        if b2a.is_floating_point():
            assert_close(b2a, b1.cpu(), tol_y, f"running statistic {i} vs 1 process")
        else:
            assert int(b2a) == 1
    # the oracle's running statistics (updated in place in p by _norm_from_state / phm_mlp)
    bnames = [f"m{i}." + k for i, m in enumerate(convs + norms) for k, _ in m.named_buffers()]
    for k, b2 in zip(bnames, shards[0]["buffers"]):
        if b2.is_floating_point():
            assert_close(b2, p[k], tol_y, f"running statistic {k} vs oracle")
