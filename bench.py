#!/usr/bin/env python3
"""Headline benchmark: graphs/sec of the PHC-GNN PHM-4 training step on
ogbg-molhiv-shaped batches (BASELINE.json metric, configs[1]: hidden 128,
4 layers, batch 512 per GPU), on N MI355X of one node.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one pre-collated, HBM-resident
synthetic batch, reproducing the reference loop body
(benchmarks/train_hiv.py:175-201): zero_grad, forward, NaN-masked BCE, backward,
(N>1: one flat-bucket RCCL all-reduce of the gradients), clip_grad_norm_(2.0),
Adam step.  Weak scaling: every rank processes its own 512-graph batches.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  "roofline":     the dominant engine kernel of the step, algorithmic work per
                  launch / its measured launch duration (HIP events recorded by
                  libphc_hip's launch profiler on the launching stream),
  "cpu_baseline": the CPU oracle (oracle/phc_oracle.py, a port of the reference's
                  op sequence) timed on this host for a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)



def _packet_capture_choice(argv, env) -> str:
    """`--packet-capture 0|1` has to act before the HIP runtime initialises, i.e. before the imports below.  With 1 the
    single-process hipGraph bench uses hipGraphLaunch's packet-capture fast path; GraphedTrainStep then alternates two static
    input sets, so the next batch's copy never targets buffers of the replay in flight.  Default 0: measured
    A/B on one MI355X the fast path is worth 0 +- 10 us per step now that the step is bound by its dependent chain on the GPU."""
    choice = env.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
    for i, a in enumerate(argv):
        if a == "--packet-capture" and i + 1 < len(argv):
            choice = argv[i + 1]
        elif a.startswith("--packet-capture="):
            choice = a.split("=", 1)[1]
    if int(env.get("WORLD_SIZE", "1")) > 1:
        choice = "0"                     # a collective queued behind a replay is not protected by double buffering
    return "1" if choice == "1" else "0"


os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = _packet_capture_choice(sys.argv[1:], os.environ)

import phc_gnn_amd  # noqa: E402,F401  (sets the HIP runtime flags it needs before HIP initialises)
import torch  # noqa: E402

PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s HBM3E
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 matrix peak
MFMA_KERNELS = ("phm_gemm_fwd", "phm_gemm_dx", "phm_dH_partial", "table_grad", "phm_gemm_ag_fwd")
# launch-profiler kernel name -> substrings of the rocprofv3 kernel names of that FAMILY in profiles/<PMC_FILE>
# (a family = every instantiation the launch profiler books under one name; traffic = launch-weighted mean over them)
PMC_NAMES = {"phm_gemm_fwd": ("gemm_hs16_kernel<128, 2, 4, 1, 4, 1>", "gemm_hs16_kernel<128, 2, 4, 1, 4, 3>",
                              "gemm_hs16_kernel<256, 2, 8, 1, 4, 1>", "gemm_hs16_kernel<256, 2, 8, 1, 4, 3>",
                              "gemm_hs16_kernel<256, 2, 8, 1, 4, 0>", "gemm_hs16_kernel<128, 2, 4, 1, 4, 0>", "mlp_fwd_fused_kernel<2, 0>"),
             # (round 6, hidden 128: both dx GEMMs of a layer and both norm backwards are ONE launch, mlp_bwd_fused_kernel; its flops
             # are booked as the two GEMMs')
             "phm_gemm_dx": ("gemm_hs16_kernel<128, 2, 4, 1, 4, 8>", "gemm_hs16_kernel<128, 2, 4, 1, 4, 12>",
                             "gemm_hs16_kernel<256, 2, 8, 1, 4, 8>", "gemm_hs16_kernel<256, 2, 8, 1, 4, 12>", "mlp_bwd_fused_kernel"),
             # the layer's aggregation inside its first GEMM (molecule batches, round 4): GEMM flops only are booked
             # (round 6, hidden 128: aggregation — <2, 1> sum, <2, 2> softmax — + both GEMMs + the inner norm in ONE launch)
             "phm_gemm_ag_fwd": ("gemm_hs_ag_kernel", "gemm_hs_agw_kernel", "mlp_fwd_fused_kernel<2, 1>", "mlp_fwd_fused_kernel<2, 2>"),
             "phm_dH_partial": ("tail_partial_kernel",), "aggregate_enc_fwd": ("aggregate_encp_direct_fwd_kernel", "aggregate_norm_fwd_kernel"),
             "aggregate_enc_dx": ("aggregate_dx_rows_kernel",), "aggregate_fwd": ("aggregate_fwd_kernel",), "aggregate_bwd": ("aggregate_bwd_kernel",),
             "bn_apply": ("bn_apply_acc_kernel",), "bn_bwd_apply": ("bn_bwd_apply_acc_kernel",), "bn_bwd_partial": ("bn_bwd_partial_kernel",),
             "bn_stats_partial": ("bn_stats_partial_kernel",), "table_grad": ("table_grad_mfma_kernel<1",)}


PMC_FILE = "r06_pmc_traffic.json"
PMC_MFMA_FILE = "r06_pmc_mfma.json"
# below this footprint a row walker's rows are served by the L2s / the 256 MiB Infinity Cache, not by HBM
# (MI355X_MICROARCH.md: 38 MB table of random rows 8.6 TB/s, 151 MB 7.4-7.9 TB/s, HBM sweep 6.0-6.1 TB/s)
INFINITY_CACHE_BYTES = 256 * 1024 * 1024
PEAK_INFINITY_CACHE_GBS = 8600.0


def pmc_traffic(kernel: str, workload: str = "molhiv_b512"):
    """HBM bytes per launch of the kernel family `kernel` in `workload` from the committed rocprofv3 PMC passes
    (FETCH_SIZE x2 + WRITE_SIZE, MI355X_MICROARCH.md §HBM; scripts/rocprof_pmc.sh <round> <workload>), or None.  The file
    is keyed by workload: a workload without its own passes reports None rather than another workload's bytes."""
    path = os.path.join(ROOT, "profiles", PMC_FILE)
    pats = PMC_NAMES.get(kernel)
    if pats is None or not os.path.exists(path):
        return None
    try:
        table = json.load(open(path)).get(workload)
    except Exception:
        return None
    if not table:
        return None
    tot = n = 0.0
    for name, rec in table.items():
        if any(p in name for p in pats):
            tot += rec.get("hbm_bytes_per_launch", 0.0) * rec.get("launches", 1)
            n += rec.get("launches", 1)
    return round(tot / n) if n else None


def pmc_mfma(kernel: str, workload: str = "molhiv_b512"):
    """MFMA-pipe utilisation of the kernel family `kernel` in `workload` from the committed rocprofv3 counter passes
    (scripts/rocprof_pmc_mfma.sh: SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES), launch-weighted over the family's instantiations:
    the share of the cycles a CU has wavefronts of the launch during which a SIMD's matrix pipe is busy), or None."""
    path = os.path.join(ROOT, "profiles", PMC_MFMA_FILE)
    pats = PMC_NAMES.get(kernel)
    if pats is None or not os.path.exists(path):
        return None
    try:
        table = json.load(open(path)).get(workload)
    except Exception:
        return None
    if not table:
        return None
    tot = n = 0.0
    for name, rec in table.items():
        if any(p in name for p in pats) and rec.get("mfma_busy") is not None:
            tot += rec["mfma_busy"] * rec.get("launches", 1)
            n += rec.get("launches", 1)
    return round(tot / n, 4) if n else None


def _flush_c_stdio():
    """RCCL prints its version banner through C stdio, which is only flushed at exit — after Python's own prints.  Flush it
    now so that the JSON line printed next is the LAST line of stdout."""
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="molhiv_b512")
    ap.add_argument("--num-batches", type=int, default=8, help="distinct resident batches cycled through")
    ap.add_argument("--fwd-bwd-only", action="store_true", help="leave grad clipping + Adam out of the step")
    ap.add_argument("--mode", choices=["graph", "eager", "refloop"], default="graph",
                    help="graph: whole step captured once into a hipGraph over bucket-padded batches, replayed per "
                         "batch (phc_gnn_amd/graphed.py); eager: harness.train_step, one launch per kernel from Python; "
                         "refloop: the loop body of the reference's train() (benchmarks/train_hiv.py:175-202) as it stands "
                         "— boolean-mask BCE, phm_weight_regularization, loss.backward(), clip_grad_norm_, "
                         "torch.optim.Adam.step(), loss.item() every step — on the mirror model, nothing of harness.py "
                         "(what a user who only swaps the package gets; molhiv workloads)")
    ap.add_argument("--optimizer", choices=["engine", "torch"], default="engine",
                    help="engine: phc_gnn_amd.optim.ClipAdam (clip + Adam in two launches); torch: clip_grad_norm_ + "
                         "torch.optim.Adam(fused, capturable)")
    ap.add_argument("--cpu-steps", type=int, default=20, help="timed CPU-oracle steps (0 = skip the CPU baseline)")
    ap.add_argument("--cpu-threads", type=int, default=6, help="reference setting: torch.set_num_threads(6)")
    ap.add_argument("--cpu-all-cores-max", type=int, default=32,
                    help="run the all-cores CPU baseline only on hosts with at most this many logical CPUs")
    ap.add_argument("--cpu-many-threads", type=int, default=32,
                    help="hosts with more logical CPUs than --cpu-all-cores-max: thread count of the secondary CPU baseline")
    ap.add_argument("--profile-steps", type=int, default=10)
    ap.add_argument("--packet-capture", choices=["0", "1"], default="0",
                    help="hipGraphLaunch's packet-capture fast path (read before HIP initialises, see _packet_capture_choice)")
    ap.add_argument("--weight-decay", type=float, default=0.001,
                    help="explicit weight regulariser of the reference loop body (train_hiv.py:181-196, --weightdecay default "
                         "0.001, p = 2): one more launch per step; 0 leaves it out")
    ap.add_argument("--layers", type=int, default=1,
                    help="synth_* workloads: 1 = one PHMGINEConv layer (SURVEY 8d); L > 1 = L stacked layers with the model-level "
                         "norm + relu + skip between them (ShardedGINEStack)")
    ap.add_argument("--min-time", type=float, default=6.0,
                    help="keep timing --steps-step windows until this many seconds (and >= 5 windows) are covered; the "
                         "median window is reported (0: a single window)")
    return ap.parse_args()


def make_grad_sync(model, world, optimizer=None):
    """One flat-bucket RCCL all-reduce (mean) of the ~0.3 MB gradient per step (phc_gnn_amd/dist.py)."""
    from phc_gnn_amd.dist import FlatGradSync
    return FlatGradSync(list(model.parameters()), optimizer=optimizer)      # engine optimizer: packed in one launch, no unpack


def cpu_baseline(args, workload, model, gb, threads=None, steps=None, warmup=3, budget_s=None):
    """Time the CPU oracle on the same batch shape: same step contents.  ``budget_s``: stop timing once this many seconds are
    spent (at least one timed step is always taken; the sample says how many were)."""
    from oracle import phc_oracle as O
    from phc_gnn_amd import harness
    w = harness.WORKLOADS[workload]
    threads = threads or args.cpu_threads
    steps = steps or args.cpu_steps
    torch.set_num_threads(threads)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    names = [k for k, _ in model.named_parameters()]
    for k in names:
        sd[k].requires_grad_()
    params = [sd[k] for k in names]
    opt = torch.optim.Adam(params, lr=1e-3)
    x, ei, ea, batch = (torch.from_numpy(a) for a in (gb.x, gb.edge_index, gb.edge_attr, gb.batch))
    y = torch.from_numpy(gb.y)
    kw = dict(phm_dim=w["phm_dim"], num_layers=w["layers"], num_atom_feats=x.shape[1], num_bond_feats=ea.shape[1],
              num_dn_hidden=len(w["d_units"]), mlp=w.get("mlp", True), sc_type=w["sc_type"], training=True,
              num_graphs=gb.num_graphs, msg_aggr=w.get("msg_aggr", "sum"))

    def step():
        opt.zero_grad(set_to_none=True)
        logits = O.skip_connect_add_forward(sd, x, ei, ea, batch, **kw)
        loss = torch.nn.functional.l1_loss(logits, y) if w.get("regression") else O.bce_loss_masked(logits, y)
        if args.weight_decay > 0:                              # same step contents as the GPU side (train_hiv.py:181-196)
            loss = loss + 1e-3 * args.weight_decay * O.phm_weight_regularization([sd[k] for k in names if k.endswith(".W")], p=2)
        loss.backward()
        if not args.fwd_bwd_only:
            torch.nn.utils.clip_grad_norm_(params, max_norm=2.0)
            opt.step()

    for _ in range(warmup):
        step()
    t0 = time.perf_counter()
    done = 0
    for _ in range(steps):
        step()
        done += 1
        if budget_s is not None and time.perf_counter() - t0 > budget_s:
            break
    steps = done
    dt = time.perf_counter() - t0
    return {"value": round(gb.num_graphs * steps / dt, 1), "unit": "graphs/s", "cores": threads,
            "kind": "port",
            "sample": f"{steps} steps of one {gb.num_graphs}-graph batch (N={gb.num_nodes}, E={gb.num_edges}), "
                      f"same step contents, {dt:.1f} s, oracle/phc_oracle.py on torch CPU (7-8 % slower than the reference itself at "
                      f"equal threads: BASELINE.md 3a)"}


LARGE_GRAPHS = {
    # BASELINE configs[4]: one graph, node-range sharded over the ranks (strong scaling: the graph is fixed)
    "synth_10m50m": dict(nodes=10_000_000, edges=50_000_000, phm_dim=4, hidden=128),
    "synth_1m5m": dict(nodes=1_000_000, edges=5_000_000, phm_dim=4, hidden=128),        # quick look / CI-sized
}


def bench_large_graph(args, world, rank, dev):
    """BASELINE configs[4] (SURVEY.md §8d "cfg 5"): ONE PHMGINEConv layer (aggregate -> + x -> linear1 -> batch-norm ->
    relu -> linear2, layer-API form: float edge_attr [E, h]) forward + backward on a synthetic graph with iid-uniform
    edges, destination-range sharded over the ranks (phc_gnn_amd/dist.py: halo all-to-all on its own stream, global
    batch statistics, parameter gradients summed with one flat all-reduce).  A step = fwd + bwd (+ that all-reduce)."""
    import torch.distributed as dist
    from phc.hypercomplex.undirectional.messagepassing import PHMGINEConv
    from phc_gnn_amd import lib
    from phc.hypercomplex.norm import PHMNorm
    from phc_gnn_amd.dist import FlatGradSync, ShardedAggregate, ShardedGINEStack, ShardedPHMGINEConv, build_halo_plan
    g = LARGE_GRAPHS[args.workload]
    N, E, n, h = g["nodes"], g["edges"], g["phm_dim"], g["hidden"]
    if not dist.is_initialized():                      # one rank: a group of one (the exchange code path stays the same)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    gen = torch.Generator(device=dev).manual_seed(0)   # same edge list on every rank (same generator, same device type)
    ei = torch.randint(0, N, (2, E), generator=gen, device=dev, dtype=torch.int64)
    t0 = time.perf_counter()
    plan = build_halo_plan(ei, N)
    del ei
    torch.manual_seed(0)
    n_layers = max(1, args.layers)
    convs = [PHMGINEConv(h, h, phm_dim=n, phm_rule=None, norm="naive-batch-norm", activation="relu", msg_encoder="identity").to(dev).train()
             for _ in range(n_layers)]
    conv = convs[0]
    if n_layers == 1:
        agg = ShardedAggregate(plan)
        layer = ShardedPHMGINEConv(conv, agg, N).train()
        trainable = list(conv.parameters())
    else:
        norms = [PHMNorm(num_features=h, phm_dim=n, type="naive-batch-norm").to(dev).train() for _ in range(n_layers)]
        layer = ShardedGINEStack(convs, norms, plan, N, activation="relu").train()
        agg = layer.agg
        trainable = [p for m in convs + norms for p in m.parameters()]
    _ = agg.st_loc.csr, agg.st_loc.csc, agg.st_halo.csr, agg.st_halo.csc       # sorted views: once per graph
    torch.cuda.synchronize()
    t_plan = time.perf_counter() - t0
    gen2 = torch.Generator(device=dev).manual_seed(1 + rank)
    e_loc, e_halo = plan.loc_edge_index.shape[1], plan.halo_edge_index.shape[1]
    x_own = torch.randn(plan.n_own, h, generator=gen2, device=dev).requires_grad_()
    ea_loc = torch.randn(e_loc, h, generator=gen2, device=dev).requires_grad_()
    ea_halo = torch.randn(e_halo, h, generator=gen2, device=dev).requires_grad_() if e_halo else None
    dy = torch.randn(plan.n_own, h, generator=gen2, device=dev)
    sync = FlatGradSync(trainable, average=False) if world > 1 else None

    def step():
        for p in trainable:
            p.grad = None
        x_own.grad = ea_loc.grad = None
        if ea_halo is not None:
            ea_halo.grad = None
        y = layer(x_own, ea_loc, ea_halo)
        y.backward(dy)
        if sync is not None:
            sync()

    # ---- correctness at THIS size (size-independent property: any destination row can be recomputed from its own in-edges):
    # the sharded aggregate (local + halo edges, self loop) and the first PHM GEMM on a sample of 4096 owned rows against
    # plain torch ops (index_select / index_add_ / matmul with H = sum_i A_i (x) W_i), 1e-5 of the largest entry ----
    verified = None
    with torch.no_grad():
        s_full, recv = agg.forward_raw(x_own.detach(), ea_loc.detach(), ea_halo.detach() if ea_halo is not None else None, "identity", True)
        gs = torch.Generator(device=dev).manual_seed(7)
        sample = torch.randperm(plan.n_own, generator=gs, device=dev)[:4096].sort().values
        ref_rows = x_own.detach().index_select(0, sample).clone()
        pos = torch.full((plan.n_own,), -1, dtype=torch.int64, device=dev)
        pos[sample] = torch.arange(sample.numel(), device=dev)
        for ei_part, ea_part, src_rows in ((plan.loc_edge_index, ea_loc, x_own.detach()), (plan.halo_edge_index, ea_halo, recv)):
            if ei_part.shape[1] == 0:
                continue
            tgt = pos[ei_part[1]]
            sel = (tgt >= 0).nonzero().flatten()
            ref_rows.index_add_(0, tgt[sel], src_rows.index_select(0, ei_part[0][sel]) + ea_part.detach().index_select(0, sel))
        got = s_full.index_select(0, sample)
        e_agg = float((got - ref_rows).abs().max() / ref_rows.abs().max())
        l1 = conv.transform.linear1
        Hm = torch.einsum("iac,ikp->akcp", l1.phm_rule, l1.W).reshape(h, h)
        z_ref = got @ Hm + l1.b
        z_got = agg.backend.linear_fwd(got.contiguous(), l1.phm_rule, l1.W, l1.b)
        e_lin = float((z_got - z_ref).abs().max() / z_ref.abs().max())
        verified = {"rows_sampled": int(sample.numel()), "aggregate_max_rel_err": e_agg, "phm_linear_max_rel_err": e_lin}
        if e_agg > 1e-5 or e_lin > 1e-5:
            raise SystemExit(f"synthetic-graph check failed: {verified}")
        del s_full, recv, ref_rows, pos
    prof_steps = max(1, min(args.profile_steps, 3))
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    lib.prof_reset(); lib.prof_enable(True)
    for _ in range(prof_steps):
        step()
    torch.cuda.synchronize()
    lib.prof_enable(False)
    times, work = lib.prof_read(), dict(lib.WORK)
    # exchange alone (no compute queued next to it): what the overlap has to hide
    ex_ms = None
    if plan.n_halo > 0:
        from phc_gnn_amd.dist import _exchange_rows, _finish
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dist.barrier(); torch.cuda.synchronize()
        ev0.record()
        for _ in range(3):
            recv, wk = _exchange_rows(agg.backend, agg.lane, x_own.detach(), plan.send_rows, plan.n_halo, plan.recv_splits,
                                      plan.send_splits, plan.group, plan.max_pair_rows)
            _finish(agg.lane, wk)
        ev1.record(); torch.cuda.synchronize()
        ex_ms = ev0.elapsed_time(ev1) / 3
    # the overlap the layer's forward relies on, measured per rank so that ONE multi-GPU run yields it: the local-edge aggregate
    # alone, and the span of {exchange on the comm lane || local aggregate on the compute stream} as ShardedAggregate issues them
    loc_ms = span_ms = None
    with torch.no_grad():
        xo, el = x_own.detach(), ea_loc.detach()
        eh = ea_halo.detach() if ea_halo is not None else None
        e0, e1, e2, e3 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
        dist.barrier(); torch.cuda.synchronize()
        e0.record()
        for _ in range(3):
            agg.backend.aggregate_fwd(xo, el, agg.st_loc, "identity", True)
        e1.record(); torch.cuda.synchronize()
        loc_ms = e0.elapsed_time(e1) / 3
        if plan.n_halo > 0:
            from phc_gnn_amd.dist import _exchange_rows, _finish
            dist.barrier(); torch.cuda.synchronize()
            e2.record()
            for _ in range(3):
                recv, wk = _exchange_rows(agg.backend, agg.lane, xo, plan.send_rows, plan.n_halo, plan.recv_splits, plan.send_splits,
                                          plan.group, plan.max_pair_rows)
                agg.backend.aggregate_fwd(xo, el, agg.st_loc, "identity", True)
                _finish(agg.lane, wk)
            e3.record(); torch.cuda.synchronize()
            span_ms = e2.elapsed_time(e3) / 3
    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    lib.check_indices(dev)

    def rate(name, key, scale):
        if name not in times or name not in work:
            return None
        return round(work[name][key] / (times[name]["total_ms"] * 1e-3) / scale, 2)
    per_rank = {"own_nodes": plan.n_own, "local_edges": e_loc, "halo_edges": e_halo, "halo_rows": plan.n_halo,
                "aggregate_fwd_GBs": rate("aggregate_fwd", "bytes", 1e9), "aggregate_bwd_GBs": rate("aggregate_bwd", "bytes", 1e9),
                "gemm_fwd_TFs": rate("phm_gemm_fwd", "flops", 1e12), "gemm_dx_TFs": rate("phm_gemm_dx", "flops", 1e12),
                "param_grad_gemm_TFs": rate("phm_dH_partial", "flops", 1e12), "bn_apply_GBs": rate("bn_apply", "bytes", 1e9),
                "halo_all_to_all_ms_alone": None if ex_ms is None else round(ex_ms, 3),
                "local_aggregate_ms_alone": None if loc_ms is None else round(loc_ms, 3),
                "exchange_and_local_aggregate_overlapped_ms": None if span_ms is None else round(span_ms, 3),
                # 1 = the shorter of the two is hidden completely behind the longer one, 0 = they run back to back
                "overlap_efficiency": None if (span_ms is None or ex_ms is None or min(ex_ms, loc_ms) <= 0) else
                round((ex_ms + loc_ms - span_ms) / min(ex_ms, loc_ms), 3),
                "halo_bytes_received": plan.n_halo * h * 4, "plan_and_sort_s": round(t_plan, 2),
                "engine_kernels_ms_per_step": {k: round(v["total_ms"] / prof_steps, 3) for k, v in sorted(times.items())}}
    gathered = [None] * world
    if world > 1:
        dist.all_gather_object(gathered, per_rank)
    else:
        gathered = [per_rank]
    if rank == 0:
        name = max((k for k in times if k in work), key=lambda k: times[k]["total_ms"])
        t_ms, n_launch = times[name]["total_ms"], times[name]["launches"]
        if name in MFMA_KERNELS:
            ach = work[name]["flops"] / (t_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4)}
        else:
            ach = work[name]["bytes"] / (t_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 4)}
        roof.update({"traffic": pmc_traffic(name, args.workload), "kernel": name, "launches": n_launch,
                     "avg_launch_us": round(1e3 * t_ms / max(n_launch, 1), 1),
                     "traffic_source": "profiles/" + PMC_FILE + " (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE per launch of this workload)"})
        agg_b = sum(work[k]["bytes"] for k in ("aggregate_fwd", "aggregate_bwd") if k in work)
        agg_t = sum(times[k]["total_ms"] for k in ("aggregate_fwd", "aggregate_bwd") if k in times)
        result = {"metric": "edges/sec (fwd+bwd) of one PHMGINEConv PHM-4 layer on a synthetic 10M-node/50M-edge graph",
                  "value": round(E * n_layers * args.steps / dt, 1), "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                  "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                  "dtype": "f32", "data": "synthetic",
                  "config": {"workload": f"{args.workload}: {N} nodes, {E} iid-uniform directed edges, PHM-{n}, hidden {h}, "
                                         + ("one PHMGINEConv layer" if n_layers == 1 else
                                            f"{n_layers} stacked PHMGINEConv layers with norm + relu + skip between them")
                                         + f" (float edge_attr [E,{h}], batch-norm, relu), destination-range sharded over {world} "
                                         f"rank(s) (BASELINE configs[4])", "layers": n_layers,
                             "step": "fwd+bwd" + ("+flat param-grad all-reduce" if world > 1 else ""), "parallelism": f"shard{world}"},
                  "roofline": roof,
                  "aggregate_hbm": {"achieved_GBs": round(agg_b / (agg_t * 1e-3) / 1e9, 1) if agg_t else None, "peak": PEAK_HBM_GBS,
                                    "frac": round(agg_b / (agg_t * 1e-3) / 1e9 / PEAK_HBM_GBS, 4) if agg_t else None},
                  "verified_rank0": verified, "per_rank": gathered}
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        _flush_c_stdio()
        print(json.dumps(result), flush=True)          # after RCCL's own shutdown banner: the JSON line is the LAST line of stdout


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    from phc_gnn_amd import harness, lib
    lib.load()
    if args.workload in LARGE_GRAPHS:
        return bench_large_graph(args, world, rank, dev)
    w = harness.WORKLOADS[args.workload]
    model = harness.build_model(args.workload, dev, seed=0)           # same init on every rank
    model.train()
    def make_optimizer():
        if args.optimizer == "engine":
            # train_hiv.py:198-201,285: clip_grad_norm_(2.0) + Adam(lr, weight_decay 0) — the engine's two-launch form
            from phc_gnn_amd.optim import ClipAdam
            return ClipAdam(model.parameters(), lr=1e-3)
        return torch.optim.Adam(model.parameters(), lr=1e-3, fused=True, capturable=True)
    opt = make_optimizer()
    cfg = harness.StepConfig(regression=bool(w.get("regression")), grad_clip=2.0, optimizer=not args.fwd_bwd_only,
                             weight_decay=args.weight_decay, reg_p=2, lr=1e-3)
    host_batches = [harness.make_batch(args.workload, seed=1000 * rank + i) for i in range(args.num_batches)]
    batches = [harness.to_device(gb, dev) for gb in host_batches]
    # PHC_BENCH_FORCE_SYNC=1: exercise the data-parallel step (two graphs + RCCL all-reduce) on a single rank
    force_sync = world == 1 and bool(os.environ.get("PHC_BENCH_FORCE_SYNC"))
    if force_sync:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    graph_step = None
    prof = None
    if rank == 0 and args.profile_steps > 0:
        # ---- roofline of the dominant engine kernel: instrumented EAGER pass over the same steps, taken before
        # the hipGraph is captured (HIP events cannot be recorded inside a replayed graph) ----
        # (rank 0 only, so no collective in here: grad_sync=None; the pass trains a few real steps, so the weights and
        # the optimizer are put back afterwards: every rank enters the timed run from the same state)
        init_state = {k: v.detach().clone() for k, v in model.state_dict().items()}
        for i in range(3):
            harness.train_step(model, batches[i % len(batches)], opt, cfg, None)
        torch.cuda.synchronize()
        lib.prof_reset()
        lib.prof_enable(True)
        for i in range(args.profile_steps):
            harness.train_step(model, batches[i % len(batches)], opt, cfg, None)
        torch.cuda.synchronize()
        lib.prof_enable(False)
        prof = (lib.prof_read(), dict(lib.WORK))
        model.load_state_dict(init_state)
        opt = make_optimizer()                     # fresh Adam moments / step counts, as on the other ranks
    if world > 1:
        import torch.distributed as dist
        for t in list(model.state_dict().values()):
            dist.broadcast(t, src=0)
        dist.barrier()
    # (created behind the profiling pass: it is bound to the optimizer object the timed run uses)
    grad_sync = make_grad_sync(model, world, opt) if (world > 1 or force_sync) else None
    graph_step = None
    n_max = e_max = 0
    if args.mode == "graph":
        from phc_gnn_amd import graphed
        n_max, e_max = graphed.bucket_shape(host_batches)
        padded = [graphed.padded_to_device(graphed.pad_batch(gb, n_max, e_max), dev) for gb in host_batches]
        graph_step = graphed.GraphedTrainStep(model, opt, cfg, padded[0], grad_sync=grad_sync)

    if args.mode == "refloop":
        if w.get("regression") or world > 1:
            raise SystemExit("--mode refloop reproduces benchmarks/train_hiv.py's loop body: BCE workloads, one rank")
        import torch.nn.functional as F
        from phc.hypercomplex.regularization import phm_weight_regularization
        opt = torch.optim.Adam(model.parameters(), weight_decay=0.0)             # train_hiv.py:285
        kwargs = {"weight_decay": args.weight_decay, "lr": 1e-3, "p": 2, "grad_clipping": 0.0 if args.fwd_bwd_only else 2.0}
        ref_state = {"total_loss": 0.0}

    def ref_loop_body(data):
        """benchmarks/train_hiv.py:174-202, statement for statement (``data`` already on the device)."""
        mask = ~torch.isnan(data.y)
        opt.zero_grad()
        logits = model(data)
        y = data.y.to(torch.float)
        loss = F.binary_cross_entropy_with_logits(input=logits[mask], target=y[mask])
        reg_loss = torch.tensor(0.0, device=dev)
        if kwargs["weight_decay"] > 0.0:
            reg_loss += kwargs["lr"] * kwargs["weight_decay"] * phm_weight_regularization(model, device=dev, p=kwargs["p"])
        loss += reg_loss.squeeze()
        loss.backward()
        if kwargs["grad_clipping"] > 0.0:
            torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=kwargs["grad_clipping"], norm_type=2)
        if not args.fwd_bwd_only:
            opt.step()
        ref_state["total_loss"] += loss.item() * data.num_graphs

    def run(n, offset=0, eager=False):
        for i in range(n):
            j = (offset + i) % len(host_batches)
            if args.mode == "refloop":
                ref_loop_body(batches[j])
                continue
            if graph_step is not None and not eager:
                # D2D copy into the static buffers + one graph launch (PHC_GRAPH_PIPELINE=1: the batch's preprocessing graph
                # was queued on a second stream by the previous call; this call launches the step graph and queues the next)
                graph_step.step(padded[j], padded[(j + 1) % len(host_batches)] if graph_step.pipeline else None)
            else:
                harness.train_step(model, batches[j], opt, cfg, grad_sync)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    import contextlib
    # graph mode: the loop runs on the step's replay stream (see GraphedTrainStep.step: never the null stream)
    with (graph_step.on_stream() if graph_step is not None else contextlib.nullcontext()):
        run(args.warmup)
        # median of several timed windows.  One window = EXACTLY --steps steps between barrier + synchronize
        # (max over ranks); windows repeat until >= --min-time seconds and >= 5 windows have been timed (one window when
        # --min-time 0), and the MEDIAN window is what the line reports (min / max / count alongside).
        windows = []
        offset = args.warmup
        while True:
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(args.steps, offset=offset)
            torch.cuda.synchronize()
            barrier()
            dtw = time.perf_counter() - t0
            if world > 1:
                import torch.distributed as dist
                t = torch.tensor([dtw, float(len(windows) + 1 >= 5 and sum(windows) + dtw >= args.min_time)], device=dev,
                                 dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)      # slowest rank's time; stop only when a rank says so for all
                dtw, stop = float(t[0].item()), bool(t[1].item() > 0)
            else:
                stop = len(windows) + 1 >= 5 and sum(windows) + dtw >= args.min_time
            windows.append(dtw)
            offset += args.steps
            if args.min_time <= 0 or stop or len(windows) >= 400:
                break
    windows.sort()
    dt = windows[len(windows) // 2]

    lib.check_indices(dev)                     # sticky device error word: no kernel met an out-of-range index
    replicas_max_diff = None
    if world > 1:
        # the data-parallel contract: identical replicas after any number of steps (same averaged gradient, same optimizer state)
        import torch.distributed as dist
        worst = torch.zeros((), device=dev)
        for prm in model.parameters():
            ref_p = prm.detach().clone()
            dist.broadcast(ref_p, src=0)
            worst = torch.maximum(worst, (prm.detach() - ref_p).abs().max())
        dist.all_reduce(worst, op=dist.ReduceOp.MAX)
        replicas_max_diff = float(worst)
        if replicas_max_diff != 0.0 and rank == 0:
            print(f"[bench] WARNING: data-parallel replicas differ: max |param - rank0's| = {replicas_max_diff:.3e}", file=sys.stderr)
    graphs = world * w["batch_size"] * args.steps
    dataset_name = {"molhiv": "ogbg-molhiv", "molpcba": "ogbg-molpcba", "zinc": "ZINC"}[w["dataset"]]
    config_note = {"molhiv_b32": " (BASELINE configs[0])", "molhiv_b512": " (BASELINE configs[1])",
                   "zinc_b1024": " (BASELINE configs[2])",
                   "molpcba_b1024": " (BASELINE configs[3], per-GPU share)"}.get(
        args.workload, " (the reference's published model shape, benchmarks/README.md:45-91)" if "_pub_" in args.workload
        else " (configs[1] variant)")
    drop_note = "".join(f", {k} {w[k]}" for k in ("dropout_mpnn", "dropout_dn") if any(w.get(k, [])))
    result = {
        "metric": f"graphs/sec (fwd+bwd) PHC-GNN PHM-{w['phm_dim']} on {dataset_name}",
        "value": round(graphs / dt, 1), "unit": "graphs/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
        "timing": {"windows": len(windows), "steps_per_window": args.steps, "reported": "median window",
                   "ms_per_step_min": round(1e3 * windows[0] / args.steps, 4), "ms_per_step_max": round(1e3 * windows[-1] / args.steps, 4)},
        "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {dataset_name}-shaped synthetic molecules, PHM-{w['phm_dim']}, hidden "
                               f"{w['hidden']}, {w['layers']} {'PHMGINEConv' if w.get('mlp', True) else 'PHMConv'} "
                               f"layers, {w.get('msg_aggr', 'sum')} aggregation, soft-attention pooling{drop_note}, batch "
                               f"{w['batch_size']}/GPU{config_note}",
                   "nodes_per_batch": int(sum(b.num_nodes for b in host_batches) / len(host_batches)),
                   "edges_per_batch": int(sum(b.num_edges for b in host_batches) / len(host_batches)),
                   "step": "zero_grad+fwd+" + ("masked_l1" if w.get("regression") else "bce") + ("+weight_regulariser(lr*wd*mean||W||_2)" if args.weight_decay > 0 else "") + "+bwd"
                           + ("" if args.fwd_bwd_only else "+clip+adam")
                           + ("+grad_allreduce" if world > 1 else ""),
                   "launch": ("hipGraph replay of the whole step on batches padded to one bucket shape "
                              f"({n_max} nodes, {e_max} edges); per-step D2D batch copy + CSR/CSC build "
                              + ("replayed one batch ahead on a second stream" if graph_step.pipeline else "inside the step graph"))
                             if graph_step is not None
                             else ("the reference's own loop body (benchmarks/train_hiv.py:174-202 verbatim: boolean-mask BCE, "
                                   "phm_weight_regularization, loss.backward(), clip_grad_norm_, torch.optim.Adam, loss.item() "
                                   "every step) on the mirror model" if args.mode == "refloop"
                                   else "eager (harness.train_step: one launch per kernel from Python)"),
                   "graph_packet_capture": (os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") == "1")
                                           + 0 if graph_step is not None else None,
                   "static_input_sets": len(graph_step.sets) if graph_step is not None else None,
                   "parallelism": f"dp{world}" if world > 1 else "single",
                   "replicas_max_param_diff": replicas_max_diff},
    }

    if rank == 0:
        times, work = prof if prof is not None else ({}, {})
        if times:
            name = max(times, key=lambda k: times[k]["total_ms"])
            rows = int(sum(b.num_nodes for b in host_batches) / len(host_batches))

            # One roofline object per engine kernel FAMILY (every instantiation the launch profiler books under one name):
            # algorithmic work booked by ops.py / launch durations of the instrumented pass.  `rows`, `flops_per_launch` /
            # `bytes_per_launch` and `avg_launch_us` let a reader recompute `frac` from the line alone.  A row walker whose
            # whole footprint fits the 256 MiB Infinity Cache is NOT priced against HBM: the activations the previous launch
            # wrote are served by the L2s / Infinity Cache (`cache_resident`, roof = the guide's random-row figure there).
            def frac_of(kname):
                if kname not in times:
                    return None
                tms, n_l = times[kname]["total_ms"], max(times[kname]["launches"], 1)
                wk_ = work.get(kname, {"flops": 0.0, "bytes": 0.0})
                out = {"kernel": kname, "launches": n_l, "avg_launch_us": round(1e3 * tms / n_l, 3), "rows": rows}
                if kname in MFMA_KERNELS:
                    a_ = wk_["flops"] / (tms * 1e-3) / 1e12
                    out.update({"bound": "mfma", "achieved": round(a_, 3), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                "frac": round(a_ / PEAK_FP32_MFMA_TFLOPS, 4), "flops_per_launch": round(wk_["flops"] / n_l)})
                else:
                    a_ = wk_["bytes"] / (tms * 1e-3) / 1e9
                    per_launch = wk_["bytes"] / n_l
                    resident = per_launch < INFINITY_CACHE_BYTES
                    peak = PEAK_INFINITY_CACHE_GBS if resident else PEAK_HBM_GBS
                    out.update({"bound": "hbm", "achieved": round(a_, 1), "peak": peak, "unit": "GB/s", "frac": round(a_ / peak, 4),
                                "bytes_per_launch": round(per_launch), "cache_resident": resident})
                    if resident:
                        out["peak_note"] = ("footprint below the 256 MiB Infinity Cache: priced against the guide's random-row rate "
                                            "out of it (8.6 TB/s), not against HBM (8 TB/s)")
                out["traffic"] = pmc_traffic(kname, args.workload)
                if kname in MFMA_KERNELS:
                    # counter-based: SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES) from profiles/<PMC_MFMA_FILE>
                    out["mfma_busy"] = pmc_mfma(kname, args.workload)
                return out
            result["roofline_phm_gemm"] = frac_of("phm_gemm_fwd")
            # (molecule batches: the forward aggregation rides inside the layer's first GEMM — `roofline_phm_gemm_aggregate` —
            # and the row walker that is left is the backward one)
            result["roofline_aggregate"] = frac_of("aggregate_enc_fwd") or frac_of("aggregate_fwd") or frac_of("aggregate_enc_dx")
            if "phm_gemm_ag_fwd" in times:
                result["roofline_phm_gemm_aggregate"] = frac_of("phm_gemm_ag_fwd")
            roof = frac_of(name)
            roof.update({"traffic_source": "profiles/" + PMC_FILE + " (rocprofv3 --pmc passes of this workload: FETCH_SIZE x2 + "
                         "WRITE_SIZE per launch, MI355X_MICROARCH.md HBM section, launch-weighted over the family's instantiations; "
                         "null = no passes committed for this workload)",
                         "engine_kernels_us_per_step": {k: round(1e3 * v["total_ms"] / args.profile_steps, 2)
                                                        for k, v in sorted(times.items())}})
            result["roofline"] = roof
        if world == 1 and args.cpu_steps > 0:
            # (bounded: ~25 s of CPU work whatever the workload — the wide configurations take seconds per oracle step)
            result["cpu_baseline"] = cpu_baseline(args, args.workload, model, host_batches[0], budget_s=25.0,
                                                  warmup=3 if args.workload.startswith("molhiv") else 1)
            all_cores = os.cpu_count() or args.cpu_threads
            if all_cores != args.cpu_threads and all_cores <= args.cpu_all_cores_max:
                # the secondary denominator, every core of the GPU host (count stated)
                result["cpu_baseline_all_cores"] = cpu_baseline(args, args.workload, model, host_batches[0], threads=all_cores,
                                                                steps=max(5, args.cpu_steps // 2))
            elif all_cores != args.cpu_threads:
                # measured once on the round-2 GPU host (256 logical CPUs): torch's intra-op pool at that width needs 91 s per
                # 512-graph step on these small tensors — 4.3 graphs/s, 230x SLOWER than 6 threads.  On such hosts the secondary
                # denominator is taken at a BOUNDED thread count instead (min(logical CPUs / 2, --cpu-many-threads), stated in
                # `cores`), with a time budget so that the default run stays within minutes.
                many = max(args.cpu_threads + 1, min(all_cores // 2, args.cpu_many_threads))
                leg = cpu_baseline(args, args.workload, model, host_batches[0], threads=many, steps=max(3, args.cpu_steps // 4),
                                   warmup=1, budget_s=20.0)
                leg["sample"] = (f"{many} threads on a host with {all_cores} logical CPUs (all {all_cores}: 4.3 graphs/s measured once, "
                                 f"91 s per step, profiles/README.md); ") + leg["sample"]
                result["cpu_baseline_all_cores"] = leg
    if world > 1 or force_sync:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        _flush_c_stdio()
        print(json.dumps(result), flush=True)          # after RCCL's own shutdown banner: the JSON line is the LAST line of stdout


if __name__ == "__main__":
    main()
