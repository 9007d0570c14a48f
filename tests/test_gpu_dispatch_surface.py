"""The dispatch surface of the training step under stress (no reference line — this protects parity).

ops.py chooses among many launch forms per layer (fused / folded / deferred / passenger variants, each behind a module-level switch).
Two kinds of silent failure live there: a kernel that reads memory nobody wrote in THIS step (a stale image, a skipped refresh, a
recycled object id — round 4's once-in-hundreds wrong loss), and a switch combination no other test happens to exercise.

  * ``test_repeated_steps_are_bit_equal_under_allocator_pollution``: the step from the same state, >= 30 times per (workload, mode),
    with the caching allocator's free blocks filled with large finite garbage between the steps — every loss and a gradient checksum
    must equal the first run's bit for bit (the engine's reductions are order-fixed: tests/test_gpu_stat_accumulators.py).
  * ``test_every_single_toggle_flip_holds_the_oracle``: a fixed small step under each single flip of every fusion switch, held to the
    CPU oracle (logits / loss 1e-5, every gradient 3e-5 on the smooth-activation model)."""
import pytest
import torch

from tests._util import TOL, assert_close, relerr

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
REPEATS = 30

EXTRA_WORKLOADS = {
    # PHMConv layers (mlp = False; SURVEY 8d's second variant): the statistics-with-residual GEMM and the self loop's gradient hand-off
    "molhiv_b32_conv": dict(dataset="molhiv", batch_size=32, phm_dim=4, hidden=128, layers=4, d_units=[128, 32], target_dim=1,
                            sc_type="first", mlp=False),
    # a small 256-wide PHM-8 model (configs[2]'s kernels: K = 256 GEMM family, k = 32 components, sc_type "last", L1 loss)
    "zinc_small_256": dict(dataset="zinc", batch_size=48, phm_dim=8, hidden=256, layers=3, d_units=[256, 128], target_dim=1,
                           sc_type="last", regression=True),
}


def _workload(name):
    from phc_gnn_amd import harness
    if name in EXTRA_WORKLOADS:
        harness.WORKLOADS.setdefault(name, EXTRA_WORKLOADS[name])
        harness.WORKLOADS.setdefault(name + "_swish", dict(EXTRA_WORKLOADS[name], activation="swish"))
    return harness.WORKLOADS[name]


def _dirty(seed):
    """Allocate and free blocks of many sizes filled with 1e3 + noise: what torch.empty hands out next is neither zero nor the
    previous step's values."""
    g = torch.Generator(device=DEV).manual_seed(seed)
    keep = []
    for sz in (1 << 10, 1 << 12, 1 << 14, 1 << 16, 1 << 18, 1 << 20, 1 << 22, 3 << 20, 5 << 18, 7 << 14, 9 << 10):
        for _ in range(3):
            t = torch.empty(sz, device=DEV)
            t.uniform_(900.0, 1100.0, generator=g)
            keep.append(t)
    del keep


def _digest(model, loss):
    """(loss bits, per-parameter gradient sums in float64) — equal bits in, equal digest out."""
    sums = torch.stack([p.grad.double().sum() if p.grad is not None else torch.zeros((), dtype=torch.float64, device=DEV)
                        for p in model.parameters()])
    absmax = torch.stack([p.grad.abs().max().double() if p.grad is not None else torch.zeros((), dtype=torch.float64, device=DEV)
                          for p in model.parameters()])
    return loss.detach().clone().view(torch.int32), sums, absmax


@pytest.mark.parametrize("workload", ["molhiv_b32", "molhiv_b512", "zinc_small_256"])
@pytest.mark.parametrize("mode", ["eager", "graphed"])
def test_repeated_steps_are_bit_equal_under_allocator_pollution(workload, mode):
    from phc_gnn_amd import graphed, harness, lib
    from phc_gnn_amd.optim import ClipAdam
    w = _workload(workload)
    cfg = harness.StepConfig(regression=bool(w.get("regression")), optimizer=False, weight_decay=1e-3)
    gb = harness.make_batch(workload, seed=5)
    first = None
    if mode == "eager":
        for i in range(REPEATS):
            # a NEW model object per repeat, same seed: object ids, image serials and allocator blocks all differ between repeats
            model = harness.build_model(workload, DEV, seed=1).train()
            data = harness.to_device(gb, DEV)
            _dirty(i)
            loss = harness.train_step(model, data, None, cfg)
            got = _digest(model, loss)
            if first is None:
                first = got
            for a, b, what in zip(got, first, ("loss", "grad sums", "grad max")):
                assert torch.equal(a, b), f"{workload} eager repeat {i}: {what} differ from the first run"
            del model, data
    else:
        model = harness.build_model(workload, DEV, seed=1).train()
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        n_max, e_max = graphed.bucket_shape([gb])
        padded = graphed.padded_to_device(graphed.pad_batch(gb, n_max + 19, e_max + 45), DEV)
        step = graphed.GraphedTrainStep(model, ClipAdam(model.parameters(), lr=1e-3), cfg, padded, warmup=2)
        for i in range(REPEATS):
            model.load_state_dict(sd)
            _dirty(100 + i)
            loss = step.step(padded)
            got = _digest(model, loss)
            if first is None:
                first = got
            for a, b, what in zip(got, first, ("loss", "grad sums", "grad max")):
                assert torch.equal(a, b), f"{workload} replay {i}: {what} differ from the first replay"
    torch.cuda.synchronize()
    lib.check_indices(torch.device(DEV))


# every module-level fusion switch of ops.py (name -> the value that is NOT the default); the default step is the "none" case
TOGGLES = ["none", "GEMM_AGGREGATE", "READOUT_FUSED", "NORM_INTO_AGGREGATE", "MLP_NORM_FUSED", "SKIP_SUM_HANDOFF", "ENCODE_FRONT",
           "STEP_TAIL_MERGED", "GATE_COMPOSED", "TAIL_EXTRAS", "TAIL_WIDE", "TABLE_GRADS_COUNTED", "STAT_ACC", "FLAT_TABLES",
           "BN_BWD_SUMS_IN_GEMM", "DOWNSTREAM_FUSED", "STATS_WITH_RESIDUAL", "RESIDUAL_HANDOFF", "HEAD_CHAIN", "HEAD_DROPOUT", "POOL_BWD_GATE",
           "MLP_BWD_FUSED", "MLP_FWD_FUSED", "MLP_FWD_FUSED_AGG", "HEAD_LOSS", "MLP_BWD_FUSED_AGG", "HEAD_FUSED_BWD", "SOFTMAX_INTO_MLP"]
ORACLE_KW = {
    "molhiv_b32_swish": dict(phm_dim=4, num_layers=4, num_atom_feats=9, num_bond_feats=3, num_dn_hidden=2, mlp=True, sc_type="first",
                             act="swish"),
    "zinc_small_256_swish": dict(phm_dim=8, num_layers=3, num_atom_feats=1, num_bond_feats=1, num_dn_hidden=2, mlp=True, sc_type="last",
                                 act="swish"),
    "molhiv_b32_conv_swish": dict(phm_dim=4, num_layers=4, num_atom_feats=9, num_bond_feats=3, num_dn_hidden=2, mlp=False, sc_type="first",
                                  act="swish"),
}
_ORACLE: dict = {}


def _oracle(workload, sd, gb, regression):
    from oracle import phc_oracle as O
    if workload in _ORACLE:
        return _ORACLE[workload]
    names = [k for k in sd if not ("running" in k or "num_batches" in k)]
    p = {k: (v.clone().double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    p32 = {k: v.clone() for k, v in sd.items()}
    for k in names:
        p[k].requires_grad_()
    args = tuple(torch.from_numpy(a) for a in (gb.x, gb.edge_index, gb.edge_attr, gb.batch))
    y = torch.from_numpy(gb.y)
    with torch.no_grad():
        lt32 = O.skip_connect_add_forward(p32, *args, training=True, num_graphs=gb.num_graphs, **ORACLE_KW[workload])
        lo32 = torch.nn.functional.l1_loss(lt32, y) if regression else O.bce_loss_masked(lt32, y)
    lt = O.skip_connect_add_forward(p, *args, training=True, num_graphs=gb.num_graphs, **ORACLE_KW[workload])
    lo = torch.nn.functional.l1_loss(lt, y.double()) if regression else O.bce_loss_masked(lt, y.double())
    gr = torch.autograd.grad(lo, [p[k] for k in names], allow_unused=True)
    _ORACLE[workload] = (lt32, lo32, dict(zip(names, gr)))
    return _ORACLE[workload]


def check_step_against_oracle(workload, toggle="none", tag=None):
    """One small step (smooth activation, no explicit regulariser) held to the oracle: logits and loss within 1e-5 of the float32
    oracle, every gradient tensor within 3e-5 of the float64 oracle — with ``ops.<toggle>`` off its default (or none)."""
    from phc_gnn_amd import harness, ops
    w = _workload(workload[:-6])
    regression = bool(w.get("regression"))
    tag = tag or f"{toggle} flipped"
    keep = getattr(ops, toggle) if toggle != "none" else None
    try:
        if toggle != "none":
            assert isinstance(keep, bool), (toggle, keep)
            setattr(ops, toggle, not keep)
        model = harness.build_model(workload, DEV, seed=7).train()
        gb = harness.make_batch(workload, seed=13)
        sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        data = harness.to_device(gb, DEV)
        cfg = harness.StepConfig(regression=regression, optimizer=False, weight_decay=0.0)
        logits_box = {}
        hk = model.downstream.register_forward_hook(lambda m, a, out: logits_box.__setitem__("l", out.detach().clone()))
        try:
            loss = harness.train_step(model, data, None, cfg)
        finally:
            hk.remove()
        torch.cuda.synchronize()
    finally:
        if toggle != "none":
            setattr(ops, toggle, keep)
    lt32, lo32, g64 = _oracle(workload, sd, gb, regression)
    assert_close(logits_box["l"][:gb.num_graphs], lt32, TOL, f"{workload} [{tag}]: logits")
    assert_close(loss, lo32, TOL, f"{workload} [{tag}]: loss")
    bad = []
    for k, prm in model.named_parameters():
        r = g64.get(k)
        if r is None or float(r.abs().max()) < 1e-7:
            continue
        assert prm.grad is not None, k
        e = relerr(prm.grad, r)
        if e > 3e-5:
            bad.append((e, k))
    assert not bad, f"{workload} [{tag}]: " + "; ".join(f"{k}: {e:.2e}" for e, k in sorted(bad, reverse=True)[:6])


@pytest.mark.parametrize("workload", list(ORACLE_KW))
@pytest.mark.parametrize("toggle", TOGGLES)
def test_every_single_toggle_flip_holds_the_oracle(workload, toggle):
    """The step without the explicit regulariser (the oracle call is the model's loss alone), smooth activation: logits and loss
    within 1e-5 of the float32 oracle, every gradient tensor within 3e-5 of the float64 oracle — with ONE switch off its default."""
    from phc_gnn_amd import ops
    if toggle != "none" and not hasattr(ops, toggle):
        pytest.skip(f"ops.{toggle} no longer exists")
    check_step_against_oracle(workload, toggle)


# The switches the LIBRARY reads from the environment (``static const`` reads inside csrc/: a value is fixed for the life of a process).
# What is left of them after round 6's pruning are product switches — which kernel family serves a call — and each is held to the
# oracle on its non-default side in a FRESH child process (never a re-exec of a process that has touched the GPU).
ENV_FLIPS = [("PHC_GEMM_NO_HS", "1"), ("PHC_GEMM_HS_PUB", "0"), ("PHC_GEMM_AGGREGATE", "0"), ("PHC_GEMM_AGGREGATE_WIDE", "0"),
             ("PHC_DOWNSTREAM_V2", "0"), ("PHC_HEAD_CHAIN", "0"), ("PHC_READOUT_FUSED", "0"), ("PHC_POOL_BWD_GATE", "0"), ("PHC_DH_PAIR", "0"),
             ("PHC_MLP_BWD_FUSED", "0"), ("PHC_MLP_FWD_FUSED", "0"), ("PHC_CLIP_ADAM_FUSED", "0"), ("PHC_HEAD_FUSED_BWD", "0")]


@pytest.mark.parametrize("name,value", ENV_FLIPS)
def test_every_library_switch_holds_the_oracle_in_a_fresh_process(name, value):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **{name: value})
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    workloads = ["molhiv_b32_swish", "zinc_small_256_swish"] if name in ("PHC_GEMM_NO_HS", "PHC_GEMM_AGGREGATE_WIDE", "PHC_DH_PAIR", "PHC_GEMM_HS_PUB") \
        else ["molhiv_b32_swish"]
    r = subprocess.run([sys.executable, "-m", "tests.helpers.flip_child", f"{name}={value}"] + workloads, cwd=root, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "FLIP-OK" in r.stdout, (name, value, r.stdout[-2000:], r.stderr[-2000:])
