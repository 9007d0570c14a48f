"""Round-4 launch-chain fusions of the training step, each against the launches it replaces on the SAME step (same model state, same
batch) — they reorder no floating-point operation, so every output is compared bit for bit:
  * the model-level norm's element-wise pass folded into the next layer's aggregation (phc_aggregate_norm_encp_fwd; models.py:208-215
    followed by messagepassing.py:132-146), sc_type "first" and "last", ReLU and swish, padded batches under hipGraph replay;
  * the skips' gradient shares added inside the first layer's aggregation backward (phc_aggregate_dx_rows_multi; models.py:230-234)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _step(workload, toggles, graph=False):
    from phc_gnn_amd import graphed, harness, ops
    from phc_gnn_amd.optim import ClipAdam
    keep = {k: getattr(ops, k) for k in toggles}
    try:
        for k, v in toggles.items():
            setattr(ops, k, v)
        model = harness.build_model(workload, DEV, seed=3).train()
        gb = harness.make_batch(workload, seed=11)
        cfg = harness.StepConfig(regression=bool(harness.WORKLOADS[workload].get("regression")), optimizer=False, weight_decay=1e-3)
        if graph:
            n_max, e_max = graphed.bucket_shape([gb])
            padded = graphed.padded_to_device(graphed.pad_batch(gb, n_max + 37, e_max + 91), DEV)
            sd = {k: v.clone() for k, v in model.state_dict().items()}
            step = graphed.GraphedTrainStep(model, ClipAdam(model.parameters(), lr=1e-3), cfg, padded, warmup=2)
            model.load_state_dict(sd)
            loss = step.step(padded)
        else:
            loss = harness.train_step(model, harness.to_device(gb, DEV), None, cfg)
        torch.cuda.synchronize()
        return (loss.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None},
                {k: b.clone() for k, b in model.named_buffers()})
    finally:
        for k, v in keep.items():
            setattr(ops, k, v)


def _same(a, b):
    assert torch.equal(a[0], b[0]), (float(a[0]), float(b[0]))
    assert a[1].keys() == b[1].keys()
    for k in a[1]:
        assert torch.equal(a[1][k], b[1][k]), ("grad", k, float((a[1][k] - b[1][k]).abs().max()))
    for k in a[2]:
        assert torch.equal(a[2][k], b[2][k]), ("buffer", k)


@pytest.mark.parametrize("workload", ["molhiv_b32", "molhiv_b32_swish", "zinc_small_last"])
@pytest.mark.parametrize("graph", [False, True])
def test_norm_folded_into_the_next_aggregation_is_bit_identical(workload, graph):
    from phc_gnn_amd import harness
    if workload == "zinc_small_last":                                     # sc_type "last": the skip IS the parked tensor
        harness.WORKLOADS.setdefault("zinc_small_last", dict(dataset="zinc", batch_size=48, phm_dim=4, hidden=64, layers=3, d_units=[64, 32],
                                                             target_dim=1, sc_type="last", regression=True))
    on = _step(workload, dict(NORM_INTO_AGGREGATE=True), graph)
    off = _step(workload, dict(NORM_INTO_AGGREGATE=False), graph)
    _same(on, off)


@pytest.mark.parametrize("workload", ["molhiv_b32", "molpcba_small"])
def test_skip_gradient_shares_meet_in_the_first_aggregation_backward(workload):
    """Same sums in a different order (slot order inside the dx kernel instead of the n-way sum kernel's): 1e-6 of the largest entry."""
    from phc_gnn_amd import harness
    if workload == "molpcba_small":                                       # five layers: the eight-slot instantiation
        harness.WORKLOADS.setdefault("molpcba_small", dict(dataset="molpcba", batch_size=24, phm_dim=4, hidden=64, layers=5, d_units=[64, 32],
                                                           target_dim=8, sc_type="first", nan_frac=0.5))
    on = _step(workload, dict(SKIP_SUM_HANDOFF=True))
    off = _step(workload, dict(SKIP_SUM_HANDOFF=False))
    assert torch.allclose(on[0], off[0], rtol=0, atol=0)
    for k in on[1]:
        ref = off[1][k]
        assert float((on[1][k] - ref).abs().max()) <= 2e-6 * max(float(ref.abs().max()), 1e-30), k


@pytest.mark.parametrize("workload", ["molhiv_b32", "zinc_small_last", "zinc_pub_phc4_4x180_nodrop"])
def test_atom_encoder_inside_the_operand_build_launch_is_bit_identical(workload):
    """phc_phm_build_H_batched_ex: the atom encoder as passenger workgroups of the step's first launch (the same table rows added in
    the same order as phc_encode_fwd) — k % 4 != 0 models (flat table images) included."""
    from phc_gnn_amd import harness
    if workload == "zinc_small_last":
        harness.WORKLOADS.setdefault("zinc_small_last", dict(dataset="zinc", batch_size=48, phm_dim=4, hidden=64, layers=3, d_units=[64, 32],
                                                             target_dim=1, sc_type="last", regression=True))
    _same(_step(workload, dict(ENCODE_FRONT=True)), _step(workload, dict(ENCODE_FRONT=False)))


def test_atom_encoder_passenger_reports_bad_indices():
    """An atom feature outside its vocabulary is read as row 0 and raises at the next check point, as in the stand-alone launch."""
    from phc_gnn_amd import harness, lib
    model = harness.build_model("molhiv_b32", DEV, seed=3).train()
    data = harness.to_device(harness.make_batch("molhiv_b32", seed=11), DEV)
    data.x = data.x.clone()
    data.x[5, 0] = 100000
    harness.train_step(model, data, None, harness.StepConfig(optimizer=False))
    torch.cuda.synchronize()
    with pytest.raises(IndexError):
        lib.check_indices(torch.device(DEV))
