"""The training scripts' explicit weight regulariser (regularization.py:15-23, train_hiv.py:181-196) as one engine launch:
value and gradients against the reference's own (tests/golden/g11_regulariser.npz), stand-alone and inside the training
step — eager (harness.train_step) and captured (GraphedTrainStep) — with the script's defaults lr 1e-3, weightdecay 1e-3, p 2."""
import types

import numpy as np
import pytest
import torch

from tests._util import TOL, assert_close, load, sub, tensors

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class _Stack(torch.nn.Module):
    def __init__(self, ws):
        super().__init__()
        self.layers = torch.nn.ModuleList()
        for w in ws:
            m = torch.nn.Module()
            m.W = torch.nn.Parameter(w.clone())
            self.layers.append(m)


@pytest.mark.parametrize("p", [1, 2])
def test_regulariser_kernel_matches_the_reference(p):
    from phc_gnn_amd import ops
    z = tensors(load("g11_regulariser"))
    k = int(z["stack::num"])
    st = _Stack([z[f"stack::W{i}"] for i in range(k)]).to(DEV)
    base = [1e-3 * torch.randn_like(l.W) for l in st.layers]               # gradients already there: the kernel ADDS
    for l, b in zip(st.layers, base):
        l.W.grad = b.clone()
    loss = torch.tensor(0.25, device=DEV)
    reg = ops.phm_weight_regularization_(st, 0.5, p, loss=loss)
    assert_close(reg, 0.5 * z[f"stack::reg_p{p}"], TOL, "value")
    assert_close(loss, 0.25 + 0.5 * z[f"stack::reg_p{p}"], TOL, "loss += reg")
    for i, (l, b) in enumerate(zip(st.layers, base)):
        assert_close(l.W.grad - b, 0.5 * z[f"stack::dW{i}_p{p}"], 2e-5, f"dW{i}")
    assert bool(torch.isfinite(st.layers[3].W.grad).all())                 # the all-zero column: subgradient 0, no NaN


def _cfg1_model(z):
    from phc.hypercomplex.undirectional.models import PHMSkipConnectAdd
    m = PHMSkipConnectAdd(phm_dim=4, atom_encoded_dim=128, mp_layers=[128] * 4, dropout_mpnn=[0.0] * 4, mlp=True,
                          downstream_layers=[128, 32], dropout_dn=[0.0, 0.0], target_dim=1, sc_type="first", msg_aggr="sum",
                          norm_mp="naive-batch-norm", norm_dn="naive-batch-norm", pooling="softattention")
    m.load_state_dict(sub(z, "sd::"), strict=True)
    return m.to(DEV).train()


@pytest.mark.parametrize("path", ["eager", "graphed"])
def test_regularised_training_step_matches_the_reference(path):
    from phc_gnn_amd import graphed, harness, synth
    from phc_gnn_amd.optim import ClipAdam
    z8, z = tensors(load("g8_model_cfg1")), tensors(load("g11_regulariser"))
    model = _cfg1_model(z8)
    gb = synth.GraphBatch(x=z8["x"].numpy(), edge_index=z8["edge_index"].numpy(), edge_attr=z8["edge_attr"].numpy(),
                          batch=z8["batch"].numpy(), y=z8["y"].numpy().astype(np.float32), num_graphs=int(z8["y"].shape[0]))
    cfg = harness.StepConfig(optimizer=False, weight_decay=1e-3, reg_p=2, lr=1e-3)
    if path == "eager":
        loss = harness.train_step(model, harness.to_device(gb, DEV), None, cfg)
    else:
        n_max, e_max = graphed.bucket_shape([gb])
        padded = graphed.padded_to_device(graphed.pad_batch(gb, n_max, e_max), DEV)
        step = graphed.GraphedTrainStep(model, ClipAdam(model.parameters(), lr=1e-3), cfg, padded, warmup=2)
        model.load_state_dict(sub(z8, "sd::"), strict=True)
        loss = step.step(padded)
    assert_close(loss, z["model::loss"], TOL, "loss incl. the regulariser")
    worst = 0.0
    for k, ref in sub(z, "model::grad::").items():
        g = dict(model.named_parameters())[k].grad
        e = float((g.cpu() - ref).abs().max()) / float(ref.abs().max())
        worst = max(worst, e)
    assert worst < 5e-4, worst                      # same bound as the unregularised G8 gradients (tests/test_gpu_model.py)
    # and the regulariser's own share of those gradients is exact: (with) - (without) == the reference's d reg / d W
    reg_only = {k: ref - z8["grad::" + k] for k, ref in sub(z, "model::grad::").items()}
    model2 = _cfg1_model(z8)
    harness.train_step(model2, harness.to_device(gb, DEV), None, harness.StepConfig(optimizer=False))
    for k, ref in reg_only.items():
        if path == "eager":
            mine = dict(model.named_parameters())[k].grad - dict(model2.named_parameters())[k].grad
            assert float((mine.cpu() - ref).abs().max()) <= 1e-4 * float(ref.abs().max()) + 1e-7 * float(z8["grad::" + k].abs().max()), k


def test_regulariser_and_gate_gradients_inside_the_step_tail_equal_their_own_launches():
    """phc_step_tail_ex: the regulariser's cells ride in the tail's reduce launch and its gradient is added where dW is written,
    the gate map's parameter gradients are workgroups of the tail's last launch reading the dense job's dH in place — the same
    arithmetic in the same order as phc_phm_weight_reg / phc_gate_param_grads behind the tail: every gradient and the loss bit for bit."""
    from phc_gnn_amd import harness, ops, synth
    z8 = tensors(load("g8_model_cfg1"))
    gb = synth.GraphBatch(x=z8["x"].numpy(), edge_index=z8["edge_index"].numpy(), edge_attr=z8["edge_attr"].numpy(),
                          batch=z8["batch"].numpy(), y=z8["y"].numpy().astype(np.float32), num_graphs=int(z8["y"].shape[0]))
    cfg = harness.StepConfig(optimizer=False, weight_decay=1e-3, reg_p=2, lr=1e-3)
    out = {}
    keep = ops.TAIL_EXTRAS
    try:
        for mode in (False, True):
            ops.TAIL_EXTRAS = mode
            model = _cfg1_model(z8)
            loss = harness.train_step(model, harness.to_device(gb, DEV), None, cfg)
            torch.cuda.synchronize()
            out[mode] = (loss.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters()})
    finally:
        ops.TAIL_EXTRAS = keep
    assert torch.equal(out[False][0], out[True][0]), (float(out[False][0]), float(out[True][0]))
    for k, g in out[False][1].items():
        assert torch.equal(g, out[True][1][k]), k
