"""Training-step harness reproducing the reference's ``train()`` inner loop
(benchmarks/train_hiv.py:165-218) on pre-collated, device-resident synthetic
batches: zero_grad -> forward -> NaN-masked BCE -> backward -> grad clip -> Adam.
Shared by bench.py, __graft_entry__.smoke() and the tests."""
from __future__ import annotations

import types
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

from . import synth
from .ops import (atom_encode_spec, deferred_param_grads, dense_weights, expect_loss, gate_specs, masked_bce_with_logits, masked_l1,
                  phm_pairs, phm_weight_regularization_, prebuilt_H, stat_accumulators)
from .optim import ClipAdam

# BASELINE.json configs -> PHMSkipConnectAdd arguments (SURVEY.md §8d "Model settings")
WORKLOADS: Dict[str, dict] = {
    # configs[0]: the reference's own CPU-runnable case
    "molhiv_b32": dict(dataset="molhiv", batch_size=32, phm_dim=4, hidden=128, layers=4, d_units=[128, 32],
                       target_dim=1, sc_type="first"),
    # configs[1]: the configuration the metric is quoted on
    "molhiv_b512": dict(dataset="molhiv", batch_size=512, phm_dim=4, hidden=128, layers=4, d_units=[128, 32],
                        target_dim=1, sc_type="first"),
    # SURVEY §8(d) variants of configs[1]: PHMConv instead of PHMGINEConv (mlp=False), and the reference's default
    # molhiv aggregation (train_hiv.py:115-117: softmax with a learnable inverse temperature)
    "molhiv_b512_conv": dict(dataset="molhiv", batch_size=512, phm_dim=4, hidden=128, layers=4, d_units=[128, 32],
                             target_dim=1, sc_type="first", mlp=False),
    "molhiv_b512_softmax": dict(dataset="molhiv", batch_size=512, phm_dim=4, hidden=128, layers=4, d_units=[128, 32],
                                target_dim=1, sc_type="first", msg_aggr="softmax"),
    # configs[2]
    "zinc_b1024": dict(dataset="zinc", batch_size=1024, phm_dim=8, hidden=256, layers=5, d_units=[256, 128],
                       target_dim=1, sc_type="last", regression=True),
    # configs[3] (per-GPU share of the 2048 batch)
    "molpcba_b1024": dict(dataset="molpcba", batch_size=1024, phm_dim=4, hidden=256, layers=5, d_units=[768, 256],
                          target_dim=128, sc_type="first", nan_frac=0.6),
}
# The reference's PUBLISHED model shapes (benchmarks/README.md:45-91, benchmarks/run_script_zinc_phm4.sh:4-5) at the BASELINE
# batch sizes, dropout as published: per-component sizes 50 / 40 / 45 (not multiples of four), GEMM depths 200 / 180.
# ``*_pub_nodrop``: the same models without dropout (what the oracle parity tests run; dropout parity: test_gpu_dropout.py).
WORKLOADS.update({
    # ogbg-molhiv paper model: PHC-4, 2 x 200, softmax aggregation with a learnable inverse temperature
    "molhiv_pub_phc4_2x200": dict(dataset="molhiv", batch_size=512, phm_dim=4, hidden=200, layers=2, d_units=[128, 32],
                                  target_dim=1, sc_type="first", msg_aggr="softmax", dropout_mpnn=[0.2, 0.2], dropout_dn=[0.3, 0.1]),
    # ZINC 100 K model: PHC-5, 4 x 200.  The README's --input_embed_dim 180 / --d_units 200,128 do not construct
    # (models.py:46 wants one width, layers.py:231 wants 128 % 5 == 0): 200 everywhere, head 200, 130
    "zinc_pub_phc5_4x200": dict(dataset="zinc", batch_size=1024, phm_dim=5, hidden=200, layers=4, d_units=[200, 130],
                                target_dim=1, sc_type="last", regression=True, dropout_dn=[0.2, 0.1]),
    # run_script_zinc_phm4.sh: PHC-4, embed 180, 4 x 180, head 180, 80
    "zinc_pub_phc4_4x180": dict(dataset="zinc", batch_size=1024, phm_dim=4, hidden=180, layers=4, d_units=[180, 80],
                                target_dim=1, sc_type="last", regression=True, dropout_dn=[0.2, 0.1]),
    # ogbg-molpcba published model (benchmarks/run_script_pcba_phm4.sh:3-10, README.md:58-66): PHC-4, 7 x 512, PHMConv
    # (--mlp_mp False), head 768, 256 over 128 tasks, dropout 0.1 in every layer; the script's default batch of 512
    "molpcba_pub_phc4_7x512": dict(dataset="molpcba", batch_size=512, phm_dim=4, hidden=512, layers=7, d_units=[768, 256],
                                   target_dim=128, sc_type="first", mlp=False, nan_frac=0.6, dropout_mpnn=[0.1] * 7,
                                   dropout_dn=[0.3, 0.2]),
})
for _k in ("molhiv_pub_phc4_2x200", "zinc_pub_phc5_4x200", "zinc_pub_phc4_4x180", "molpcba_pub_phc4_7x512"):
    WORKLOADS[_k + "_nodrop"] = {k: v for k, v in WORKLOADS[_k].items() if not k.startswith("dropout")}
    WORKLOADS[_k + "_nodrop_swish"] = dict(WORKLOADS[_k + "_nodrop"], activation="swish")
# smooth-activation twins (parity runs: without the ReLU kink every gradient tensor is comparable to ~1e-5)
for _k in ("molhiv_b32", "molhiv_b512", "zinc_b1024", "molpcba_b1024", "molhiv_b512_softmax"):
    WORKLOADS[_k + "_swish"] = dict(WORKLOADS[_k], activation="swish")


def model_kwargs(w: dict) -> dict:
    kw = dict(phm_dim=w["phm_dim"], learn_phm=True, atom_encoded_dim=w["hidden"], mp_layers=[w["hidden"]] * w["layers"],
              dropout_mpnn=list(w.get("dropout_mpnn", [0.0] * w["layers"])), mlp=w.get("mlp", True), msg_aggr=w.get("msg_aggr", "sum"),
              norm_mp="naive-batch-norm",
              norm_dn="naive-batch-norm", pooling="softattention", activation=w.get("activation", "relu"), real_trafo="linear",
              w_init="phm", c_init="standard", downstream_layers=list(w["d_units"]),
              dropout_dn=list(w.get("dropout_dn", [0.0] * len(w["d_units"]))), target_dim=w["target_dim"], sc_type=w["sc_type"],
              msg_encoder="identity")
    if kw["msg_aggr"] == "softmax":
        kw.update(initial_beta=1.0, learn_beta=True)          # train_hiv.py defaults
    if w["dataset"] == "zinc":
        kw.update(atom_input_dims=synth.ZINC_ATOM_DIMS, bond_input_dims=synth.ZINC_BOND_DIMS)
    return kw


def build_model(workload: str, device, seed: int = 0):
    from phc.hypercomplex.undirectional.models import PHMSkipConnectAdd
    import numpy as np
    w = WORKLOADS[workload]
    torch.manual_seed(seed)
    np.random.seed(seed)
    model = PHMSkipConnectAdd(**model_kwargs(w))
    return model.to(device)


def make_batch(workload: str, seed: int) -> synth.GraphBatch:
    w = WORKLOADS[workload]
    return synth.molecule_batch(w["batch_size"], seed=seed, dataset=w["dataset"], num_tasks=w["target_dim"],
                                nan_frac=w.get("nan_frac", 0.0), regression=w.get("regression", False))


def to_device(gb: synth.GraphBatch, device) -> types.SimpleNamespace:
    """The object handed to ``model(data)``: same fields PyG's Batch exposes to
    PHMSkipConnectAdd.forward (models.py:220)."""
    y = torch.from_numpy(gb.y).to(device)
    return types.SimpleNamespace(
        x=torch.from_numpy(gb.x).to(device), edge_index=torch.from_numpy(gb.edge_index).to(device),
        edge_attr=torch.from_numpy(gb.edge_attr).to(device), batch=torch.from_numpy(gb.batch).to(device),
        y=y, y_mask=~torch.isnan(y), num_graphs=gb.num_graphs, num_nodes=gb.num_nodes, num_edges=gb.num_edges,
        batch_is_sorted=bool((gb.batch[1:] >= gb.batch[:-1]).all()))        # collated batches are; checked on the host copy


def loss_fn(logits: torch.Tensor, data, regression: bool = False) -> torch.Tensor:
    if regression:
        # benchmarks/train_zinc.py uses L1.  NaN labels (the padding graphs of a fixed-shape batch, graphed.py) stay out of the
        # mean exactly like unlabeled tasks do in the BCE; without any NaN this is F.l1_loss.  One engine kernel each way.
        return masked_l1(logits, data.y)
    # train_hiv.py:174-178: mask = ~isnan(y); BCE(logits[mask], y[mask]) — fused, without the host-synchronising
    # boolean gather of the reference
    return masked_bce_with_logits(logits, data.y)


@dataclass
class StepConfig:
    regression: bool = False
    grad_clip: float = 2.0              # --grad_clipping default, train_hiv.py
    optimizer: bool = True
    # the explicit weight regulariser of train(): loss += lr * weight_decay * phm_weight_regularization(model, p)
    # (train_hiv.py:181-196; --weightdecay defaults to 0.001, --regularization to 2, --lr to 0.001).  0 = off.
    weight_decay: float = 0.0
    reg_p: int = 2
    lr: float = 1e-3


_ONE: Dict[torch.device, torch.Tensor] = {}


def backward_seed(loss: torch.Tensor) -> torch.Tensor:
    """d loss / d loss = 1 as a resident constant: ``loss.backward()`` would launch a fill kernel for it every step."""
    one = _ONE.get(loss.device)
    if one is None:
        from .ops import register_unit_seed
        one = _ONE[loss.device] = register_unit_seed(torch.ones((), dtype=loss.dtype, device=loss.device))
    return one


def train_step(model, data, optimizer: Optional[torch.optim.Optimizer], cfg: StepConfig,
               grad_sync=None) -> torch.Tensor:
    """One iteration of the reference loop body (train_hiv.py:175-201) without the
    host-synchronising ``loss.item()``."""
    if optimizer is not None:
        optimizer.zero_grad(set_to_none=True)
    else:
        for p in model.parameters():
            p.grad = None
    advance_dropout_rng(model, data)
    # operand matrices of all PHM layers: one launch; dW/dA/db of all layers: three batched launches after the backward
    # (stat_accumulators: the batch-norm statistics of the step travel through zeroed accumulator slots, no merge launches)
    # (the atom encoder rides in the operand-build launch: atom_encode_spec)
    with stat_accumulators(data.x.device), prebuilt_H(phm_pairs(model), dense_weights(model), gate_specs(model),
                                                      encode=atom_encode_spec(model, data)), deferred_param_grads():
        with expect_loss(getattr(data, "y", None), "l1" if cfg.regression else "bce"):     # (a head that can leaves the loss behind)
            logits = model(data)
        loss = loss_fn(logits, data, cfg.regression)
        loss.backward(backward_seed(loss))
        parked = park_weight_regulariser(model, loss, cfg)       # rides in the step tail's launches when it can
    if not parked:
        apply_weight_regulariser(model, loss, cfg)
    if grad_sync is not None:
        grad_sync(model)
    if optimizer is not None and cfg.optimizer:
        optimizer_step(model, optimizer, cfg)
    return loss


def advance_dropout_rng(model, data) -> None:
    """New dropout masks for this step (one tiny launch, only for models that drop anything: the reference's defaults do,
    --dropout_mpnn 0.3,0.1 / --dropout_dn 0.3,0.1; the BASELINE configs do not)."""
    from .ops import model_uses_dropout, rng_advance
    if model_uses_dropout(model):
        rng_advance(data.x.device)


def park_weight_regulariser(model, loss: torch.Tensor, cfg: StepConfig) -> bool:
    """Inside the step's ``deferred_param_grads`` block, after ``backward``: hand the regulariser to the step tail (no launch of
    its own: gradient added where dW is written, value added to ``loss``).  True = taken (or nothing to do)."""
    if cfg.weight_decay <= 0.0:
        return True
    from .ops import regularised_weights, request_weight_regulariser
    return request_weight_regulariser(regularised_weights(model), cfg.lr * cfg.weight_decay, cfg.reg_p, loss.detach())


def apply_weight_regulariser(model, loss: torch.Tensor, cfg: StepConfig) -> None:
    """train_hiv.py:181-196 as ONE engine launch after the backward pass: d reg / d W added into the W gradients, the value
    into ``loss`` (what the reference logs).  No-op for weight_decay 0."""
    if cfg.weight_decay > 0.0:
        phm_weight_regularization_(model, cfg.lr * cfg.weight_decay, cfg.reg_p, loss=loss.detach())


def optimizer_step(model, optimizer, cfg: StepConfig) -> None:
    """train_hiv.py:198-201: clip_grad_norm_(max_norm) then the Adam step — two launches with the engine's
    ClipAdam, PyTorch's multi-tensor kernels for any other optimizer."""
    if isinstance(optimizer, ClipAdam):
        optimizer.step(max_norm=cfg.grad_clip)
        return
    if cfg.grad_clip > 0:
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=cfg.grad_clip, norm_type=2)
    optimizer.step()
