#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REAL reference.

Runs ONLY in the build container: it imports ``phc.*`` from /root/reference
(read-only) with the stand-ins in tests/golden/_standins/ for the three
third-party packages this image lacks, feeds seeded inputs through the
reference's own functions/modules, and stores inputs + outputs as .npz.
The committed .npz files are data; no reference source travels.

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz

Every parameter is set explicitly from a seeded generator — never through
``reset_parameters`` — because the reference leaves ``b[out/n]`` uninitialised
(phc/hypercomplex/layers.py:276-278).
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
if not os.path.isdir(REF):
    sys.exit("make_golden.py needs the reference checkout at /root/reference (build container only)")
sys.path.insert(0, os.path.join(HERE, "_standins"))
sys.path.insert(0, REF)
# the repo root stays OFF sys.path: the reference's ``phc`` is a namespace package (no __init__.py), so the
# build's own regular ``phc`` package would shadow it
sys.path[:] = [p for p in sys.path if os.path.abspath(p or os.getcwd()) != REPO]

import numpy as np  # noqa: E402
import torch  # noqa: E402

import phc  # noqa: E402
assert phc.__path__[0].startswith(REF), phc.__path__

from phc.hypercomplex.kronecker import (kronecker_product, kronecker_product_single,  # noqa: E402
                                        kronecker_product_einsum_batched)
from phc.hypercomplex.layers import PHMLinear, matvec_product_new, phm_dropout  # noqa: E402
from phc.hypercomplex.utils import get_multiplication_matrices  # noqa: E402
from phc.hypercomplex.norm import PHMNorm  # noqa: E402
from phc.hypercomplex.pooling import PHMGlobalSumPooling, PHMSoftAttentionPooling  # noqa: E402
from phc.hypercomplex.downstream import PHMDownstreamNet  # noqa: E402
from phc.hypercomplex.undirectional.messagepassing import PHMMessagePassing  # noqa: E402
from phc.hypercomplex.undirectional.models import PHMSkipConnectAdd  # noqa: E402
from phc.quaternion.qr import RealP  # noqa: E402
from torch_geometric.data import Batch  # noqa: E402

import importlib.util  # noqa: E402
_spec = importlib.util.spec_from_file_location("phc_synth", os.path.join(REPO, "phc_gnn_amd", "synth.py"))
synth = importlib.util.module_from_spec(_spec)          # the seeded molecule generator only (numpy, no package import)
sys.modules["phc_synth"] = synth
_spec.loader.exec_module(synth)

torch.set_num_threads(4)


def npy(t):
    return t.detach().cpu().numpy()


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: (npy(v) if torch.is_tensor(v) else np.asarray(v)) for k, v in arrays.items()})
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.0f} KiB  ({len(arrays)} arrays)")


def fill_module(module, gen):
    """Deterministic, well-conditioned values for every parameter and buffer."""
    for name, p in module.named_parameters():
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "phm_rule":
            n = p.shape[0]
            base = torch.stack(get_multiplication_matrices(n, type="standard"), 0)
            p.data = base + 0.05 * torch.randn(p.shape, generator=gen)
        elif leaf == "W":
            p.data = torch.randn(p.shape, generator=gen) / np.sqrt(p.shape[0] * p.shape[1])
        elif leaf == "b":
            p.data = 0.1 * torch.randn(p.shape, generator=gen)
        elif leaf == "weight" and p.dim() == 1:            # batch-norm scale
            p.data = 0.5 + torch.rand(p.shape, generator=gen)
        elif leaf == "bias":
            p.data = 0.1 * torch.randn(p.shape, generator=gen)
        elif leaf == "beta":
            p.data = torch.tensor(1.0 + 0.25 * float(torch.randn((), generator=gen)))
        else:                                               # embeddings, nn.Linear weights
            p.data = torch.randn(p.shape, generator=gen) / np.sqrt(max(p.shape[-1], 1))
    for name, buf in module.named_buffers():
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "running_mean":
            buf.copy_(0.1 * torch.randn(buf.shape, generator=gen))
        elif leaf == "running_var":
            buf.copy_(0.5 + torch.rand(buf.shape, generator=gen))


def state_arrays(module, prefix="sd::"):
    return {prefix + k: v for k, v in module.state_dict().items()}


# ---------------------------------------------------------------------------
def g1_phm_linear():
    """G1: matvec_product_new forward + autograd backward."""
    gen = torch.Generator().manual_seed(101)
    cases = [  # (n, in, out, N, bias)
        (4, 128, 128, 721, True), (4, 128, 128, 1, True), (4, 128, 128, 7, False),
        (2, 64, 32, 37, True), (3, 48, 96, 37, True), (5, 200, 200, 64, True),
        (8, 256, 256, 130, True), (8, 64, 128, 7, True), (4, 128, 256, 33, True),
        (4, 256, 256, 65, True), (1, 16, 24, 9, True), (4, 8, 4, 5, True),
    ]
    out = {"num_cases": len(cases)}
    for ci, (n, fin, fout, N, bias) in enumerate(cases):
        A = torch.randn(n, n, n, generator=gen).requires_grad_()
        W = (torch.randn(n, fin // n, fout // n, generator=gen) / np.sqrt(fin)).requires_grad_()
        b = (0.1 * torch.randn(fout, generator=gen)).requires_grad_() if bias else None
        x = torch.randn(N, fin, generator=gen).requires_grad_()
        y = matvec_product_new(W=W, x=x, bias=b, phm_rule=A)
        dy = torch.randn(N, fout, generator=gen)
        grads = torch.autograd.grad(y, [x, W, A] + ([b] if bias else []), dy)
        d = dict(x=x, A=A, W=W, y=y, dy=dy, dx=grads[0], dW=grads[1], dA=grads[2])
        if bias:
            d.update(b=b, db=grads[3])
        d["meta"] = np.array([n, fin, fout, N, int(bias)])
        out.update({f"c{ci}::{k}": v for k, v in d.items()})
    save("g1_phm_linear", **out)


def g2_g3_g4_kronecker_rules():
    """G2 H builder (incl. the shapes of test_kronecker_product.py:13-19), G3
    standard rules, G4 quaternion known answer (test_realrepr_sumkronecker.py)."""
    gen = torch.Generator().manual_seed(202)
    out = {}
    A = torch.randn(4, 4, 4, generator=gen)
    B = torch.randn(4, 16, 8, generator=gen)
    ke = kronecker_product_einsum_batched(A, B)
    ks = torch.stack([kronecker_product_single(a, b) for a, b in zip(A, B)], 0)
    kb = kronecker_product(A, B)
    assert torch.allclose(ks, kb) and torch.allclose(ks, ke)      # the reference's own test
    out.update({"k0::A": A, "k0::B": B, "k0::kron": ke, "k0::H": ke.sum(0)})
    for ci, (n, kin, kout) in enumerate([(2, 5, 3), (3, 4, 4), (5, 8, 6), (8, 6, 4)], start=1):
        A = torch.randn(n, n, n, generator=gen)
        B = torch.randn(n, kin, kout, generator=gen)
        ke = kronecker_product_einsum_batched(A, B)
        out.update({f"k{ci}::A": A, f"k{ci}::B": B, f"k{ci}::kron": ke, f"k{ci}::H": ke.sum(0)})
    out["num_kron"] = 5
    for n in (1, 2, 3, 4, 5, 8):
        out[f"rule::{n}"] = torch.stack(get_multiplication_matrices(n, type="standard"), 0)
    Q = [torch.randn(16, 8, generator=gen) for _ in range(4)]
    rule = torch.stack(get_multiplication_matrices(4, type="standard"), 0)
    realp = RealP(*Q)
    ksum = kronecker_product(rule, torch.stack(Q, 0)).sum(0)
    assert torch.allclose(realp, ksum)                            # the reference's own test
    out.update({"quat::Q": torch.stack(Q, 0), "quat::RealP": realp})
    # corollary: PHMLinear(c_init=standard, n=4) has H == RealP(W_0..W_3)
    lin = PHMLinear(32, 64, phm_dim=4, c_init="standard")
    W = torch.randn(4, 8, 16, generator=gen)
    H = kronecker_product_einsum_batched(lin.phm_rule.data, W).sum(0)
    assert torch.allclose(H, RealP(W[0], W[1], W[2], W[3]))
    out.update({"quat::W": W, "quat::H": H})
    save("g2_kronecker_rules", **out)


def _to_batch(gb):
    return Batch(x=torch.from_numpy(gb.x), edge_index=torch.from_numpy(gb.edge_index),
                 edge_attr=torch.from_numpy(gb.edge_attr), batch=torch.from_numpy(gb.batch),
                 y=torch.from_numpy(gb.y), num_graphs=gb.num_graphs)


def g5_convs():
    """G5: PHMConv / PHMGINEConv (aggr add) forward + backward."""
    gen = torch.Generator().manual_seed(505)
    mol = synth.molecule_batch(6, seed=5, isolated_frac=0.1)
    ei_mol = torch.from_numpy(mol.edge_index)
    n_mol = mol.num_nodes
    ei_rag = torch.from_numpy(synth.ragged_edges(96, seed=6, hub_degree=150))
    cases = [  # (graph, n, h, mlp, same_dim, self_loops, msg_encoder, norm)
        ("mol", 4, 64, False, True, True, "identity", None),
        ("mol", 4, 64, False, False, True, "identity", None),
        ("mol", 4, 64, False, True, False, "relu", None),
        ("mol", 4, 64, True, True, True, "identity", "naive-batch-norm"),
        ("mol", 4, 64, True, True, False, "relu", "naive-batch-norm"),
        ("mol", 4, 64, True, True, True, "swish", None),
        ("rag", 4, 32, False, True, True, "identity", None),
        ("rag", 8, 64, True, True, True, "lrelu", "naive-batch-norm"),
        ("rag", 5, 40, False, False, True, "elu", None),
        ("mol", 2, 24, True, True, True, "identity", "naive-batch-norm"),
    ]
    out = {"num_cases": len(cases)}
    for ci, (g, n, h, mlp, same_dim, loops, enc, norm) in enumerate(cases):
        ei, N = (ei_mol, n_mol) if g == "mol" else (ei_rag, 96)
        conv = PHMMessagePassing(h, h, phm_dim=n, phm_rule=None, learn_phm=True, bias=True,
                                 add_self_loops=loops, norm=norm, activation="relu", w_init="phm",
                                 c_init="standard", aggr="add", mlp=mlp, same_dim=same_dim, msg_encoder=enc)
        fill_module(conv, gen)
        conv.train()
        sd_before = {k: v.clone() for k, v in conv.state_dict().items()}
        x = torch.randn(N, h, generator=gen).requires_grad_()
        ea = torch.randn(ei.shape[1], h, generator=gen).requires_grad_()
        y = conv(x, ei, ea)
        dy = torch.randn(N, h, generator=gen)
        params = [p for p in conv.parameters()]
        grads = torch.autograd.grad(y, [x, ea] + params, dy)
        d = dict(edge_index=ei, x=x, edge_attr=ea, y=y, dy=dy, dx=grads[0], dea=grads[1])
        d.update({"sd::" + k: v for k, v in sd_before.items()})
        d.update({"sd_after::" + k: v for k, v in conv.state_dict().items() if "running" in k})
        d.update({"grad::" + k: gv for (k, _), gv in zip(conv.named_parameters(), grads[2:])})
        d["meta"] = np.array([n, h, int(mlp), int(same_dim), int(loops), int(norm is not None)])
        d["msg_encoder"] = np.array(enc)
        out.update({f"c{ci}::{k}": v for k, v in d.items()})
    save("g5_convs", **out)


def g5b_softmax():
    """softmax aggregation variants (N2 row)."""
    gen = torch.Generator().manual_seed(515)
    ei = torch.from_numpy(synth.ragged_edges(64, seed=7, hub_degree=70))
    out = {}
    cases = [(4, 32, True, "identity"), (4, 32, False, "relu")]
    for ci, (n, h, mlp, enc) in enumerate(cases):
        conv = PHMMessagePassing(h, h, phm_dim=n, phm_rule=None, norm="naive-batch-norm" if mlp else None,
                                 aggr="softmax", mlp=mlp, msg_encoder=enc, initial_beta=1.0, learn_beta=True)
        fill_module(conv, gen)
        conv.train()
        sd_before = {k: v.clone() for k, v in conv.state_dict().items()}
        x = torch.randn(64, h, generator=gen).requires_grad_()
        ea = torch.randn(ei.shape[1], h, generator=gen).requires_grad_()
        y = conv(x, ei, ea)
        dy = torch.randn(64, h, generator=gen)
        grads = torch.autograd.grad(y, [x, ea] + list(conv.parameters()), dy)
        d = dict(edge_index=ei, x=x, edge_attr=ea, y=y, dy=dy, dx=grads[0], dea=grads[1])
        d.update({"sd::" + k: v for k, v in sd_before.items()})
        d.update({"grad::" + k: gv for (k, _), gv in zip(conv.named_parameters(), grads[2:])})
        d["meta"] = np.array([n, h, int(mlp)])
        d["msg_encoder"] = np.array(enc)
        out.update({f"c{ci}::{k}": v for k, v in d.items()})
    out["num_cases"] = len(cases)
    save("g5b_softmax_convs", **out)


def g6_norm():
    gen = torch.Generator().manual_seed(606)
    out = {}
    for ci, (n, h, N) in enumerate([(4, 64, 50), (8, 64, 33), (3, 30, 17)]):
        norm = PHMNorm(num_features=h, phm_dim=n, type="naive-batch-norm")
        fill_module(norm, gen)
        sd = {k: v.clone() for k, v in norm.state_dict().items()}
        x = torch.randn(N, h, generator=gen).requires_grad_()
        norm.train()
        y = norm(x)
        dy = torch.randn(N, h, generator=gen)
        grads = torch.autograd.grad(y, [x] + list(norm.parameters()), dy)
        sd_after = {k: v.clone() for k, v in norm.state_dict().items()}
        norm.eval()
        y_eval = norm(x)
        d = dict(x=x, y_train=y, dy=dy, dx=grads[0], y_eval=y_eval, meta=np.array([n, h, N]))
        d.update({"sd::" + k: v for k, v in sd.items()})
        d.update({"sd_after::" + k: v for k, v in sd_after.items()})
        d.update({"grad::" + k: gv for (k, _), gv in zip(norm.named_parameters(), grads[1:])})
        out.update({f"c{ci}::{k}": v for k, v in d.items()})
    out["num_cases"] = 3
    save("g6_norm", **out)


def g7_heads():
    gen = torch.Generator().manual_seed(707)
    mol = synth.molecule_batch(9, seed=9)
    batch = torch.from_numpy(mol.batch)
    N = mol.num_nodes
    out = {"batch": batch}
    n, h = 4, 64
    x = torch.randn(N, h, generator=gen).requires_grad_()
    pool = PHMSoftAttentionPooling(embed_dim=h, phm_dim=n, phm_rule=None)
    fill_module(pool, gen)
    y = pool(x, batch)
    dy = torch.randn(y.shape, generator=gen)
    grads = torch.autograd.grad(y, [x] + list(pool.parameters()), dy)
    out.update({"pool::x": x, "pool::y": y, "pool::dy": dy, "pool::dx": grads[0]})
    out.update({"pool::sd::" + k: v for k, v in pool.state_dict().items()})
    out.update({"pool::grad::" + k: gv for (k, _), gv in zip(pool.named_parameters(), grads[1:])})
    out["sumpool::y"] = PHMGlobalSumPooling(phm_dim=n)(x.detach(), batch)
    dn = PHMDownstreamNet(in_features=h, phm_dim=n, phm_rule=None, hidden_layers=[32, 16], out_features=3,
                          activation="relu", bias=True, norm="naive-batch-norm", w_init="phm",
                          c_init="standard", dropout=[0.0, 0.0])
    fill_module(dn, gen)
    dn.train()
    sd = {k: v.clone() for k, v in dn.state_dict().items()}
    xin = torch.randn(9, h, generator=gen).requires_grad_()
    yd = dn(xin)
    dyd = torch.randn(yd.shape, generator=gen)
    grads = torch.autograd.grad(yd, [xin] + list(dn.parameters()), dyd)
    out.update({"dn::x": xin, "dn::y": yd, "dn::dy": dyd, "dn::dx": grads[0]})
    out.update({"dn::sd::" + k: v for k, v in sd.items()})
    out.update({"dn::grad::" + k: gv for (k, _), gv in zip(dn.named_parameters(), grads[1:])})
    save("g7_heads", **out)


def _model_fixture(name, gb, seed, **kw):
    gen = torch.Generator().manual_seed(seed)
    model = PHMSkipConnectAdd(**kw)
    fill_module(model, gen)
    data = _to_batch(gb)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model.eval()
    with torch.no_grad():
        logits_eval = model(data)
    model.train()
    logits = model(data)
    y = data.y.to(torch.float)
    if kw.get("_regression"):
        loss = torch.nn.functional.l1_loss(logits, y)
    else:
        mask = ~torch.isnan(y)
        loss = torch.nn.functional.binary_cross_entropy_with_logits(logits[mask], y[mask])
    loss.backward()
    out = dict(x=data.x, edge_index=data.edge_index, edge_attr=data.edge_attr, batch=data.batch, y=data.y,
               logits_eval=logits_eval, logits_train=logits, loss=loss)
    out.update({"sd::" + k: v for k, v in sd.items()})
    out.update({"sd_after::" + k: v for k, v in model.state_dict().items() if "running" in k})
    out.update({"grad::" + k: p.grad for k, p in model.named_parameters() if p.grad is not None})
    out["num_params"] = model.get_number_of_params_()
    save(name, **out)


def g8_models():
    """G8: full PHMSkipConnectAdd — BASELINE config 1 (B=32) and two small variants."""
    common = dict(norm_mp="naive-batch-norm", norm_dn="naive-batch-norm", msg_aggr="sum",
                  pooling="softattention", activation="relu", real_trafo="linear", w_init="phm",
                  c_init="standard", learn_phm=True)
    _model_fixture("g8_model_cfg1", synth.molecule_batch(32, seed=0), 808,
                   phm_dim=4, atom_encoded_dim=128, mp_layers=[128] * 4, dropout_mpnn=[0.0] * 4, mlp=True,
                   downstream_layers=[128, 32], dropout_dn=[0.0, 0.0], target_dim=1, sc_type="first", **common)
    _model_fixture("g8_model_conv_small", synth.molecule_batch(8, seed=1, num_tasks=5, nan_frac=0.4), 818,
                   phm_dim=4, atom_encoded_dim=32, mp_layers=[32] * 2, dropout_mpnn=[0.0] * 2, mlp=False,
                   downstream_layers=[32, 16], dropout_dn=[0.0, 0.0], target_dim=5, sc_type="first", **common)
    gz = synth.molecule_batch(8, seed=2, dataset="zinc", regression=True)
    m = PHMSkipConnectAdd  # noqa: F841
    kw = dict(phm_dim=8, atom_input_dims=synth.ZINC_ATOM_DIMS, bond_input_dims=synth.ZINC_BOND_DIMS,
              atom_encoded_dim=64, mp_layers=[64] * 3, dropout_mpnn=[0.0] * 3, mlp=True,
              downstream_layers=[64, 32], dropout_dn=[0.0, 0.0], target_dim=1, sc_type="last", **common)
    # l1 loss for the ZINC-like regression variant
    gen = torch.Generator().manual_seed(828)
    model = PHMSkipConnectAdd(**kw)
    fill_module(model, gen)
    data = _to_batch(gz)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model.eval()
    with torch.no_grad():
        logits_eval = model(data)
    model.train()
    logits = model(data)
    loss = torch.nn.functional.l1_loss(logits, data.y)
    loss.backward()
    out = dict(x=data.x, edge_index=data.edge_index, edge_attr=data.edge_attr, batch=data.batch, y=data.y,
               logits_eval=logits_eval, logits_train=logits, loss=loss)
    out.update({"sd::" + k: v for k, v in sd.items()})
    out.update({"sd_after::" + k: v for k, v in model.state_dict().items() if "running" in k})
    out.update({"grad::" + k: p.grad for k, p in model.named_parameters() if p.grad is not None})
    out["num_params"] = model.get_number_of_params_()
    save("g8_model_zinc_small", **out)


def g8_model_softmax():
    """G8 with the reference's default / paper-best molhiv message aggregation (train_hiv.py:115-117):
    msg_aggr='softmax' with a learnable inverse temperature per layer."""
    common = dict(norm_mp="naive-batch-norm", norm_dn="naive-batch-norm", msg_aggr="softmax",
                  pooling="softattention", activation="relu", real_trafo="linear", w_init="phm",
                  c_init="standard", learn_phm=True, initial_beta=1.0, learn_beta=True)
    _model_fixture("g8_model_softmax_small", synth.molecule_batch(8, seed=3), 838,
                   phm_dim=4, atom_encoded_dim=64, mp_layers=[64] * 3, dropout_mpnn=[0.0] * 3, mlp=True,
                   downstream_layers=[64, 32], dropout_dn=[0.0, 0.0], target_dim=1, sc_type="first", **common)


def _relerr(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)


def _loss_of(logits, y, regression):
    y = y.to(logits.dtype)
    if regression:
        return torch.nn.functional.l1_loss(logits, y)
    mask = ~torch.isnan(y)
    return torch.nn.functional.binary_cross_entropy_with_logits(logits[mask], y[mask])


def _fullwidth_fixture(name, workload_batches, seed, kw, regression=False, store_grad32=True, adam_steps=0):
    """Full-width model fixture (BASELINE configs[1..3] widths): the reference in float32 (its own arithmetic) AND the
    same reference modules cast to float64 on the same parameters and batch.  The float64 run is the yardstick for
    the gradient tolerance: ``err32::<param>`` = max|ref32 - ref64| / max|ref64| is how far the REFERENCE's own fp32
    evaluation sits from the exact result; a GPU fp32 evaluation in a different summation order cannot be asked to sit
    closer to the reference than the reference sits to the truth.

    ``adam_steps`` > 0: additionally the reference loop body (train_hiv.py:173-201 with weight_decay 0: zero_grad,
    forward, masked BCE, backward, clip_grad_norm_(2.0), Adam(lr 1e-3) step) over the first ``adam_steps`` batches,
    float32 and float64; stored: the parameters afterwards and the per-step losses."""
    import copy
    gen = torch.Generator().manual_seed(seed)
    model = PHMSkipConnectAdd(**kw)
    fill_module(model, gen)
    gb = workload_batches[0]
    data = _to_batch(gb)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model64 = copy.deepcopy(model).double()

    model.eval()
    with torch.no_grad():
        logits_eval = model(data)
    model.train()
    logits = model(data)
    loss = _loss_of(logits, data.y, regression)
    loss.backward()

    model64.train()
    logits64 = model64(_to_batch(gb))
    assert logits64.dtype == torch.float64
    loss64 = _loss_of(logits64, data.y, regression)
    loss64.backward()

    out = dict(x=data.x, edge_index=data.edge_index, edge_attr=data.edge_attr, batch=data.batch, y=data.y,
               logits_eval=logits_eval, logits_train=logits, loss=loss,
               logits_train64=logits64.float(), loss64=loss64.float(),
               err32_logits=np.array(_relerr(logits, logits64)), err32_loss=np.array(_relerr(loss, loss64)))
    out.update({"sd::" + k: v for k, v in sd.items()})
    out.update({"sd_after::" + k: v for k, v in model.state_dict().items() if "running" in k})
    p64 = dict(model64.named_parameters())
    worst = (0.0, None)
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        if store_grad32:
            out["grad::" + k] = p.grad
        out["grad64::" + k] = p64[k].grad.float()
        e = _relerr(p.grad, p64[k].grad)
        out["err32::" + k] = np.array(e)
        if float(p64[k].grad.abs().max()) > 1e-7 and e > worst[0]:
            worst = (e, k)
    out["num_params"] = model.get_number_of_params_()
    print(f"  {name}: N={gb.num_nodes} E={gb.num_edges} params={out['num_params']} loss={float(loss):.6f} "
          f"ref32-vs-ref64: logits {float(out['err32_logits']):.2e} loss {float(out['err32_loss']):.2e} "
          f"worst grad {worst[0]:.2e} ({worst[1]})")

    if adam_steps > 0:
        finals, losses = {}, {}
        for tag, dt in (("32", torch.float32), ("64", torch.float64)):
            m = PHMSkipConnectAdd(**kw)
            m.load_state_dict(sd)
            m = m.to(dt).train()
            opt = torch.optim.Adam(m.parameters(), lr=1e-3)
            ls = []
            for gbi in workload_batches[:adam_steps]:
                d = _to_batch(gbi)
                opt.zero_grad()
                lo = _loss_of(m(d), d.y, regression)
                lo.backward()
                torch.nn.utils.clip_grad_norm_(m.parameters(), max_norm=2.0, norm_type=2)
                opt.step()
                ls.append(float(lo))
            finals[tag], losses[tag] = {k: v.clone() for k, v in m.state_dict().items()}, ls
        out["adam::steps"] = np.array(adam_steps)
        out["adam::batch_seeds"] = np.array([int(s) for s in workload_batches[adam_steps:]] or [0])
        out["adam::losses32"] = np.array(losses["32"], dtype=np.float64)
        out["adam::losses64"] = np.array(losses["64"], dtype=np.float64)
        for k, v in finals["64"].items():
            if v.is_floating_point():
                out["adam::sd64::" + k] = v.float()
                out["adam::err32::" + k] = np.array(_relerr(finals["32"][k], v))
    save(name, **out)


def g10_fullwidth():
    """Full-width fixtures for BASELINE configs[1], [2], [3] (model widths / depths / task counts of the configs; the
    batch is the config's for molhiv and a small one for the two 256-wide models to keep the files small)."""
    common = dict(norm_mp="naive-batch-norm", norm_dn="naive-batch-norm", msg_aggr="sum", pooling="softattention",
                  activation="relu", real_trafo="linear", w_init="phm", c_init="standard", learn_phm=True)
    # configs[1]: molhiv PHM-4, hidden 128, 4 layers, batch 512 (seeds 0, 1, 2 = the generator seeds of the Adam batches)
    hiv = [synth.molecule_batch(512, seed=s) for s in (0, 1, 2)]
    _fullwidth_fixture("g10_molhiv_b512", hiv, 1010,
                       dict(phm_dim=4, atom_encoded_dim=128, mp_layers=[128] * 4, dropout_mpnn=[0.0] * 4, mlp=True,
                            downstream_layers=[128, 32], dropout_dn=[0.0, 0.0], target_dim=1, sc_type="first", **common),
                       adam_steps=3)
    # the same model and batch with a smooth activation: no ReLU kink, so every fp32 evaluation order must agree with
    # the float64 run tensor by tensor (the tight check of the full-width pipelines)
    _fullwidth_fixture("g10_molhiv_b512_swish", hiv[:1], 1010,
                       dict(phm_dim=4, atom_encoded_dim=128, mp_layers=[128] * 4, dropout_mpnn=[0.0] * 4, mlp=True,
                            downstream_layers=[128, 32], dropout_dn=[0.0, 0.0], target_dim=1, sc_type="first",
                            **dict(common, activation="swish")), store_grad32=False)
    # configs[2]: ZINC PHM-8, hidden 256, 5 layers (batch 48)
    zinc = [synth.molecule_batch(48, seed=20, dataset="zinc", regression=True)]
    _fullwidth_fixture("g10_zinc_h256", zinc, 1020,
                       dict(phm_dim=8, atom_input_dims=synth.ZINC_ATOM_DIMS, bond_input_dims=synth.ZINC_BOND_DIMS,
                            atom_encoded_dim=256, mp_layers=[256] * 5, dropout_mpnn=[0.0] * 5, mlp=True,
                            downstream_layers=[256, 128], dropout_dn=[0.0, 0.0], target_dim=1, sc_type="last", **common),
                       regression=True)
    # configs[3]: molpcba PHM-4, hidden 256, 5 layers, 128 tasks with ~60 % unlabeled entries (batch 64)
    pcba = [synth.molecule_batch(64, seed=30, dataset="molpcba", num_tasks=128, nan_frac=0.6)]
    _fullwidth_fixture("g10_molpcba_h256", pcba, 1030,
                       dict(phm_dim=4, atom_encoded_dim=256, mp_layers=[256] * 5, dropout_mpnn=[0.0] * 5, mlp=True,
                            downstream_layers=[768, 256], dropout_dn=[0.0, 0.0], target_dim=128, sc_type="first", **common),
                       store_grad32=False)


def g10_fullwidth_swish256():
    """Smooth-activation twins of the two 256-wide fixtures (same seeds, parameters and batches as g10_zinc_h256 /
    g10_molpcba_h256, activation swish): no ReLU kink, so every gradient tensor of PHM-8 / k = 32 / hidden 256 and of
    PHM-4 / k = 64 / 128 NaN-masked tasks must agree with the float64 run to ~1e-5 on every execution path."""
    common = dict(norm_mp="naive-batch-norm", norm_dn="naive-batch-norm", msg_aggr="sum", pooling="softattention",
                  activation="swish", real_trafo="linear", w_init="phm", c_init="standard", learn_phm=True)
    zinc = [synth.molecule_batch(48, seed=20, dataset="zinc", regression=True)]
    _fullwidth_fixture("g10_zinc_h256_swish", zinc, 1020,
                       dict(phm_dim=8, atom_input_dims=synth.ZINC_ATOM_DIMS, bond_input_dims=synth.ZINC_BOND_DIMS,
                            atom_encoded_dim=256, mp_layers=[256] * 5, dropout_mpnn=[0.0] * 5, mlp=True,
                            downstream_layers=[256, 128], dropout_dn=[0.0, 0.0], target_dim=1, sc_type="last", **common),
                       regression=True, store_grad32=False)
    pcba = [synth.molecule_batch(64, seed=30, dataset="molpcba", num_tasks=128, nan_frac=0.6)]
    _fullwidth_fixture("g10_molpcba_h256_swish", pcba, 1030,
                       dict(phm_dim=4, atom_encoded_dim=256, mp_layers=[256] * 5, dropout_mpnn=[0.0] * 5, mlp=True,
                            downstream_layers=[768, 256], dropout_dn=[0.0, 0.0], target_dim=128, sc_type="first", **common),
                       store_grad32=False)


def g8_yardsticks():
    """Float64 yardsticks for the four G8 model fixtures (the G10 tests' allowance, tests/test_gpu_fullwidth.py): the SAME models
    (same seeds as g8_models / g8_model_softmax: the parameters are checked against the committed fixtures) run in float64 on the
    same batch; stored per fixture: grad64::<param> and err32::<param> = how far the reference's own float32 gradient sits from it."""
    import copy
    common = dict(norm_mp="naive-batch-norm", norm_dn="naive-batch-norm", msg_aggr="sum", pooling="softattention", activation="relu",
                  real_trafo="linear", w_init="phm", c_init="standard", learn_phm=True)
    cases = [
        ("g8_model_cfg1", synth.molecule_batch(32, seed=0), 808, False,
         dict(phm_dim=4, atom_encoded_dim=128, mp_layers=[128] * 4, dropout_mpnn=[0.0] * 4, mlp=True, downstream_layers=[128, 32],
              dropout_dn=[0.0, 0.0], target_dim=1, sc_type="first", **common)),
        ("g8_model_conv_small", synth.molecule_batch(8, seed=1, num_tasks=5, nan_frac=0.4), 818, False,
         dict(phm_dim=4, atom_encoded_dim=32, mp_layers=[32] * 2, dropout_mpnn=[0.0] * 2, mlp=False, downstream_layers=[32, 16],
              dropout_dn=[0.0, 0.0], target_dim=5, sc_type="first", **common)),
        ("g8_model_zinc_small", synth.molecule_batch(8, seed=2, dataset="zinc", regression=True), 828, True,
         dict(phm_dim=8, atom_input_dims=synth.ZINC_ATOM_DIMS, bond_input_dims=synth.ZINC_BOND_DIMS, atom_encoded_dim=64,
              mp_layers=[64] * 3, dropout_mpnn=[0.0] * 3, mlp=True, downstream_layers=[64, 32], dropout_dn=[0.0, 0.0], target_dim=1,
              sc_type="last", **common)),
        ("g8_model_softmax_small", synth.molecule_batch(8, seed=3), 838, False,
         dict(phm_dim=4, atom_encoded_dim=64, mp_layers=[64] * 3, dropout_mpnn=[0.0] * 3, mlp=True, downstream_layers=[64, 32],
              dropout_dn=[0.0, 0.0], target_dim=1, sc_type="first",
              **dict(common, msg_aggr="softmax", initial_beta=1.0, learn_beta=True))),
    ]
    for name, gb, seed, regression, kw in cases:
        gen = torch.Generator().manual_seed(seed)
        model = PHMSkipConnectAdd(**kw)
        fill_module(model, gen)
        have = np.load(os.path.join(HERE, name + ".npz"))
        for k, v in model.state_dict().items():
            assert np.array_equal(npy(v), have["sd::" + k]), (name, k)          # the very model of the committed fixture
        model64 = copy.deepcopy(model).double()
        out = {}
        grads = {}
        for tag, m in (("32", model), ("64", model64)):
            m.train()
            data = _to_batch(gb)
            logits = m(data)
            _loss_of(logits, data.y, regression).backward()
            grads[tag] = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
        worst = 0.0
        for k, g64 in grads["64"].items():
            out["grad64::" + k] = g64.float()
            e = _relerr(grads["32"][k], g64)
            out["err32::" + k] = np.array(e)
            if float(g64.abs().max()) >= 1e-7:
                worst = max(worst, e)
        print(f"  {name}: reference float32 vs float64 gradients: worst {worst:.2e}")
        save("g8y_" + name[3:], **out)


def g12_published():
    """The reference's PUBLISHED model shapes (benchmarks/README.md:45-91, benchmarks/run_script_zinc_phm4.sh:4-5), dropout 0
    (dropout has its own fixture, g9): widths whose per-component size is not a multiple of four and GEMM depths of 180 / 200.

      g12_hiv_phc4_2x200      ogbg-molhiv paper model: PHC-4, embed 200, 2 x 200 (k = 50), softmax aggregation with a learnable
                              inverse temperature, head 128, 32
      g12_zinc_phc5_4x200     ZINC 100 K model: PHC-5, 4 x 200 (k = 40), head 200, 130.  The README's command line does not construct:
                              --input_embed_dim 180 trips PHMSkipConnectAdd's width assertion (models.py:46) and --d_units
                              200,128 PHMLinear's divisibility assertion (layers.py:231); nearest shapes that do: 200, 130
      g12_zinc_phc4_4x180     run_script_zinc_phm4.sh: PHC-4, embed 180, 4 x 180 (k = 45), head 180, 80
    and a smooth-activation twin of each (no ReLU kink: the tight gradient check)."""
    common = dict(norm_mp="naive-batch-norm", norm_dn="naive-batch-norm", pooling="softattention", real_trafo="linear",
                  w_init="phm", c_init="standard", learn_phm=True)
    hiv = [synth.molecule_batch(64, seed=40)]
    zinc = [synth.molecule_batch(48, seed=41, dataset="zinc", regression=True)]
    zkw = dict(atom_input_dims=synth.ZINC_ATOM_DIMS, bond_input_dims=synth.ZINC_BOND_DIMS)
    for act in ("relu", "swish"):
        sfx = "" if act == "relu" else "_swish"
        _fullwidth_fixture("g12_hiv_phc4_2x200" + sfx, hiv, 1210,
                           dict(phm_dim=4, atom_encoded_dim=200, mp_layers=[200, 200], dropout_mpnn=[0.0, 0.0], mlp=True,
                                msg_aggr="softmax", initial_beta=1.0, learn_beta=True, downstream_layers=[128, 32],
                                dropout_dn=[0.0, 0.0], target_dim=1, sc_type="first", activation=act, **common),
                           store_grad32=False)
        _fullwidth_fixture("g12_zinc_phc5_4x200" + sfx, zinc, 1220,
                           dict(phm_dim=5, atom_encoded_dim=200, mp_layers=[200] * 4, dropout_mpnn=[0.0] * 4, mlp=True,
                                msg_aggr="sum", downstream_layers=[200, 130], dropout_dn=[0.0, 0.0], target_dim=1,
                                sc_type="last", activation=act, **common, **zkw), regression=True, store_grad32=False)
        _fullwidth_fixture("g12_zinc_phc4_4x180" + sfx, zinc, 1230,
                           dict(phm_dim=4, atom_encoded_dim=180, mp_layers=[180] * 4, dropout_mpnn=[0.0] * 4, mlp=True,
                                msg_aggr="sum", downstream_layers=[180, 80], dropout_dn=[0.0, 0.0], target_dim=1,
                                sc_type="last", activation=act, **common, **zkw), regression=True, store_grad32=False)


def g12_pcba_published():
    """The reference's published ogbg-molpcba model (benchmarks/run_script_pcba_phm4.sh:3-10, README.md:58-66) with dropout 0:
    PHC-4, embed 512, 7 x 512 with --mlp_mp False (PHMConv: ONE PHM layer per message-passing layer, messagepassing.py:55-74),
    head 768, 256 over 128 tasks with ~60 % unlabeled entries; a 24-graph batch keeps the file small (the parameters dominate it).
    ``g12_pcba_phc4_7x512_swish``: the smooth-activation twin."""
    common = dict(norm_mp="naive-batch-norm", norm_dn="naive-batch-norm", pooling="softattention", real_trafo="linear",
                  w_init="phm", c_init="standard", learn_phm=True)
    pcba = [synth.molecule_batch(24, seed=42, dataset="molpcba", num_tasks=128, nan_frac=0.6)]
    for act in ("relu", "swish"):
        sfx = "" if act == "relu" else "_swish"
        _fullwidth_fixture("g12_pcba_phc4_7x512" + sfx, pcba, 1240,
                           dict(phm_dim=4, atom_encoded_dim=512, mp_layers=[512] * 7, dropout_mpnn=[0.0] * 7, mlp=False,
                                msg_aggr="sum", downstream_layers=[768, 256], dropout_dn=[0.0, 0.0], target_dim=128,
                                sc_type="first", activation=act, **common), store_grad32=False)


def g11_regulariser():
    """phm_weight_regularization (regularization.py:15-23) value and gradients for p = 1, 2 on a stack of PHM layers, plus
    the regularised loop body of train() (train_hiv.py:173-201 with the script's defaults lr 1e-3, weightdecay 1e-3, p 2):
    total loss and the W gradients of the cfg-1 model on its 32-graph batch."""
    from phc.hypercomplex.regularization import phm_weight_regularization
    gen = torch.Generator().manual_seed(1111)
    out = {}

    class Stack(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.layers = torch.nn.ModuleList([PHMLinear(128, 128, phm_dim=4), PHMLinear(128, 32, phm_dim=4), PHMLinear(64, 64, phm_dim=8),
                                               PHMLinear(30, 45, phm_dim=3), PHMLinear(32, 4, phm_dim=4)])
    st = Stack()
    fill_module(st, gen)
    with torch.no_grad():
        st.layers[3].W[:, 2, 5] = 0.0                              # a zero column: the p = 2 subgradient there is 0
    for i, l in enumerate(st.layers):
        out[f"stack::W{i}"] = l.W.detach().clone()
    for p in (1, 2):
        for l in st.layers:
            l.W.grad = None
        reg = phm_weight_regularization(st, p=p)
        reg.backward()
        out[f"stack::reg_p{p}"] = reg.detach()
        for i, l in enumerate(st.layers):
            out[f"stack::dW{i}_p{p}"] = l.W.grad.clone()
    out["stack::num"] = np.array(len(st.layers))
    # the regularised training loss of the cfg-1 model
    z = np.load(os.path.join(HERE, "g8_model_cfg1.npz"))
    common = dict(norm_mp="naive-batch-norm", norm_dn="naive-batch-norm", msg_aggr="sum", pooling="softattention", activation="relu",
                  real_trafo="linear", w_init="phm", c_init="standard", learn_phm=True)
    model = PHMSkipConnectAdd(phm_dim=4, atom_encoded_dim=128, mp_layers=[128] * 4, dropout_mpnn=[0.0] * 4, mlp=True,
                              downstream_layers=[128, 32], dropout_dn=[0.0, 0.0], target_dim=1, sc_type="first", **common)
    model.load_state_dict({k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd::")})
    model.train()
    data = Batch(x=torch.from_numpy(z["x"]), edge_index=torch.from_numpy(z["edge_index"]), edge_attr=torch.from_numpy(z["edge_attr"]),
                 batch=torch.from_numpy(z["batch"]), y=torch.from_numpy(z["y"]), num_graphs=int(z["y"].shape[0]))
    logits = model(data)
    y = data.y.to(torch.float)
    mask = ~torch.isnan(y)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(logits[mask], y[mask])
    reg_loss = 1e-3 * 1e-3 * phm_weight_regularization(model, p=2)
    loss = loss + reg_loss
    loss.backward()
    out["model::loss"] = loss.detach()
    out["model::reg"] = reg_loss.detach()
    for k, prm in model.named_parameters():
        if k.endswith(".W"):
            out["model::grad::" + k] = prm.grad.clone()
    save("g11_regulariser", **out)


def g9_dropout():
    """G9: phm_dropout(same=True) mask structure (test_ops_equal_quaternion.py:64-103)."""
    torch.manual_seed(909)
    n, k, N, p = 4, 16, 40, 0.3
    x = torch.randn(N, n * k) + 3.0
    y = phm_dropout(x, phm_dim=n, p=p, training=True, same=True)
    kept = (y != 0).reshape(N, n, k)
    assert bool((kept == kept[:, :1]).all())
    ratio = (y[y != 0] / x[y != 0])
    assert torch.allclose(ratio, torch.full_like(ratio, 1 / (1 - p)))
    save("g9_dropout", x=x, y=y, meta=np.array([n, k, N]), p=np.array(p))


if __name__ == "__main__":
    makers = [g1_phm_linear, g2_g3_g4_kronecker_rules, g5_convs, g5b_softmax, g6_norm, g7_heads, g8_models,
              g8_model_softmax, g9_dropout, g10_fullwidth, g10_fullwidth_swish256, g11_regulariser, g12_published, g12_pcba_published, g8_yardsticks]
    wanted = sys.argv[1:]                       # e.g. `make_golden.py g8_model_softmax` rewrites only that fixture
    for make in makers:
        if not wanted or make.__name__ in wanted:
            make()
