"""Pin the CPU oracle against the golden vectors produced by the real reference
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import phc_oracle as O
from tests._util import assert_close, case, load, sub, tensors

TOL_ORACLE = 5e-6      # same fp32 ATen kernels on both sides; only op grouping differs


def test_g2_kronecker_and_H():
    z = load("g2_kronecker_rules")
    for i in range(int(z["num_kron"])):
        A, B = torch.from_numpy(z[f"k{i}::A"]), torch.from_numpy(z[f"k{i}::B"])
        assert_close(O.kron_batched(A, B), z[f"k{i}::kron"], 1e-7, f"kron {i}")
        assert_close(O.kron_sum_H(A, B), z[f"k{i}::H"], 1e-6, f"H {i}")


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 8])
def test_g3_standard_rules(n):
    z = load("g2_kronecker_rules")
    assert torch.equal(O.standard_rule(n), torch.from_numpy(z[f"rule::{n}"]))


def test_g4_quaternion_known_answer():
    z = load("g2_kronecker_rules")
    Q = torch.from_numpy(z["quat::Q"])
    assert torch.equal(O.quaternion_real_repr(*Q), torch.from_numpy(z["quat::RealP"]))
    assert_close(O.kron_sum_H(O.standard_rule(4), Q), z["quat::RealP"], 1e-7, "Σ A_i⊗Q_i == RealP")
    W = torch.from_numpy(z["quat::W"])
    assert_close(O.kron_sum_H(O.standard_rule(4), W), z["quat::H"], 1e-7)
    assert torch.equal(O.quaternion_real_repr(*W), torch.from_numpy(z["quat::H"]))


def test_g1_phm_linear_fwd_bwd():
    z = load("g1_phm_linear")
    for ci in range(int(z["num_cases"])):
        c = case(z, ci)
        b = c.get("b")
        x = c["x"].clone().requires_grad_()
        A = c["A"].clone().requires_grad_()
        W = c["W"].clone().requires_grad_()
        bb = b.clone().requires_grad_() if b is not None else None
        y = O.phm_linear(x, A, W, bb)
        assert_close(y, c["y"], TOL_ORACLE, f"case {ci} y")
        grads = torch.autograd.grad(y, [x, W, A] + ([bb] if b is not None else []), c["dy"])
        ex = O.phm_linear_backward(c["x"], c["A"], c["W"], c["dy"])
        for name, g_auto, g_explicit in zip(("dx", "dW", "dA", "db"), grads, ex):
            assert_close(g_auto, c[name], TOL_ORACLE, f"case {ci} {name} (autograd)")
            assert_close(g_explicit, c[name], TOL_ORACLE, f"case {ci} {name} (explicit formula)")


def _conv_forward(c, p, training=True):
    n, h, mlp, same_dim, loops, norm = [int(v) for v in c["meta"]]
    enc = str(c["msg_encoder"])
    if mlp:
        return O.phm_gine_conv(c["x_in"], c["edge_index"], c["ea_in"], p, "transform.transform.", n, "relu",
                               bool(norm), training, bool(loops), enc)
    q = "transform.transform."
    return O.phm_conv(c["x_in"], c["edge_index"], c["ea_in"], p[q + "phm_rule"], p[q + "W"], p.get(q + "b"),
                      bool(loops), bool(same_dim), enc)


def test_g5_convs_fwd_bwd():
    z = load("g5_convs")
    for ci in range(int(z["num_cases"])):
        c = case(z, ci)
        p = {k: v.clone() for k, v in sub(c, "sd::").items()}
        leaves = {k: v.requires_grad_() for k, v in p.items() if v.is_floating_point() and "running" not in k}
        c["x_in"] = c["x"].clone().requires_grad_()
        c["ea_in"] = c["edge_attr"].clone().requires_grad_()
        y = _conv_forward(c, p)
        assert_close(y, c["y"], TOL_ORACLE, f"case {ci} y")
        names = list(sub(c, "grad::").keys())
        grads = torch.autograd.grad(y, [c["x_in"], c["ea_in"]] + [leaves[k] for k in names], c["dy"])
        assert_close(grads[0], c["dx"], TOL_ORACLE, f"case {ci} dx")
        assert_close(grads[1], c["dea"], TOL_ORACLE, f"case {ci} dea")
        for k, g in zip(names, grads[2:]):
            assert_close(g, c["grad::" + k], 2e-5, f"case {ci} grad {k}")
        for k, v in sub(c, "sd_after::").items():
            assert_close(p[k], v, TOL_ORACLE, f"case {ci} {k}")


def test_g5b_softmax_convs():
    z = load("g5b_softmax_convs")
    for ci in range(int(z["num_cases"])):
        c = case(z, ci)
        n, h, mlp = [int(v) for v in c["meta"]]
        enc = str(c["msg_encoder"])
        p = sub(c, "sd::")
        agg = O.softmax_aggregate(c["x"], c["edge_index"], c["edge_attr"], p["transform.beta"], enc)
        if mlp:
            y = O.phm_mlp(agg + c["x"], {k: v.clone() for k, v in p.items()}, "transform.transform.", n, "relu",
                          True, True)
        else:
            q = "transform.transform."
            y = O.phm_linear(agg, p[q + "phm_rule"], p[q + "W"], p.get(q + "b")) + c["x"]
        assert_close(y, c["y"], TOL_ORACLE, f"softmax case {ci}")


def test_g6_norm():
    z = load("g6_norm")
    for ci in range(int(z["num_cases"])):
        c = case(z, ci)
        n, h, N = [int(v) for v in c["meta"]]
        p = {k: v.clone() for k, v in sub(c, "sd::").items()}
        x = c["x"].clone().requires_grad_()
        w = [p[f"bn.bn.{i}.weight"].requires_grad_() for i in range(n)]
        b = [p[f"bn.bn.{i}.bias"].requires_grad_() for i in range(n)]
        y = O._norm_from_state(x, p, "", n, True)
        assert_close(y, c["y_train"], TOL_ORACLE)
        g = torch.autograd.grad(y, [x] + w + b, c["dy"])
        assert_close(g[0], c["dx"], TOL_ORACLE)
        for i in range(n):
            assert_close(g[1 + i], c[f"grad::bn.bn.{i}.weight"], TOL_ORACLE)
            assert_close(g[1 + n + i], c[f"grad::bn.bn.{i}.bias"], TOL_ORACLE)
            assert_close(p[f"bn.bn.{i}.running_mean"], c[f"sd_after::bn.bn.{i}.running_mean"], TOL_ORACLE)
            assert_close(p[f"bn.bn.{i}.running_var"], c[f"sd_after::bn.bn.{i}.running_var"], TOL_ORACLE)
        assert_close(O._norm_from_state(c["x"], p, "", n, False), c["y_eval"], TOL_ORACLE)


def test_g7_heads():
    z = tensors(load("g7_heads"))
    p = sub(z, "pool::sd::")
    y = O.soft_attention_pool(z["pool::x"], z["batch"], p["linear.phm_rule"], p["linear.W"], p["linear.b"],
                              p["real_trafo.affine.weight"], p["real_trafo.affine.bias"], 4)
    assert_close(y, z["pool::y"], TOL_ORACLE, "soft attention pooling")
    assert_close(O.global_add_pool(z["pool::x"], z["batch"]), z["sumpool::y"], TOL_ORACLE, "global sum pooling")
    pd = {k: v.clone() for k, v in sub(z, "dn::sd::").items()}
    yd = O.downstream_net(z["dn::x"], pd, "", 4, 2, "relu", True, True)
    assert_close(yd, z["dn::y"], TOL_ORACLE, "downstream")


MODEL_CFGS = {
    "g8_model_cfg1": dict(phm_dim=4, num_layers=4, num_atom_feats=9, num_bond_feats=3, num_dn_hidden=2, mlp=True,
                          sc_type="first"),
    "g8_model_conv_small": dict(phm_dim=4, num_layers=2, num_atom_feats=9, num_bond_feats=3, num_dn_hidden=2,
                                mlp=False, sc_type="first"),
    "g8_model_zinc_small": dict(phm_dim=8, num_layers=3, num_atom_feats=1, num_bond_feats=1, num_dn_hidden=2,
                                mlp=True, sc_type="last"),
    "g8_model_softmax_small": dict(phm_dim=4, num_layers=3, num_atom_feats=9, num_bond_feats=3, num_dn_hidden=2,
                                   mlp=True, sc_type="first", msg_aggr="softmax"),
}


@pytest.mark.parametrize("name", list(MODEL_CFGS))
def test_g8_full_model(name):
    z = tensors(load(name))
    cfg = MODEL_CFGS[name]
    p = {k: v.clone() for k, v in sub(z, "sd::").items()}
    args = (z["x"], z["edge_index"], z["edge_attr"], z["batch"])
    with torch.no_grad():
        le = O.skip_connect_add_forward({k: v.clone() for k, v in p.items()}, *args, training=False, **cfg)
    assert_close(le, z["logits_eval"], TOL_ORACLE, "eval logits")
    gnames = list(sub(z, "grad::").keys())
    for k in gnames:
        p[k].requires_grad_()
    lt = O.skip_connect_add_forward(p, *args, training=True, **cfg)
    assert_close(lt, z["logits_train"], TOL_ORACLE, "train logits")
    if name == "g8_model_zinc_small":
        loss = torch.nn.functional.l1_loss(lt, z["y"])
    else:
        loss = O.bce_loss_masked(lt, z["y"])
    assert_close(loss, torch.as_tensor(z["loss"]), TOL_ORACLE, "loss")
    grads = torch.autograd.grad(loss, [p[k] for k in gnames])
    worst = max(float((g - z["grad::" + k]).abs().max()) / max(float(z["grad::" + k].abs().max()), 1e-30)
                for k, g in zip(gnames, grads) if float(z["grad::" + k].abs().max()) > 1e-7)
    assert worst < 5e-4, worst      # fp32 autograd through 4 BN layers: order-of-ops noise
    for k, v in sub(z, "sd_after::").items():
        assert_close(p[k], v, TOL_ORACLE, k)


FULLWIDTH_CFGS = {
    "g10_molhiv_b512": dict(phm_dim=4, num_layers=4, num_atom_feats=9, num_bond_feats=3, num_dn_hidden=2, mlp=True,
                            sc_type="first"),
    "g10_molhiv_b512_swish": dict(phm_dim=4, num_layers=4, num_atom_feats=9, num_bond_feats=3, num_dn_hidden=2, mlp=True,
                                  sc_type="first", act="swish"),
    "g10_zinc_h256": dict(phm_dim=8, num_layers=5, num_atom_feats=1, num_bond_feats=1, num_dn_hidden=2, mlp=True,
                          sc_type="last"),
    "g10_molpcba_h256": dict(phm_dim=4, num_layers=5, num_atom_feats=9, num_bond_feats=3, num_dn_hidden=2, mlp=True,
                             sc_type="first"),
    "g10_zinc_h256_swish": dict(phm_dim=8, num_layers=5, num_atom_feats=1, num_bond_feats=1, num_dn_hidden=2, mlp=True,
                                sc_type="last", act="swish"),
    "g10_molpcba_h256_swish": dict(phm_dim=4, num_layers=5, num_atom_feats=9, num_bond_feats=3, num_dn_hidden=2, mlp=True,
                                   sc_type="first", act="swish"),
}
# the reference's published shapes (benchmarks/README.md:45-91, run_script_zinc_phm4.sh:4-5; make_golden.py::g12_published):
# component sizes 50 / 40 / 45, GEMM depths 200 / 180, the softmax aggregation of the molhiv paper model
for _sfx, _act in (("", "relu"), ("_swish", "swish")):
    FULLWIDTH_CFGS["g12_hiv_phc4_2x200" + _sfx] = dict(phm_dim=4, num_layers=2, num_atom_feats=9, num_bond_feats=3, num_dn_hidden=2,
                                                      mlp=True, sc_type="first", msg_aggr="softmax", act=_act)
    FULLWIDTH_CFGS["g12_zinc_phc5_4x200" + _sfx] = dict(phm_dim=5, num_layers=4, num_atom_feats=1, num_bond_feats=1, num_dn_hidden=2,
                                                       mlp=True, sc_type="last", act=_act)
    FULLWIDTH_CFGS["g12_zinc_phc4_4x180" + _sfx] = dict(phm_dim=4, num_layers=4, num_atom_feats=1, num_bond_feats=1, num_dn_hidden=2,
                                                       mlp=True, sc_type="last", act=_act)
    # the published ogbg-molpcba model (run_script_pcba_phm4.sh:3-10; make_golden.py::g12_pcba_published): 7 x 512, PHMConv
    FULLWIDTH_CFGS["g12_pcba_phc4_7x512" + _sfx] = dict(phm_dim=4, num_layers=7, num_atom_feats=9, num_bond_feats=3, num_dn_hidden=2,
                                                       mlp=False, sc_type="first", act=_act)
L1_FIXTURES = ("g10_zinc_h256", "g12_zinc")


@pytest.mark.parametrize("name", list(FULLWIDTH_CFGS))
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_g10_fullwidth_models(name, dtype):
    """The oracle at the BASELINE configs' full widths (configs[1] at its own batch of 512) against the reference's
    float32 AND float64 runs.  The float64 oracle must reproduce the float64 reference to ~1e-9 (the restatement is the
    same function); the float32 oracle sits as far from it as the reference's own float32 run does (``err32::*``)."""
    z = tensors(load(name))
    cfg = FULLWIDTH_CFGS[name]
    p = {k: (v.clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in sub(z, "sd::").items()}
    gnames = list(sub(z, "grad64::").keys())
    for k in gnames:
        p[k].requires_grad_()
    lt = O.skip_connect_add_forward(p, z["x"], z["edge_index"], z["edge_attr"], z["batch"], training=True, **cfg)
    assert lt.dtype == dtype
    y = z["y"].to(dtype)
    loss = torch.nn.functional.l1_loss(lt, y) if name.startswith(L1_FIXTURES) else O.bce_loss_masked(lt, y)
    grads = torch.autograd.grad(loss, [p[k] for k in gnames])
    exact = dtype == torch.float64
    # gradient yardstick: the largest float32-vs-float64 distance of any of the reference's own gradient tensors (the
    # ReLU kink makes WHICH tensors deviate a matter of summation order; see tests/test_gpu_fullwidth.py)
    worst32 = max(float(z["err32::" + k]) for k in gnames if float(z["grad64::" + k].abs().max()) >= 1e-7)
    # float64 results are stored rounded to float32: 6e-8 relative
    assert_close(lt, z["logits_train64"], 2e-7 if exact else max(TOL_ORACLE, 3 * float(z["err32_logits"])), "logits vs float64")
    if not exact:
        assert_close(lt, z["logits_train"], TOL_ORACLE, "train logits")
        assert_close(loss, torch.as_tensor(z["loss"]), TOL_ORACLE, "loss")
    for k, g in zip(gnames, grads):
        ref = z["grad64::" + k]
        if float(ref.abs().max()) < 1e-7:
            continue
        allow = 1e-6 if exact else max(1e-5, 3.0 * worst32)
        assert_close(g, ref, allow, f"{name} {dtype} grad {k}")


def test_g11_weight_regulariser():
    z = tensors(load("g11_regulariser"))
    k = int(z["stack::num"])
    for p in (1, 2):
        ws = [z[f"stack::W{i}"].clone().requires_grad_() for i in range(k)]
        reg = O.phm_weight_regularization(ws, p=p)
        assert_close(reg, z[f"stack::reg_p{p}"], TOL_ORACLE, f"reg p={p}")
        grads = torch.autograd.grad(reg, ws)
        for i, g in enumerate(grads):
            assert_close(g, z[f"stack::dW{i}_p{p}"], TOL_ORACLE, f"dW{i} p={p}")
