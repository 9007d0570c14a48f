"""CPU-side checks: the C-ABI library loads and exports every declared symbol,
the drop-in package keeps the reference's API surface, and the host logic
(initialisers, helpers, dropout) behaves like the reference.  No GPU needed."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from tests._util import assert_close, load, tensors

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(REPO, "include", "phc_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(phc_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from phc_gnn_amd import lib
    handle = lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(handle, name), f"libphc_hip.so does not export {name}"
        assert name in lib.SIGNATURES, f"{name} has no ctypes signature"
    assert handle.phc_version() == lib.ABI_VERSION == 8
    assert handle.phc_error_string(-3).decode().startswith("workspace")


def test_workspace_queries_run_without_a_gpu():
    from phc_gnn_amd import lib
    h = lib.load()
    assert h.phc_phm_linear_bwd_ws_bytes(13000, 4, 32, 32) >= 4 * (128 * 128 + 128) * 2
    assert h.phc_csr_ws_bytes(1000, 100) >= 4 * 1100
    assert h.phc_phm_linear_bwd_ws_bytes(-1, 4, 32, 32) == 0


def test_argument_validation_without_a_gpu():
    from phc_gnn_amd import lib
    h = lib.load()
    assert h.phc_phm_linear_fwd(None, None, None, None, None, 5, 4, 8, 8, None, None) == -1
    assert b"null" in h.phc_last_error()
    assert h.phc_phm_linear_fwd(None, None, None, None, None, 5, 99, 8, 8, None, None) == -2
    assert h.phc_aggregate_fwd(None, None, None, None, None, 4, 16, 77, 0, None, None) == -1


def test_cpu_forward_is_refused_loudly():
    from phc.hypercomplex.layers import PHMLinear
    from phc_gnn_amd.lib import PhcError
    lin = PHMLinear(8, 8, phm_dim=2)
    with pytest.raises(PhcError, match="no CPU fallback"):
        lin(torch.randn(3, 8))


def test_phmlinear_api_surface():
    from phc.hypercomplex.layers import PHMLinear
    lin = PHMLinear(128, 256, phm_dim=4, c_init="standard")
    assert [k for k, _ in lin.named_parameters()] == ["phm_rule", "W", "b"]
    assert tuple(lin.W.shape) == (4, 32, 64) and tuple(lin.phm_rule.shape) == (4, 4, 4) and tuple(lin.b.shape) == (256,)
    assert sum(p.numel() for p in lin.parameters()) == 8512          # phc/README.md:28-30
    assert (lin.b[:64] == 0).all() and (lin.b[64:] == 0.2).all()
    assert PHMLinear(8, 8, 2, bias=False).b is None
    assert not PHMLinear(8, 8, 2, learn_phm=False).phm_rule.requires_grad
    with pytest.raises(AssertionError):
        PHMLinear(10, 8, phm_dim=4)
    z = load("g2_kronecker_rules")
    assert torch.equal(lin.phm_rule.data, torch.from_numpy(z["rule::4"]))


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 8])
def test_standard_rules_match_reference(n):
    from phc.hypercomplex.utils import get_multiplication_matrices
    z = load("g2_kronecker_rules")
    assert torch.equal(torch.stack(get_multiplication_matrices(n, "standard")), torch.from_numpy(z[f"rule::{n}"]))
    r = torch.stack(get_multiplication_matrices(n, "random"))
    assert r.shape == (n, n, n) and float(r.abs().max()) <= 1.0


def test_kronecker_helpers_match_reference():
    from phc.hypercomplex import kronecker as K
    z = load("g2_kronecker_rules")
    for i in range(int(z["num_kron"])):
        A, B = torch.from_numpy(z[f"k{i}::A"]), torch.from_numpy(z[f"k{i}::B"])
        ref = torch.from_numpy(z[f"k{i}::kron"])
        assert_close(K.kronecker_product_einsum_batched(A, B), ref, 1e-7)
        assert_close(K.kronecker_product(A, B), ref, 1e-7)
        assert_close(torch.stack([K.kronecker_product_single(a, b) for a, b in zip(A, B)]), ref, 1e-7)
        assert_close(torch.stack([K._kronecker_product(a, b) for a, b in zip(A, B)]), ref, 1e-7)


def test_model_state_dict_keys_match_reference_fixture():
    from phc.hypercomplex.undirectional.models import PHMSkipConnectAdd
    z = load("g8_model_cfg1")
    ref_keys = [k[4:] for k in z.files if k.startswith("sd::")]
    m = PHMSkipConnectAdd(phm_dim=4, atom_encoded_dim=128, mp_layers=[128] * 4, dropout_mpnn=[0.0] * 4, mlp=True,
                          msg_aggr="sum", downstream_layers=[128, 32], dropout_dn=[0.0, 0.0])
    assert list(m.state_dict().keys()) == ref_keys
    assert m.get_number_of_params_() == int(z["num_params"]) == 79401
    for attr in ("convs", "norms", "pooling", "downstream", "atomencoder", "bondencoders", "phm_dim"):
        assert hasattr(m, attr)
    from phc.quaternion.regularization import get_model_blocks
    covered = sum(len(b[0]["params"]) for b in (get_model_blocks(m, a, lr=1e-3) for a in
                  ("atomencoder", "bondencoders", "convs", "norms", "pooling", "downstream")) if b)
    assert covered == len(list(m.parameters()))            # benchmarks/train_hiv.py:279-282


def test_whole_module_pickle_roundtrip(tmp_path):
    from phc.hypercomplex.undirectional.models import PHMSkipConnectAdd
    m = PHMSkipConnectAdd(phm_dim=2, atom_encoded_dim=16, mp_layers=[16, 16], dropout_mpnn=[0.0, 0.0], mlp=True,
                          msg_aggr="sum", downstream_layers=[8], dropout_dn=[0.0])
    path = tmp_path / "model.pt"
    torch.save(m, path)                                      # benchmarks/train_hiv.py:344
    m2 = torch.load(path, weights_only=False)
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))


def test_regularizers_find_weights_by_name():
    from phc.hypercomplex.layers import PHMMLP
    from phc.hypercomplex.regularization import multiplication_rule_regularization, phm_weight_regularization
    mlp = PHMMLP(16, 16, phm_dim=4, phm_rule=None)
    w = phm_weight_regularization(mlp, p=2)
    expect = mlp.linear1.W.norm(p=2, dim=0).mean() + mlp.linear2.W.norm(p=2, dim=0).mean()
    assert torch.allclose(w, expect)
    r = multiplication_rule_regularization(mlp, p=1)
    assert torch.allclose(r, mlp.linear1.phm_rule.norm(p=1) + mlp.linear2.phm_rule.norm(p=1))


def test_phm_dropout_same_mask_structure():
    """reference test phc/hypercomplex/tests/test_ops_equal_quaternion.py:64-103 (G9)."""
    from phc.hypercomplex.layers import phm_dropout
    z = load("g9_dropout")
    n, k, N = [int(v) for v in z["meta"]]
    p = float(z["p"])
    x = torch.from_numpy(z["x"])
    torch.manual_seed(0)
    y = phm_dropout(x, phm_dim=n, p=p, training=True, same=True)
    kept = (y != 0).reshape(N, n, k)
    assert bool((kept == kept[:, :1]).all())
    assert torch.allclose(y[y != 0] / x[y != 0], torch.full_like(y[y != 0], 1 / (1 - p)))
    assert 0.5 < float(kept.float().mean()) < 0.9
    ref_kept = torch.from_numpy(z["y"] != 0).reshape(N, n, k)
    assert bool((ref_kept == ref_kept[:, :1]).all())
    assert phm_dropout(x, n, p=p, training=False) is x and phm_dropout(x, n, p=0.0) is x


def test_phm_init_statistics():
    from phc.hypercomplex.inits import phm_init, unitary_init
    np.random.seed(0)
    torch.manual_seed(0)
    n, kin, kout = 4, 64, 96
    w = phm_init(n, kin, kout, transpose=False)
    assert w.shape == (n, kin, kout)
    assert phm_init(n, kin, kout).shape == (n, kout, kin)
    # |w|^2 summed over components is chi^2(n) * s^2 with s^2 = 2 / (n (kin + kout))  => mean 2/(kin+kout)
    mag2 = (w ** 2).sum(0)
    assert abs(float(mag2.mean()) - 2.0 / (kin + kout)) < 0.1 * 2.0 / (kin + kout)
    v = unitary_init(n, 5, 7)
    assert torch.allclose(v.norm(dim=0), torch.ones(5, 7), atol=1e-6) and float(v[0].abs().max()) == 0.0


def test_synthetic_batches_are_seeded_and_shaped():
    from phc_gnn_amd import synth
    a, b = synth.molecule_batch(64, seed=3), synth.molecule_batch(64, seed=3)
    assert np.array_equal(a.edge_index, b.edge_index) and np.array_equal(a.x, b.x)
    assert a.edge_index.max() < a.num_nodes and (np.diff(a.batch) >= 0).all()
    assert (a.edge_index[0, 0::2] == a.edge_index[1, 1::2]).all()      # each bond = two directed edges
    assert 20 < a.num_nodes / 64 < 31 and 1.0 < a.num_edges / 2 / a.num_nodes < 1.2
    assert (a.x.max(0) < np.array(synth.OGB_ATOM_DIMS)).all()


def test_bucket_shape_prefers_whole_gemm_rounds_inside_its_slack():
    """graphed.bucket_shape: every batch fits with at least 128 padding nodes; when a whole number of 4 096-row GEMM rounds lies inside
    the slack window it is taken (24 200 nodes: 24 576 instead of 24 832), otherwise the window's upper end stays."""
    from types import SimpleNamespace as NS
    from phc_gnn_amd import graphed
    for n, want in ((24200, 24576), (24460, 25088), (13100, 13568), (27400, 28160), (500, 640)):
        n_max, e_max = graphed.bucket_shape([NS(num_nodes=n, num_edges=2 * n), NS(num_nodes=n - 50, num_edges=2 * n + 10)])
        assert n_max == want, (n, n_max)
        assert n_max >= n + 128 and n_max % 128 == 0
        assert e_max >= 2 * n + 10 and e_max % 128 == 0
