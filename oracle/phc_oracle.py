"""CPU ORACLE for the PHC message-passing path  --  TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement (plain PyTorch CPU ops, fp32 or fp64) of the
reference's algorithm for the hot path named in BASELINE.json / SURVEY.md §8.
It is the checker for the HIP kernels, never the product:

  * only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg
    of ``bench.py`` may import it;
  * nothing under ``phc/`` or ``phc_gnn_amd/`` imports it, and the product
    raises if its HIP library is missing rather than falling back to this.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the real
reference from /root/reference (in the build container only, with stand-ins for
the three uninstalled third-party packages) and dumps inputs/outputs to
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every function
below against those vectors, including the reference's own known-answer tests
(phc/hypercomplex/tests/test_kronecker_product.py:10-28,
test_realrepr_sumkronecker.py:14-33).  The arithmetic of torch-scatter 2.0.5 /
torch-geometric 1.6.1 (``scatter(reduce='add')``, ``MessagePassing.propagate``,
``global_add_pool``) is not in /root/reference; it is restated here from the
packages' documented behaviour and anchored on the
reference's call sites.

Every function cites the reference lines it follows (paths relative to
/root/reference).  The op sequence deliberately mirrors the reference's
(einsum-Kronecker -> sum -> matmul; index_select -> add -> scatter-add; one
batch-norm per component; one embedding lookup per component and feature) so
that timing this file on host cores is a faithful CPU baseline ("port").
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ----------------------------------------------------------------------------
# a1: Kronecker-sum weight  H = sum_i A_i (x) W_i
# ----------------------------------------------------------------------------
def kron_batched(A: Tensor, W: Tensor) -> Tensor:
    """phc/hypercomplex/kronecker.py:35-48 — batched Kronecker product,
    (b,a,c),(b,k,p) -> (b, a*k, c*p)."""
    b, a, c = A.shape
    _, k, p = W.shape
    return torch.einsum("bac,bkp->bakcp", A, W).reshape(b, a * k, c * p)


def kron_sum_H(A: Tensor, W: Tensor) -> Tensor:
    """phc/hypercomplex/layers.py:214 — H[a*k_in+κ, c*k_out+π] = Σ_i A[i,a,c]·W[i,κ,π]."""
    return kron_batched(A, W).sum(0)


# ----------------------------------------------------------------------------
# a2/a3: PHM affine map
# ----------------------------------------------------------------------------
def phm_linear(x: Tensor, A: Tensor, W: Tensor, b: Optional[Tensor]) -> Tensor:
    """phc/hypercomplex/layers.py:198-219 (matvec_product_new): y = x·H (+ b)."""
    y = torch.matmul(x, kron_sum_H(A, W))
    if b is not None:
        y = y + b
    return y


def phm_linear_backward(x: Tensor, A: Tensor, W: Tensor, dy: Tensor):
    """a4 — explicit gradients of y = x·H(A,W) + b (what autograd derives from
    layers.py:214-217).  Returns dx, dW, dA, db.  Used to cross-check the HIP
    backward independently of torch.autograd."""
    n, kin, kout = W.shape
    H = kron_sum_H(A, W)
    dx = dy @ H.t()
    dH = (x.t() @ dy).reshape(n, kin, n, kout)           # [a, κ, c, π]
    dW = torch.einsum("iac,akcp->ikp", A, dH)
    dA = torch.einsum("ikp,akcp->iac", W, dH)
    db = dy.sum(0)
    return dx, dW, dA, db


# ----------------------------------------------------------------------------
# activations (phc/quaternion/activations.py:134-147)
# ----------------------------------------------------------------------------
# ---- ReLU gates as data (tests only) -------------------------------------------------------------------------------------
# With ReLU a handful of pre-activations sit within an ulp of the kink; which side they fall on depends on the fp32 summation
# order of whoever evaluates them, and one flipped gate moves whole gradient tensors by 1e-3.  To compare an implementation's
# gradients TIGHTLY nonetheless, the oracle can (a) record the gates of its own run in call order and (b) be run with gates
# imposed from outside: relu(x) := x * gate, gate a constant 0/1 tensor — the exact gradient of the function "with these gates".
_GATES = {"impose": None, "record": None, "pre": None}


class relu_gates:
    """``with relu_gates(impose=[g0, g1, ...])``: the i-th relu call of the block multiplies by g_i instead of testing the sign.
    ``with relu_gates(record=True) as r``: r.gates / r.pre hold every relu call's (x > 0) and its input, in call order."""

    def __init__(self, impose=None, record=False):
        self.impose, self.want_record = (list(impose) if impose is not None else None), record
        self.gates, self.pre = [], []

    def __enter__(self):
        _GATES["impose"] = list(self.impose) if self.impose is not None else None
        _GATES["record"] = self.gates if self.want_record else None
        _GATES["pre"] = self.pre if self.want_record else None
        return self

    def __exit__(self, *exc):
        left = _GATES["impose"]
        _GATES["impose"] = _GATES["record"] = _GATES["pre"] = None
        if exc[0] is None and left:
            raise RuntimeError(f"relu_gates: {len(left)} imposed gates were not consumed")


def activation(name: str, x: Tensor) -> Tensor:
    name = name.lower()
    if name == "identity":
        return x
    if name == "relu":
        if _GATES["record"] is not None:
            _GATES["record"].append((x > 0).detach())
            _GATES["pre"].append(x.detach())
        if _GATES["impose"] is not None:
            g = _GATES["impose"].pop(0)
            if tuple(g.shape) != tuple(x.shape):
                raise RuntimeError(f"relu_gates: gate {tuple(g.shape)} for an input {tuple(x.shape)}")
            return x * g.to(x.dtype)
        return F.relu(x)
    if name == "lrelu":
        return F.leaky_relu(x)
    if name == "elu":
        return F.elu(x)
    if name == "selu":
        return F.selu(x)
    if name == "swish":                                   # activations.py:13-35
        return x * torch.sigmoid(x)
    raise ValueError(name)


# ----------------------------------------------------------------------------
# a6: message passing with sum aggregation
# ----------------------------------------------------------------------------
def scatter_sum_rows(msg: Tensor, index: Tensor, num_rows: int) -> Tensor:
    """torch_scatter.scatter(src, index, dim=-2, dim_size=N, reduce='add') as
    called from PyG's aggregate; rows nobody points at stay 0."""
    out = msg.new_zeros((num_rows, msg.shape[1]))
    return out.index_add_(0, index, msg)


def propagate_add(x: Tensor, edge_index: Tensor, edge_attr: Tensor, msg_encoder: str = "identity") -> Tensor:
    """phc/hypercomplex/undirectional/messagepassing.py:60,72-74 / :136,144-146
    through PyG 1.6.1 ``propagate`` (flow source_to_target):
    out[i] = Σ_{e: edge_index[1,e]=i} φ(x[edge_index[0,e]] + edge_attr[e])."""
    assert x.shape[-1] == edge_attr.shape[-1]            # messagepassing.py:73,145
    x_j = x.index_select(0, edge_index[0])
    msg = activation(msg_encoder, x_j + edge_attr)
    return scatter_sum_rows(msg, edge_index[1], x.shape[0])


def softmax_aggregate(x: Tensor, edge_index: Tensor, edge_attr: Tensor, beta: Tensor,
                      msg_encoder: str = "identity", eps: float = 1e-12) -> Tensor:
    """messagepassing.py:207-214 / :293-300 — per destination and per column,
    softmax(β·m) weighted sum of the messages m (torch-scatter 2.0.5
    scatter_softmax: max-shifted, sum + eps in the denominator)."""
    n = x.shape[0]
    dst = edge_index[1]
    m = activation(msg_encoder, x.index_select(0, edge_index[0]) + edge_attr)
    s = m * beta
    idx = dst.view(-1, 1).expand_as(s)
    mx = s.new_zeros((n, s.shape[1])).scatter_reduce(0, idx, s, reduce="amax", include_self=False)
    ex = (s - mx.index_select(0, dst)).exp()
    den = scatter_sum_rows(ex, dst, n)
    w = ex / (den.index_select(0, dst) + eps)
    return scatter_sum_rows(m * w, dst, n)


def batch_norm_naive(x: Tensor, phm_dim: int, weight: Sequence[Tensor], bias: Sequence[Tensor],
                     running_mean: Sequence[Tensor], running_var: Sequence[Tensor],
                     training: bool, momentum: float = 0.1, eps: float = 1e-5) -> Tensor:
    """phc/hypercomplex/norm.py:30-35 — reshape to [n, N, k], one BatchNorm1d(k)
    per component (running stats updated in place when training), concatenate."""
    k = x.shape[1] // phm_dim
    parts = x.reshape(x.shape[0], phm_dim, k).permute(1, 0, 2)
    outs = [F.batch_norm(parts[c], running_mean[c], running_var[c], weight[c], bias[c],
                         training, momentum, eps) for c in range(phm_dim)]
    return torch.cat(outs, dim=-1)


def phm_mlp(x: Tensor, p: Dict[str, Tensor], prefix: str, phm_dim: int, act: str,
            norm: bool, training: bool) -> Tensor:
    """phc/hypercomplex/layers.py:349-355 — linear1 -> (norm) -> act -> linear2."""
    x = phm_linear(x, p[prefix + "linear1.phm_rule"], p[prefix + "linear1.W"], p.get(prefix + "linear1.b"))
    if norm:
        x = _norm_from_state(x, p, prefix + "norm.", phm_dim, training)
    x = activation(act, x)
    return phm_linear(x, p[prefix + "linear2.phm_rule"], p[prefix + "linear2.W"], p.get(prefix + "linear2.b"))


def _propagate(x: Tensor, edge_index: Tensor, edge_attr: Tensor, msg_encoder: str, beta: Optional[Tensor]) -> Tensor:
    """Sum aggregation, or (``beta`` given) the *Softmax convs' aggregate, messagepassing.py:207-214 / :293-300."""
    if beta is None:
        return propagate_add(x, edge_index, edge_attr, msg_encoder)
    return softmax_aggregate(x, edge_index, edge_attr, beta, msg_encoder)


def phm_conv(x: Tensor, edge_index: Tensor, edge_attr: Tensor, A: Tensor, W: Tensor, b: Optional[Tensor],
             add_self_loops: bool = True, same_dim: bool = True, msg_encoder: str = "identity",
             beta: Optional[Tensor] = None) -> Tensor:
    """messagepassing.py:55-70 (PHMConv.forward); with ``beta``: PHMConvSoftmax.forward (:216-231), same body."""
    agg = _propagate(x, edge_index, edge_attr, msg_encoder, beta)
    if same_dim:
        y = phm_linear(agg, A, W, b)
        return y + x if add_self_loops else y
    if add_self_loops:
        agg = agg + x
    return phm_linear(agg, A, W, b)


def phm_gine_conv(x: Tensor, edge_index: Tensor, edge_attr: Tensor, p: Dict[str, Tensor], prefix: str,
                  phm_dim: int, act: str = "relu", norm: bool = True, training: bool = True,
                  add_self_loops: bool = True, msg_encoder: str = "identity",
                  beta: Optional[Tensor] = None) -> Tensor:
    """messagepassing.py:132-142 (PHMGINEConv.forward): PHMMLP(agg + x); with ``beta``: PHMGINEConvSoftmax.forward
    (:302-311), same body."""
    agg = _propagate(x, edge_index, edge_attr, msg_encoder, beta)
    if add_self_loops:
        agg = agg + x
    return phm_mlp(agg, p, prefix, phm_dim, act, norm, training)


# ----------------------------------------------------------------------------
# encoders (adjacent rows: a9 inputs, N1)
# ----------------------------------------------------------------------------
def integer_encoder(idx: Tensor, tables: Sequence[Tensor]) -> Tensor:
    """phc/quaternion/encoder.py:43-51 — sum over features f of table_f[idx[:, f]]."""
    if idx.dim() == 1:
        idx = idx.unsqueeze(1)
    out = 0
    for f in range(idx.shape[1]):
        out = out + F.embedding(idx[:, f], tables[f])
    return out


def phm_encoder(idx: Tensor, p: Dict[str, Tensor], prefix: str, phm_dim: int, num_feats: int) -> Tensor:
    """phc/hypercomplex/encoder.py:31-34 — n independent integer encoders stacked
    on dim 1 -> [*, n, k]; the model flattens to component-major [*, n*k]
    (models.py:225,239-240)."""
    comps = [integer_encoder(idx, [p[f"{prefix}encoders.{c}.embeddings.{f}.weight"] for f in range(num_feats)])
             for c in range(phm_dim)]
    return torch.stack(comps, dim=1)


# ----------------------------------------------------------------------------
# a10 / a11: read-out and downstream head
# ----------------------------------------------------------------------------
def global_add_pool(x: Tensor, batch: Tensor, num_graphs: Optional[int] = None) -> Tensor:
    """torch_geometric.nn.global_add_pool as used at phc/hypercomplex/pooling.py:14."""
    b = int(batch.max()) + 1 if num_graphs is None else num_graphs
    return x.new_zeros((b, x.shape[1])).index_add_(0, batch, x)


def soft_attention_pool(x: Tensor, batch: Tensor, A: Tensor, W: Tensor, b: Optional[Tensor],
                        lin_w: Tensor, lin_b: Optional[Tensor], phm_dim: int,
                        num_graphs: Optional[int] = None) -> Tensor:
    """phc/hypercomplex/pooling.py:57-66 with real_trafo='linear'
    (layers.py:399-414): gate = σ(Linear_{h->k}(PHMLinear(x))) broadcast over
    the n components, then per-graph sum."""
    h = x.shape[1]
    g = torch.sigmoid(F.linear(phm_linear(x, A, W, b), lin_w, lin_b))         # [N, k]
    gated = (x.view(x.shape[0], phm_dim, h // phm_dim) * g.unsqueeze(1)).reshape(x.shape[0], h)
    return global_add_pool(gated, batch, num_graphs)


def _norm_from_state(x: Tensor, p: Dict[str, Tensor], prefix: str, phm_dim: int, training: bool) -> Tensor:
    """PHMNorm(type='naive-batch-norm') parameters live at <prefix>bn.bn.<c>.* (norm.py:20,58)."""
    q = prefix + "bn.bn."
    return batch_norm_naive(
        x, phm_dim,
        [p[f"{q}{c}.weight"] for c in range(phm_dim)], [p[f"{q}{c}.bias"] for c in range(phm_dim)],
        [p[f"{q}{c}.running_mean"] for c in range(phm_dim)], [p[f"{q}{c}.running_var"] for c in range(phm_dim)],
        training)


def downstream_net(x: Tensor, p: Dict[str, Tensor], prefix: str, phm_dim: int, num_hidden: int,
                   act: str, norm: bool, training: bool) -> Tensor:
    """phc/hypercomplex/downstream.py:90-120 with dropout 0 and real_trafo='linear'."""
    for i in range(num_hidden + 1):
        q = f"{prefix}affine.{i}."
        x = phm_linear(x, p[q + "phm_rule"], p[q + "W"], p.get(q + "b"))
        if i < num_hidden:
            if norm:
                x = _norm_from_state(x, p, f"{prefix}norm.{i}.", phm_dim, training)
            x = activation(act, x)
    return F.linear(x, p[prefix + "real_trafo.affine.weight"], p.get(prefix + "real_trafo.affine.bias"))


# ----------------------------------------------------------------------------
# a9: PHMSkipConnectAdd forward (models.py:219-249), dropout 0
# ----------------------------------------------------------------------------
def skip_connect_add_forward(p: Dict[str, Tensor], x_idx: Tensor, edge_index: Tensor, edge_attr_idx: Tensor,
                             batch: Tensor, *, phm_dim: int, num_layers: int, num_atom_feats: int,
                             num_bond_feats: int, num_dn_hidden: int, mlp: bool = True,
                             act: str = "relu", norm_mp: bool = True, norm_dn: bool = True,
                             pooling: str = "softattention", sc_type: str = "first",
                             msg_encoder: str = "identity", add_self_loops: bool = True,
                             training: bool = True, num_graphs: Optional[int] = None,
                             msg_aggr: str = "sum") -> Tensor:
    """State-dict driven restatement of PHMSkipConnectAdd.forward for
    msg_aggr in {'sum', 'softmax'}, dropout 0.  ``p`` uses the reference's parameter names (the softmax convs keep
    their inverse temperature at ``convs.<l>.transform.beta``, messagepassing.py:197,283)."""
    assert msg_aggr in ("sum", "add", "softmax")
    n_nodes = x_idx.shape[0]
    atom = phm_encoder(x_idx, p, "atomencoder.", phm_dim, num_atom_feats).reshape(n_nodes, -1)   # :223-225
    x = atom
    for l in range(num_layers):
        skip = atom if (l == 0 or sc_type == "first") else x                                     # :227-236
        ea = phm_encoder(edge_attr_idx, p, f"bondencoders.{l}.", phm_dim, num_bond_feats)
        ea = ea.reshape(ea.shape[0], -1)                                                          # :238-240
        beta = p[f"convs.{l}.transform.beta"] if msg_aggr == "softmax" else None
        if mlp:
            y = phm_gine_conv(x, edge_index, ea, p, f"convs.{l}.transform.transform.", phm_dim, act,
                              norm_mp, training, add_self_loops, msg_encoder, beta)
        else:
            q = f"convs.{l}.transform.transform."
            y = phm_conv(x, edge_index, ea, p[q + "phm_rule"], p[q + "W"], p.get(q + "b"),
                         add_self_loops, True, msg_encoder, beta)
        if norm_mp:
            y = _norm_from_state(y, p, f"norms.{l}.", phm_dim, training)                         # :208-209
        x = activation(act, y) + skip                                                             # :211-215
    if pooling == "softattention":
        out = soft_attention_pool(x, batch, p["pooling.linear.phm_rule"], p["pooling.linear.W"],
                                  p.get("pooling.linear.b"), p["pooling.real_trafo.affine.weight"],
                                  p.get("pooling.real_trafo.affine.bias"), phm_dim, num_graphs)
    else:
        out = global_add_pool(x, batch, num_graphs)
    return downstream_net(out, p, "downstream.", phm_dim, num_dn_hidden, act, norm_dn, training)


def bce_loss_masked(logits: Tensor, y: Tensor) -> Tensor:
    """benchmarks/train_hiv.py:174-178 — NaN-masked BCE-with-logits."""
    mask = ~torch.isnan(y)
    return F.binary_cross_entropy_with_logits(logits[mask], y[mask])


# ----------------------------------------------------------------------------
# contribution-matrix initialiser (known-answer source)
# ----------------------------------------------------------------------------
def standard_rule(n: int) -> Tensor:
    """phc/hypercomplex/utils.py:61-78 — n=2 complex, n=4 quaternion (utils.py:5-22),
    otherwise A_0 = I and A_i = diag(+1,-1,+1,...)·Pⁱ with P the right-shift
    permutation (np.roll(eye, 1, axis=1))."""
    if n == 2:
        return torch.tensor([[[1., 0.], [0., 1.]], [[0., -1.], [1., 0.]]])
    if n == 4:
        return torch.tensor([
            [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]],
            [[0, -1, 0, 0], [1, 0, 0, 0], [0, 0, 0, -1], [0, 0, 1, 0]],
            [[0, 0, -1, 0], [0, 0, 0, 1], [1, 0, 0, 0], [0, -1, 0, 0]],
            [[0, 0, 0, -1], [0, 0, -1, 0], [0, 1, 0, 0], [1, 0, 0, 0]]], dtype=torch.float32)
    P = torch.roll(torch.eye(n), shifts=1, dims=1)
    sign = torch.diag(torch.tensor([1.0 if j % 2 == 0 else -1.0 for j in range(n)]))
    mats = [torch.eye(n)]
    for i in range(1, n):
        mats.append(sign @ torch.linalg.matrix_power(P, i))
    return torch.stack(mats, 0)


def quaternion_real_repr(Wr: Tensor, Wi: Tensor, Wj: Tensor, Wk: Tensor) -> Tensor:
    """phc/quaternion/qr.py:10-25 (RealP) — the known answer for the standard
    n=4 rule: Σ_i A_i ⊗ W_i == RealP(W_0..W_3)
    (phc/hypercomplex/tests/test_realrepr_sumkronecker.py:14-33)."""
    return torch.cat([
        torch.cat([Wr, -Wi, -Wj, -Wk], 1),
        torch.cat([Wi, Wr, -Wk, Wj], 1),
        torch.cat([Wj, Wk, Wr, -Wi], 1),
        torch.cat([Wk, -Wj, Wi, Wr], 1)], 0)


# ----------------------------------------------------------------------------
# explicit weight regulariser of the training scripts
# ----------------------------------------------------------------------------
def phm_weight_regularization(weights: Sequence[Tensor], p: int = 2) -> Tensor:
    """phc/hypercomplex/regularization.py:15-23 — sum over the PHM weights W [n, k_in, k_out] of
    ``W.norm(p, dim=0).mean()`` (added to the loss as lr * weight_decay * this, benchmarks/train_hiv.py:181-196)."""
    reg = torch.zeros((), dtype=weights[0].dtype)
    for w in weights:
        reg = reg + w.norm(p=p, dim=0).mean()
    return reg
