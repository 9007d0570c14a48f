"""Seeded synthetic inputs for the PHC message-passing path.

The reference trains on ogbg-molhiv / ogbg-molpcba / ZINC through PyG data
loaders (benchmarks/train_hiv.py:457-493); none of those files exist offline
(benchmarks/dataset/*/ holds only .gitkeep), so every test and bench.py run
uses the generators below (SURVEY.md §8d "Synthetic inputs").

A *batch* is what PyG's collation hands to ``PHMSkipConnectAdd.forward``
(models.py:219-221): ``x`` int64 [N, F_atom], ``edge_index`` int64 [2, E]
(row 0 = source j, row 1 = target i, NOT sorted by target), ``edge_attr``
int64 [E, F_bond], ``batch`` int64 [N] (sorted graph id per node), ``y``.
Everything here is numpy on the host; callers move it to HBM.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import numpy as np

# ogb 1.2.4 vocabularies (ogb.utils.features; consumed at models.py:18-21).
OGB_ATOM_DIMS: List[int] = [119, 4, 12, 12, 10, 6, 6, 2, 2]
OGB_BOND_DIMS: List[int] = [5, 6, 2]
# ZINC-12k vocabularies (benchmarks/train_zinc.py:513-514).
ZINC_ATOM_DIMS: List[int] = [28]
ZINC_BOND_DIMS: List[int] = [4]

# public dataset means: (atoms per graph, bonds per atom)
DATASET_STATS = {
    "molhiv": (25.5, 27.5 / 25.5),
    "molpcba": (26.0, 28.1 / 26.0),
    "zinc": (23.2, 24.9 / 23.2),
}


@dataclass
class GraphBatch:
    """Collated mini-batch of small graphs (host arrays)."""
    x: np.ndarray            # int64 [N, F_atom]
    edge_index: np.ndarray   # int64 [2, E]
    edge_attr: np.ndarray    # int64 [E, F_bond]
    batch: np.ndarray        # int64 [N]
    y: np.ndarray            # float32 [B, T] (NaN = unlabeled task)
    num_graphs: int

    @property
    def num_nodes(self) -> int:
        return int(self.x.shape[0])

    @property
    def num_edges(self) -> int:
        return int(self.edge_index.shape[1])


def _graph_sizes(rng: np.random.Generator, num_graphs: int, mean_nodes: float,
                 sigma: float = 0.45, lo: int = 2, hi: int = 222) -> np.ndarray:
    mu = np.log(mean_nodes) - 0.5 * sigma * sigma
    n = np.rint(rng.lognormal(mu, sigma, size=num_graphs)).astype(np.int64)
    return np.clip(n, lo, hi)


def molecule_batch(num_graphs: int, seed: int = 0, dataset: str = "molhiv",
                   atom_dims: Optional[List[int]] = None,
                   bond_dims: Optional[List[int]] = None,
                   num_tasks: int = 1, nan_frac: float = 0.0,
                   regression: bool = False,
                   isolated_frac: float = 0.0) -> GraphBatch:
    """Molecule-like graphs: random spanning tree + ring closures, every bond
    emitted as two adjacent directed edges (u->v, v->u).

    ``isolated_frac`` > 0 appends nodes without any edge (the reference strips
    them with RemoveIsolatedNodes, train_hiv.py:171-172; the kernels must cope
    with empty destination segments regardless).
    """
    mean_nodes, bonds_per_atom = DATASET_STATS[dataset]
    if atom_dims is None:
        atom_dims = ZINC_ATOM_DIMS if dataset == "zinc" else OGB_ATOM_DIMS
    if bond_dims is None:
        bond_dims = ZINC_BOND_DIMS if dataset == "zinc" else OGB_BOND_DIMS
    rng = np.random.default_rng(seed)
    sizes = _graph_sizes(rng, num_graphs, mean_nodes)

    src_l, dst_l, battr_l, batch_l = [], [], [], []
    offset = 0
    for g, nv in enumerate(sizes):
        nv = int(nv)
        # spanning tree: node v attaches to one of the few nodes before it
        # (chain-like with short branches, as in drug-like molecules)
        child = np.arange(1, nv, dtype=np.int64)
        back = rng.geometric(0.55, size=nv - 1)
        parent = np.maximum(child - back, 0)
        n_ring = max(int(round((bonds_per_atom - 1.0) * nv + 1)), 0)
        n_ring = min(n_ring, max(nv - 2, 0))
        if n_ring > 0 and nv > 3:
            u = rng.integers(0, nv, size=n_ring)
            span = rng.integers(2, 7, size=n_ring)      # 3..7-membered rings
            v = np.minimum(u + span, nv - 1)
            keep = u != v
            u, v = u[keep], v[keep]
        else:
            u = v = np.zeros(0, dtype=np.int64)
        bu = np.concatenate([child, u]) + offset
        bv = np.concatenate([parent, v]) + offset
        nb = bu.shape[0]
        battr = np.stack([rng.integers(0, d, size=nb) for d in bond_dims], axis=1)
        s = np.empty(2 * nb, dtype=np.int64)
        d = np.empty(2 * nb, dtype=np.int64)
        s[0::2], d[0::2] = bu, bv
        s[1::2], d[1::2] = bv, bu
        src_l.append(s)
        dst_l.append(d)
        battr_l.append(np.repeat(battr, 2, axis=0))
        n_iso = int(rng.binomial(nv, isolated_frac)) if isolated_frac > 0 else 0
        batch_l.append(np.full(nv + n_iso, g, dtype=np.int64))
        offset += nv + n_iso

    n_total = offset
    edge_index = np.stack([np.concatenate(src_l), np.concatenate(dst_l)], axis=0)
    edge_attr = np.concatenate(battr_l, axis=0).astype(np.int64)
    batch = np.concatenate(batch_l)
    x = np.stack([rng.integers(0, d, size=n_total) for d in atom_dims], axis=1).astype(np.int64)
    if regression:
        y = rng.standard_normal((num_graphs, num_tasks)).astype(np.float32)
    else:
        y = (rng.random((num_graphs, num_tasks)) < 0.035).astype(np.float32)
    if nan_frac > 0:
        y[rng.random(y.shape) < nan_frac] = np.nan
    return GraphBatch(x=x, edge_index=edge_index, edge_attr=edge_attr, batch=batch,
                      y=y, num_graphs=num_graphs)


def random_edges(num_nodes: int, num_edges: int, seed: int = 0):
    """iid-uniform directed edges (BASELINE config 5 shape)."""
    rng = np.random.default_rng(seed)
    src = rng.integers(0, num_nodes, size=num_edges, dtype=np.int64)
    dst = rng.integers(0, num_nodes, size=num_edges, dtype=np.int64)
    return np.stack([src, dst], axis=0)


def ragged_edges(num_nodes: int, seed: int = 0, hub_degree: int = 300,
                 multi_edge: bool = True):
    """Edge list that exercises the edge cases the aggregation kernels must
    survive: empty destination segments, one very long segment, repeated
    (src, dst) pairs and self edges, all in shuffled order."""
    rng = np.random.default_rng(seed)
    src = [rng.integers(0, num_nodes, size=3 * num_nodes)]
    dst = [rng.integers(0, max(num_nodes // 2, 1), size=3 * num_nodes)]   # upper half stays empty
    src.append(rng.integers(0, num_nodes, size=hub_degree))
    dst.append(np.zeros(hub_degree, dtype=np.int64))                      # hub destination 0
    if multi_edge:
        src.append(np.full(5, 1, dtype=np.int64))
        dst.append(np.full(5, 2, dtype=np.int64))                         # 5 parallel edges 1->2
    src.append(np.arange(min(4, num_nodes), dtype=np.int64))
    dst.append(np.arange(min(4, num_nodes), dtype=np.int64))              # explicit self edges
    src = np.concatenate(src).astype(np.int64)
    dst = np.concatenate(dst).astype(np.int64)
    p = rng.permutation(src.shape[0])
    return np.stack([src[p], dst[p]], axis=0)
