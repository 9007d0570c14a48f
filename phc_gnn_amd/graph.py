"""Device-resident sorted edge lists for one ``edge_index`` (built by
``phc_csr_build``), shared by every layer and by forward and backward.

The reference re-derives the scatter pattern from the raw int64 ``edge_index``
on every ``propagate`` call (PyG 1.6.1, call sites
phc/hypercomplex/undirectional/messagepassing.py:60,136).  Here the
destination-sorted view (forward) and the source-sorted view (backward) are
built once per batch and cached by tensor identity.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Optional, Tuple

import torch

from . import lib as L

_CACHE_SIZE = 16
BUCKETED_MIN_ITEMS, BUCKETED_MIN_KEYS = 1 << 20, 1 << 16
_cache: "OrderedDict[tuple, EdgeStructure]" = OrderedDict()


def _build_sorted(key: torch.Tensor, other: Optional[torch.Tensor], num_keys: int, out=None, num_other: int = 0):
    """(rowptr[num_keys+1], col[E], perm[E]) int32 on key.device; ``out``: rebuild into these three tensors.
    ``num_other``: valid range of the ``other`` values (they become gather indices)."""
    lib = L.load()
    dev = L.require_device(key, other)
    E = key.numel()
    if out is not None:
        rowptr, col, perm = out
    else:
        rowptr = torch.empty(num_keys + 1, dtype=torch.int32, device=dev)
        col = torch.empty(E, dtype=torch.int32, device=dev)
        perm = torch.empty(E, dtype=torch.int32, device=dev)
    if E >= BUCKETED_MIN_ITEMS and num_keys >= BUCKETED_MIN_KEYS and num_keys <= 16384 * 4096:
        # large graphs: partition by key range first, then sort bucket by bucket in LDS (same result, ~2x faster at 10 M / 50 M;
        # rows longer than 2048 items — a hub of a power-law graph — are sorted by a workgroup-wide bitonic network in both paths)
        ws_bytes = lib.phc_csr_bucketed_ws_bytes(E, num_keys)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        L.check(lib.phc_csr_build_bucketed(L.ptr(key), L.ptr(other), E, num_keys, num_other, L.ptr(rowptr), L.ptr(col), L.ptr(perm),
                                           L.ptr(ws), ws_bytes, L.stream_ptr(dev)), "phc_csr_build_bucketed")
        L.maybe_check_indices(dev)
        return rowptr, col, perm
    ws_bytes = lib.phc_csr_ws_bytes(E, num_keys)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    L.check(lib.phc_csr_build(L.ptr(key), L.ptr(other), E, num_keys, num_other, L.ptr(rowptr), L.ptr(col), L.ptr(perm),
                              L.ptr(ws), ws_bytes, L.stream_ptr(dev)), "phc_csr_build")
    L.maybe_check_indices(dev)
    return rowptr, col, perm


class EdgeStructure:
    """CSR (by target) and CSC (by source) of a directed edge list.

    ``num_src`` rows exist on the source side, ``num_dst`` on the target side
    (equal for an ordinary graph; for a destination-range shard the targets are
    the owned nodes and the sources are owned + halo rows)."""

    def __init__(self, edge_index: torch.Tensor, num_src: int, num_dst: Optional[int] = None):
        if edge_index.dim() != 2 or edge_index.shape[0] != 2:
            raise L.PhcError(f"edge_index must be [2, E], got {tuple(edge_index.shape)}")
        if edge_index.dtype != torch.int64:
            edge_index = edge_index.to(torch.int64)
        L.require_device(edge_index)
        self.edge_index = edge_index            # keeps the storage alive while cached
        self.num_src = int(num_src)
        self.num_dst = int(num_src if num_dst is None else num_dst)
        self.num_edges = int(edge_index.shape[1])
        self._src = edge_index[0].contiguous()
        self._dst = edge_index[1].contiguous()
        self._csr: Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = None
        self._csc: Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = None
        self._codes: dict = {}
        self._code_counts = None          # (codes tensor, [num_dst, 16 | 32] per-destination table-row counts) from the pair build
        self._codes_args = None
        self._codes_orig: dict = {}
        self._codes_orig_args = None
        self._codes_src: dict = {}
        self._codes_src_args = None
        self._ell = self._ell_t = None    # first four in- / out-edges per row (pair build with codes)

    def packed_codes(self, bond_idx: torch.Tensor, vocab_c, F: int) -> torch.Tensor:
        """uint32 per edge in destination-sorted order: the F table rows of the edge's integer features, one
        byte each (built once per bond tensor, shared by all layers)."""
        key = (bond_idx.data_ptr(), bond_idx._version, tuple(vocab_c))
        codes = self._codes.get(key)
        if codes is None:
            codes = torch.empty(self.num_edges, dtype=torch.int32, device=bond_idx.device)
            self._pack_codes(bond_idx, vocab_c, F, codes)
            self._codes = {key: codes}
            self._codes_args = (bond_idx, vocab_c, F)
        return codes

    def _pack_codes(self, bond_idx, vocab_c, F, codes, sorted_order: bool = True, by_source: bool = False) -> None:
        perm = self.csc[2] if by_source else (self.csr[2] if sorted_order else None)
        L.check(L.load().phc_pack_edge_codes(L.ptr(bond_idx), L.ptr(perm), vocab_c, self.num_edges, F, L.ptr(codes),
                                             L.stream_ptr(bond_idx.device)), "phc_pack_edge_codes")

    def packed_codes_orig(self, bond_idx: torch.Tensor, vocab_c, F: int) -> torch.Tensor:
        """The same codes in ORIGINAL edge order (the source-sorted backward walk addresses edges by their original id)."""
        key = (bond_idx.data_ptr(), bond_idx._version, tuple(vocab_c))
        codes = self._codes_orig.get(key)
        if codes is None:
            codes = torch.empty(self.num_edges, dtype=torch.int32, device=bond_idx.device)
            self._pack_codes(bond_idx, vocab_c, F, codes, sorted_order=False)
            self._codes_orig = {key: codes}
            self._codes_orig_args = (bond_idx, vocab_c, F)
        return codes

    def packed_codes_src(self, bond_idx: torch.Tensor, vocab_c, F: int) -> torch.Tensor:
        """The same codes in SOURCE-sorted order (the softmax backward walks the out-edges of a source and wants an edge's
        code in the same round trip as its target index, not one hop later through the original edge id)."""
        key = (bond_idx.data_ptr(), bond_idx._version, tuple(vocab_c))
        codes = self._codes_src.get(key)
        if codes is None:
            codes = torch.empty(self.num_edges, dtype=torch.int32, device=bond_idx.device)
            self._pack_codes(bond_idx, vocab_c, F, codes, by_source=True)
            self._codes_src = {key: codes}
            self._codes_src_args = (bond_idx, vocab_c, F)
        return codes

    def rebuild(self) -> None:
        """Recompute every view that has been built — same tensors, new contents — after the edge list (and the bond
        features behind the packed codes) changed IN PLACE: the per-batch preprocessing of a fixed-shape step whose
        static buffers were reloaded (graphed.GraphedTrainStep captures this as its own small graph)."""
        pair_rebuilt = False
        if self._csr is not None and self._csc is not None and self.num_edges > 0 and \
                0 < max(self.num_src, self.num_dst) <= self.PAIR_MAX_ROWS:
            self._build_pair(reuse=True)
            pair_rebuilt = True
        else:
            if self._csr is not None:
                _build_sorted(self._dst, self._src, self.num_dst, out=self._csr, num_other=self.num_src)
            if self._csc is not None:
                _build_sorted(self._src, self._dst, self.num_src, out=self._csc, num_other=self.num_dst)
        if not pair_rebuilt:
            self._ell = self._ell_t = None                            # (only the pair build writes them)
        if not (pair_rebuilt and self._codes_args is not None):      # the pair launches repacked the sorted codes already
            self._code_counts = None                                  # (and the counts that go with them; stale otherwise)
            for codes in self._codes.values():
                self._pack_codes(*self._codes_args, codes)
        for codes in self._codes_orig.values():
            self._pack_codes(*self._codes_orig_args, codes, sorted_order=False)
        if not (pair_rebuilt and getattr(self, "_pair_packed_src", False)):      # (... and the source-sorted codes, when it carries them)
            for codes in self._codes_src.values():
                self._pack_codes(*self._codes_src_args, codes, by_source=True)

    PAIR_MAX_ROWS = 32767

    def _build_pair(self, reuse: bool = False, codes_for=None, codes_src: bool = False) -> None:
        """Both views side by side (phc_csr_build_pair: 5 launches instead of 22); ``codes_for = (bond_idx, vocab_c, F)``:
        the packed edge codes of that bond tensor come out of the same launches."""
        lib = L.load()
        dev = self._src.device
        E = self.num_edges

        def alloc(rows):
            return (torch.empty(rows + 1, dtype=torch.int32, device=dev), torch.empty(E, dtype=torch.int32, device=dev),
                    torch.empty(E, dtype=torch.int32, device=dev))
        csr, csc = (self._csr, self._csc) if reuse else (alloc(self.num_dst), alloc(self.num_src))
        ws_bytes = lib.phc_csr_pair_ws_bytes(E, self.num_src, self.num_dst)
        # resident workspace: zeroed ONCE per (device, stream, E, rows) and shared by every structure of those sizes — each build leaves
        # its counter arrays zeroed again, so neither a zero launch nor a fill is part of a (captured) per-batch build
        # (keyed by the sizes, not the byte count: the zeroed counter arrays sit at size-dependent offsets, the rest is scratch)
        key = (dev, torch.cuda.current_stream(dev).cuda_stream, int(E), int(self.num_src), int(self.num_dst))
        ws = _PAIR_WS_PINNED.get(key)
        if ws is None:
            ws = _PAIR_WS.get(key)
            if ws is not None and torch.cuda.is_current_stream_capturing():
                _PAIR_WS_PINNED[key] = ws                     # a captured step holds this address from now on: never evicted
        if ws is None:
            if torch.cuda.is_current_stream_capturing():
                ws = torch.zeros(ws_bytes, dtype=torch.uint8, device=dev)        # (first seen inside a capture: a captured fill)
            else:
                ws = _PAIR_WS[key] = torch.zeros(ws_bytes, dtype=torch.uint8, device=dev)
                while len(_PAIR_WS) > _PAIR_WS_MAX:           # eager training over many batch shapes: keep the recent ones
                    _PAIR_WS.pop(next(iter(_PAIR_WS)))
        elif key in _PAIR_WS:
            _PAIR_WS[key] = _PAIR_WS.pop(key)                 # most recently used last
        self._pair_ws = ws
        codes = bond = vocab_c = counts = None
        F = 0
        ell, ell_t = (getattr(self, "_ell", None), getattr(self, "_ell_t", None)) if reuse else (None, None)
        if reuse and self._codes and self._codes_args is not None:
            bond, vocab_c, F = self._codes_args
            codes = next(iter(self._codes.values()))
            cc = getattr(self, "_code_counts", None)
            counts = cc[1] if (cc is not None and cc[0] is codes) else None
        elif codes_for is not None:
            bond, vocab_c, F = codes_for
            codes = torch.empty(E, dtype=torch.int32, device=dev)
            rows = sum(int(v) for v in list(vocab_c)[:F])
            if rows <= 32:                                   # per-destination table-row counts (phc_step_tail's bond operand)
                counts = torch.empty((self.num_dst, 16 if rows <= 16 else 32), dtype=torch.float32, device=dev)
            # the first four in- / out-edges of every row side by side (the aggregating fused launches read a row's edges in the round
            # trip of its row pointers: csrc/mlp_fused.hip)
            ell = torch.empty((self.num_dst, 8), dtype=torch.int32, device=dev)
            ell_t = torch.empty((self.num_src, 4), dtype=torch.int32, device=dev)
        if codes is None:
            ell = None                                       # (ell carries the packed codes)
        # the same codes in SOURCE-sorted order (the softmax aggregation's backward walk: packed_codes_src) out of the same launches
        csrc = None
        if codes is not None:
            if reuse:
                if self._codes_src and self._codes_src_args is not None and self._codes_src_args[0] is bond and \
                        tuple(self._codes_src_args[1]) == tuple(vocab_c):
                    csrc = next(iter(self._codes_src.values()))
            elif codes_src:
                csrc = torch.empty(E, dtype=torch.int32, device=dev)
        L.check(lib.phc_csr_build_pair_codes_resident_ell_src(L.ptr(self._src), L.ptr(self._dst), E, self.num_src, self.num_dst, L.ptr(csr[0]),
                                                     L.ptr(csr[1]), L.ptr(csr[2]), L.ptr(csc[0]), L.ptr(csc[1]), L.ptr(csc[2]),
                                                     L.ptr(bond), vocab_c, F, L.ptr(codes), L.ptr(counts), L.ptr(ell), L.ptr(ell_t),
                                                     L.ptr(csrc), L.ptr(ws), ws_bytes, L.stream_ptr(dev)),
                "phc_csr_build_pair_codes_resident_ell_src")
        self._pair_packed_src = csrc is not None
        if csrc is not None and not reuse:
            self._codes_src = {(bond.data_ptr(), bond._version, tuple(vocab_c)): csrc}
            self._codes_src_args = (bond, vocab_c, F)
        self._ell, self._ell_t = ell, ell_t
        self._code_counts = (codes, counts) if counts is not None else None
        L.maybe_check_indices(dev)
        self._csr, self._csc = csr, csc
        if codes is not None and not reuse:
            self._codes = {(bond.data_ptr(), bond._version, tuple(vocab_c)): codes}
            self._codes_args = (bond, vocab_c, F)
        self._pair_packed = codes is not None

    def prepare(self, for_backward: bool, codes_for=None, codes_src: bool = False) -> "EdgeStructure":
        """Build what a forward (and, with ``for_backward``, its backward) will walk.  A differentiable forward wants
        the source-sorted view later: both views are then built side by side (5 launches instead of 22).  ``codes_for
        = (int64 bond tensor, vocab_c, F)``: the caller will ask for ``packed_codes`` of it next — they ride along (``codes_src``: and for
        ``packed_codes_src`` in its backward)."""
        if (for_backward and self._csr is None and self._csc is None and self.num_edges > 0
                and 0 < max(self.num_src, self.num_dst) <= self.PAIR_MAX_ROWS):
            self._build_pair(codes_for=codes_for, codes_src=codes_src)
        return self

    @property
    def csr(self):
        """rowptr[num_dst+1], col = source of each in-edge, perm = original edge id."""
        if self._csr is None:
            self._csr = _build_sorted(self._dst, self._src, self.num_dst, num_other=self.num_src)
        return self._csr

    @property
    def csc(self):
        """rowptr_t[num_src+1], col_t = target of each out-edge, perm_t = original edge id."""
        if self._csc is None:
            self._csc = _build_sorted(self._src, self._dst, self.num_src, num_other=self.num_dst)
        return self._csc


_IOTA: dict = {}
_PAIR_WS_MAX = 16
_PAIR_WS_PINNED: dict = {}   # entries a captured hipGraph refers to
_PAIR_WS: dict = {}          # (device, stream, E, n_src, n_dst) -> zeroed workspace of the pair build (EdgeStructure._build_pair)


def _iota(n: int, dev) -> torch.Tensor:
    """arange(n) int32 on dev, grown on demand (identity member list of sorted segments)."""
    t = _IOTA.get(dev)
    if t is None or t.numel() < n:
        t = torch.arange(max(n, 1024), dtype=torch.int32, device=dev)
        _IOTA[dev] = t
    return t


class SegmentStructure:
    """Rows grouped by an int64 segment id (``batch`` -> graphs), for read-out.

    ``is_sorted=True`` (the collated ``batch`` vector of a PyG mini-batch is non-decreasing by
    construction) takes the cheap path: segment boundaries by binary search, members in place."""

    def __init__(self, batch: torch.Tensor, num_segments: int, is_sorted: bool = False):
        L.require_device(batch)
        self.batch = batch if batch.dtype == torch.int64 else batch.to(torch.int64)
        self.batch = self.batch.contiguous()
        self.num_segments = int(num_segments)
        self.is_sorted = bool(is_sorted)
        self._csr = None

    @property
    def csr(self):
        if self._csr is None:
            if self.is_sorted:
                dev = self.batch.device
                n = self.batch.numel()
                rowptr = torch.empty(self.num_segments + 1, dtype=torch.int32, device=dev)
                self._sorted_ptr(rowptr)
                self._csr = (rowptr, _iota(n, dev), None)
            else:
                self._csr = _build_sorted(self.batch, None, self.num_segments)
        return self._csr

    def _sorted_ptr(self, rowptr) -> None:
        dev = self.batch.device
        L.check(L.load().phc_segment_ptr_sorted(L.ptr(self.batch), self.batch.numel(), self.num_segments, L.ptr(rowptr),
                                                L.stream_ptr(dev)), "phc_segment_ptr_sorted")

    def rebuild(self) -> None:
        """Recompute the built view in place after ``batch`` changed in place (see EdgeStructure.rebuild)."""
        if self._csr is None:
            return
        if self.is_sorted:
            self._sorted_ptr(self._csr[0])
        else:
            _build_sorted(self.batch, None, self.num_segments, out=self._csr)


def _key(t: torch.Tensor, *extra):
    return (t.data_ptr(), t._version, tuple(t.shape), t.dtype, t.device, *extra)


def edge_structure(edge_index: torch.Tensor, num_src: int, num_dst: Optional[int] = None) -> EdgeStructure:
    k = _key(edge_index, "edge", num_src, num_dst)
    st = _cache.get(k)
    if st is None:
        st = EdgeStructure(edge_index, num_src, num_dst)
        _cache[k] = st
        st._pin = edge_index                    # the key's data_ptr stays valid while the entry lives
        while len(_cache) > _CACHE_SIZE:
            _cache.popitem(last=False)
    else:
        _cache.move_to_end(k)
    return st


def segment_structure(batch: torch.Tensor, num_segments: Optional[int] = None, is_sorted: bool = False) -> SegmentStructure:
    if num_segments is None:
        # same host sync the reference pays in global_add_pool (batch.max() + 1)
        num_segments = int(batch.max().item()) + 1 if batch.numel() else 0
    k = _key(batch, "seg", num_segments, is_sorted)
    st = _cache.get(k)
    if st is None:
        st = SegmentStructure(batch, num_segments, is_sorted)
        _cache[k] = st
        st._pin = batch
        while len(_cache) > _CACHE_SIZE:
            _cache.popitem(last=False)
    else:
        _cache.move_to_end(k)
    return st


def clear_structure_cache() -> None:
    _cache.clear()


def cached_structures_of(*tensors) -> list:
    """The cached structures that were built from any of these tensors (by storage address)."""
    ptrs = {t.data_ptr() for t in tensors}
    return [st for st in _cache.values() if st._pin.data_ptr() in ptrs]


def remove_isolated_nodes(x: Optional[torch.Tensor], edge_index: torch.Tensor, batch: Optional[torch.Tensor] = None,
                          num_nodes: Optional[int] = None, sync: bool = True):
    """Drop the nodes that no edge touches and re-index the edge list — the per-batch transform of the reference's
    training loop (benchmarks/utils.py:36-46 / PyG ``RemoveIsolatedNodes``, train_hiv.py:171-172), on the device.

    x: int64 [N, F] node features (or None), edge_index: int64 [2, E], batch: int64 [N] (or None).
    ``sync=True`` returns (x', edge_index', batch', kept_index) trimmed to the kept nodes (one host read of the count,
    which the reference pays too — it runs this on the CPU); ``sync=False`` returns full-size buffers whose rows
    [0, n_kept) are valid plus the device int32 count: (x', edge_index', batch', kept_index, n_kept), for
    fixed-shape batches.  edge_attr is untouched by the transform (edges are never removed)."""
    lib = L.load()
    dev = L.require_device(edge_index, x, batch)
    ei = edge_index if edge_index.dtype == torch.int64 else edge_index.to(torch.int64)
    ei = ei.contiguous()
    E = ei.shape[1]
    N = int(num_nodes if num_nodes is not None else (x.shape[0] if x is not None else batch.shape[0]))
    x_ = None if x is None else x.to(torch.int64).contiguous().reshape(N, -1)
    F = 0 if x_ is None else x_.shape[1]
    b_ = None if batch is None else batch.to(torch.int64).contiguous()
    ei_out = torch.empty_like(ei)
    x_out = None if x_ is None else torch.zeros_like(x_)
    b_out = None if b_ is None else torch.zeros_like(b_)
    kept = torch.zeros(N, dtype=torch.int64, device=dev)
    n_kept = torch.zeros((), dtype=torch.int32, device=dev)
    ws_bytes = lib.phc_remove_isolated_ws_bytes(N)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    L.check(lib.phc_remove_isolated_nodes(L.ptr(ei[0]) if E else None, L.ptr(ei[1]) if E else None, E, N, L.ptr(x_), F, L.ptr(b_),
                                          L.ptr(ei_out[0]) if E else None, L.ptr(ei_out[1]) if E else None, L.ptr(x_out),
                                          L.ptr(b_out), L.ptr(kept), L.ptr(n_kept), L.ptr(ws), ws_bytes, L.stream_ptr(dev)),
            "phc_remove_isolated_nodes")
    if not sync:
        return x_out, ei_out, b_out, kept, n_kept
    k = int(n_kept.item())
    return (None if x_out is None else x_out[:k]), ei_out, (None if b_out is None else b_out[:k]), kept[:k]
