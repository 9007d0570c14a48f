"""hipGraph-replayed training step on fixed-shape (padded) batches.

The molecule-scale step is launch-bound (≈200 kernels of a few µs each), so the whole step —
CSR/CSC build, forward, loss, backward, gradient all-reduce, clipping, Adam — is captured ONCE
into a hipGraph and replayed per batch.  Batches differ in size, so they are padded to one
bucket shape:

  * nodes  N..N_max-1  are padding nodes with feature indices 0, owned by PAD_GRAPHS extra graph ids
    B..B+PAD_GRAPHS-1 in round-robin blocks (no single long read-out segment);
  * edges  E..E_max-1  connect padding nodes to themselves, spread over the padding nodes;
  * the extra graphs have NaN labels (masked out of the loss exactly like unlabeled tasks);
  * ``n_valid_nodes`` / ``n_valid_graphs`` are DEVICE int32 scalars read by the batch-norm kernels:
    padding rows stay out of the statistics, are written as 0 and receive zero gradient, so no
    padding row ever reaches a parameter gradient (tests/test_gpu_model.py::test_graphed_step_*).

Per step: one device-to-device copy of the padded batch into the static buffers + one graph launch.
(Reference loop being reproduced: benchmarks/train_hiv.py:165-218.)
"""
from __future__ import annotations

import gc
import os
import types
import warnings
from typing import List, Optional

import numpy as np
import torch

from . import synth
from .graph import cached_structures_of, clear_structure_cache
from .harness import StepConfig, advance_dropout_rng, apply_weight_regulariser, loss_fn, optimizer_step, park_weight_regulariser
from .ops import atom_encode_spec, deferred_param_grads, dense_weights, expect_loss, gate_specs, phm_pairs, prebuilt_H, stat_accumulators


# rows of one full round of the row-blocked GEMM launches: 16-row steps x one workgroup per CU of a 256-CU part (the 256-wide layers;
# the 128-wide ones run two workgroups per CU: 8 192 rows, a multiple).  A padded node count just ABOVE a whole number of rounds costs
# every such launch a nearly empty extra round (24 832 rows = 6.06 rounds: seven passes of the step loop where 24 576 take six).
ROUND_ROWS = int(__import__("os").environ.get("PHC_BUCKET_ROUND_ROWS", "4096"))


def bucket_shape(batches: List[synth.GraphBatch], slack: float = 0.02, multiple: int = 128):
    """(N_max, E_max) covering every batch with a little slack, plus room for padding nodes.  Inside the slack window a node count that
    is a whole number of GEMM rounds (ROUND_ROWS) is preferred: the padding is there to be cheap."""
    n = max(b.num_nodes for b in batches)
    e = max(b.num_edges for b in batches)
    n_max = int(np.ceil(n * (1 + slack) / multiple) * multiple) + multiple
    if ROUND_ROWS > 0:
        whole = n_max // ROUND_ROWS * ROUND_ROWS
        # (only while half of the relative slack and `multiple` padding nodes remain — ADVICE r5: a caller that sizes the bucket from
        # a SAMPLE of its batches relies on the slack; the padding graphs live in the padding nodes)
        if whole >= int(np.ceil(n * (1 + slack / 2))) + multiple:
            n_max = whole
    e_max = int(np.ceil(e * (1 + slack) / multiple) * multiple)
    return n_max, e_max


PAD_GRAPHS = 32


def pad_batch(gb: synth.GraphBatch, n_max: int, e_max: int) -> dict:
    """Host-side padding of one collated batch to the bucket shape (part of collation)."""
    N, E, B = gb.num_nodes, gb.num_edges, gb.num_graphs
    if N >= n_max or E > e_max:
        raise ValueError(f"batch ({N} nodes, {E} edges) does not fit the bucket ({n_max}, {e_max})")
    if N > 1 and not bool((gb.batch[1:] >= gb.batch[:-1]).all()):
        raise ValueError("pad_batch: `batch` must be non-decreasing (a collated mini-batch); sort the nodes by graph first")
    n_pad = n_max - N
    x = np.zeros((n_max, gb.x.shape[1]), dtype=np.int64)
    x[:N] = gb.x
    ei = np.empty((2, e_max), dtype=np.int64)
    ei[:, :E] = gb.edge_index
    pad_nodes = N + (np.arange(e_max - E, dtype=np.int64) % n_pad)
    ei[0, E:] = pad_nodes
    ei[1, E:] = pad_nodes
    ea = np.zeros((e_max, gb.edge_attr.shape[1]), dtype=np.int64)
    ea[:E] = gb.edge_attr
    batch = np.empty(n_max, dtype=np.int64)
    batch[:N] = gb.batch
    per = -(-n_pad // PAD_GRAPHS)                                      # contiguous blocks keep `batch` sorted
    batch[N:] = B + np.minimum(np.arange(n_pad, dtype=np.int64) // per, PAD_GRAPHS - 1)
    y = np.full((B + PAD_GRAPHS, gb.y.shape[1]), np.nan, dtype=np.float32)
    y[:B] = gb.y
    return dict(x=x, edge_index=ei, edge_attr=ea, batch=batch, y=y, n_valid_nodes=np.array(N, dtype=np.int32),
                num_graphs=B, num_nodes=N, num_edges=E)


FLAT_TENSORS = ("x", "edge_index", "edge_attr", "batch", "y", "n_valid_nodes")


def _flat_layout(p: dict):
    """Byte layout of the batch arrays inside one flat buffer (16-byte aligned segments)."""
    layout, off = {}, 0
    for k in FLAT_TENSORS:
        a = np.asarray(p[k])
        layout[k] = (off, a.shape, a.dtype)
        off += -(-a.nbytes // 16) * 16
    return layout, off


def padded_to_device(p: dict, device) -> dict:
    """One host buffer, ONE host-to-device copy; the per-array tensors are views into the flat device buffer,
    so loading a batch into the static buffers of a captured step is a single device-to-device copy."""
    layout, total = _flat_layout(p)
    host = np.zeros(total, dtype=np.uint8)
    for k, (off, shape, dt) in layout.items():
        a = np.ascontiguousarray(p[k])
        host[off:off + a.nbytes] = a.reshape(-1).view(np.uint8)
    flat = torch.from_numpy(host).to(device)
    out = {k: v for k, v in p.items() if k not in layout}
    out.update(_flat_views(flat, layout))
    out["_flat"], out["_layout"] = flat, layout
    return out


def _flat_views(flat: torch.Tensor, layout: dict) -> dict:
    views = {}
    for k, (off, shape, dt) in layout.items():
        tdt = torch.from_numpy(np.empty(0, dtype=dt)).dtype
        nbytes = int(np.prod(shape, dtype=np.int64)) * np.dtype(dt).itemsize
        views[k] = flat[off:off + nbytes].view(tdt).view(tuple(shape))
    return views


class _BatchSet:
    """One set of static batch buffers with everything captured over it."""
    __slots__ = ("static_flat", "static", "data", "graph", "graph2", "loss", "logits", "prep", "structs", "ev_prep", "ev_main",
                 "ready")


class GraphedTrainStep:
    """Static buffers + one captured hipGraph for the whole training step.

    ``pipeline`` (opt-in, also ``PHC_GRAPH_PIPELINE=1``): the batch-dependent preprocessing — the copy of
    the next batch into static buffers, the CSR/CSC build, the packed edge codes, the read-out segment pointers; 8
    launches that depend on the batch only, not on the parameters — is captured as its OWN small graph over a second
    set of static buffers and replayed on a second stream while the previous step still runs (``step(batch,
    next_batch)``: a data-loader prefetch).  The step graph proper is captured once per buffer set and finds the
    structures prebuilt.  Two events per step order the two streams.  Measured on ROCm 7.0 with the graph packet
    capture disabled (phc_gnn_amd/__init__.py): bit-identical results, but 0.87 ms instead of 0.81 ms per molhiv step —
    the host enqueues graph nodes one by one (~8 us each, ~780 us per step, as long as the GPU needs), so a second graph
    launch and two events per step cost more host time than the 8 launches they take off the GPU's critical path.
    Hence off by default; it pays once graph launches are cheap again."""

    TENSORS = FLAT_TENSORS

    def __init__(self, model, optimizer, cfg: StepConfig, example: dict, grad_sync=None, warmup: int = 3,
                 pipeline: Optional[bool] = None, double_buffer: Optional[bool] = None):
        """``grad_sync``: optional ``phc_gnn_amd.dist.FlatGradSync`` (data-parallel ranks).  The step is then
        captured as TWO graphs — [forward, loss, backward, pack gradients into the flat bucket] and
        [unpack, clip, Adam] — with the RCCL all-reduce of the bucket issued eagerly between them."""
        self.model, self.optimizer, self.cfg, self.grad_sync = model, optimizer, cfg, grad_sync
        if pipeline is None:
            pipeline = os.environ.get("PHC_GRAPH_PIPELINE", "0") == "1"
        self.pipeline = bool(pipeline)
        # ``double_buffer``: TWO sets of static batch buffers, each with its own captured step, used alternately.  The next
        # batch's copy then never targets buffers a replay in flight may still read — which is the only thing the "graph
        # packet capture" fast path of HIP 7.0 fails to order (module docstring of phc_gnn_amd/__init__.py).  Default: on
        # exactly when that fast path is on (DEBUG_CLR_GRAPH_PACKET_CAPTURE != 0); single-process runs only — a collective
        # queued behind a replay is not protected by this.
        packet_capture = os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "1") != "0"
        if double_buffer is None:
            double_buffer = packet_capture and not self.pipeline
        self.double_buffer = bool(double_buffer) and not self.pipeline
        if packet_capture and grad_sync is not None:
            raise RuntimeError("GraphedTrainStep: the graph packet capture fast path (DEBUG_CLR_GRAPH_PACKET_CAPTURE != 0) does not "
                               "order a replay against the collective queued behind it; keep it off for data-parallel runs")
        dev = example["x"].device
        B = int(example["num_graphs"])
        n_valid_graphs = torch.tensor(B, dtype=torch.int32, device=dev)
        n_valid_graphs._phc_max_valid = B           # (host-side bound for kernels whose grid depends on it: ops.downstream_head)
        self.sets: List[_BatchSet] = []
        for _ in range(2 if (self.pipeline or self.double_buffer) else 1):
            bs = _BatchSet()
            if "_flat" in example:                  # batch arrays live in one flat buffer: one copy per step
                bs.static_flat = example["_flat"].clone()
                bs.static = _flat_views(bs.static_flat, example["_layout"])
            else:
                bs.static_flat = None
                bs.static = {k: example[k].clone() for k in self.TENSORS}
            bs.data = types.SimpleNamespace(
                x=bs.static["x"], edge_index=bs.static["edge_index"], edge_attr=bs.static["edge_attr"],
                batch=bs.static["batch"], y=bs.static["y"], num_graphs=B + PAD_GRAPHS,
                n_valid_nodes=bs.static["n_valid_nodes"], n_valid_graphs=n_valid_graphs, all_labeled=False,
                batch_is_sorted=True)                  # pad_batch keeps `batch` non-decreasing (and checks its input)
            bs.graph = bs.graph2 = bs.loss = bs.logits = bs.prep = None
            bs.structs = []
            bs.ev_prep, bs.ev_main, bs.ready = torch.cuda.Event(), torch.cuda.Event(), False
            self.sets.append(bs)
        self.layout = example.get("_layout")
        self._cur = 0
        self.device = dev
        self._pairs = phm_pairs(model)
        self._dense = dense_weights(model)
        self._gates = gate_specs(model)
        self.replay_stream = torch.cuda.Stream(device=dev)     # never the capture stream, never the null stream
        self.prep_stream = torch.cuda.Stream(device=dev) if self.pipeline else None
        from .ops import register_unit_seed
        self._seed = register_unit_seed(torch.ones((), dtype=torch.float32, device=dev))   # d loss / d loss, resident
        if packet_capture and not self.double_buffer:
            warnings.warn("DEBUG_CLR_GRAPH_PACKET_CAPTURE is not 0: on HIP 7.0 pipelined hipGraph replays race with the "
                          "batch copies queued behind them (see phc_gnn_amd/__init__.py)", RuntimeWarning)
        # Warm-up (allocator pools, lazy library initialisation, pointer tables) and capture run on ONE
        # dedicated stream, so that autograd's per-parameter AccumulateGrad nodes, the optimizer state and
        # the captured kernels all agree on the stream (no cross-stream edges inside the graph).
        self.stream = torch.cuda.Stream(device=dev)
        self.stream2 = torch.cuda.Stream(device=dev) if grad_sync is not None else None
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self.stream):
            for _ in range(warmup):
                self._body(self.sets[0])
        self.stream.synchronize()
        pool = None
        for bs in self.sets:
            gc.collect()                       # drop the warm-up autograd graphs before capturing
            clear_structure_cache()
            if self.pipeline:
                # one more eager step over THIS set fills the structure cache with exactly what the step walks; the
                # preprocessing graph recomputes those structures in place, the step graph is captured with the cache
                # still holding them (no static buffer is written in between: the cache keys carry tensor versions)
                with torch.cuda.stream(self.stream):
                    self._body(bs)
                self.stream.synchronize()
                bs.structs = cached_structures_of(bs.data.edge_index, bs.data.batch)
                bs.prep = torch.cuda.CUDAGraph()
                cap = torch.cuda.Stream(device=dev)
                with torch.cuda.graph(bs.prep, stream=cap):
                    for st in bs.structs:
                        st.rebuild()
                gc.collect()
            self.optimizer.zero_grad(set_to_none=True)
            bs.graph = torch.cuda.CUDAGraph()
            kw = {} if pool is None else {"pool": pool}
            if self.grad_sync is None:
                with torch.cuda.graph(bs.graph, stream=self.stream, **kw):
                    bs.loss = self._body(bs)
            else:
                with torch.cuda.graph(bs.graph, stream=self.stream, **kw):
                    bs.loss = self._front(bs)
                # second half on its own stream (the first capture stream is never touched again), same memory pool
                bs.graph2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(bs.graph2, pool=bs.graph.pool(), stream=self.stream2):
                    self._back()
            pool = bs.graph.pool()             # the step graphs of the two sets never run concurrently: one pool
            # Drop the captured autograd graph (its AccumulateGrad nodes are bound to the capture stream; a later
            # eager backward of the same model would otherwise enqueue work on that stream).  The loss buffer
            # itself stays: it is the graph's output.
            bs.loss = bs.loss.detach()
        clear_structure_cache()
        gc.collect()
        # NOTE: only a device-wide synchronize here.  Making the default stream wait on the capture stream
        # (Stream.wait_stream -> an event recorded on the just-captured stream) left the graph in a state where
        # later eager allocations made its replays fault on ROCm 7.x (the packet-capture ordering bug, DESIGN.md §5;
        # scripts/dbg/debug_pipelined.py reproduces the family).
        torch.cuda.synchronize(dev)
        from .lib import check_indices
        check_indices(dev)                     # the example batch went through every index-reading kernel: bad ids raise here

    def check(self) -> None:
        """IndexError if any replay since the last check met an out-of-range node id / feature index (one 4-byte read;
        the kernels read such entries as 0 instead of dereferencing them)."""
        from .lib import check_indices
        check_indices(self.device)

    def __del__(self):
        try:
            from .ops import unregister_unit_seed
            unregister_unit_seed(self._seed)
        except Exception:
            pass

    # the single-set attributes of the non-pipelined form (tests, scripts)
    @property
    def static(self):
        return self.sets[self._cur].static

    @property
    def static_flat(self):
        return self.sets[self._cur].static_flat

    @property
    def data(self):
        return self.sets[self._cur].data

    @property
    def loss(self):
        return self.sets[self._cur].loss

    @property
    def logits(self):
        return self.sets[self._cur].logits

    @property
    def graph(self):
        return self.sets[self._cur].graph

    @property
    def graph2(self):
        return self.sets[self._cur].graph2

    def _front(self, bs: _BatchSet):
        """zero_grad, forward, loss, backward (+ pack the gradients into the flat all-reduce bucket)."""
        self.optimizer.zero_grad(set_to_none=True)
        advance_dropout_rng(self.model, bs.data)
        with stat_accumulators(self.device), prebuilt_H(self._pairs, self._dense, self._gates,
                                                        encode=atom_encode_spec(self.model, bs.data)), deferred_param_grads():
            with expect_loss(bs.data.y, "l1" if self.cfg.regression else "bce"):
                logits = self.model(bs.data)
            loss = loss_fn(logits, bs.data, self.cfg.regression)
            loss.backward(self._seed)
            parked = park_weight_regulariser(self.model, loss, self.cfg)
        if not parked:
            apply_weight_regulariser(self.model, loss, self.cfg)
        bs.logits = logits.detach()            # static output buffer of the captured step (rows of padding graphs included)
        if self.grad_sync is not None:
            self.grad_sync.pack()
        return loss

    def _back(self):
        """(unpack the reduced bucket,) clip, Adam."""
        if self.grad_sync is not None:
            self.grad_sync.unpack()
        if self.cfg.optimizer:
            optimizer_step(self.model, self.optimizer, self.cfg)

    def _body(self, bs: _BatchSet):
        loss = self._front(bs)
        if self.grad_sync is not None:
            self.grad_sync.reduce()
        self._back()
        return loss

    def _load_into(self, bs: _BatchSet, padded: dict) -> None:
        if bs.static_flat is not None and padded.get("_flat") is not None and \
                padded["_flat"].numel() == bs.static_flat.numel():
            bs.static_flat.copy_(padded["_flat"], non_blocking=True)
            return
        for k in self.TENSORS:
            bs.static[k].copy_(padded[k], non_blocking=True)

    def load(self, padded: dict) -> None:
        """Device-to-device copy of one padded, HBM-resident batch into the (current) static buffers."""
        self._load_into(self.sets[self._cur], padded)

    def step(self, padded: Optional[dict] = None, next_padded: Optional[dict] = None) -> torch.Tensor:
        """Load one padded batch and replay; ``next_padded`` (pipelined form) is the batch of the NEXT call, whose
        preprocessing starts now on the second stream.  Stream-ordered with the caller's current stream, but the copy
        and the graph launch themselves are issued on a dedicated NON-DEFAULT stream: on ROCm 7.2 a hipGraphLaunch into
        the legacy null stream is not ordered against the null-stream work queued behind it, so the next batch's copy
        overwrote the static buffers of a replay still in flight (scripts/dbg/debug_bench_like.py reproduces the fault)."""
        cur = torch.cuda.current_stream(self.device)
        if cur != torch.cuda.default_stream(self.device):
            return self._issue(padded, next_padded)    # caller already runs on a side stream (`with step.on_stream():`)
        rs = self.replay_stream                # default stream: hop over to the replay stream and back (two event waits)
        rs.wait_stream(cur)
        with torch.cuda.stream(rs):
            loss = self._issue(padded, next_padded)
        cur.wait_stream(rs)
        return loss

    def _prepare(self, bs: _BatchSet, padded: dict) -> None:
        """Queue one batch's preprocessing on the second stream: it may start as soon as the step that last read this
        buffer set has finished."""
        ps = self.prep_stream
        ps.wait_event(bs.ev_main)
        with torch.cuda.stream(ps):
            self._load_into(bs, padded)
            bs.prep.replay()
            bs.ev_prep.record(ps)
        bs.ready = padded

    def _issue(self, padded: Optional[dict], next_padded: Optional[dict] = None) -> torch.Tensor:
        main = torch.cuda.current_stream(self.device)
        if self.double_buffer:
            if padded is not None:
                self._cur = 1 - self._cur                      # the other set: nothing in flight reads it
                self._load_into(self.sets[self._cur], padded)
            bs = self.sets[self._cur]
        elif not self.pipeline:
            bs = self.sets[0]
            if padded is not None:
                self._load_into(bs, padded)
        else:
            if padded is not None:
                nxt = self.sets[1 - self._cur]
                if nxt.ready is padded:                        # prefetched by the previous call
                    self._cur = 1 - self._cur
                else:
                    # no prefetch: this call's own preprocessing, still on the second stream (the buffer set that is
                    # not being read by the step in flight)
                    self._cur = 1 - self._cur
                    self.prep_stream.wait_stream(main)
                    self._prepare(self.sets[self._cur], padded)
            bs = self.sets[self._cur]
            if bs.ready is not False:
                main.wait_event(bs.ev_prep)
        bs.graph.replay()
        if bs.graph2 is not None:
            self.grad_sync.reduce()            # eager RCCL all-reduce of the flat bucket (stream-ordered)
            bs.graph2.replay()
        if self.pipeline:
            bs.ev_main.record(main)
            if next_padded is not None:
                self._prepare(self.sets[1 - self._cur], next_padded)
        return bs.loss

    def on_stream(self):
        """Context manager that makes the replay stream current: a training loop run inside it pays no per-step
        cross-stream event waits (`torch.cuda.synchronize()` / `stream.synchronize()` still bracket it as usual)."""
        return torch.cuda.stream(self.replay_stream)
