"""Gradient clipping + Adam for a whole model in two launches (``phc_clip_adam_step``).

The reference's optimizer step is ``clip_grad_norm_(model.parameters(), max_norm)`` followed by
``torch.optim.Adam(lr, weight_decay=0).step()`` (benchmarks/train_hiv.py:198-201, :285).  PyTorch executes that
as ~21 multi-tensor launches over the ~230 small parameter tensors of a PHC model; ``ClipAdam`` does the same
arithmetic (fp32, torch's fused-Adam formulas) with one sums-of-squares launch and one update launch whose workgroups derive the
norm / clip coefficient from the partial sums themselves.  There is no CPU path."""
from __future__ import annotations

import ctypes as C
from typing import Iterable, List, Optional

import torch

from . import lib as L


# clip + Adam as ONE launch whose workgroups exchange the sums of squares inside it (csrc/optim.hip, round 6); 0 = the two launches
FUSED_STEP = __import__("os").environ.get("PHC_CLIP_ADAM_FUSED", "1") != "0"


class ClipAdam:
    """Drop-in for the (clip_grad_norm_, Adam) pair of the training loop.  API: ``zero_grad(set_to_none=True)``,
    ``step(max_norm=None)`` (returns the device scalar of the total gradient norm), ``state_dict`` /
    ``load_state_dict``.  Parameters must be contiguous fp32 tensors on one ROCm device."""

    def __init__(self, params: Iterable[torch.Tensor], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 max_norm: float = 0.0, write_back_clipped: bool = True):
        self.params: List[torch.Tensor] = [p for p in params]
        if not self.params:
            raise ValueError("ClipAdam: no parameters")
        self.lr, self.betas, self.eps, self.max_norm = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(max_norm)
        self.write_back_clipped = bool(write_back_clipped)
        lib = L.load()
        self.device = L.require_device(*self.params)
        for p in self.params:
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise L.PhcError("ClipAdam: parameters must be contiguous fp32 tensors")
        self.group_size = lib.phc_clip_adam_max_tensors()
        chunk = lib.phc_clip_adam_chunk()
        total = sum(p.numel() for p in self.params)
        dev = self.device
        self.exp_avg_flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq_flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg, self.exp_avg_sq = [], []
        off = 0
        for p in self.params:
            self.exp_avg.append(self.exp_avg_flat[off:off + p.numel()].view_as(p))
            self.exp_avg_sq.append(self.exp_avg_sq_flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.step_count = torch.zeros(len(self.params), dtype=torch.float32, device=dev)   # per tensor, as torch.optim.Adam
        self.scalars = torch.zeros(2, dtype=torch.float32, device=dev)        # total norm | clip coefficient
        # groups of <= group_size tensors: device tables {param, m, v, numel} and chunk lists {tensor-in-group, offset}
        self.groups = []
        chunk_base = 0
        for g0 in range(0, len(self.params), self.group_size):
            idx = list(range(g0, min(g0 + self.group_size, len(self.params))))
            table = torch.tensor([[self.params[i].data_ptr(), self.exp_avg[i].data_ptr(), self.exp_avg_sq[i].data_ptr(),
                                   self.params[i].numel()] for i in idx], dtype=torch.int64).to(dev)
            chunks = [(t, o) for t, i in enumerate(idx) for o in range(0, max(self.params[i].numel(), 1), chunk)]
            self.groups.append(dict(idx=idx, table=table, chunks=torch.tensor(chunks, dtype=torch.int32).to(dev),
                                    num_chunks=len(chunks), chunk_base=chunk_base, steps=self.step_count[g0:g0 + len(idx)]))
            chunk_base += len(chunks)
        self.total_chunks = chunk_base
        self.partial = torch.empty(self.total_chunks, dtype=torch.float32, device=dev)
        # the one-launch form (phc_clip_adam_fused): one parameter group whose chunks fit a co-resident launch
        self._fused_ws = None
        if FUSED_STEP and len(self.groups) == 1 and lib.phc_clip_adam_fused_supported(len(self.params), self.total_chunks):
            self._fused_ws = torch.zeros(lib.phc_clip_adam_fused_ws_bytes(self.total_chunks), dtype=torch.uint8, device=dev)
        self._ptrs = [p.data_ptr() for p in self.params]

    # -- torch.optim-like surface ------------------------------------------------------------------
    def zero_grad(self, set_to_none: bool = True) -> None:
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def _check_tables(self) -> None:
        if any(p.data_ptr() != q for p, q in zip(self.params, self._ptrs)):
            raise L.PhcError("ClipAdam: a parameter's storage moved (module.to(), load_state_dict(assign=True)); "
                             "construct the optimizer after the model is on its device")

    def _grad_pointers(self, grp):
        """Device pointers of a group's ``.grad`` tensors (None = no gradient), validated: the kernels read them as dense fp32
        of the parameter's shape, anything else raises instead of being read as garbage (``step`` and ``pack_grads`` alike)."""
        arr = (C.c_void_p * len(grp["idx"]))()
        for t, i in enumerate(grp["idx"]):
            p = self.params[i]
            g = p.grad
            if g is None:
                arr[t] = None
                continue
            if g.dtype != torch.float32 or g.shape != p.shape or g.device != p.device:
                raise L.PhcError(f"ClipAdam: gradient of parameter {i} is {g.dtype} {tuple(g.shape)} on {g.device}, expected "
                                 f"torch.float32 {tuple(p.shape)} on {p.device}")
            if not g.is_contiguous():
                g = g.contiguous()
                p.grad = g
            arr[t] = g.data_ptr()
        return arr

    @torch.no_grad()
    def step(self, max_norm: Optional[float] = None) -> torch.Tensor:
        lib = L.load()
        self._check_tables()
        mn = self.max_norm if max_norm is None else float(max_norm)
        stream = L.stream_ptr(self.device)
        calls = [(grp, self._grad_pointers(grp)) for grp in self.groups]
        single = len(calls) == 1

        def call(grp, arr, phases):
            L.check(lib.phc_clip_adam_step(L.ptr(grp["table"]), arr, len(grp["idx"]), L.ptr(grp["chunks"]), grp["num_chunks"],
                                           grp["chunk_base"], self.total_chunks, phases, mn, self.lr, self.betas[0],
                                           self.betas[1], self.eps, L.ptr(grp["steps"]), L.ptr(self.scalars),
                                           L.ptr(self.partial), 1 if self.write_back_clipped else 0, stream),
                    "phc_clip_adam_step")
        if single and self._fused_ws is not None:
            grp, arr = calls[0]
            L.check(lib.phc_clip_adam_fused(L.ptr(grp["table"]), arr, len(grp["idx"]), L.ptr(grp["chunks"]), self.total_chunks, mn, self.lr,
                                            self.betas[0], self.betas[1], self.eps, L.ptr(grp["steps"]), L.ptr(self.scalars),
                                            L.ptr(self._fused_ws), self._fused_ws.numel(), 1 if self.write_back_clipped else 0, stream),
                    "phc_clip_adam_fused")
        elif single:
            call(calls[0][0], calls[0][1], 5)               # sums of squares, then the update (which derives the norm itself)
        else:
            for grp, arr in calls:
                call(grp, arr, 1)
            for grp, arr in calls:
                call(grp, arr, 4)
        return self.scalars[0]

    # -- data-parallel gradient bucket (phc_gnn_amd.dist.FlatGradSync) ---------------------------------
    def grad_bucket(self) -> torch.Tensor:
        """The flat fp32 buffer (same layout as the moment buffers) the gradients are packed into / reduced in / read from."""
        if getattr(self, "_grad_flat", None) is None:
            self._grad_flat = torch.zeros_like(self.exp_avg_flat)
            self._grad_views, off = [], 0
            for p in self.params:
                self._grad_views.append(self._grad_flat[off:off + p.numel()].view_as(p))
                off += p.numel()
        return self._grad_flat

    @torch.no_grad()
    def pack_grads(self, scale: float = 1.0) -> torch.Tensor:
        """All ``.grad`` tensors -> the bucket, times ``scale``, one launch per parameter group (a missing gradient packs zeros)."""
        lib = L.load()
        self._check_tables()
        flat = self.grad_bucket()
        stream = L.stream_ptr(self.device)
        for grp in self.groups:
            arr = self._grad_pointers(grp)
            L.check(lib.phc_pack_grads(L.ptr(grp["table"]), arr, len(grp["idx"]), L.ptr(grp["chunks"]), grp["num_chunks"],
                                       L.ptr(self.exp_avg_flat), L.ptr(flat), float(scale), stream), "phc_pack_grads")
        return flat

    def adopt_bucket_grads(self) -> None:
        """``p.grad`` := the parameter's view of the bucket (no copy, no launch): the optimizer step reads the reduced values in place."""
        self.grad_bucket()
        for p, v in zip(self.params, self._grad_views):
            p.grad = v

    def state_dict(self) -> dict:
        return dict(step=self.step_count.clone(), exp_avg=self.exp_avg_flat.clone(), exp_avg_sq=self.exp_avg_sq_flat.clone(),
                    lr=self.lr, betas=self.betas, eps=self.eps, max_norm=self.max_norm)

    def load_state_dict(self, sd: dict) -> None:
        self.step_count.copy_(sd["step"])
        self.exp_avg_flat.copy_(sd["exp_avg"])
        self.exp_avg_sq_flat.copy_(sd["exp_avg_sq"])
        self.lr, self.betas, self.eps, self.max_norm = float(sd["lr"]), tuple(sd["betas"]), float(sd["eps"]), float(sd["max_norm"])
