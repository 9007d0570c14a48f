"""Differentiable operators of the PHC message-passing path, each a thin
``torch.autograd.Function`` over the C ABI of libphc_hip.so.

These are the engine-side counterparts of the reference's functional layer:
``matvec_product_new`` (phc/hypercomplex/layers.py:198-219), PyG
``propagate(aggr='add')`` as used by PHMConv / PHMGINEConv
(phc/hypercomplex/undirectional/messagepassing.py:55-74,132-146) and
``global_add_pool`` (phc/hypercomplex/pooling.py:14).
All inputs must be fp32 tensors on a ROCm device; there is no CPU path.
"""
from __future__ import annotations

import contextlib
import threading
from typing import Optional

import torch

from . import lib as L
from .graph import EdgeStructure, SegmentStructure, edge_structure, segment_structure

_SLOW_PATHS_SEEN: set = set()


def warn_slow_path(tag: str, what: str) -> None:
    """One RuntimeWarning per process and tag: a module of the ``phc`` mirror is about to run on plain PyTorch ops because the
    engine has no kernel for that case (the results are the reference's; only the speed differs)."""
    if tag not in _SLOW_PATHS_SEEN:
        _SLOW_PATHS_SEEN.add(tag)
        import warnings
        warnings.warn(f"phc: {what} runs on PyTorch ops, not on the HIP engine", RuntimeWarning, stacklevel=3)


def _act_id(name: str) -> int:
    try:
        return L.ACT_IDS[name.lower()]
    except KeyError:
        raise L.PhcError(f"unknown message activation '{name}' (known: {sorted(L.ACT_IDS)})") from None


# ---------------------------------------------------------------------------
# Deferred parameter gradients
# ---------------------------------------------------------------------------
# The dW / dA / db chain of a PHM layer (dH = xᵀ·dy as split-K slabs -> reduce -> contract) is off the critical
# path of the backward pass: nothing but the optimizer reads it.  At molecule scale one layer's chain cannot fill
# 256 CUs, so inside ``deferred_param_grads()`` every layer's backward only runs its dx GEMM, keeps (x, dy) alive
# and files a job; when the block exits, ALL layers' parameter gradients are produced by three batched launches
# (phc_phm_param_grads) and written to ``param.grad`` (accumulating into an existing ``.grad``, like autograd).
# Only leaf parameters are deferred; anything else (a weight that is itself computed, double backward, a backward
# that runs after the block has exited) takes the per-layer path.  Tensor hooks on deferred parameters do not fire.
class _Deferred:
    def __init__(self):
        self.active = False
        self.jobs = []
        self.table_jobs = []
        self.atom_jobs = []
        self.beta_jobs = []
        self.reg = None                    # a weight regulariser waiting for the step tail (request_weight_regulariser)
        self.lock = threading.Lock()       # the backward pass runs on autograd's device thread


_DEFER = _Deferred()


class _AutoStep:                           # (process-wide, like the contexts it opens: the backward's hooks run on autograd's device thread)
    stack = None                           # the ExitStack of a step block a model forward opened by itself (auto_step_forward)
    in_forward = False


_AUTO = _AutoStep()


def _deferred_reset() -> None:
    _DEFER.jobs, _DEFER.table_jobs, _DEFER.atom_jobs, _DEFER.beta_jobs, _DEFER.reg = [], [], [], [], None


def _deferred_flush() -> None:
    """Everything filed since the block opened -> the batched launches; the block is closed afterwards."""
    _DEFER.active = False
    jobs, _DEFER.jobs = _DEFER.jobs, []
    tjobs, _DEFER.table_jobs = _DEFER.table_jobs, []
    ajobs, _DEFER.atom_jobs = _DEFER.atom_jobs, []
    bjobs, _DEFER.beta_jobs = _DEFER.beta_jobs, []
    reg, _DEFER.reg = _DEFER.reg, None
    if bjobs:
        _flush_beta_grads(bjobs)
    if STEP_TAIL_MERGED and len(ajobs) > 1:
        # several row-contiguous encoder jobs (a model whose bond encoders run on the [E, h] tensor path, e.g. softmax
        # aggregation): the tail launch takes the largest, the others run on their own first
        ajobs.sort(key=lambda aj: aj["idx"].shape[0] * aj["ts"].R)
        _flush_encoder_grads(ajobs[:-1])
        ajobs = ajobs[-1:]
    if STEP_TAIL_MERGED and _flush_step_tail(jobs, tjobs, ajobs, reg):
        return
    if tjobs:
        _flush_table_grads(tjobs)
    _flush_encoder_grads(ajobs)
    if jobs:
        _flush_param_grads(jobs)
    if reg is not None:                                      # the regulariser could not ride in a step tail: its own launch
        _weight_reg_launch(reg["Ws"], reg["coef"], reg["p"], reg["loss"])


@contextlib.contextmanager
def deferred_param_grads():
    """Wrap forward + ``loss.backward()`` of a training step; parameter gradients of the PHM layers are complete
    when the block exits."""
    if _AUTO.stack is not None and not _AUTO.in_forward:
        close_auto_step(flush=False)                         # (a block a forward opened by itself whose backward never came)
    if _DEFER.active:                       # nested: the outermost block flushes
        yield
        return
    _DEFER.active = True
    _deferred_reset()
    try:
        yield
    except BaseException:
        _DEFER.active = False
        _deferred_reset()
        raise
    _deferred_flush()


# the explicit weight regulariser and the gate map's parameter gradients inside the step tail's last two launches
TAIL_EXTRAS = __import__("os").environ.get("PHC_TAIL_EXTRAS", "1") != "0"
# 256-wide layers inside the fused step tail (register-resident 128 x 64 tiles beside the table-gradient workgroups, regulariser and
# gate-map gradients as passengers) instead of phc_phm_param_grads + separate table / regulariser / gate launches
TAIL_WIDE = __import__("os").environ.get("PHC_TAIL_WIDE", "1") != "0"


def request_weight_regulariser(Ws, coef: float, p: int, loss: Optional[torch.Tensor]) -> bool:
    """Inside ``deferred_param_grads``: park ``coef * sum_W mean ||W[:, kappa, pi]||_p`` for the step tail — its gradient is
    added to the dW tensors as the tail writes them, its value to ``loss`` (in place), without a launch of its own.  False: no
    deferred block is open (or the switch is off) and the caller runs ``phm_weight_regularization_``."""
    if not (_DEFER.active and TAIL_EXTRAS):
        return False
    _DEFER.reg = dict(Ws=list(Ws), coef=float(coef), p=int(p), loss=loss)
    return True


# bond-table gradients of identity-message layers from per-node in-edge counts instead of per-edge gathers
TABLE_GRADS_COUNTED = __import__("os").environ.get("PHC_TABLE_GRADS_COUNTED", "1") != "0"

# all deferred gradients of a step (PHM / dense layers, bond tables, atom tables) in three launches (phc_step_tail)
STEP_TAIL_MERGED = __import__("os").environ.get("PHC_STEP_TAIL_MERGED", "1") != "0"


def _assign_grads(params, grads) -> None:
    for prm, g in zip(params, grads):
        if prm is None or g is None:
            continue
        if prm.grad is None:
            prm.grad = g
        else:
            prm.grad.add_(g)


def _flush_atom_grads(aj) -> None:
    """Atom-encoder table gradients on their own (phc_encode_bwd)."""
    lib = L.load()
    ts, idx_, dout_ = aj["ts"], aj["idx"], aj["dout"]
    dev = dout_.device
    N = idx_.shape[0]
    with torch.cuda.device(dev):
        flat = torch.empty(ts.n * ts.R * ts.k, dtype=torch.float32, device=dev)
        ws_bytes = lib.phc_encode_bwd_ws_bytes(N, ts.F, ts.vocab_c, ts.n, ts.k)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        L.check(lib.phc_encode_bwd(L.ptr(dout_), L.ptr(idx_), N, ts.F, ts.vocab_c, ts.n, ts.k, L.ptr(flat), L.ptr(ws), ws_bytes,
                                   L.stream_ptr(dev)), "phc_encode_bwd")
        if L.PROF_ON:
            L.account("table_grad", 1.0 * N * ts.F * ts.n * ts.k, N * (8.0 * ts.F + 4.0 * ts.n * ts.k))
    _assign_grads(aj["params"], ts.grad_views(flat))


def _flush_beta_grads(bjobs) -> None:
    """d loss / d beta of the softmax-aggregation layers: their per-workgroup partial sums -> scalars, 16 per launch."""
    lib = L.load()
    for c0 in range(0, len(bjobs), 16):
        chunk = bjobs[c0:c0 + 16]
        dev = chunk[0][0].device
        with torch.cuda.device(dev):
            out = torch.empty(len(chunk), dtype=torch.float32, device=dev)
            srcs = (_C.c_void_p * len(chunk))(*[L.ptr(p) for p, _ in chunk])
            lens = (_C.c_int32 * len(chunk))(*[p.numel() for p, _ in chunk])
            outs = (_C.c_void_p * len(chunk))(*[L.ptr(out) + 4 * i for i in range(len(chunk))])
            L.check(lib.phc_vec_sums(srcs, lens, outs, len(chunk), L.stream_ptr(dev)), "phc_vec_sums")
        _assign_grads([b for _, b in chunk], [out[i].reshape(b.shape) for i, (_, b) in enumerate(chunk)])


def _flush_encoder_grads(ajobs) -> None:
    """Table gradients of encoders on the tensor path: the encoders that read one index tensor (the bond encoders of
    all layers) go through phc_encode_bwd_batched — two launches for the group — the rest one by one."""
    groups = {}
    for aj in ajobs:
        ts = aj["ts"]
        groups.setdefault((aj["idx"].data_ptr(), tuple(aj["idx"].shape), ts.F, ts.n, ts.k, tuple(ts.vocab), aj["dout"].device), []).append(aj)
    lib = L.load()
    for js in groups.values():
        if len(js) == 1:
            _flush_atom_grads(js[0])
            continue
        ts, idx_ = js[0]["ts"], js[0]["idx"]
        dev = js[0]["dout"].device
        N = idx_.shape[0]
        with torch.cuda.device(dev):
            per = ts.n * ts.R * ts.k
            flat = torch.empty(len(js) * per, dtype=torch.float32, device=dev)
            arr = (L.TableGradJob * len(js))()
            for i, aj in enumerate(js):
                arr[i] = L.TableGradJob(L.ptr(aj["dout"]), None, None, L.ptr(flat) + 4 * i * per)
            ws_bytes = lib.phc_encode_bwd_batched_ws_bytes(N, ts.F, ts.vocab_c, ts.n, ts.k, len(js))
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            L.check(lib.phc_encode_bwd_batched(arr, len(js), L.ptr(idx_), N, ts.F, ts.vocab_c, ts.n, ts.k, L.ptr(ws), ws_bytes,
                                               L.stream_ptr(dev)), "phc_encode_bwd_batched")
            if L.PROF_ON:
                for _ in js:
                    L.account("table_grad", 1.0 * N * ts.F * ts.n * ts.k, N * (8.0 * ts.F + 4.0 * ts.n * ts.k))
        for i, aj in enumerate(js):
            _assign_grads(aj["params"], aj["ts"].grad_views(flat[i * per:(i + 1) * per]))


# Order of the dH family's jobs inside the step tail.  Workgroups start in job order, two per CU, and a node-level job's workgroup
# runs six times longer than one of the head's 1 056-row jobs.  The one-launch heads queue their jobs (in forward order) when THEIR
# backward runs, i.e. first in the list: measured, those belong at the END ("head_last": zinc_pub_phc4_4x180 1.479 -> 1.426 ms once
# its head was fused, molpcba_b1024 2.071 -> 2.064, molhiv_b512 0.577 -> 0.567, zinc_b1024 and the 2 x 200 model unchanged).  A
# plain sort by rows ("desc") does the same for them but also moves the per-layer head's LDS-tiled jobs of zinc_pub_phc5_4x200 (130
# columns: not register-resident) behind everything: 1.575 -> 1.606 ms; "asc" equals the order of arrival everywhere.  Empirical.
TAIL_JOB_ORDER = __import__("os").environ.get("PHC_TAIL_JOB_ORDER", "head_last")   # "none", "head_last", "asc", "desc"


def _flush_step_tail(jobs, tjobs, ajobs, reg=None) -> bool:
    """Everything in one phc_step_tail call when the shapes allow it; False: the caller runs the separate flushes (``reg``, a
    parked weight regulariser, is then still the caller's to launch)."""
    if not jobs or len(jobs) > 16 or len(ajobs) > 1:
        return False
    dev = jobs[0]["x"].device
    if any(jb["x"].device != dev for jb in jobs):
        return False
    for jb in jobs:                                         # 128 x 128 tile jobs go through phc_phm_param_grads
        n, kin, kout = _job_dims(jb)
        if jb["x"].shape[0] >= 65536 or (n * kin >= 256 and n * kout >= 256 and not TAIL_WIDE):
            return False
    keys = {tj["key"] for tj in tjobs}
    if len(keys) > 1 or len(tjobs) > 8:
        return False
    ts_b = tjobs[0]["ts"] if tjobs else None
    ts_a = ajobs[0]["ts"] if ajobs else None
    if ts_b is None and ts_a is None:
        return False
    n, k = (ts_b.n, ts_b.k) if ts_b is not None else (ts_a.n, ts_a.k)
    if (ts_a is not None and (ts_a.n, ts_a.k) != (n, k)) or (tjobs and tjobs[0]["bond"].device != dev) or \
            (ajobs and ajobs[0]["dout"].device != dev):
        return False
    lib = L.load()
    if TAIL_JOB_ORDER in ("asc", "desc"):
        jobs.sort(key=lambda jb: jb["x"].shape[0], reverse=(TAIL_JOB_ORDER == "desc"))
    elif TAIL_JOB_ORDER == "head_last":
        jobs.sort(key=lambda jb: 1 if jb.get("late") else 0)          # (stable: everything else keeps its order of arrival)
    with torch.cuda.device(dev):
        # PHM / dense jobs
        total = 0
        for jb in jobs:
            jn, jkin, jkout = _job_dims(jb)
            jb["off"] = total
            total += jn * jkin * jkout + jn ** 3 + jn * jkout
        flat = torch.empty(total, dtype=torch.float32, device=dev)
        arr = (L.GradJob * len(jobs))()
        outs = []
        for i, jb in enumerate(jobs):
            jn, jkin, jkout = _job_dims(jb)
            o = jb["off"]
            dW = flat[o:o + jn * jkin * jkout].view(jn, jkin, jkout)
            dA = flat[o + jn * jkin * jkout:o + jn * jkin * jkout + jn ** 3].view(jn, jn, jn)
            db = flat[o + jn * jkin * jkout + jn ** 3:o + jn * jkin * jkout + jn ** 3 + jn * jkout]
            need_A, need_W, need_b = jb["need"]
            if jb.get("dense"):
                dW = dW.view(jkout, jkin)
            xf = jb.get("xf")
            arr[i] = L.GradJob(L.ptr(jb["x"]), L.ptr(jb["dy"]), L.ptr(jb["A"]), L.ptr(jb["W"]), L.ptr(dW) if need_W else None,
                               L.ptr(dA) if need_A else None, L.ptr(db) if need_b else None, jb["x"].shape[0], jn, jkin, jkout,
                               1 if jb.get("dense") else 0, L.ptr(xf[0]) if xf else None, L.ptr(xf[1]) if xf else None,
                               xf[2] if xf else 0, 0)
            outs.append((dA if need_A else None, dW if need_W else None, db if need_b else None))
        # ---- extras of the tail's last two launches: the gate map's parameter gradients, the weight regulariser ----
        extras, keep, gate_i, gate_out = None, [], -1, None
        if TAIL_EXTRAS:
            gate_jobs = [i for i, jb in enumerate(jobs) if jb.get("gate") is not None]
            ex = L.TailExtras()
            ex.gate_job = -1
            if len(gate_jobs) == 1:
                gate_i = gate_jobs[0]
                gt = jobs[gate_i]["gate"]
                gn, gkin, gkout = gt["W"].shape
                nA, nW, nb, nWg, nbg = gt["need"]
                gdA = torch.empty_like(gt["A"]) if nA else None
                gdW = torch.empty_like(gt["W"]) if nW else None
                gdb = torch.empty(gn * gkout, dtype=torch.float32, device=dev) if nb else None
                gdWg = torch.empty_like(gt["Wg"]) if nWg else None
                dbg = outs[gate_i][2]                                  # the dense job's bias gradient = sum dg
                arr[gate_i].dW = None                                  # dM^T is read from the job's dH in place: nobody needs it written
                ex.gate_job, ex.gate_n, ex.gate_kin, ex.gate_kout, ex.gate_kg = gate_i, gn, gkin, gkout, gt["Wg"].shape[0]
                ex.gate_H, ex.gate_w, ex.gate_phm_b, ex.gate_A, ex.gate_W = L.ptr(gt["H"]), L.ptr(gt["Wg"]), L.ptr(gt["b"]), L.ptr(gt["A"]), L.ptr(gt["W"])
                ex.gate_dA, ex.gate_dW, ex.gate_db, ex.gate_dWg = L.ptr(gdA), L.ptr(gdW), L.ptr(gdb), L.ptr(gdWg)
                gate_out = (gdA, gdW, gdb, gdWg, dbg if nbg else None)
                extras = ex
            if reg is not None:
                Ws = reg["Ws"]
                by_ptr = {jb["W"].data_ptr(): i for i, jb in enumerate(jobs) if not jb.get("dense") and jb["need"][1]}
                where = []
                for w in Ws:
                    if gate_i >= 0 and w.data_ptr() == jobs[gate_i]["gate"]["W"].data_ptr() and jobs[gate_i]["gate"]["need"][1]:
                        where.append(-2)
                    else:
                        where.append(by_ptr.get(w.data_ptr(), -1))
                ok = all(j != -1 for j in where) and 0 < len(Ws) <= lib.phc_tail_reg_max_jobs() and reg["p"] in (1, 2) and \
                    all(w.dim() == 3 and w.shape[0] <= 16 and w.dtype == torch.float32 and w.is_contiguous() and w.device == dev for w in Ws)
                if ok:
                    k_ = len(Ws)
                    W_arr = (_C.c_void_p * k_)(*[w.data_ptr() for w in Ws])
                    n_arr = (_C.c_int32 * k_)(*[w.shape[0] for w in Ws])
                    c_arr = (_C.c_int32 * k_)(*[w.shape[1] * w.shape[2] for w in Ws])
                    j_arr = (_C.c_int32 * k_)(*where)
                    keep += [W_arr, n_arr, c_arr, j_arr]
                    ex.reg_num, ex.reg_p, ex.reg_coef = k_, reg["p"], reg["coef"]
                    ex.reg_W, ex.reg_n, ex.reg_cells, ex.reg_job = (_C.cast(a_, _C.c_void_p) for a_ in (W_arr, n_arr, c_arr, j_arr))
                    reg_value = torch.empty((), dtype=torch.float32, device=dev)
                    reg_ws = torch.empty(lib.phc_tail_extras_ws_floats(_C.byref(ex)), dtype=torch.float32, device=dev)
                    keep += [reg_value, reg_ws]
                    ex.reg_value, ex.reg_loss_inout, ex.reg_ws = L.ptr(reg_value), L.ptr(reg["loss"]), L.ptr(reg_ws)
                    extras = ex
                    reg = None                                         # consumed
        # bond tables
        tarr, tflat, per = None, None, 0
        E = F_b = act = b_ndst = 0
        bond = src = dst = vocab_b = b_codes = b_rowptr = b_counts = None
        if tjobs:
            st = tjobs[0]["st"]
            per = ts_b.n * ts_b.R * ts_b.k
            tflat = torch.empty(len(tjobs) * per, dtype=torch.float32, device=dev)
            tarr = (L.TableGradJob * len(tjobs))()
            for i, tj in enumerate(tjobs):
                tarr[i] = L.TableGradJob(L.ptr(tj["dout"]), L.ptr(tj["x"]), L.ptr(tj["ts"].ptrs), L.ptr(tflat) + 4 * i * per)
            E, F_b, act = st.num_edges, ts_b.F, tjobs[0]["act"]
            bond, src, dst, vocab_b = tjobs[0]["bond"], st._src, st._dst, ts_b.vocab_c
            if tjobs[0].get("codes") is not None:
                b_codes, b_rowptr, b_ndst = tjobs[0]["codes"], st.csr[0], st.num_dst
                cc = getattr(st, "_code_counts", None)       # per-destination table-row counts out of the pair build
                if cc is not None and cc[0] is b_codes:
                    b_counts = cc[1]
        # atom tables
        aflat = None
        N_a = F_a = 0
        a_dout = a_idx = vocab_a = None
        if ajobs:
            aj = ajobs[0]
            aflat = torch.empty(ts_a.n * ts_a.R * ts_a.k, dtype=torch.float32, device=dev)
            N_a, F_a, a_dout, a_idx, vocab_a = aj["idx"].shape[0], ts_a.F, aj["dout"], aj["idx"], ts_a.vocab_c
        ws_bytes = lib.phc_step_tail_ws_bytes(arr, len(jobs), len(tjobs), E, b_ndst, F_b, vocab_b, N_a, F_a, vocab_a, n, k)
        if ws_bytes == 0:
            return False
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        L.check(lib.phc_step_tail_ex(arr, len(jobs), tarr, len(tjobs), L.ptr(bond), vocab_b, L.ptr(src), L.ptr(dst), E, F_b, act,
                                     L.ptr(b_rowptr), L.ptr(b_codes), L.ptr(b_counts), b_ndst, L.ptr(a_dout), L.ptr(a_idx), N_a, F_a, vocab_a,
                                     L.ptr(aflat), n, k, L.ptr(ws), ws_bytes, _C.byref(extras) if extras is not None else None,
                                     L.stream_ptr(dev)), "phc_step_tail_ex")
        if L.PROF_ON:
            for jb in jobs:
                jn, jkin, jkout = _job_dims(jb)
                N = jb["x"].shape[0]
                L.account("phm_dH_partial", 2.0 * N * jn * jkin * jn * jkout, 4.0 * N * jn * (jkin + jkout) + 4.0 * jn * jkin * jn * jkout)
    for i, (jb, grads) in enumerate(zip(jobs, outs)):
        if i == gate_i:                                                # the gate map's gradients came out of the tail itself
            _assign_grads(jb["gate"]["params"], gate_out)
            continue
        _assign_grads(jb["params"], grads)
        if jb.get("post") is not None:
            jb["post"](grads)
    for i, tj in enumerate(tjobs):
        _assign_grads(tj["params"], tj["ts"].grad_views(tflat[i * per:(i + 1) * per]))
    if ajobs:
        _assign_grads(ajobs[0]["params"], ts_a.grad_views(aflat))
    if reg is not None:                                                # a parked regulariser this tail could not take
        _weight_reg_launch(reg["Ws"], reg["coef"], reg["p"], reg["loss"])
    return True


def _flush_table_grads(jobs) -> None:
    """Bond-table gradients of all message-passing layers that share one edge list: two launches per group."""
    lib = L.load()
    groups = {}
    for jb in jobs:
        groups.setdefault(jb["key"], []).append(jb)
    for js in groups.values():
        j0 = js[0]
        ts, st, bond = j0["ts"], j0["st"], j0["bond"]
        dev = bond.device
        with torch.cuda.device(dev):
            per = ts.n * ts.R * ts.k
            flat = torch.empty(len(js) * per, dtype=torch.float32, device=dev)
            arr = (L.TableGradJob * len(js))()
            for i, jb in enumerate(js):
                arr[i] = L.TableGradJob(L.ptr(jb["dout"]), L.ptr(jb["x"]), L.ptr(jb["ts"].ptrs), L.ptr(flat) + 4 * i * per)
            if j0.get("codes") is not None:
                ws_bytes = lib.phc_table_grads_counted_ws_bytes(st.num_dst, ts.F, ts.vocab_c, ts.n, ts.k, len(js))
                ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
                L.check(lib.phc_table_grads_counted(arr, len(js), L.ptr(st.csr[0]), L.ptr(j0["codes"]), st.num_dst, st.num_edges,
                                                    ts.vocab_c, ts.F, ts.n, ts.k, L.ptr(ws), ws_bytes, L.stream_ptr(dev)),
                        "phc_table_grads_counted")
            else:
                ws_bytes = lib.phc_table_grads_ws_bytes(st.num_edges, ts.F, ts.vocab_c, ts.n, ts.k, len(js))
                ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
                L.check(lib.phc_table_grads_batched(arr, len(js), L.ptr(bond), ts.vocab_c, L.ptr(st._src), L.ptr(st._dst),
                                                    st.num_edges, ts.F, ts.n, ts.k, j0["act"], L.ptr(ws), ws_bytes,
                                                    L.stream_ptr(dev)), "phc_table_grads_batched")
            if L.PROF_ON:
                E, Nd, h = st.num_edges, st.num_dst, ts.n * ts.k
                for _ in js:
                    L.account("table_grad", 1.0 * E * ts.F * h, Nd * 4.0 * h + E * (4.0 + 8.0 * ts.F) + 4.0 * (Nd + 1))
            for i, jb in enumerate(js):
                views = jb["ts"].grad_views(flat[i * per:(i + 1) * per])
                for prm, g in zip(jb["params"], views):
                    if prm is None:
                        continue
                    if prm.grad is None:
                        prm.grad = g
                    else:
                        prm.grad.add_(g)


def _job_dims(jb):
    """(n, kin, kout) of a parameter-gradient job; a dense job carries the nn.Linear weight [out, in]."""
    if jb.get("dense"):
        fo, fi = jb["W"].shape
        return 1, fi, fo
    return tuple(jb["W"].shape)


def _flush_param_grads(jobs, return_grads: bool = False):
    lib = L.load()
    by_dev = {}
    for jb in jobs:
        by_dev.setdefault(jb["x"].device, []).append(jb)
    for dev, js in by_dev.items():
        total = 0
        for jb in js:
            n, kin, kout = _job_dims(jb)
            jb["off"] = total
            total += n * kin * kout + n * n * n + n * kout
        with torch.cuda.device(dev):
            flat = torch.empty(total, dtype=torch.float32, device=dev)
            arr = (L.GradJob * len(js))()
            outs = []
            for i, jb in enumerate(js):
                n, kin, kout = _job_dims(jb)
                o = jb["off"]
                dW = flat[o:o + n * kin * kout].view(n, kin, kout)
                dA = flat[o + n * kin * kout:o + n * kin * kout + n ** 3].view(n, n, n)
                db = flat[o + n * kin * kout + n ** 3:o + n * kin * kout + n ** 3 + n * kout]
                need_A, need_W, need_b = jb["need"]
                if jb.get("dense"):
                    dW = dW.view(kout, kin)                # written transposed: the nn.Linear layout [out, in]
                xf = jb.get("xf")
                arr[i] = L.GradJob(L.ptr(jb["x"]), L.ptr(jb["dy"]), L.ptr(jb["A"]), L.ptr(jb["W"]),
                                   L.ptr(dW) if need_W else None, L.ptr(dA) if need_A else None,
                                   L.ptr(db) if need_b else None, jb["x"].shape[0], n, kin, kout, 1 if jb.get("dense") else 0,
                                   L.ptr(xf[0]) if xf else None, L.ptr(xf[1]) if xf else None, xf[2] if xf else 0, 0)
                outs.append((dA if need_A else None, dW if need_W else None, db if need_b else None))
            ws_bytes = lib.phc_phm_param_grads_ws_bytes(arr, len(js))
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            L.check(lib.phc_phm_param_grads(arr, len(js), L.ptr(ws), ws_bytes, L.stream_ptr(dev)), "phc_phm_param_grads")
            if L.PROF_ON:
                for jb in js:
                    n, kin, kout = _job_dims(jb)
                    N = jb["x"].shape[0]
                    L.account("phm_dH_partial", 2.0 * N * n * kin * n * kout, 4.0 * N * n * (kin + kout) + 4.0 * n * kin * n * kout)
            if return_grads:
                return outs
            for jb, grads in zip(js, outs):
                if jb.get("post") is not None:
                    jb["post"](grads)
                for prm, g in zip(jb["params"], grads):
                    if prm is None or g is None:
                        continue
                    if prm.grad is None:
                        prm.grad = g
                    else:
                        prm.grad.add_(g)


# ---------------------------------------------------------------------------
# Prebuilt operand matrices
# ---------------------------------------------------------------------------
# H = Σ_i A_i ⊗ W_i only changes when the optimizer steps.  Inside ``prebuilt_H(pairs)`` H and Hᵀ of every listed
# (phm_rule, W) pair are assembled by ONE launch and each PHM layer's forward / dx GEMM loads its operand panel
# from them (phc_gemm_prebuilt) instead of re-assembling it from A and W in every workgroup.  The block must not
# span an in-place update of the weights (one training step: forward + backward, optimizer step after it).
class _Prebuilt:
    encoded = None        # (idx data_ptr, idx shape, table-pointer array data_ptr, out): the atom encoding the build launch produced

    def __init__(self):
        self.table = {}


_PREBUILT = _Prebuilt()


def phm_pairs(model) -> list:
    """(phm_rule, W) of every PHM linear layer of ``model`` (modules exposing both as fp32 parameters / tensors)."""
    pairs = []
    for m in model.modules():
        A, W = getattr(m, "phm_rule", None), getattr(m, "W", None)
        if torch.is_tensor(A) and torch.is_tensor(W) and W.dim() == 3 and A.dim() == 3:
            pairs.append((A, W))
    return pairs


def dense_weights(model) -> list:
    """``nn.Linear`` weights [out, in] that the engine's dense map (``dense_linear``) can serve from a prebuilt
    operand: in a multiple of 4, in and out <= 768 (the hypercomplex -> real heads of the read-out; molpcba: 512 -> 128; the
    published 200-wide models: 200 -> 50)."""
    out = []
    for m in model.modules():
        if isinstance(m, torch.nn.Linear) and m.in_features % 4 == 0 and m.in_features <= PREBUILT_MAX_WIDTH \
                and m.out_features <= PREBUILT_MAX_WIDTH:
            out.append(m.weight)
    return out


def gate_specs(model) -> list:
    """(phm_rule, W, b, gate_weight, gate_bias) of every soft-attention read-out of ``model`` whose gate
    ``Linear(PHMLinear(x))`` the engine evaluates as ONE linear map (``gate_linear``; pooling.py:57-66)."""
    out = []
    for m in model.modules():
        lin, rt = getattr(m, "linear", None), getattr(m, "real_trafo", None)
        aff = getattr(rt, "affine", None)
        if lin is not None and isinstance(aff, torch.nn.Linear) and torch.is_tensor(getattr(lin, "W", None)) \
                and torch.is_tensor(getattr(lin, "phm_rule", None)) and gate_composable(lin.W, aff.weight):
            out.append((lin.phm_rule, lin.W, getattr(lin, "b", None), aff.weight, aff.bias))
    return out


def _one(dev) -> torch.Tensor:
    one = _ONES.get(dev)
    if one is None:
        one = torch.ones(1, 1, 1, dtype=torch.float32, device=dev)
        _ONES[dev] = one
    return one


# widest layer side whose operand matrices are prebuilt (phc_gemm_prebuilt[_stats] take K <= 768: the 768-wide head of the
# molpcba configuration); the fused variants (operand transform, backward sums, the H-stationary kernel) stop at 256
PREBUILT_MAX_WIDTH = 768
FUSED_MAX_WIDTH = 256


# the step's atom encoder as passenger workgroups of the operand-build launch (phc_phm_build_H_batched_ex)
ENCODE_FRONT = __import__("os").environ.get("PHC_ENCODE_FRONT", "1") != "0"


def atom_encode_spec(model, data):
    """(idx, TableSet, tables) of ``model.atomencoder`` applied to ``data.x`` when that is the integer / sum encoder the engine
    evaluates in one kernel, else None: what ``prebuilt_H(encode=...)`` needs to run the encoder inside the operand-build launch."""
    enc = getattr(model, "atomencoder", None)
    tables = getattr(enc, "engine_tables", None) if enc is not None else None
    x = getattr(data, "x", None)
    if tables is None or x is None or not x.is_cuda or x.dim() != 2:
        return None
    return x, tableset_for(enc, tables), tables


@contextlib.contextmanager
def prebuilt_H(pairs, dense=(), gates=(), encode=None):
    if _AUTO.stack is not None and not _AUTO.in_forward:
        close_auto_step(flush=False)                       # (a block a forward opened by itself whose backward never came)
    if _PREBUILT.table or not (pairs or dense or gates):   # nested (outermost wins) or nothing to do
        yield
        return
    outer_step = getattr(_FLAT_STEP, "packed", None) is None
    if outer_step:
        _FLAT_STEP.packed = set()                          # flat table images (FlatTableSet) are refreshed once inside this block
        try:
            FlatTableSet.refresh_all()                     # ... all the live ones by one launch per device, right here
        except BaseException:
            _FLAT_STEP.packed = None                       # (a failed refresh must not leave the thread "inside a step")
            raise
    try:
        lib = L.load()
    except BaseException:
        if outer_step:
            _FLAT_STEP.packed = None
        raise
    by_dev = {}
    for w in dense:
        if w.is_cuda and w.dtype == torch.float32 and w.is_contiguous() and w.dim() == 2:
            by_dev.setdefault(w.device, []).append((None, w))
    for A, W in pairs:
        n, kin, kout = W.shape
        if A.is_cuda and W.is_cuda and A.dtype == W.dtype == torch.float32 and A.is_contiguous() and W.is_contiguous() \
                and (n * kout) <= PREBUILT_MAX_WIDTH and (n * kin) <= PREBUILT_MAX_WIDTH and (n * kin) % 4 == 0 and (n * kout) % 4 == 0:
            by_dev.setdefault(W.device, []).append((A, W))
    gates_by_dev = {}
    if GATE_COMPOSED:
        for gs in gates:
            A, W, Wg = gs[0], gs[1], gs[3]
            if A.is_cuda and all(t is None or (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()) for t in gs) \
                    and gate_composable(W, Wg):
                gates_by_dev.setdefault(W.device, []).append(gs)
                by_dev.setdefault(W.device, [])
    try:
        for dev, prs in by_dev.items():
            L.require_device(torch.empty(0, device=dev))   # (first engine op on a device: error word, load-time self check)
            with torch.cuda.device(dev):
                gts = gates_by_dev.get(dev, [])
                # per PHM pair: H, H^T and their k4-interleaved copies (the H-stationary GEMM's operand form)
                total = sum(4 * (W.shape[0] * W.shape[1]) * (W.shape[0] * W.shape[2]) if A is not None else W.numel()
                            for A, W in prs)
                # per gate map: M [in, kg], M^T and the composed bias [kg]
                total += sum(2 * (g[1].shape[0] * g[1].shape[1]) * g[3].shape[0] + g[3].shape[0] + 4 for g in gts)
                flat = torch.empty(total, dtype=torch.float32, device=dev)
                # the step's statistics accumulators are cleared by this launch too (stat_accumulators entered just before)
                arena = _STAT_ARENAS.get((dev.type, dev.index if dev.index is not None else torch.cuda.current_device()))
                zero = arena.take_pending_zero() if (arena is not None and arena.active) else None
                arr = (L.HJob * (len(prs) + len(gts) + (1 if zero else 0)))()
                if zero:
                    arr[len(prs) + len(gts)] = L.HJob(None, None, L.ptr(zero[0]), None, 1, zero[1] // 256, 256, 2, None, None)
                off = 0
                for i, (A, W, b, Wg, bg) in enumerate(gts):
                    n, kin, kout = W.shape
                    kg, inw = Wg.shape[0], n * kin
                    M, MT = flat[off:off + inw * kg].view(inw, kg), flat[off + inw * kg:off + 2 * inw * kg].view(kg, inw)
                    c = flat[off + 2 * inw * kg:off + 2 * inw * kg + kg]
                    off += 2 * inw * kg + kg
                    off = (off + 3) // 4 * 4                  # 16-byte aligned segments
                    arr[i] = L.HJob(L.ptr(A), L.ptr(W), L.ptr(M), L.ptr(MT), n, kin, kout, 3, None, None, L.ptr(Wg), L.ptr(bg),
                                    L.ptr(b), L.ptr(c), kg)            # (first in the launch: two dependent phases per workgroup)
                    _PREBUILT.table[("gate", A.data_ptr(), W.data_ptr(), Wg.data_ptr())] = (
                        M, MT, c, tuple(t._version if t is not None else -1 for t in (A, W, b, Wg, bg)))
                for i, (A, W) in enumerate(prs):
                    if A is None:                            # nn.Linear weight [out, in]: H = W^T, and W itself is H^T
                        fo, fi = W.shape
                        H = flat[off:off + fi * fo].view(fi, fo)
                        off += fi * fo
                        arr[len(gts) + i] = L.HJob(None, L.ptr(W), L.ptr(H), None, 1, fi, fo, 1, None, None)
                        _PREBUILT.table[("dense", W.data_ptr())] = (H, W, W._version, W._version, None, None)
                        continue
                    n, kin, kout = W.shape
                    sz = (n * kin) * (n * kout)
                    H, HT = flat[off:off + sz].view(n * kin, n * kout), flat[off + sz:off + 2 * sz].view(n * kout, n * kin)
                    Hk4 = HTk4 = None
                    if max(n * kin, n * kout) <= FUSED_MAX_WIDTH:      # the k4 forms feed the H-stationary kernel only
                        Hk4, HTk4 = flat[off + 2 * sz:off + 3 * sz], flat[off + 3 * sz:off + 4 * sz]
                    off += 4 * sz
                    arr[len(gts) + i] = L.HJob(L.ptr(A), L.ptr(W), L.ptr(H), L.ptr(HT), n, kin, kout, 0, L.ptr(Hk4), L.ptr(HTk4))
                    _PREBUILT.table[(A.data_ptr(), W.data_ptr())] = (H, HT, A._version, W._version, Hk4, HTk4)
                ef = None
                if encode is not None and ENCODE_FRONT and encode[0].device == dev and len(arr) > 0:
                    idx, ts, tables = encode
                    ts = ts.vector_form(tables)
                    idx_ = _idx64(idx)
                    if idx_.shape[1] == ts.F and lib.phc_encode_front_supported(idx_.shape[0], ts.F, ts.n, ts.k):
                        enc_out = torch.empty((idx_.shape[0], ts.n * ts.k), dtype=torch.float32, device=dev)
                        ef = L.EncodeFront(L.ptr(idx_), L.ptr(ts.ptrs), _C.cast(ts.vocab_c, _C.c_void_p), L.ptr(enc_out), idx_.shape[0],
                                           ts.F, ts.n, ts.k)
                        _PREBUILT.encoded = (idx_.data_ptr(), tuple(idx_.shape), ts.ptrs.data_ptr(), enc_out, idx_)
                L.check(lib.phc_phm_build_H_batched_ex(arr, len(arr), _C.byref(ef) if ef is not None else None, L.stream_ptr(dev)),
                        "phc_phm_build_H_batched")
                if ef is not None and L.PROF_ON:
                    L.account("encode_fwd", 0.0, idx_.shape[0] * (8.0 * ts.F + 4.0 * ts.n * ts.k) + 4.0 * ts.R * ts.n * ts.k)
        yield
    finally:
        _PREBUILT.table = {}
        _PREBUILT.encoded = None
        if outer_step:
            _FLAT_STEP.packed = None


def _prebuilt_for(A, W):
    ent = _PREBUILT.table.get((A.data_ptr(), W.data_ptr()) if A is not None else ("dense", W.data_ptr()))
    if ent is None or ent[3] != W._version or (A is not None and ent[2] != A._version):   # touched since the build
        return None
    return ent[0], ent[1], ent[4], ent[5]            # H, H^T, and their k4 forms (None for a dense weight)


# ---------------------------------------------------------------------------
# PHM linear
# ---------------------------------------------------------------------------
# PHMConv (mlp = False): the statistics of `transform(agg) + x` out of the transform's GEMM (phc_gemm_prebuilt_stats_res) instead of a
# statistics pass + a finalize launch in front of the norm's apply pass
STATS_WITH_RESIDUAL = __import__("os").environ.get("PHC_STATS_WITH_RESIDUAL", "1") != "0"
# ... and the self loop's share of dx parked for the aggregation's backward walk (PHC_RESIDUAL_HANDOFF=0: autograd adds the two)
RESIDUAL_HANDOFF = __import__("os").environ.get("PHC_RESIDUAL_HANDOFF", "1") != "0"


class _PHMLinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, A, W, b, residual, bn_stats=None, dx_handoff=None, res_handoff=None):
        lib = L.load()
        ctx.dx_handoff = dx_handoff
        if dx_handoff is not None:
            dx_handoff.accepted = True
        # the residual's gradient (= dy) parked for another consumer of the residual tensor whose backward runs later and adds it
        # inside its own kernel (PHMConv: the self loop's share of dx goes into the aggregation's backward walk, no add launch)
        ctx.res_handoff = res_handoff if (res_handoff is not None and res_handoff.accepted and residual is not None) else None
        dev = L.require_device(x, A, W, b, residual)
        n, kin, kout = W.shape
        if tuple(A.shape) != (n, n, n):
            raise L.PhcError(f"phm_rule must be [{n},{n},{n}], got {tuple(A.shape)}")
        if x.shape[-1] != n * kin:
            raise L.PhcError(f"x has {x.shape[-1]} features, PHM weight expects {n * kin}")
        x2 = L.f32c(x.reshape(-1, n * kin))
        A_, W_, b_ = L.f32c(A), L.f32c(W), L.f32c(b)
        res2 = L.f32c(residual.reshape(-1, n * kout)) if residual is not None else None
        N = x2.shape[0]
        y = torch.empty((N, n * kout), dtype=torch.float32, device=dev)
        pre = _prebuilt_for(A_, W_) if (_PREBUILT.table and x2.data_ptr() % 16 == 0) else None
        ctx.HT = ctx.HTk4 = None
        if pre is not None and bn_stats is not None and N > 0 and x.dim() == 2 and \
                (res2 is None or (STATS_WITH_RESIDUAL and res2.data_ptr() % 16 == 0 and (n * kout) % 4 == 0)):
            # (with a residual — PHMConv's self loop behind the transform — the statistics are those of the stored sum)
            sp, sc, want_num = bn_stats.allocate(dev, n * kout, lib.phc_gemm_stats_max_partials())
            num = _C.c_int(0)
            L.check(lib.phc_gemm_prebuilt_stats_res(L.ptr(x2), L.ptr(pre[0]), L.ptr(pre[2]), L.ptr(b_), L.ptr(res2), N, n * kin, n * kout,
                                                    L.ptr(y), L.ptr(bn_stats.n_valid), L.ptr(sp), L.ptr(sc),
                                                    _C.byref(num) if want_num else None, L.stream_ptr(dev)), "phc_gemm_prebuilt_stats")
            bn_stats.num = num.value
            ctx.HT, ctx.HTk4 = pre[1], pre[3]
        elif pre is not None:
            L.check(lib.phc_gemm_prebuilt(L.ptr(x2), L.ptr(pre[0]), L.ptr(pre[2]), L.ptr(b_), L.ptr(res2), N, n * kin, n * kout, L.ptr(y), 0,
                                          L.stream_ptr(dev)), "phc_gemm_prebuilt")
            ctx.HT, ctx.HTk4 = pre[1], pre[3]    # dx = dy . H^T from the same build (the weights do not change before it)
        else:
            L.check(lib.phc_phm_linear_fwd(L.ptr(x2), L.ptr(A_), L.ptr(W_), L.ptr(b_), L.ptr(res2), N, n, kin, kout,
                                           L.ptr(y), L.stream_ptr(dev)), "phc_phm_linear_fwd")
        if L.PROF_ON:
            L.account("phm_gemm_fwd", 2.0 * N * n * kin * n * kout,
                      4.0 * (N * n * (kin + kout) + n * kin * kout + n ** 3 + n * kout))
        ctx.save_for_backward(x2, A_, W_)
        # deferral is decided here: leaf fp32 parameters used in place (their .grad is written at flush time)
        ctx.defer_params = None
        if _DEFER.active and N > 0 and A_ is A and W_ is W and (b is None or b_ is b) and \
                all(p is None or (p.is_leaf and p.grad_fn is None) for p in (A, W, b)):
            ctx.defer_params = (A, W, b)
        ctx.has_bias = b is not None
        ctx.has_res = residual is not None
        ctx.x_shape = x.shape
        ctx.res_shape = residual.shape if residual is not None else None
        return y.reshape(*x.shape[:-1], n * kout)

    @staticmethod
    def backward(ctx, dy):
        lib = L.load()
        x2, A, W = ctx.saved_tensors
        dev = x2.device
        n, kin, kout = W.shape
        N = x2.shape[0]
        dy2 = L.f32c(dy.reshape(-1, n * kout))
        need_x, need_A, need_W, need_b, need_res = ctx.needs_input_grad[:5]
        need_b = need_b and ctx.has_bias
        dx = torch.empty_like(x2) if need_x else None
        # the other consumer's share of dx (DxHandoff): added inside the dx GEMM where that path runs, by one add otherwise
        extra = ctx.dx_handoff.take() if ctx.dx_handoff is not None else None
        if extra is not None and (not need_x or tuple(extra.shape) != tuple(x2.shape)):
            raise L.PhcError("DxHandoff: a parked gradient needs a consumer that computes dx of the same shape")
        need_param = need_A or need_W or need_b
        if need_param and ctx.defer_params is not None and _DEFER.active and not torch.is_grad_enabled():
            # dx only; the parameter gradients of all layers are produced together when the deferral block exits
            if need_x and ctx.HT is not None and dy2.data_ptr() % 16 == 0 and (extra is None or extra.data_ptr() % 16 == 0):
                L.check(lib.phc_gemm_prebuilt(L.ptr(dy2), L.ptr(ctx.HT), L.ptr(ctx.HTk4), None, L.ptr(extra), N, n * kout, n * kin, L.ptr(dx), 1,
                                              L.stream_ptr(dev)), "phc_gemm_prebuilt(dx)")
                extra = None
            else:
                L.check(lib.phc_phm_linear_bwd(L.ptr(x2), L.ptr(A), L.ptr(W), L.ptr(dy2), N, n, kin, kout, L.ptr(dx), None,
                                               None, None, None, 0, L.stream_ptr(dev)), "phc_phm_linear_bwd")
            if extra is not None:
                dx += extra
            if L.PROF_ON and need_x:
                L.account("phm_gemm_dx", 2.0 * N * n * kin * n * kout, 4.0 * (N * n * (kin + kout) + n * kin * kout + n ** 3))
            pA, pW, pb = ctx.defer_params
            with _DEFER.lock:
                _DEFER.jobs.append(dict(x=x2, dy=dy2, A=A, W=W, need=(need_A, need_W, need_b),
                                        params=(pA if need_A else None, pW if need_W else None, pb if need_b else None)))
            return (dx.reshape(ctx.x_shape) if need_x else None, None, None, None, _res_grad(ctx, dy, dy2, need_res), None, None, None)
        dW = torch.empty_like(W) if need_param else None
        dA = torch.empty_like(A) if need_A else None
        db = torch.empty(n * kout, dtype=torch.float32, device=dev) if need_b else None
        ws_bytes = lib.phc_phm_linear_bwd_ws_bytes(N, n, kin, kout) if need_param else 0
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev) if need_param else None
        L.check(lib.phc_phm_linear_bwd(L.ptr(x2), L.ptr(A), L.ptr(W), L.ptr(dy2), N, n, kin, kout, L.ptr(dx), L.ptr(dW),
                                       L.ptr(dA), L.ptr(db), L.ptr(ws), ws_bytes, L.stream_ptr(dev)),
                "phc_phm_linear_bwd")
        if L.PROF_ON:
            gemm = 2.0 * N * n * kin * n * kout
            io = 4.0 * N * n * (kin + kout)
            if need_x:
                L.account("phm_gemm_dx", gemm, io + 4.0 * (n * kin * kout + n ** 3))
            if need_param:
                L.account("phm_dH_partial", gemm, io + 4.0 * n * kin * n * kout)
        if extra is not None:
            dx += extra
        return (dx.reshape(ctx.x_shape) if need_x else None, dA, dW if need_W else None, db, _res_grad(ctx, dy, dy2, need_res), None, None, None)


def _res_grad(ctx, dy, dy2, need_res):
    """The residual input's gradient of ``_PHMLinearFn``: dy itself — returned, or parked in the hand-off (then autograd gets None
    for this input and the consumer behind the hand-off adds the share)."""
    if not (need_res and ctx.has_res):
        return None
    if ctx.res_handoff is not None and dy2.dim() == 2 and tuple(dy2.shape) == tuple(ctx.res_shape):
        ctx.res_handoff.park(dy2)
        return None
    return dy.reshape(ctx.res_shape)


def phm_linear_dx_raw(dy: torch.Tensor, A: torch.Tensor, W: torch.Tensor) -> torch.Tensor:
    """dx = dy . H^T (one launch, no autograd bookkeeping): the backward-data GEMM of a PHM linear layer."""
    lib = L.load()
    dev = L.require_device(dy, A, W)
    n, kin, kout = W.shape
    dy2, A_, W_ = L.f32c(dy), L.f32c(A.detach()), L.f32c(W.detach())
    N = dy2.shape[0]
    dx = torch.empty((N, n * kin), dtype=torch.float32, device=dev)
    if N:
        # (x is only read for the parameter gradients; the entry point wants a non-null pointer)
        L.check(lib.phc_phm_linear_bwd(L.ptr(dy2), L.ptr(A_), L.ptr(W_), L.ptr(dy2), N, n, kin, kout, L.ptr(dx), None, None, None,
                                       None, 0, L.stream_ptr(dev)), "phc_phm_linear_bwd(dx)")
        if L.PROF_ON:
            L.account("phm_gemm_dx", 2.0 * N * n * kin * n * kout, 4.0 * (N * n * (kin + kout) + n * kin * kout + n ** 3))
    return dx


def phm_linear_param_grads_raw(x: torch.Tensor, dy: torch.Tensor, A: torch.Tensor, W: torch.Tensor, has_bias: bool):
    """(dA, dW, db | None) of y = x.H(A, W) + b from x and dy alone (no dx): dH = x^T.dy as fixed-order split-K slabs,
    then the two contractions with A and W."""
    lib = L.load()
    dev = L.require_device(x, dy, A, W)
    n, kin, kout = W.shape
    x2, dy2, A_, W_ = L.f32c(x), L.f32c(dy), L.f32c(A.detach()), L.f32c(W.detach())
    N = x2.shape[0]
    dW, dA = torch.empty_like(W_), torch.empty_like(A_)
    db = torch.empty(n * kout, dtype=torch.float32, device=dev) if has_bias else None
    ws_bytes = lib.phc_phm_linear_bwd_ws_bytes(N, n, kin, kout)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    L.check(lib.phc_phm_linear_bwd(L.ptr(x2), L.ptr(A_), L.ptr(W_), L.ptr(dy2), N, n, kin, kout, None, L.ptr(dW), L.ptr(dA),
                                   L.ptr(db), L.ptr(ws), ws_bytes, L.stream_ptr(dev)), "phc_phm_linear_bwd(params)")
    if L.PROF_ON:
        L.account("phm_dH_partial", 2.0 * N * n * kin * n * kout, 4.0 * N * n * (kin + kout) + 4.0 * n * kin * n * kout)
    return dA, dW, db


# ---------------------------------------------------------------------------
# statistics accumulators (csrc/phc_common.h): column sums that the workgroups of one launch ADD into 64-bit fixed-point
# words with integer atomics, read as totals by the next launch — no merge launch in between.  They must be zero when the
# producing launch starts: a per-device arena hands out slots inside ``with stat_accumulators(dev):`` (one training step:
# harness.train_step / GraphedTrainStep) and zeroes what the previous step used with ONE memset on entry.
# ---------------------------------------------------------------------------
STAT_ACC = __import__("os").environ.get("PHC_STAT_ACC", "1") != "0"
_STAT_ARENAS: dict = {}


class _StatArena:
    WORDS = 1 << 18                    # 2 MB of int64 words per device: ~128 norms of width 128 per step

    def __init__(self, dev):
        self.buf = torch.zeros(self.WORDS, dtype=torch.int64, device=dev)
        self.used = 0                  # words handed out since the last reset
        self.dirty = 0                 # words that may be non-zero
        self.pending_zero = 0          # words of the previous step still to be cleared (stat_accumulators.__enter__)
        self.active = False

    def take_pending_zero(self):
        """(tensor, float count) of the region a zero-fill job must clear (rounded up to whole 256-float blocks), or None."""
        if not self.pending_zero:
            return None
        floats = min((2 * self.pending_zero + 255) // 256 * 256, 2 * self.WORDS)
        self.pending_zero = 0
        return self.buf, floats

    def flush_pending_zero(self):
        if self.pending_zero:
            self.buf[:self.pending_zero].zero_()
            self.pending_zero = 0

    def slot(self, h: int):
        lib = L.load()
        words = lib.phc_stat_acc_words(h)
        if not self.active or self.used + words > self.WORDS:
            return None
        self.flush_pending_zero()
        t = self.buf[self.used:self.used + words].view(2 * lib.phc_stat_acc_replicas(), 2 * h)
        self.used += words
        self.dirty = max(self.dirty, self.used)
        return t


class stat_accumulators:
    """Context of one training step on ``dev``: statistics-emitting launches inside it use accumulator slots."""

    def __init__(self, dev):
        self.dev = torch.device(dev)

    def __enter__(self):
        if not STAT_ACC or self.dev.type != "cuda":
            self.arena = None
            return self
        key = (self.dev.type, self.dev.index if self.dev.index is not None else torch.cuda.current_device())
        arena = _STAT_ARENAS.get(key)
        if arena is None:
            arena = _STAT_ARENAS[key] = _StatArena(self.dev)
        # everything the previous step added into is cleared before this step's first accumulator user: by the operand-build
        # launch that opens a training step (prebuilt_H, entered right after this context: a zero-fill job in its table) or, when
        # no build follows, by one memset at the first slot()
        arena.pending_zero = arena.dirty
        arena.used = 0
        arena.active = True
        self.arena = arena
        return self

    def __exit__(self, *exc):
        if self.arena is not None:
            self.arena.active = False
        return False


def stat_acc_slot(dev, h: int):
    """A zeroed accumulator [2 * replicas, 2 h] (int64) of the running step, or None outside ``stat_accumulators``."""
    if h % 4:                              # the accumulator kernels address float4 column groups (a partial last 32-column block is fine)
        return None
    dev = torch.device(dev)
    arena = _STAT_ARENAS.get((dev.type, dev.index if dev.index is not None else torch.cuda.current_device()))
    return arena.slot(h) if arena is not None else None


class BNStatsRequest:
    """Ask a PHM linear layer for the batch-norm statistics of its output (training mode): when the layer runs on a
    prebuilt operand matrix, its GEMM emits per-column (sum, M2) partials from registers and the normalisation that
    follows (``batch_norm_act`` on the very tensor this layer returned) skips its own pass over the data.
    ``consumer``: "apply" / "linear" — which op follows (``batch_norm_act`` or ``bn_act_linear``); inside ``stat_accumulators`` the
    statistics travel through an accumulator either way (``counts`` stays None), outside as partial rows."""

    def __init__(self, n_valid: Optional[torch.Tensor] = None, consumer: str = "linear"):
        self.n_valid = n_valid
        self.consumer = consumer
        self.partial = self.counts = None
        self.num = 0

    def allocate(self, dev, h: int, max_partials: int):
        """Output buffers of the producing launch: (partial, counts, want_num)."""
        acc = stat_acc_slot(dev, h)        # (either consumer reads accumulator totals: bn_apply_acc / the operand transform's prologue)
        if acc is not None:
            self.partial, self.counts = acc, None
            return acc, None, False
        self.partial = torch.empty((max_partials, 2 * h), dtype=torch.float32, device=dev)
        self.counts = torch.empty(max_partials, dtype=torch.float32, device=dev)
        return self.partial, self.counts, True


# sc_type "first" models: the skips' gradient shares added inside the first layer's aggregation backward (DxSumHandoff) instead of
# an n-way sum launch (PHC_SKIP_SUM_HANDOFF=0 restores the launch)
SKIP_SUM_HANDOFF = __import__("os").environ.get("PHC_SKIP_SUM_HANDOFF", "1") != "0"


class DxHandoff:
    """Gradient hand-over between two consumers of ONE tensor x (the soft-attention read-out: x feeds the gate's PHM layer and
    the gated sum).  The consumer whose backward runs first parks its dx here instead of returning it; the PHM layer's backward
    adds it inside its dx GEMM (the GEMM's residual operand) — autograd then sees one gradient for x and launches no add."""

    def __init__(self):
        self.tensor = None
        self.accepted = False      # set by the consumer that will add the parked gradient (its forward ran): nobody parks before

    def park(self, t):
        self.tensor = t

    def take(self):
        t, self.tensor = self.tensor, None
        return t


class DxSumHandoff(DxHandoff):
    """The same for a tensor with MANY parked shares: the encoded atoms of an ``sc_type = "first"`` model are the skip tensor of
    every layer (models.py:230-234) and the first conv's input.  Each layer's fused MLP + norm node parks its share of the
    gradient here instead of returning it; the FIRST layer's aggregation backward — the last to run — adds all of them inside
    its dx kernel (``phc_aggregate_dx_rows_multi``): the n-way sum launch in front of the atom-table gradient disappears."""
    MAX = 8

    def __init__(self):
        super().__init__()
        self.tensors = []

    def park(self, t):
        self.tensors.append(t)

    def take(self):
        ts, self.tensors = self.tensors, []
        return ts if ts else None


# the dx of a layer's sum aggregation built inside the backward launch of the fused MLP + norm node BELOW it (round 6,
# phc_mlp_norm_bwd_desc.agg_src): three aggregate_dx_rows launches less per molhiv step; 0 = the launch of its own
MLP_BWD_FUSED_AGG = __import__("os").environ.get("PHC_MLP_BWD_FUSED_AGG", "1") != "0"


class PendingDx:
    """dx of a sum aggregation (identity message function) that has NOT been launched: ``dx`` is allocated and was returned to
    autograd as the gradient of the aggregation's input, its rows are still to be written.  The one consumer — the backward of the
    fused MLP + norm node that produced that input (``DxLink``) — builds them inside its own launch (``phc_mlp_norm_bwd_fused`` with
    ``agg_src``), or calls ``materialize`` first: the launch that would have run in the aggregation's backward, identical rows."""

    def __init__(self, gsrc, residual, st, add_self, dx):
        self.gsrc, self.residual, self.st, self.add_self, self.dx = gsrc, residual, st, add_self, dx

    def materialize(self):
        lib = L.load()
        st = self.st
        rowptr_t, col_t, _ = st.csc
        h = self.dx.shape[1]
        L.check(lib.phc_aggregate_dx_rows(L.ptr(self.gsrc), L.ptr(rowptr_t), L.ptr(col_t), st.num_src, st.num_dst, h,
                                          1 if self.add_self else 0, L.ptr(self.residual), L.ptr(self.dx), L.stream_ptr(self.dx.device)),
                "phc_aggregate_dx_rows")


class DxLink:
    """Backward-side mirror of ``PendingAggregate``: joins the fused MLP + norm node of layer l - 1 (the producer of x) with the sum
    aggregation of layer l (x's ONLY autograd consumer: the model loop hands the link over only then — the skip's share travels
    through a DxHandoff / DxSumHandoff).  ``ready``: the node's backward can take the one-launch form with the aggregation inside."""

    def __init__(self):
        self.ready = False
        self.pending: Optional[PendingDx] = None

    def take(self):
        p, self.pending = self.pending, None
        return p


def phm_linear(x: torch.Tensor, phm_rule: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor] = None,
               residual: Optional[torch.Tensor] = None, bn_stats: Optional[BNStatsRequest] = None,
               dx_handoff: Optional[DxHandoff] = None, res_handoff: Optional[DxHandoff] = None) -> torch.Tensor:
    """y = x · (Σ_i phm_rule[i] ⊗ W[i]) + bias (+ residual)."""
    y = _PHMLinearFn.apply(x, phm_rule, W, bias, residual, bn_stats, dx_handoff, res_handoff)
    if bn_stats is not None and bn_stats.partial is not None:
        y._phc_bn = bn_stats                 # consumed by batch_norm_act if it is handed this very tensor
    return y


_ONES: dict = {}


class _DenseLinearFn(torch.autograd.Function):
    """y = x·weightᵀ + b on a prebuilt operand (H = weightᵀ from the step's batched build; Hᵀ is the weight itself),
    dweight / dbias deferred to the batched parameter-gradient launch (written transposed, i.e. as [out, in])."""

    @staticmethod
    def forward(ctx, x, weight, bias, H):
        lib = L.load()
        dev = x.device
        x_ = L.f32c(x)
        N = x_.shape[0]
        fo, fi = weight.shape
        y = torch.empty((N, fo), dtype=torch.float32, device=dev)
        L.check(lib.phc_gemm_prebuilt(L.ptr(x_), L.ptr(H), None, L.ptr(bias), None, N, fi, fo, L.ptr(y), 0, L.stream_ptr(dev)),
                "phc_gemm_prebuilt(dense)")
        if L.PROF_ON:
            L.account("phm_gemm_fwd", 2.0 * N * fi * fo, 4.0 * (N * (fi + fo) + fi * fo + fo))
        ctx.save_for_backward(x_, weight)
        ctx.params = (weight, bias)
        ctx.H = H
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = L.load()
        x_, weight = ctx.saved_tensors
        dev = x_.device
        N = x_.shape[0]
        fo, fi = weight.shape
        dy_ = L.f32c(dy)
        need_x, need_w, need_b = ctx.needs_input_grad[:3]
        dx = None
        if need_x:
            dx = torch.empty_like(x_)
            if fo % 4 == 0 and dy_.data_ptr() % 16 == 0:
                L.check(lib.phc_gemm_prebuilt(L.ptr(dy_), L.ptr(weight), None, None, None, N, fo, fi, L.ptr(dx), 1, L.stream_ptr(dev)),
                        "phc_gemm_prebuilt(dense dx)")
            else:
                # an output width that is not a multiple of four (the 50-wide gate of the published 200-wide model) cannot be the
                # contraction of the streaming kernel: dx through the tiled kernel, on the prebuilt H = weight^T as the n = 1 weight
                L.check(lib.phc_phm_linear_bwd(L.ptr(x_), L.ptr(_one(dev)), L.ptr(ctx.H), L.ptr(dy_), N, 1, fi, fo, L.ptr(dx), None,
                                               None, None, None, 0, L.stream_ptr(dev)), "phc_phm_linear_bwd(dense dx)")
            if L.PROF_ON:
                L.account("phm_gemm_dx", 2.0 * N * fi * fo, 4.0 * (N * (fi + fo) + fi * fo))
        pw, pb = ctx.params
        if (need_w or need_b) and _DEFER.active and not torch.is_grad_enabled() and pw.is_leaf and (pb is None or pb.is_leaf):
            with _DEFER.lock:                               # job in PHM form: n = 1, W = weightᵀ [1, in, out], dW transposed
                _DEFER.jobs.append(dict(x=x_, dy=dy_, A=_one(dev), W=weight, need=(False, need_w, need_b and pb is not None),
                                        params=(None, pw if need_w else None, pb if (need_b and pb is not None) else None),
                                        dense=True))
            return dx, None, None, None
        dw = dy_.t().mm(x_) if need_w else None
        db = dy_.sum(0) if (need_b and pb is not None) else None
        return dx, dw, db, None


class _FanoutFn(torch.autograd.Function):
    """n aliases of one tensor whose gradients are summed by ONE kernel (phc_sum_n) instead of n-1 pairwise adds."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.n = n
        ctx.set_materialize_grads(False)     # a consumer that parked its share elsewhere (DxSumHandoff) hands back None, not zeros
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        gs = [L.f32c(g) for g in grads if g is not None]
        if not gs:
            return None, None
        if len(gs) == 1:
            return gs[0], None
        lib = L.load()
        dev = L.require_device(*gs)
        out = torch.empty_like(gs[0])
        for c0 in range(0, len(gs), 16):                    # 16 sources per launch
            part = gs[c0:c0 + 16] if c0 == 0 else [out] + gs[c0:c0 + 15]
            arr = (_C.c_void_p * len(part))(*[g.data_ptr() for g in part])
            L.check(lib.phc_sum_n(arr, len(part), out.numel(), L.ptr(out), L.stream_ptr(dev)), "phc_sum_n")
        return out, None


def fanout(x: torch.Tensor, n: int):
    """``n`` aliases of ``x`` for ``n`` consumers; in the backward pass their gradients are summed in one pass."""
    if n <= 1 or not (torch.is_grad_enabled() and x.requires_grad) or not x.is_cuda:
        return tuple(x for _ in range(max(n, 1)))
    return _FanoutFn.apply(x, n)


class _SmallLinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        lib = L.load()
        dev = L.require_device(x, weight, bias)
        x_, w_, b_ = L.f32c(x), L.f32c(weight), L.f32c(bias)
        N, (fo, fi) = x_.shape[0], w_.shape
        y = torch.empty((N, fo), dtype=torch.float32, device=dev)
        L.check(lib.phc_small_linear_fwd(L.ptr(x_), L.ptr(w_), L.ptr(b_), N, fi, fo, L.ptr(y), L.stream_ptr(dev)),
                "phc_small_linear_fwd")
        ctx.save_for_backward(x_, w_)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = L.load()
        x_, w_ = ctx.saved_tensors
        dev = x_.device
        N, (fo, fi) = x_.shape[0], w_.shape
        dy_ = L.f32c(dy)
        need_x, need_w, need_b = ctx.needs_input_grad
        need_b = need_b and ctx.has_bias
        dx = torch.empty_like(x_) if need_x else None
        dw = torch.empty_like(w_) if need_w else None
        db = torch.empty(fo, dtype=torch.float32, device=dev) if need_b else None
        L.check(lib.phc_small_linear_bwd(L.ptr(dy_), L.ptr(x_), L.ptr(w_), N, fi, fo, L.ptr(dx), L.ptr(dw), L.ptr(db),
                                         L.stream_ptr(dev)), "phc_small_linear_bwd")
        return dx, dw, db


def small_linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = x · weightᵀ + bias for a tiny ``nn.Linear`` (in·out <= 64): the last map of the read-out on B rows."""
    return _SmallLinearFn.apply(x, weight, bias)


def dense_linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = x · weightᵀ + bias for an ``nn.Linear`` weight [out, in], on the PHM kernels: a dense map is the
    phm_dim = 1 case (A = [[[1]]], W = weightᵀ).  Used for the hypercomplex→real head
    (phc/hypercomplex/layers.py:386-414), whose K = N gradient GEMM is a poor fit for the BLAS heuristics."""
    dev = L.require_device(x, weight)
    if _PREBUILT.table and x.dim() == 2 and x.is_contiguous() and x.data_ptr() % 16 == 0 and (bias is None or bias.is_contiguous()):
        pre = _prebuilt_for(None, weight)
        if pre is not None:
            return _DenseLinearFn.apply(x, weight, bias, pre[0])
    return _PHMLinearFn.apply(x, _one(dev), weight.t().unsqueeze(0), bias, None, None)


def phm_build_operands(pairs):
    """[(H, H^T, Hk4, H^Tk4)] of the listed (phm_rule, W) pairs, assembled by ONE launch (phc_phm_build_H_batched) — the
    operand forms of ``phm_linear_pre`` / ``phm_linear_dx_pre`` (no autograd; the sharded layer of dist.py keeps its own)."""
    lib = L.load()
    if not pairs:
        return []
    dev = L.require_device(*[t for pr in pairs for t in pr])
    total = sum(4 * (W.shape[0] * W.shape[1]) * (W.shape[0] * W.shape[2]) for _, W in pairs)
    flat = torch.empty(total, dtype=torch.float32, device=dev)
    arr = (L.HJob * len(pairs))()
    out, off = [], 0
    for i, (A, W) in enumerate(pairs):
        n, kin, kout = W.shape
        if (n * kin) % 4 or (n * kout) % 4:
            raise L.PhcError("phm_build_operands: widths must be multiples of 4")
        sz = (n * kin) * (n * kout)
        H, HT = flat[off:off + sz].view(n * kin, n * kout), flat[off + sz:off + 2 * sz].view(n * kout, n * kin)
        Hk4, HTk4 = flat[off + 2 * sz:off + 3 * sz], flat[off + 3 * sz:off + 4 * sz]
        off += 4 * sz
        arr[i] = L.HJob(L.ptr(L.f32c(A.detach())), L.ptr(L.f32c(W.detach())), L.ptr(H), L.ptr(HT), n, kin, kout, 0, L.ptr(Hk4), L.ptr(HTk4))
        out.append((H, HT, Hk4, HTk4))
    L.check(lib.phc_phm_build_H_batched(arr, len(pairs), L.stream_ptr(dev)), "phc_phm_build_H_batched")
    return out


def phm_linear_pre(x: torch.Tensor, pre, bias: Optional[torch.Tensor], dx: bool = False) -> torch.Tensor:
    """y = x . H + bias (``dx=False``) or dx = x . H^T (``dx=True``, bias ignored) on prebuilt operands (``phm_build_operands``):
    the H-stationary GEMM where it serves the shape, the LDS-staged one otherwise.  No autograd bookkeeping."""
    lib = L.load()
    dev = L.require_device(x)
    x_ = L.f32c(x)
    M, Mk4 = (pre[1], pre[3]) if dx else (pre[0], pre[2])
    K, Nout = M.shape
    y = torch.empty((x_.shape[0], Nout), dtype=torch.float32, device=dev)
    L.check(lib.phc_gemm_prebuilt(L.ptr(x_), L.ptr(M), L.ptr(Mk4), None if (dx or bias is None) else L.ptr(L.f32c(bias)), None, x_.shape[0], K, Nout, L.ptr(y),
                                  1 if dx else 0, L.stream_ptr(dev)), "phc_gemm_prebuilt")
    if L.PROF_ON:
        L.account("phm_gemm_dx" if dx else "phm_gemm_fwd", 2.0 * x_.shape[0] * K * Nout, 4.0 * (x_.shape[0] * (K + Nout) + K * Nout))
    return y


def phm_build_H(phm_rule: torch.Tensor, W: torch.Tensor) -> torch.Tensor:
    """Materialise H = Σ_i A_i ⊗ W_i  [n*kin, n*kout] (inspection / tests; not differentiable)."""
    lib = L.load()
    dev = L.require_device(phm_rule, W)
    n, kin, kout = W.shape
    H = torch.empty((n * kin, n * kout), dtype=torch.float32, device=dev)
    L.check(lib.phc_phm_build_H(L.ptr(L.f32c(phm_rule.detach())), L.ptr(L.f32c(W.detach())), n, kin, kout, L.ptr(H),
                                L.stream_ptr(dev)), "phc_phm_build_H")
    return H


# ---------------------------------------------------------------------------
# neighbour aggregation
# ---------------------------------------------------------------------------
class _AggregateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ea, st: EdgeStructure, act: int, add_self: bool):
        lib = L.load()
        dev = L.require_device(x, ea)
        x_ = L.f32c(x)
        ea_ = L.f32c(ea)
        h = x_.shape[1]
        if x_.shape[0] < st.num_src:
            raise L.PhcError(f"x has {x_.shape[0]} rows, edge structure expects {st.num_src} source rows")
        if ea_ is not None and tuple(ea_.shape) != (st.num_edges, h):
            raise L.PhcError(f"edge_attr must be [{st.num_edges}, {h}], got {tuple(ea_.shape)}")
        rowptr, col, perm = st.csr
        out = torch.empty((st.num_dst, h), dtype=torch.float32, device=dev)
        L.check(lib.phc_aggregate_fwd(L.ptr(x_), L.ptr(ea_), L.ptr(rowptr), L.ptr(col), L.ptr(perm), st.num_dst, h, act,
                                      1 if add_self else 0, L.ptr(out), L.stream_ptr(dev)), "phc_aggregate_fwd")
        if L.PROF_ON:
            E, N = st.num_edges, st.num_dst
            L.account("aggregate_fwd", 2.0 * E * h,
                      E * (4.0 * h * (2 if ea is not None else 1) + 4) + N * 4.0 * h * (2 if add_self else 1) + 4.0 * (N + 1))
        ctx.st, ctx.act, ctx.add_self, ctx.has_ea = st, act, add_self, ea is not None
        if act != 0:
            ctx.save_for_backward(x_, ea_) if ea is not None else ctx.save_for_backward(x_)
        ctx.x_rows = x_.shape[0]
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = L.load()
        st: EdgeStructure = ctx.st
        dout_ = L.f32c(dout)
        dev = dout_.device
        h = dout_.shape[1]
        x_ = ea_ = None
        if ctx.act != 0:
            saved = ctx.saved_tensors
            x_ = saved[0]
            ea_ = saved[1] if ctx.has_ea else None
        need_x, need_ea = ctx.needs_input_grad[0], ctx.needs_input_grad[1] and ctx.has_ea
        rowptr_t, col_t, perm_t = st.csc
        dx = torch.empty((ctx.x_rows, h), dtype=torch.float32, device=dev)
        if ctx.x_rows > st.num_src:
            dx[st.num_src:].zero_()
        dea = torch.empty((st.num_edges, h), dtype=torch.float32, device=dev) if need_ea else None
        L.check(lib.phc_aggregate_bwd(L.ptr(dout_), L.ptr(x_), L.ptr(ea_), L.ptr(rowptr_t), L.ptr(col_t), L.ptr(perm_t),
                                      st.num_src, st.num_dst, h, ctx.act, 1 if ctx.add_self else 0, L.ptr(dx), L.ptr(dea),
                                      L.stream_ptr(dev)), "phc_aggregate_bwd")
        if L.PROF_ON:
            E, Ns = st.num_edges, st.num_src
            extra = (E * 4.0 * h * (1 if ctx.has_ea else 0) + Ns * 4.0 * h) if ctx.act != 0 else 0.0
            L.account("aggregate_bwd", 1.0 * E * h,
                      E * (4.0 * h * (2 if need_ea else 1) + 4) + Ns * 4.0 * h * (2 if ctx.add_self else 1)
                      + 4.0 * (Ns + 1) + extra)
        return (dx if need_x else None), dea, None, None, None


def aggregate_fwd_raw(x: torch.Tensor, ea: Optional[torch.Tensor], st: EdgeStructure, msg_encoder, add_self: bool,
                      accumulate_into: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Non-differentiable forward launch (used by phc_gnn_amd.dist to compose the sharded operator).
    ``accumulate_into`` [num_dst, h]: the sums are ADDED onto this tensor inside the kernel (and it is returned)."""
    lib = L.load()
    dev = L.require_device(x, ea, accumulate_into)
    act = msg_encoder if isinstance(msg_encoder, int) else _act_id(msg_encoder)
    x_, ea_ = L.f32c(x), L.f32c(ea)
    h = x_.shape[1]
    rowptr, col, perm = st.csr
    if accumulate_into is not None:
        out = accumulate_into
        if out.dtype != torch.float32 or not out.is_contiguous() or tuple(out.shape) != (st.num_dst, h):
            raise L.PhcError("accumulate_into must be contiguous fp32 [num_dst, h]")
    else:
        out = torch.empty((st.num_dst, h), dtype=torch.float32, device=dev)
    L.check(lib.phc_aggregate_fwd(L.ptr(x_), L.ptr(ea_), L.ptr(rowptr), L.ptr(col), L.ptr(perm), st.num_dst, h, act,
                                  (1 if add_self else 0) | (2 if accumulate_into is not None else 0), L.ptr(out),
                                  L.stream_ptr(dev)), "phc_aggregate_fwd")
    if L.PROF_ON:
        E, N = st.num_edges, st.num_dst
        L.account("aggregate_fwd", 2.0 * E * h,
                  E * (4.0 * h * (2 if ea is not None else 1) + 4) + N * 4.0 * h * (1 + (1 if add_self else 0)
                  + (1 if accumulate_into is not None else 0)) + 4.0 * (N + 1))
    return out


def aggregate_bwd_raw(dout: torch.Tensor, x: Optional[torch.Tensor], ea: Optional[torch.Tensor], st: EdgeStructure,
                      msg_encoder, add_self: bool, need_dea: bool):
    """Non-differentiable backward launch: returns (dx [num_src, h], dea [E, h] or None)."""
    lib = L.load()
    dev = L.require_device(dout, x, ea)
    act = msg_encoder if isinstance(msg_encoder, int) else _act_id(msg_encoder)
    dout_, x_, ea_ = L.f32c(dout), L.f32c(x), L.f32c(ea)
    h = dout_.shape[1]
    rowptr_t, col_t, perm_t = st.csc
    dx = torch.empty((st.num_src, h), dtype=torch.float32, device=dev)
    dea = torch.empty((st.num_edges, h), dtype=torch.float32, device=dev) if need_dea else None
    L.check(lib.phc_aggregate_bwd(L.ptr(dout_), L.ptr(x_) if act != 0 else None, L.ptr(ea_) if act != 0 else None,
                                  L.ptr(rowptr_t), L.ptr(col_t), L.ptr(perm_t), st.num_src, st.num_dst, h, act,
                                  1 if add_self else 0, L.ptr(dx), L.ptr(dea), L.stream_ptr(dev)), "phc_aggregate_bwd")
    if L.PROF_ON:
        E, Ns = st.num_edges, st.num_src
        extra = (E * 4.0 * h * (1 if ea is not None else 0) + Ns * 4.0 * h) if act != 0 else 0.0
        L.account("aggregate_bwd", 1.0 * E * h,
                  E * (4.0 * h * (2 if need_dea else 1) + 4) + Ns * 4.0 * h * (2 if add_self else 1) + 4.0 * (Ns + 1) + extra)
    return dx, dea


def aggregate_dx_rows_raw(dout: torch.Tensor, st: EdgeStructure, add_self: bool, extra: Optional[torch.Tensor] = None):
    """dx [num_src, h] of an identity-message aggregation by the row walker (no autograd bookkeeping):
    dx[s] = sum over out-edges of dout[dst(e)] (+ dout[s] with the self loop) (+ extra[s], another consumer's share of the same
    tensor's gradient — added on the way out, no add launch).  None when the shapes are outside the walker's range."""
    lib = L.load()
    dev = L.require_device(dout, extra)
    dout_ = L.f32c(dout)
    h = dout_.shape[1]
    if h % 4 or dout_.data_ptr() % 16 or st.num_src != st.num_dst or (extra is not None and (extra.data_ptr() % 16 or not extra.is_contiguous()
                                                                                          or tuple(extra.shape) != (st.num_src, h))):
        return None
    rowptr_t, col_t, _ = st.csc
    dx = torch.empty((st.num_src, h), dtype=torch.float32, device=dev)
    L.check(lib.phc_aggregate_dx_rows(L.ptr(dout_), L.ptr(rowptr_t), L.ptr(col_t), st.num_src, st.num_dst, h, 1 if add_self else 0,
                                      L.ptr(extra), L.ptr(dx), L.stream_ptr(dev)), "phc_aggregate_dx_rows")
    if L.PROF_ON:
        E, Ns = st.num_edges, st.num_src
        L.account("aggregate_enc_dx", 1.0 * E * h, E * (4.0 * h + 4) + Ns * 4.0 * h * (2 if add_self else 1) + 4.0 * (Ns + 1))
    return dx


def gather_rows(x: torch.Tensor, rows_i64: torch.Tensor) -> torch.Tensor:
    """out[r] = x[rows[r]] (engine kernel; staging of the halo exchange's send buffer)."""
    lib = L.load()
    dev = L.require_device(x, rows_i64)
    x_ = L.f32c(x)
    rows = rows_i64 if rows_i64.dtype == torch.int64 else rows_i64.to(torch.int64)
    rows = rows.contiguous()
    out = torch.empty((rows.numel(), x_.shape[1]), dtype=torch.float32, device=dev)
    if rows.numel():
        L.check(lib.phc_gather_rows_i64(L.ptr(x_), L.ptr(rows), rows.numel(), x_.shape[1], L.ptr(out), L.stream_ptr(dev)),
                "phc_gather_rows_i64")
        if L.PROF_ON:
            L.account("gather_rows", 0.0, rows.numel() * (8.0 * x_.shape[1] + 8))
    return out


# ---------------------------------------------------------------------------
# row-sharded batch-norm: the engine phases between the host's collectives (phc_gnn_amd.dist.ShardedNorm)
# ---------------------------------------------------------------------------
def bn_moments(x: torch.Tensor) -> torch.Tensor:
    """[2h] = column sums | M2 (around this shard's own column means) of the rows of x."""
    lib = L.load()
    dev = L.require_device(x)
    x_ = L.f32c(x)
    N, h = x_.shape
    moments = torch.empty(2 * h, dtype=torch.float32, device=dev)
    ws_bytes = lib.phc_bn_ws_bytes(N, h)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    L.check(lib.phc_bn_moments(L.ptr(x_), N, h, L.ptr(moments), L.ptr(ws), ws_bytes, L.stream_ptr(dev)), "phc_bn_moments")
    if L.PROF_ON:
        L.account("bn_stats_partial", 3.0 * N * h, 4.0 * N * h)
    return moments


def bn_fwd_from_moments(x, moments_all, counts_all, n_total, weights, biases, running_mean, running_var, nbt, momentum, eps,
                        act: int, residual=None):
    """(out, save_stat): normalise the local rows with the statistics merged from every shard's moments."""
    lib = L.load()
    dev = L.require_device(x, moments_all, counts_all, residual)
    x_, res_ = L.f32c(x), L.f32c(residual)
    N, h = x_.shape
    n_comp = len(running_mean) if running_mean else len(weights)
    out = torch.empty_like(x_)
    save_stat = torch.empty(6 * h, dtype=torch.float32, device=dev)
    L.check(lib.phc_bn_fwd_from_moments(
        L.ptr(x_), N, L.ptr(moments_all), L.ptr(counts_all), int(counts_all.numel()), int(n_total),
        _ptr_array(weights) if weights else None, _ptr_array(biases) if biases else None,
        _ptr_array(running_mean) if running_mean else None, _ptr_array(running_var) if running_var else None,
        _ptr_array(nbt) if nbt else None, n_comp, h, float(momentum), float(eps), act, L.ptr(res_), L.ptr(out),
        L.ptr(save_stat), L.stream_ptr(dev)), "phc_bn_fwd_from_moments")
    if L.PROF_ON:
        L.account("bn_apply", 2.0 * N * h, 4.0 * N * h * (3 if residual is not None else 2))
    return out, save_stat


def bn_bwd_sums(dy, x, save_stat, act: int) -> torch.Tensor:
    """[2h] = this shard's partial dweight | dbias (= sum g*x_hat | sum g over its rows)."""
    lib = L.load()
    dev = L.require_device(dy, x)
    dy_, x_ = L.f32c(dy), L.f32c(x)
    N, h = x_.shape
    dwb = torch.empty(2 * h, dtype=torch.float32, device=dev)
    ws_bytes = lib.phc_bn_ws_bytes(N, h)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    L.check(lib.phc_bn_bwd_sums(L.ptr(dy_), L.ptr(x_), L.ptr(save_stat), N, h, act, L.ptr(dwb), L.ptr(ws), ws_bytes,
                                L.stream_ptr(dev)), "phc_bn_bwd_sums")
    if L.PROF_ON:
        L.account("bn_bwd_partial", 6.0 * N * h, 8.0 * N * h)
    return dwb


def bn_bwd_apply(dy, x, save_stat, dwb_global, n_total: int, act: int) -> torch.Tensor:
    lib = L.load()
    dev = L.require_device(dy, x, dwb_global)
    dy_, x_ = L.f32c(dy), L.f32c(x)
    N, h = x_.shape
    dx = torch.empty_like(x_)
    ws = torch.empty(2 * h, dtype=torch.float32, device=dev)
    L.check(lib.phc_bn_bwd_apply(L.ptr(dy_), L.ptr(x_), L.ptr(save_stat), L.ptr(dwb_global), int(n_total), N, h, act, L.ptr(dx),
                                 L.ptr(ws), 8 * h, L.stream_ptr(dev)), "phc_bn_bwd_apply")
    if L.PROF_ON:
        L.account("bn_bwd_apply", 8.0 * N * h, 12.0 * N * h)
    return dx


# ---- the same norm with its four passes folded into the GEMMs on either side (dist._ShardedGINEFn, engine backend) ----
# linear1 -> norm -> act -> linear2 on a row shard: the statistics leave linear1's GEMM as integer totals (read back as this rank's
# moments for the all-gather), linear2's GEMM normalises its operand while it streams in, and in the backward the element-wise norm
# backward runs inside linear1's dx GEMM on the all-reduced (integer: exact) column sums.  What remains a pass of its own is the
# backward's column sums (phc_bn_bwd_partial_acc): as a GEMM epilogue they cost more than the pass at 10 M rows
# (scripts/dbg/gemm_variants_at_scale.py).
def sharded_norm_fusable(h: int, k_in: int) -> bool:
    """Whether the folded forms serve a norm of width h behind a linear [k_in -> h] (both in {128, 256}: the H-stationary GEMM's range)."""
    return bool(L.load().phc_gemm_bwd_apply_supported(int(h), int(k_in)))


def new_stat_acc(h: int, dev) -> torch.Tensor:
    """A zeroed statistics accumulator of width h outside the per-step arena (the sharded layer keeps it across collectives)."""
    return torch.zeros(int(L.load().phc_stat_acc_words(int(h))), dtype=torch.int64, device=dev)


def stat_acc_read(acc: torch.Tensor, h: int, count: int = 0) -> torch.Tensor:
    """[2h] floats of an accumulator: its two totals (count = 0) or sum | M2 of a statistics epilogue over ``count`` rows."""
    lib = L.load()
    out = torch.empty(2 * h, dtype=torch.float32, device=acc.device)
    L.check(lib.phc_stat_acc_read(L.ptr(acc), int(h), int(count), L.ptr(out), L.stream_ptr(acc.device)), "phc_stat_acc_read")
    return out


def gemm_pre_stats(x: torch.Tensor, pre, bias: Optional[torch.Tensor]):
    """(y, acc): y = x . H + bias on prebuilt operands with the statistics epilogue (sum y | sum y^2 per column) in a fresh accumulator."""
    lib = L.load()
    dev = L.require_device(x)
    x_ = L.f32c(x)
    M, Mk4 = pre[0], pre[2]
    K, Nout = M.shape
    N = x_.shape[0]
    y = torch.empty((N, Nout), dtype=torch.float32, device=dev)
    acc = new_stat_acc(Nout, dev)
    L.check(lib.phc_gemm_prebuilt_stats(L.ptr(x_), L.ptr(M), L.ptr(Mk4), None if bias is None else L.ptr(L.f32c(bias)), N, K, Nout, L.ptr(y), None,
                                        L.ptr(acc), None, None, L.stream_ptr(dev)), "phc_gemm_prebuilt_stats")
    if L.PROF_ON:
        L.account("phm_gemm_fwd", 2.0 * N * K * Nout, 4.0 * (N * (K + Nout) + K * Nout))
    return y, acc


def bn_finalize_from_moments(moments_all, counts_all, n_total, weights, biases, running_mean, running_var, nbt, momentum, eps, h: int):
    """save_stat [6h] of the statistics merged from every shard's moments (running statistics updated) — ``bn_fwd_from_moments``
    without its element-wise pass."""
    lib = L.load()
    dev = L.require_device(moments_all, counts_all)
    n_comp = len(running_mean) if running_mean else len(weights)
    save_stat = torch.empty(6 * h, dtype=torch.float32, device=dev)
    L.check(lib.phc_bn_fwd_from_moments(
        None, 0, L.ptr(moments_all), L.ptr(counts_all), int(counts_all.numel()), int(n_total),
        _ptr_array(weights) if weights else None, _ptr_array(biases) if biases else None,
        _ptr_array(running_mean) if running_mean else None, _ptr_array(running_var) if running_var else None,
        _ptr_array(nbt) if nbt else None, n_comp, h, float(momentum), float(eps), 0, None, None,
        L.ptr(save_stat), L.stream_ptr(dev)), "phc_bn_fwd_from_moments(finalize)")
    return save_stat


def gemm_pre_xf(x: torch.Tensor, save_stat: torch.Tensor, act: int, pre, bias: Optional[torch.Tensor], want_stats: bool = False):
    """y = act(batch_norm(x)) . H + bias with the norm (save_stat) applied to the operand as it streams in; ``want_stats``:
    (y, acc) with the column statistics of y (sum | sum of squares) in a fresh accumulator."""
    lib = L.load()
    dev = L.require_device(x, save_stat)
    x_ = L.f32c(x)
    M, Mk4 = pre[0], pre[2]
    K, Nout = M.shape
    N = x_.shape[0]
    y = torch.empty((N, Nout), dtype=torch.float32, device=dev)
    acc = new_stat_acc(Nout, dev) if want_stats else None
    L.check(lib.phc_gemm_prebuilt_xf(L.ptr(x_), L.ptr(save_stat), int(act), None, L.ptr(M), L.ptr(Mk4), None if bias is None else L.ptr(L.f32c(bias)),
                                     N, K, Nout, L.ptr(y), L.ptr(acc), None, None, L.stream_ptr(dev)), "phc_gemm_prebuilt_xf")
    if L.PROF_ON:
        L.account("phm_gemm_fwd", 2.0 * N * K * Nout, 4.0 * (N * (K + Nout) + K * Nout))
    return (y, acc) if want_stats else y


def bn_bwd_sums_acc(dy: torch.Tensor, x: torch.Tensor, save_stat: torch.Tensor, act: int) -> torch.Tensor:
    """This shard's sum g | sum g x_hat as a fresh accumulator (``bn_bwd_sums`` in the form that all-reduces exactly)."""
    lib = L.load()
    dev = L.require_device(dy, x)
    dy_, x_ = L.f32c(dy), L.f32c(x)
    N, h = x_.shape
    acc = new_stat_acc(h, dev)
    L.check(lib.phc_bn_bwd_partial_acc(L.ptr(dy_), L.ptr(x_), L.ptr(save_stat), None, N, h, int(act), L.ptr(acc), L.stream_ptr(dev)),
            "phc_bn_bwd_partial_acc")
    if L.PROF_ON:
        L.account("bn_bwd_partial", 6.0 * N * h, 8.0 * N * h)
    return acc


def gemm_pre_bwd_apply(dy: torch.Tensor, x: torch.Tensor, save_stat: torch.Tensor, sums_acc: torch.Tensor, count: int, act: int, pre):
    """(dz, ds): dz = the norm backward of (dy, x) on the column sums in ``sums_acc`` (taken over ``count`` rows: every shard's),
    ds = dz . H^T — one launch (phc_gemm_bwd_apply)."""
    lib = L.load()
    dev = L.require_device(dy, x, sums_acc)
    dy_, x_ = L.f32c(dy), L.f32c(x)
    N, K = x_.shape
    MT = pre[1]
    Nout = MT.shape[1]
    dz = torch.empty_like(x_)
    ds = torch.empty((N, Nout), dtype=torch.float32, device=dev)
    d = L.GemmBaDesc()
    d.dout, d.bn_x, d.bn_stat, d.sums_acc, d.n_valid_dev = L.ptr(dy_), L.ptr(x_), L.ptr(save_stat), L.ptr(sums_acc), None
    d.dz, d.dwb, d.Mk4, d.y = L.ptr(dz), None, L.ptr(pre[3]), L.ptr(ds)
    d.N, d.K, d.Nout, d.act = N, K, Nout, int(act)
    d.bw_act, d.bw_x, d.bw_stat, d.bw_sums_acc = 0, None, None, None
    d.count = int(count)
    L.check(lib.phc_gemm_bwd_apply(_C.byref(d), L.stream_ptr(dev)), "phc_gemm_bwd_apply")
    if L.PROF_ON:
        L.account("phm_gemm_dx", 2.0 * N * K * Nout, 4.0 * (4 * N * K + K * Nout))
    return dz, ds


def phm_param_grads_xf(x: torch.Tensor, save_stat: torch.Tensor, act: int, dy: torch.Tensor, A: torch.Tensor, W: torch.Tensor, has_bias: bool):
    """(dA, dW, db | None) of y = act(batch_norm(x)) . H(A, W) + b: the normalised operand is rebuilt from x inside the launch."""
    job = dict(x=L.f32c(x), dy=L.f32c(dy), A=L.f32c(A.detach()), W=L.f32c(W.detach()), need=(True, True, bool(has_bias)),
               params=(None, None, None), xf=(save_stat, None, int(act)))
    (dA, dW, db), = _flush_param_grads([job], return_grads=True)
    return dA, dW, db


def aggregate_sum(x: torch.Tensor, edge_index, edge_attr: Optional[torch.Tensor], msg_encoder: str = "identity",
                  add_self_loops: bool = False, num_dst: Optional[int] = None) -> torch.Tensor:
    """out[i] = Σ_{e: edge_index[1,e]=i} φ(x[edge_index[0,e]] + edge_attr[e])  (+ x[i]).

    ``edge_index`` is the reference's int64 [2, E] tensor (row 0 = source,
    row 1 = target) or an ``EdgeStructure`` built from one."""
    st = edge_index if isinstance(edge_index, EdgeStructure) else edge_structure(edge_index, x.shape[0], num_dst)
    st.prepare(for_backward=torch.is_grad_enabled() and x.requires_grad)
    return _AggregateFn.apply(x, edge_attr, st, _act_id(msg_encoder), bool(add_self_loops))


# ---------------------------------------------------------------------------
# softmax neighbour aggregation
# ---------------------------------------------------------------------------
class _AggregateSoftmaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ea, beta, st: EdgeStructure, act: int, add_self: bool = False):
        lib = L.load()
        dev = L.require_device(x, ea, beta)
        x_, ea_ = L.f32c(x), L.f32c(ea)
        beta_ = L.f32c(beta.reshape(1))
        h = x_.shape[1]
        if ea_ is not None and tuple(ea_.shape) != (st.num_edges, h):
            raise L.PhcError(f"edge_attr must be [{st.num_edges}, {h}], got {tuple(ea_.shape)}")
        rowptr, col, perm = st.csr
        out = torch.empty((st.num_dst, h), dtype=torch.float32, device=dev)
        lse = torch.empty((st.num_dst, h), dtype=torch.float32, device=dev)
        if add_self and x_.shape[0] < st.num_dst:
            raise L.PhcError(f"self loop: x has {x_.shape[0]} rows, {st.num_dst} destination rows")
        out_self = torch.empty_like(out) if add_self else None          # aggregate + x: what the layer returns
        L.check(lib.phc_aggregate_softmax_fwd_self(L.ptr(x_), L.ptr(ea_), L.ptr(rowptr), L.ptr(col), L.ptr(perm), L.ptr(beta_),
                                                   st.num_dst, h, act, L.ptr(out), L.ptr(out_self), L.ptr(lse),
                                                   L.stream_ptr(dev)), "phc_aggregate_softmax_fwd_self")
        if L.PROF_ON:
            E, N = st.num_edges, st.num_dst
            L.account("aggregate_softmax_fwd", 8.0 * E * h, E * (4.0 * h * (2 if ea is not None else 1) + 8) + N * 8.0 * h)
        ctx.st, ctx.act, ctx.has_ea, ctx.add_self = st, act, ea is not None, bool(add_self)
        # a leaf nn.Parameter beta can take its gradient at the end of the backward pass (deferred_param_grads)
        ctx.beta_param = beta if (_DEFER.active and beta.is_leaf and beta.grad_fn is None and beta.requires_grad) else None
        ctx.save_for_backward(x_, ea_ if ea is not None else x_.new_empty(0), beta_, out, lse)
        return out_self if add_self else out

    @staticmethod
    def backward(ctx, dout):
        lib = L.load()
        st: EdgeStructure = ctx.st
        x_, ea_, beta_, out, lse = ctx.saved_tensors
        ea_ = ea_ if ctx.has_ea else None
        dout_ = L.f32c(dout)
        dev = dout_.device
        h = dout_.shape[1]
        rowptr_t, col_t, perm_t = st.csc
        dx = torch.empty((st.num_src, h), dtype=torch.float32, device=dev)
        need_ea = ctx.needs_input_grad[1] and ctx.has_ea
        dea = torch.empty((st.num_edges, h), dtype=torch.float32, device=dev) if need_ea else None
        nparts = lib.phc_softmax_bwd_num_partials(st.num_src)
        parts = torch.empty(nparts, dtype=torch.float32, device=dev)
        L.check(lib.phc_aggregate_softmax_bwd_self(L.ptr(dout_), L.ptr(out), L.ptr(lse), L.ptr(x_), L.ptr(ea_), L.ptr(beta_),
                                                   L.ptr(rowptr_t), L.ptr(col_t), L.ptr(perm_t), st.num_src, h, ctx.act,
                                                   min(st.num_dst, st.num_src) if ctx.add_self else 0, L.ptr(dx), L.ptr(dea),
                                                   L.ptr(parts), L.stream_ptr(dev)), "phc_aggregate_softmax_bwd_self")
        if L.PROF_ON:
            E, N = st.num_edges, st.num_src
            L.account("aggregate_softmax_bwd", 12.0 * E * h, E * (4.0 * h * (5 if need_ea else 4) + 8) + N * 8.0 * h)
        dbeta = None
        if ctx.needs_input_grad[2]:
            beta_param = ctx.beta_param
            if beta_param is not None and _DEFER.active and not torch.is_grad_enabled():
                # d/d beta of all softmax layers of the step: one phc_vec_sums launch at the end of the backward pass
                with _DEFER.lock:
                    _DEFER.beta_jobs.append((parts, beta_param))
            else:
                dbeta = parts.sum().reshape(())
        return (dx if ctx.needs_input_grad[0] else None), dea, dbeta, None, None, None


def aggregate_softmax(x: torch.Tensor, edge_index, edge_attr: Optional[torch.Tensor], beta: torch.Tensor,
                      msg_encoder: str = "identity", num_dst: Optional[int] = None, add_self: bool = False) -> torch.Tensor:
    """out[i] = Σ_e softmax_e(β·m_e)·m_e over the in-edges of i, per feature column, m_e = φ(x[src_e] + edge_attr[e])
    (messagepassing.py:211-214 semantics; empty neighbourhoods give 0).  ``add_self``: + x[i], the layer's self loop
    (messagepassing.py:222-231), inside the same two launches."""
    st = edge_index if isinstance(edge_index, EdgeStructure) else edge_structure(edge_index, x.shape[0], num_dst)
    st.prepare(for_backward=torch.is_grad_enabled() and x.requires_grad)
    return _AggregateSoftmaxFn.apply(x, edge_attr, beta, st, _act_id(msg_encoder), add_self)


class _AggregateSoftmaxEncFn(torch.autograd.Function):
    """Softmax aggregation with the bond encoder read in place (packed codes + tables, no [E, h] forward tensor)."""

    @staticmethod
    def forward(ctx, x, bond_idx, beta, st: EdgeStructure, act: int, add_self: bool, ts: "TableSet", pending, lazy_box, dx_handoff, *tables):
        lib = L.load()
        dev = L.require_device(x, bond_idx, beta)
        # the skip connections' shares of dx (DxHandoff / DxSumHandoff) are added inside this op's backward walk
        # (phc_aggregate_softmax_enc_bwd_res) when x's rows are the walk's source rows
        ctx.dx_handoff = dx_handoff if (dx_handoff is not None and x.shape[0] == st.num_src and x.shape[1] % 4 == 0) else None
        if ctx.dx_handoff is not None:
            ctx.dx_handoff.accepted = True
        # `pending`: x = act(norm(y)) (+ skip) has not been evaluated yet (PendingNorm): the walk does it on the fly for every row it
        # gathers and writes x itself once (phc_aggregate_softmax_norm_enc_fwd, round 5) — or the stand-alone pass runs right here
        fuse_norm = pending is not None and not pending.done and NORM_INTO_AGGREGATE and act == 0 and x.is_contiguous() and \
            x.shape[0] == st.num_src == st.num_dst and st.num_edges > 0 and x.shape[0] * x.shape[1] <= NORM_INTO_AGGREGATE_MAX_WORK and \
            bool(lib.phc_aggregate_softmax_norm_supported(st.num_dst, x.shape[1], ts.F, ts.R))
        if pending is not None and not pending.done and not fuse_norm:
            pending.materialize()
        x_ = L.f32c(x)
        bond_ = _idx64(bond_idx)
        beta_ = L.f32c(beta.reshape(1))
        h = x_.shape[1]
        if h != ts.n * ts.k:
            raise L.PhcError(f"x has {h} features, bond encoder produces {ts.n * ts.k}")
        if tuple(bond_.shape) != (st.num_edges, ts.F):
            raise L.PhcError(f"edge_attr indices must be [{st.num_edges}, {ts.F}], got {tuple(bond_.shape)}")
        if add_self and x_.shape[0] < st.num_dst:
            raise L.PhcError(f"self loop: x has {x_.shape[0]} rows, {st.num_dst} destination rows")
        rowptr, col, _ = st.csr
        codes = st.packed_codes(bond_, ts.vocab_c, ts.F)
        out = torch.empty((st.num_dst, h), dtype=torch.float32, device=dev)
        lse = torch.empty((st.num_dst, h), dtype=torch.float32, device=dev)
        out_self = torch.empty_like(out) if add_self else None
        # `lazy_box` (a dict): the caller feeds this aggregate straight into the fused MLP + norm node — at hidden 128 nothing is
        # launched here: that node's one-launch forward walks the rows itself (mlp_fused.hip, AGG = 2) and writes out / lse /
        # out_self on the way; anybody else who reads them first calls PendingSoftmaxAggregate.materialize (the launch below)
        lazy = lazy_box is not None and MLP_FWD_FUSED and MLP_FWD_FUSED_AGG and SOFTMAX_INTO_MLP and act == 0 and h == 128 and \
            st.num_edges > 0 and x_.shape[0] == st.num_src == st.num_dst and x_.data_ptr() == x.data_ptr() and x_.data_ptr() % 16 == 0 and \
            getattr(st, "_ell", None) is not None and len(ts.key) <= 16 and (pending is None or pending.done or fuse_norm) and \
            _supported("phc_mlp_norm_fwd_fused_agg_supported", st.num_dst, h, ts.n, ts.F, ts.R)
        # (one description of the launch — PendingSoftmaxAggregate.materialize —, run now or left to the consumer)
        pa = PendingSoftmaxAggregate(x_, codes, ts, st, bool(add_self), out_self if add_self else out, pending if fuse_norm else None,
                                     beta_, out, lse, act)
        if fuse_norm:
            assert x_.data_ptr() == pending.out.data_ptr()
        if lazy:
            lazy_box["pa"] = pa
        else:
            pa.materialize()
        ctx.st, ctx.act, ctx.add_self, ctx.ts = st, act, bool(add_self), ts
        ctx.beta_param = beta if (_DEFER.active and beta.is_leaf and beta.grad_fn is None and beta.requires_grad) else None
        ctx.defer_tables = tuple(tables) if (_DEFER.active and st.num_edges > 0 and
                                             all(t.is_leaf and t.grad_fn is None for t in tables)) else None
        ctx.save_for_backward(x_, bond_, beta_, out, lse)
        return out_self if add_self else out

    @staticmethod
    def backward(ctx, dout):
        lib = L.load()
        st: EdgeStructure = ctx.st
        ts: TableSet = ctx.ts
        x_, bond_, beta_, out, lse = ctx.saved_tensors
        dout_ = L.f32c(dout)
        dev = dout_.device
        h = dout_.shape[1]
        rowptr_t, col_t, perm_t = st.csc
        codes_o = st.packed_codes_src(bond_, ts.vocab_c, ts.F)
        need_t = any(ctx.needs_input_grad[10:])
        dx = torch.empty((st.num_src, h), dtype=torch.float32, device=dev)
        dea = torch.empty((st.num_edges, h), dtype=torch.float32, device=dev) if need_t else None
        nparts = lib.phc_softmax_bwd_num_partials(st.num_src)
        parts = torch.empty(nparts, dtype=torch.float32, device=dev)
        extra = ctx.dx_handoff.take() if ctx.dx_handoff is not None else None
        extras = extra if isinstance(extra, list) else ([extra] if extra is not None else [])
        if extras and (not ctx.needs_input_grad[0] or any(tuple(e.shape) != (st.num_src, h) for e in extras)):
            raise L.PhcError("DxHandoff: a parked gradient needs a consumer that computes dx of the same shape")
        in_walk = [e for e in extras if e.is_contiguous() and e.dtype == torch.float32 and e.data_ptr() % 16 == 0][:8] if len(extras) <= 8 else []
        rest = [e for e in extras if not any(e is w for w in in_walk)]
        arr = (_C.c_void_p * max(len(in_walk), 1))(*[e.data_ptr() for e in in_walk])
        L.check(lib.phc_aggregate_softmax_enc_bwd_res(L.ptr(dout_), L.ptr(out), L.ptr(lse), L.ptr(x_), L.ptr(codes_o), L.ptr(ts.ptrs),
                                                      ts.vocab_c, ts.F, ts.n, ts.k, L.ptr(beta_), L.ptr(rowptr_t), L.ptr(col_t),
                                                      L.ptr(perm_t), st.num_src, ctx.act,
                                                      min(st.num_dst, st.num_src) if ctx.add_self else 0, arr if in_walk else None,
                                                      len(in_walk), L.ptr(dx), L.ptr(dea), L.ptr(parts), L.stream_ptr(dev)),
                "phc_aggregate_softmax_enc_bwd_res")
        for e in rest:                                       # (shapes / layouts the walk does not take: plain adds)
            dx += e
        if L.PROF_ON:
            E, N = st.num_edges, st.num_src
            L.account("aggregate_softmax_bwd", 12.0 * E * h, E * (4.0 * h * (4 if need_t else 3) + 12) + N * 8.0 * h)
        dbeta = None
        if ctx.needs_input_grad[2]:
            if ctx.beta_param is not None and _DEFER.active and not torch.is_grad_enabled():
                with _DEFER.lock:
                    _DEFER.beta_jobs.append((parts, ctx.beta_param))
            else:
                dbeta = parts.sum().reshape(())
        tgrads = [None] * ts.num_tables
        if need_t:
            if ctx.defer_tables is not None and _DEFER.active and not torch.is_grad_enabled():
                params = [t if ng else None for t, ng in zip(ctx.defer_tables, ctx.needs_input_grad[10:])]
                with _DEFER.lock:                            # d enc = dea: an encoder job over the edge rows
                    _DEFER.atom_jobs.append(dict(dout=dea, idx=bond_, ts=ts, params=params))
            else:
                flat = torch.empty(ts.n * ts.R * ts.k, dtype=torch.float32, device=dev)
                ws_bytes = lib.phc_encode_bwd_ws_bytes(st.num_edges, ts.F, ts.vocab_c, ts.n, ts.k)
                ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
                L.check(lib.phc_encode_bwd(L.ptr(dea), L.ptr(bond_), st.num_edges, ts.F, ts.vocab_c, ts.n, ts.k, L.ptr(flat),
                                           L.ptr(ws), ws_bytes, L.stream_ptr(dev)), "phc_encode_bwd")
                tgrads = list(ts.grad_views(flat))
        return ((dx if ctx.needs_input_grad[0] else None), None, dbeta, None, None, None, None, None, None, None, *tgrads)


def aggregate_softmax_encoded(x: torch.Tensor, edge_index, bond_idx: torch.Tensor, ts: "TableSet", tables, beta: torch.Tensor,
                              msg_encoder: str = "identity", add_self: bool = False, num_dst: Optional[int] = None,
                              pending=None, lazy: bool = False, dx_handoff=None) -> torch.Tensor:
    """aggregate_softmax with ``edge_attr = bond encoder(bond_idx)`` evaluated inside the kernels (needs
    ``ts.fusable_into_aggregate`` and F <= 4, sum(vocab) <= 256: the packed-code limits)."""
    st = edge_index if isinstance(edge_index, EdgeStructure) else edge_structure(edge_index, x.shape[0], num_dst)
    ts = ts.vector_form(tables)
    bond_idx = _idx64(bond_idx)
    st.prepare(for_backward=torch.is_grad_enabled() and x.requires_grad, codes_for=(bond_idx, ts.vocab_c, ts.F), codes_src=True)
    box = {} if (lazy and torch.is_grad_enabled()) else None
    res = _AggregateSoftmaxEncFn.apply(x, bond_idx, beta, st, _act_id(msg_encoder), add_self, ts, pending, box, dx_handoff,
                                       *[t for comp in tables for t in comp])
    if box and box.get("pa") is not None:
        res._phc_pending_agg = box["pa"]
    return res


# ---------------------------------------------------------------------------
# read-out: per-graph sum
# ---------------------------------------------------------------------------
class _SegmentSumFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, st: SegmentStructure):
        lib = L.load()
        dev = L.require_device(x)
        x_ = L.f32c(x)
        h = x_.shape[1]
        rowptr, col, _ = st.csr
        out = torch.empty((st.num_segments, h), dtype=torch.float32, device=dev)
        L.check(lib.phc_aggregate_fwd(L.ptr(x_), None, L.ptr(rowptr), L.ptr(col), None, st.num_segments, h, 0, 0,
                                      L.ptr(out), L.stream_ptr(dev)), "phc_aggregate_fwd(segment sum)")
        if L.PROF_ON:
            L.account("aggregate_fwd", 1.0 * x_.shape[0] * h,
                      x_.shape[0] * (4.0 * h + 4) + st.num_segments * 4.0 * h + 4.0 * (st.num_segments + 1))
        ctx.st = st
        ctx.n_rows = x_.shape[0]
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = L.load()
        st: SegmentStructure = ctx.st
        dout_ = L.f32c(dout)
        dev = dout_.device
        h = dout_.shape[1]
        dx = torch.empty((ctx.n_rows, h), dtype=torch.float32, device=dev)
        L.check(lib.phc_gather_rows_i64(L.ptr(dout_), L.ptr(st.batch), ctx.n_rows, h, L.ptr(dx), L.stream_ptr(dev)),
                "phc_gather_rows_i64")
        if L.PROF_ON:
            L.account("gather_rows", 0.0, ctx.n_rows * (8.0 * h + 8))
        return dx, None


def global_add_pool(x: torch.Tensor, batch, num_graphs: Optional[int] = None, is_sorted: bool = False) -> torch.Tensor:
    """out[g] = Σ_{i: batch[i]=g} x[i]  (torch_geometric.nn.global_add_pool semantics).
    ``is_sorted=True`` promises a non-decreasing ``batch`` (PyG collation) and skips the grouping sort."""
    st = batch if isinstance(batch, SegmentStructure) else segment_structure(batch, num_graphs, is_sorted)
    return _SegmentSumFn.apply(x, st)




# the gate layer's backward-data map inside the gated sum's backward launch (phc_gated_pool_bwd_gate; k = 32 / 64, composed gate)
POOL_BWD_GATE = __import__("os").environ.get("PHC_POOL_BWD_GATE", "1") != "0"


class _GatedSegmentSumFn(torch.autograd.Function):
    """out[g, c*k+j] = Σ_{r in graph g} x[r, c*k+j]·σ(gate[r, j]) over rows that are already grouped by graph."""

    @staticmethod
    def forward(ctx, x, gate, st: SegmentStructure, n_comp: int, dx_handoff=None, pending_gate=None):
        lib = L.load()
        ctx.dx_handoff = dx_handoff
        dev = L.require_device(x, gate)
        if pending_gate is not None and not (st.is_sorted and st._csr is None):
            pending_gate.materialize()
            pending_gate = None
        x_, g_ = L.f32c(x), L.f32c(gate)
        N, h = x_.shape
        if h % n_comp or tuple(g_.shape) != (N, h // n_comp):
            raise L.PhcError(f"gate must be [{N}, {h // n_comp}] for x [{N}, {h}] with {n_comp} components")
        out = torch.empty((st.num_segments, h), dtype=torch.float32, device=dev)
        pg = pending_gate
        if pg is not None and not pg.run_with_pool(st, n_comp, out):
            pg.materialize()                     # outside the one-launch form after all: finish norm + gate before the pool reads them
            pg = None
        if pg is not None:
            pass
        elif st.is_sorted and st._csr is None:
            # collated batch: the workgroups find their rows in the sorted `batch` vector themselves (no pointer launch)
            L.check(lib.phc_gated_pool_fwd_sorted(L.ptr(x_), L.ptr(g_), L.ptr(st.batch), N, st.num_segments, h, n_comp,
                                                  L.ptr(out), L.stream_ptr(dev)), "phc_gated_pool_fwd_sorted")
        else:
            L.check(lib.phc_gated_pool_fwd(L.ptr(x_), L.ptr(g_), L.ptr(st.csr[0]), st.num_segments, h, n_comp, L.ptr(out),
                                           L.stream_ptr(dev)), "phc_gated_pool_fwd")
        if L.PROF_ON:
            L.account("gated_pool_fwd", 2.0 * N * h, 4.0 * (N * (h + h // n_comp) + st.num_segments * h))
        ctx.save_for_backward(x_, g_, st.batch)
        ctx.meta = (st.num_segments, n_comp)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = L.load()
        x_, g_, batch = ctx.saved_tensors
        G, n_comp = ctx.meta
        dout_ = L.f32c(dout)
        dev = dout_.device
        N, h = x_.shape
        dx = torch.empty_like(x_) if ctx.needs_input_grad[0] else None
        dg = torch.empty_like(g_) if ctx.needs_input_grad[1] else None
        ho = ctx.dx_handoff
        MT = getattr(ho, "gate_MT", None) if (ho is not None and ho.accepted) else None
        if POOL_BWD_GATE and MT is not None and dx is not None and dg is not None and N > 0 and tuple(MT.shape) == (g_.shape[1], h) \
                and MT.is_contiguous() and lib.phc_gated_pool_bwd_gate_supported(h, n_comp) and dout_.data_ptr() % 16 == 0:
            # the gate layer's dx = dg . M^T rides in this launch: dx is x's whole gradient from the read-out
            L.check(lib.phc_gated_pool_bwd_gate(L.ptr(dout_), L.ptr(x_), L.ptr(g_), L.ptr(batch), N, G, h, n_comp, L.ptr(MT), L.ptr(dx),
                                                L.ptr(dg), L.stream_ptr(dev)), "phc_gated_pool_bwd_gate")
            if L.PROF_ON:
                L.account("gated_pool_bwd", 4.0 * N * h + 2.0 * N * h * g_.shape[1], 4.0 * (N * (2 * h + 2 * h // n_comp) + G * h))
            ho.tensor, ho.complete = dx, True
            return None, dg, None, None, None, None
        L.check(lib.phc_gated_pool_bwd(L.ptr(dout_), L.ptr(x_), L.ptr(g_), L.ptr(batch), N, G, h, n_comp, L.ptr(dx), L.ptr(dg),
                                       L.stream_ptr(dev)), "phc_gated_pool_bwd")
        if L.PROF_ON:
            L.account("gated_pool_bwd", 4.0 * N * h, 4.0 * (N * (2 * h + 2 * h // n_comp) + G * h))
        if ctx.dx_handoff is not None and ctx.dx_handoff.accepted and dx is not None:
            ctx.dx_handoff.tensor = dx                      # added inside the gate layer's dx GEMM (DxHandoff)
            dx = None
        return dx, dg, None, None, None, None


def gated_segment_sum(x: torch.Tensor, gate: torch.Tensor, batch, n_comp: int, num_graphs: Optional[int] = None,
                      is_sorted: bool = False, dx_handoff=None) -> torch.Tensor:
    """Soft-attention read-out (phc/hypercomplex/pooling.py:57-66): per-graph sum of x scaled by σ(gate), the gate
    [N, h/n_comp] broadcast over the n_comp components.  One kernel each way when ``batch`` is sorted (collated
    mini-batches are); an unsorted ``batch`` takes the sigmoid / multiply / segment-sum route."""
    st = batch if isinstance(batch, SegmentStructure) else segment_structure(batch, num_graphs, is_sorted)
    pg = getattr(gate, "_phc_pending_gate", None)
    if pg is not None:
        gate._phc_pending_gate = None
    if not st.is_sorted:
        if pg is not None:
            pg.materialize()
        N, h = x.shape
        gated = (x.view(N, n_comp, h // n_comp) * torch.sigmoid(gate).unsqueeze(1)).reshape(N, h)
        return _SegmentSumFn.apply(gated, st)
    return _GatedSegmentSumFn.apply(x, gate, st, n_comp, dx_handoff, pg)


# ---------------------------------------------------------------------------
# The read-out's gate as ONE linear map (pooling.py:57-66: Linear_{h->kg}(PHMLinear_{in->h}(x)), nothing between the two maps):
#   gate = x . M + c,   M = H(A, W) . Wg^T [in, kg],   c = b . Wg^T + bg
# M, M^T and c are assembled by the step's operand-build launch (phc_phm_H_job.flags == 3); forward one GEMM instead of two and no
# [N, h] intermediate; backward dx = dg . M^T (K = kg); parameter gradients from dM^T = dg^T . x (a dense job of the step tail) and one
# small launch behind it (phc_gate_param_grads: dH = dM . Wg re-evaluated inside the contractions, dWg = dM^T . H + dbg (x) b).
# Same function as the two-map form up to fp32 round-off.  PHC_GATE_COMPOSED=0 restores the two GEMMs.
# ---------------------------------------------------------------------------
GATE_COMPOSED = __import__("os").environ.get("PHC_GATE_COMPOSED", "1") != "0"


def gate_composable(W: torch.Tensor, gate_weight: torch.Tensor) -> bool:
    """Shapes the composed form serves: a square PHM layer [h -> h] (h % 4 == 0, h <= 256) in front of Linear [h -> kg], kg % 4 == 0,
    gate_weight small enough for the LDS staging of phc_gate_param_grads."""
    if W.dim() != 3 or gate_weight.dim() != 2:
        return False
    n, kin, kout = W.shape
    h, kg = n * kout, gate_weight.shape[0]
    return kin == kout and gate_weight.shape[1] == h and h % 4 == 0 and kg % 4 == 0 and h <= FUSED_MAX_WIDTH and (kg + n) * (h + n) + n ** 3 <= 30000


def _gate_operands(A, W, b, Wg, bg):
    """(M [in, kg], M^T [kg, in], c [kg]): from the step's operand table when it is current, else built now (one launch)."""
    ent = _PREBUILT.table.get(("gate", A.data_ptr(), W.data_ptr(), Wg.data_ptr())) if _PREBUILT.table else None
    if ent is not None and ent[3] == tuple(t._version if t is not None else -1 for t in (A, W, b, Wg, bg)):
        return ent[0], ent[1], ent[2]
    lib = L.load()
    dev = W.device
    n, kin, kout = W.shape
    kg, inw = Wg.shape[0], n * kin
    M = torch.empty((inw, kg), dtype=torch.float32, device=dev)
    MT = torch.empty((kg, inw), dtype=torch.float32, device=dev)
    c = torch.empty(kg, dtype=torch.float32, device=dev)
    arr = (L.HJob * 1)()
    arr[0] = L.HJob(L.ptr(A), L.ptr(W), L.ptr(M), L.ptr(MT), n, kin, kout, 3, None, None, L.ptr(Wg), L.ptr(bg), L.ptr(b), L.ptr(c), kg)
    L.check(lib.phc_phm_build_H_batched(arr, 1, L.stream_ptr(dev)), "phc_phm_build_H_batched(gate)")
    return M, MT, c


class _GateLinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, A, W, b, Wg, bg, dx_handoff=None, lazy_box=None):
        lib = L.load()
        dev = L.require_device(x, A, W, b, Wg, bg)
        n, kin, kout = W.shape
        inw, kg = n * kin, Wg.shape[0]
        x_ = L.f32c(x)
        A_, W_, b_, Wg_, bg_ = (L.f32c(t.detach()) if t is not None else None for t in (A, W, b, Wg, bg))
        N = x_.shape[0]
        M, MT, c = _gate_operands(A_, W_, b_, Wg_, bg_)
        gate = torch.empty((N, kg), dtype=torch.float32, device=dev)
        pend = lazy_box.get("pend") if lazy_box is not None else None      # the layer in front's un-applied norm (PendingNorm), if any
        lazy = lazy_box is not None and lazy_box.get("lazy") and READOUT_FUSED and N > 0 and x_.data_ptr() % 16 == 0 and \
            bool(lib.phc_readout_fused_supported(N, 1, inw, n, kg)) and (c is None or c.data_ptr() % 16 == 0)
        if lazy:
            # nothing is launched here: the gated sum that consumes this gate builds it itself (phc_readout_fused_fwd);
            # anybody else calls PendingGate.materialize first
            lazy_box["pg"] = PendingGate(x_, M, c, gate, pend if (pend is not None and not pend.done) else None)
        else:
            if pend is not None:
                pend.materialize()
            if N:
                L.check(lib.phc_gemm_prebuilt(L.ptr(x_), L.ptr(M), None, L.ptr(c), None, N, inw, kg, L.ptr(gate), 0, L.stream_ptr(dev)),
                        "phc_gemm_prebuilt(gate)")
                if L.PROF_ON:
                    L.account("phm_gemm_fwd", 2.0 * N * inw * kg, 4.0 * (N * (inw + kg) + inw * kg))
        ctx.dx_handoff = dx_handoff
        if dx_handoff is not None:
            dx_handoff.accepted = True
            dx_handoff.gate_MT = MT                          # the gated sum's backward may apply this map itself (POOL_BWD_GATE)
            dx_handoff.complete = False
        pre = _prebuilt_for(A_, W_) if _PREBUILT.table else None
        ctx.H = pre[0] if pre is not None else None          # the PHM layer's own operand matrix, for dWg = dM^T . H
        ctx.MT = MT
        ctx.params = (A, W, b, Wg, bg)
        ctx.save_for_backward(x_, A_, W_, Wg_)
        ctx.b_ = b_
        return gate

    @staticmethod
    def backward(ctx, dg):
        lib = L.load()
        x_, A_, W_, Wg_ = ctx.saved_tensors
        b_ = ctx.b_
        A, W, b, Wg, bg = ctx.params
        dev = x_.device
        n, kin, kout = W_.shape
        inw, kg = n * kin, Wg_.shape[0]
        N = x_.shape[0]
        dg_ = L.f32c(dg)
        sp = L.stream_ptr(dev)
        dx = None
        extra = ctx.dx_handoff.take() if ctx.dx_handoff is not None else None
        if extra is not None and getattr(ctx.dx_handoff, "complete", False):
            # the gated sum's backward launch already added dg . M^T to its share (phc_gated_pool_bwd_gate): nothing to launch here
            ctx.dx_handoff.complete = False
            if not ctx.needs_input_grad[0]:
                raise L.PhcError("gate_linear: a complete dx without a consumer")
            dx = extra
        elif ctx.needs_input_grad[0]:
            dx = torch.empty((N, inw), dtype=torch.float32, device=dev)
            if N:
                L.check(lib.phc_gemm_prebuilt(L.ptr(dg_), L.ptr(ctx.MT), None, None, L.ptr(L.f32c(extra)) if extra is not None else None, N, kg,
                                              inw, L.ptr(dx), 1, sp), "phc_gemm_prebuilt(gate dx)")
                if L.PROF_ON:
                    L.account("phm_gemm_dx", 2.0 * N * inw * kg, 4.0 * (N * (inw + kg) + inw * kg))
            elif extra is not None:
                dx = extra
        elif extra is not None:
            raise L.PhcError("gate_linear: a parked dx share without a dx to add it to")
        need = ctx.needs_input_grad
        nA, nW, nb, nWg, nbg = need[1], need[2], need[3] and b is not None, need[4], need[5] and bg is not None
        if not (nA or nW or nb or nWg or nbg) or N == 0:
            return dx, None, None, None, None, None, None, None
        H = ctx.H
        results = {}

        def post(grads):
            # grads = (None, dM^T [kg, in] in the nn.Linear layout, dbg [kg]) of the dense job (x, dg)
            _, dMT, dbg = grads
            Hm = H if H is not None else phm_build_operands([(A_, W_)])[0][0]
            dA = torch.empty_like(A_) if nA else None
            dW = torch.empty_like(W_) if nW else None
            db = torch.empty(n * kout, dtype=torch.float32, device=dev) if nb else None
            dWg = torch.empty_like(Wg_) if nWg else None
            L.check(lib.phc_gate_param_grads(L.ptr(dMT), L.ptr(dbg), L.ptr(Hm), L.ptr(Wg_), L.ptr(b_), L.ptr(A_), L.ptr(W_), n, kin, kout, kg,
                                             L.ptr(dA), L.ptr(dW), L.ptr(db), L.ptr(dWg), L.stream_ptr(dev)), "phc_gate_param_grads")
            results["g"] = (dA, dW, db, dWg, dbg if nbg else None)

        job = dict(x=x_, dy=dg_, A=_one(dev), W=Wg_, need=(False, True, True), params=(None, None, None), dense=True, post=post)
        deferred = _DEFER.active and not torch.is_grad_enabled() and all(p is None or p.is_leaf for p in (A, W, b, Wg, bg))
        if deferred:
            def post_assign(grads):
                post(grads)
                _assign_grads((A if nA else None, W if nW else None, b if nb else None, Wg if nWg else None, bg if nbg else None),
                              results["g"])
            job["post"] = post_assign
            if H is not None:
                # what the step tail needs to produce these gradients inside its own last launch (phc_step_tail_ex) instead
                job["gate"] = dict(H=H, Wg=Wg_, b=b_, A=A_, W=W_, need=(nA, nW, nb, nWg, nbg),
                                   params=(A if nA else None, W if nW else None, b if nb else None, Wg if nWg else None,
                                           bg if nbg else None))
            with _DEFER.lock:
                _DEFER.jobs.append(job)
            return dx, None, None, None, None, None, None, None
        job["post"] = None
        (grads,) = _flush_param_grads([job], return_grads=True)
        post(grads)
        dA, dW, db, dWg, dbg = results["g"]
        return dx, dA, dW, db, dWg, dbg, None, None


def gate_linear(x: torch.Tensor, phm_rule, W, b, gate_weight, gate_bias, dx_handoff=None, lazy: bool = False) -> torch.Tensor:
    """Pre-sigmoid gate of the soft-attention read-out, ``Linear(PHMLinear(x))`` as one linear map (see above).  ``lazy``: the gate
    goes straight into ``gated_segment_sum`` over a sorted batch — the launch may be left to that node (``PendingGate``)."""
    pend = getattr(x, "_phc_pending", None)
    want = bool(lazy and torch.is_grad_enabled())
    box = dict(pend=pend, lazy=want) if (want or pend is not None) else None
    gate = _GateLinearFn.apply(x, phm_rule, W, b, gate_weight, gate_bias, dx_handoff, box)
    if box and box.get("pg") is not None:
        gate._phc_pending_gate = box["pg"]
        if box["pg"].pending is not None:
            x._phc_pending = None                            # rides with the gate now
    elif pend is not None:
        x._phc_pending = None                                # (materialised inside the node)
    return gate


READOUT_FUSED = __import__("os").environ.get("PHC_READOUT_FUSED", "1") != "0"


class PendingGate:
    """gate = x . M + c of the soft-attention read-out that has NOT been launched (``gate`` is allocated): the gated per-graph sum
    builds it itself (phc_readout_fused_fwd, csrc/readout.hip) together with the layer in front's un-applied norm (``pending``);
    ``materialize`` = the launches that would have run (norm pass, gate GEMM)."""

    def __init__(self, x_, M, c, gate, pending):
        self.x_, self.M, self.c, self.gate, self.pending = x_, M, c, gate, pending
        self.done = False

    def materialize(self):
        if self.done:
            return
        lib = L.load()
        if self.pending is not None:
            self.pending.materialize()
        N, inw = self.x_.shape
        kg = self.gate.shape[1]
        L.check(lib.phc_gemm_prebuilt(L.ptr(self.x_), L.ptr(self.M), None, L.ptr(self.c), None, N, inw, kg, L.ptr(self.gate), 0,
                                      L.stream_ptr(self.x_.device)), "phc_gemm_prebuilt(gate)")
        if L.PROF_ON:
            L.account("phm_gemm_fwd", 2.0 * N * inw * kg, 4.0 * (N * (inw + kg) + inw * kg))
        self.done = True

    def run_with_pool(self, st, n_comp, out) -> bool:
        """Norm (if pending) + gate + gated per-graph sum in one launch; False = outside the fused form (nothing launched)."""
        lib = L.load()
        N, h = self.x_.shape
        kg = self.gate.shape[1]
        if self.done or not st.is_sorted or not lib.phc_readout_fused_supported(N, st.num_segments, h, n_comp, kg):
            return False
        d = L.ReadoutDesc()
        d.batch_sorted_i64, d.N, d.num_graphs, d.h, d.n_comp, d.kg = L.ptr(st.batch), N, st.num_segments, h, n_comp, kg
        d.M, d.c, d.gate, d.out = L.ptr(self.M), L.ptr(self.c), L.ptr(self.gate), L.ptr(out)
        keep = None
        pn = self.pending if (self.pending is not None and not self.pending.done) else None
        if pn is not None:
            keep = pn._norm_args()
            w, bb, rm, rv, nbt, no = keep
            cast = lambda a: _C.cast(a, _C.c_void_p) if a is not None else None
            assert self.x_.data_ptr() == pn.out.data_ptr()
            d.x, d.norm_acc, d.n_valid_dev = L.ptr(pn.y2), L.ptr(pn.acc2), L.ptr(pn.n_valid)
            d.weight, d.bias, d.running_mean, d.running_var, d.num_batches_tracked = cast(w), cast(bb), cast(rm), cast(rv), cast(nbt)
            d.norm_n_comp, d.norm_act, d.momentum, d.eps = no, pn.act_o, float(pn.outer["momentum"]), float(pn.outer["eps"])
            d.skip, d.save_stat, d.x_out = L.ptr(pn.skip), L.ptr(pn.save_o), L.ptr(pn.out)
        else:
            d.x = L.ptr(self.x_)
        L.check(lib.phc_readout_fused_fwd(_C.byref(d), L.stream_ptr(self.x_.device)), "phc_readout_fused_fwd")
        if pn is not None:
            pn.done = True
        self.done = True
        return True


# Tensors known to hold exactly 1.0 for their whole life (harness.backward_seed, GraphedTrainStep's seed): a backward
# that receives one of them as its upstream gradient may skip the multiplication by it.
_UNIT_SEEDS = set()


def register_unit_seed(t: torch.Tensor) -> torch.Tensor:
    _UNIT_SEEDS.add((t.data_ptr(), t.device))
    return t


def unregister_unit_seed(t: torch.Tensor) -> None:
    """Call before the tensor is released: its address may be handed to a tensor that does not hold 1."""
    _UNIT_SEEDS.discard((t.data_ptr(), t.device))


def _masked_loss_forward(kind, ctx, logits, y):
    """Shared forward of the NaN-masked mean losses; ``kind``: entry-point family phc_<kind>_masked_{fwd, fwd_unit, bwd}."""
    lib = L.load()
    dev = L.require_device(logits, y)
    z_, y_ = L.f32c(logits), L.f32c(y)
    if z_.shape != y_.shape:
        raise L.PhcError(f"logits {tuple(z_.shape)} vs labels {tuple(y_.shape)}")
    total = z_.numel()
    out = torch.empty(3, dtype=torch.float32, device=dev)             # loss | sum | labeled count
    ctx.dz_unit = None
    ctx.kind = kind
    if total <= (1 << 12) and ctx.needs_input_grad[0]:
        # one workgroup: loss and d loss / d logits (for an upstream gradient of 1) in the same launch
        ctx.dz_unit = torch.empty_like(z_)
        L.check(getattr(lib, f"phc_{kind}_masked_fwd_unit")(L.ptr(z_), L.ptr(y_), total, L.ptr(out), L.ptr(out) + 4, L.ptr(ctx.dz_unit),
                                                           L.stream_ptr(dev)), f"phc_{kind}_masked_fwd_unit")
    else:
        ws_bytes = lib.phc_bce_masked_ws_bytes(total)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev) if total > (1 << 12) else None
        L.check(getattr(lib, f"phc_{kind}_masked_fwd")(L.ptr(z_), L.ptr(y_), total, L.ptr(out), L.ptr(out) + 4, L.ptr(ws),
                                                      ws_bytes if ws is not None else 0, L.stream_ptr(dev)), f"phc_{kind}_masked_fwd")
    ctx.save_for_backward(z_, y_, out)
    ctx.shape = logits.shape
    return out[0]


def _masked_loss_backward(ctx, dloss):
    lib = L.load()
    z_, y_, out = ctx.saved_tensors
    if ctx.dz_unit is not None and dloss.numel() == 1 and (dloss.data_ptr(), dloss.device) in _UNIT_SEEDS:
        return ctx.dz_unit.reshape(ctx.shape), None              # upstream gradient is the resident 1: nothing to launch
    dl = L.f32c(dloss.reshape(1))
    dz = torch.empty_like(z_)
    L.check(getattr(lib, f"phc_{ctx.kind}_masked_bwd")(L.ptr(z_), L.ptr(y_), L.ptr(dl), L.ptr(out) + 4, z_.numel(), L.ptr(dz),
                                                      L.stream_ptr(dz.device)), f"phc_{ctx.kind}_masked_bwd")
    return dz.reshape(ctx.shape), None


class _MaskedBCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, y):
        return _masked_loss_forward("bce", ctx, logits, y)

    @staticmethod
    def backward(ctx, dloss):
        return _masked_loss_backward(ctx, dloss)


class _MaskedL1Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, y):
        return _masked_loss_forward("l1", ctx, pred, y)

    @staticmethod
    def backward(ctx, dloss):
        return _masked_loss_backward(ctx, dloss)


# The training loss inside the prediction head's forward launch (phc_downstream_desc.loss_*, ABI 8): a step that knows its labels
# before the model runs says so with ``expect_loss``; a head whose shapes allow it (the 16-row kernels, at most 4096 logits) then
# leaves the loss and d loss / d logits behind, and ``masked_bce_with_logits`` / ``masked_l1`` on those logits and labels launch nothing.
HEAD_LOSS = __import__("os").environ.get("PHC_HEAD_LOSS", "1") != "0"
# ... and the head's BACKWARD in that same launch (phc_downstream_desc.fused_bwd; csrc/downstream.hip downstream2_fb_kernel); 0 = its own launch
HEAD_FUSED_BWD = __import__("os").environ.get("PHC_HEAD_FUSED_BWD", "1") != "0"
_LOSS_HINT = threading.local()
_LOSS_KINDS = {"bce": 0, "l1": 1}


@contextlib.contextmanager
def expect_loss(y: Optional[torch.Tensor], kind: str):
    """``y``: the labels the step will hand to ``masked_bce_with_logits`` (kind "bce") / ``masked_l1`` ("l1") with the model's output."""
    prev = getattr(_LOSS_HINT, "spec", None)
    ok = HEAD_LOSS and y is not None and y.is_cuda and y.dtype == torch.float32 and y.is_contiguous() and y.dim() == 2
    _LOSS_HINT.spec = (y, _LOSS_KINDS[kind]) if ok else None
    try:
        yield
    finally:
        _LOSS_HINT.spec = prev


_DN_LOSS_GRANULES: dict = {}


def _dn_loss_granules(dev, B: int, total: int) -> torch.Tensor:
    """The logits' hand-over words of the head's in-launch loss, per device, stream and row count (zero before the first use, owned by
    the kernels from then on: generation-tagged granules, as ``_dn_buffers``' partial buffer whose generation they share)."""
    key = (dev, torch.cuda.current_stream(dev).cuda_stream, B, total)
    buf = _DN_LOSS_GRANULES.get(key)
    if buf is None:
        buf = _DN_LOSS_GRANULES[key] = torch.zeros(total, dtype=torch.int64, device=dev)
    return buf


class _HeadLossFn(torch.autograd.Function):
    """The masked mean loss the head's forward launch already computed: (loss | sum | count, d loss / d logits for an upstream 1)."""

    @staticmethod
    def forward(ctx, logits, out3, dz_unit):
        ctx.save_for_backward(dz_unit)
        ctx.shape = logits.shape
        return out3[0]

    @staticmethod
    def backward(ctx, dloss):
        (dz_unit,) = ctx.saved_tensors
        if dloss.numel() == 1 and (dloss.data_ptr(), dloss.device) in _UNIT_SEEDS:
            return dz_unit.reshape(ctx.shape), None, None        # upstream gradient is the resident 1: nothing to launch
        return (dz_unit * dloss).reshape(ctx.shape), None, None


def _head_loss(logits: torch.Tensor, y: torch.Tensor, kind: int):
    res = getattr(logits, "_phc_loss", None)
    if res is None or res[3] != kind or res[2].data_ptr() != y.data_ptr() or tuple(res[2].shape) != tuple(y.shape) or \
            res[2]._version != res[4] or tuple(logits.shape) != tuple(y.shape):
        return None
    return _HeadLossFn.apply(logits, res[0], res[1])


def masked_l1(pred: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """mean |pred - y| over the entries whose target is not NaN (benchmarks/train_zinc.py: torch.nn.L1Loss; the padding graphs
    of a fixed-shape batch carry NaN targets) — one kernel each way instead of a dozen element-wise launches."""
    done = _head_loss(pred, y, 1)
    return done if done is not None else _MaskedL1Fn.apply(pred, y)


def masked_bce_with_logits(logits: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """mean BCE-with-logits over the labeled entries, NaN in ``y`` = unlabeled (benchmarks/train_hiv.py:174-178:
    ``mask = ~isnan(y); criterion(logits[mask], y[mask])``) — one kernel each way, no host synchronisation."""
    done = _head_loss(logits, y, 0)
    return done if done is not None else _MaskedBCEFn.apply(logits, y)


# ---------------------------------------------------------------------------
# integer-feature encoders (atom / bond embeddings) read in place
# ---------------------------------------------------------------------------
import ctypes as _C   # noqa: E402
import weakref as _weakref  # noqa: E402


class TableSet:
    """The n*F embedding tables of one PHM encoder as the engine sees them: a device
    array of table pointers (entry c*F+f) plus the host-side vocabulary sizes.

    The tables stay the reference's separate ``nn.Embedding`` weights; the pointer
    array is rebuilt only when a weight has moved (``.to()``, ``load_state_dict``
    with assign)."""

    def __init__(self, tables):
        self.refresh(tables)

    def refresh(self, tables):
        self.n = len(tables)
        self.F = len(tables[0])
        self.k = int(tables[0][0].shape[1])
        self.vocab = [int(t.shape[0]) for t in tables[0]]
        self.R = sum(self.vocab)
        self.vocab_c = (_C.c_int32 * self.F)(*self.vocab)
        flat = [t for comp in tables for t in comp]
        for t in flat:
            if t.dtype != torch.float32 or not t.is_contiguous() or t.shape[1] != self.k:
                raise L.PhcError("embedding tables must be contiguous fp32 [V_f, k]")
        self.dev = L.require_device(*flat)
        self.key = tuple(t.data_ptr() for t in flat)
        self.ptrs = torch.tensor(self.key, dtype=torch.int64, device=self.dev)
        offs = [0]
        for v in self.vocab:
            offs.append(offs[-1] + v)
        self.offsets = offs

    def current(self, tables) -> "TableSet":
        flat = [t for comp in tables for t in comp]
        if tuple(t.data_ptr() for t in flat) != self.key:
            self.refresh(tables)
        return self

    def grad_views(self, flat_grad: torch.Tensor):
        """Per-table [V_f, k] views of the flat [c][f][V_f][k] gradient buffer."""
        out = []
        for c in range(self.n):
            for f in range(self.F):
                b = (c * self.R + self.offsets[f]) * self.k
                out.append(flat_grad[b:b + self.vocab[f] * self.k].view(self.vocab[f], self.k))
        return out

    @property
    def fusable_into_aggregate(self) -> bool:
        return self.k % 4 == 0 and self.R * self.n * self.k * 4 <= 48 * 1024 and self.F <= 16

    @property
    def num_tables(self) -> int:
        """Number of nn.Embedding weights behind this set (= gradients an autograd node returns for it)."""
        return self.n * self.F

    def vector_form(self, tables) -> "TableSet":
        """The table set the vector kernels can read: this one when k % 4 == 0 (or when no float4 form exists: (n*k) % 4 != 0);
        otherwise its flat ONE-component image (``FlatTableSet``), refreshed from the tables' current values."""
        if self.k % 4 == 0 or (self.n * self.k) % 4 != 0 or not FLAT_TABLES:
            return self
        flat = getattr(self, "_flat_form", None)
        if flat is None or flat.parent_key != self.key:
            flat = self._flat_form = FlatTableSet(self)
        flat.refresh_image()
        return flat


FLAT_TABLES = __import__("os").environ.get("PHC_FLAT_TABLES", "1") != "0"      # A/B switch
_FLAT_STEP = threading.local()           # .packed: SERIAL numbers of the images refreshed in the running step (prebuilt_H context)


class FlatTableSet(TableSet):
    """One-component image of an encoder whose component size is not a multiple of four (the reference's published 180- /
    200-wide models: k = 45, 50): ``image[off_f + r, c*k + j] = table(c, f)[r, j]``, a ``[R, n*k]`` buffer whose rows are 16-byte
    aligned whatever k is.  To the kernels it is an encoder with n = 1, k = n*k and F tables ``image[off_f : off_f + V_f]`` — every
    vector path (float4 encoders, the aggregation with the bond embedding read in place, the MFMA / register-resident table
    gradients) applies unchanged.  Costs: one small launch to refresh the image when the tables have changed (once per training
    step inside ``prebuilt_H``; before every use outside) and one to turn the image's gradient into the per-table gradients."""

    def __init__(self, parent: TableSet):
        self.parent = parent
        self.parent_key = parent.key
        self.n, self.F, self.k = 1, parent.F, parent.n * parent.k
        self.vocab, self.R, self.vocab_c, self.offsets, self.dev = parent.vocab, parent.R, parent.vocab_c, parent.offsets, parent.dev
        h, pk = self.k, parent.k
        self.image = torch.zeros((self.R, h), dtype=torch.float32, device=self.dev)
        self.key = tuple(self.image.data_ptr() + 4 * parent.offsets[f] * h for f in range(self.F))
        self.ptrs = torch.tensor(self.key, dtype=torch.int64, device=self.dev)
        pack = (L.Copy2dJob * (parent.n * parent.F))()
        unpack = (L.Copy2dJob * (parent.n * parent.F))()
        for c in range(parent.n):
            for f in range(parent.F):
                i = c * parent.F + f
                V, off = parent.vocab[f], parent.offsets[f]
                # image <- table(c, f);  per-table gradient block [c][f] of the [n][R][k] buffer <- image gradient
                pack[i] = L.Copy2dJob(parent.key[i], None, 0, off * h + c * pk, pk, h, V, pk)
                unpack[i] = L.Copy2dJob(None, None, off * h + c * pk, (c * parent.R + off) * pk, h, pk, V, pk)
        self.max_elems = max(parent.vocab) * pk
        self._pack_jobs = torch.frombuffer(bytearray(bytes(pack)), dtype=torch.uint8).to(self.dev)
        self._unpack_jobs = torch.frombuffer(bytearray(bytes(unpack)), dtype=torch.uint8).to(self.dev)
        self.num_jobs = parent.n * parent.F
        if FlatTableSet._LIVE is None:
            FlatTableSet._LIVE = _weakref.WeakSet()
        FlatTableSet._LIVE.add(self)
        FlatTableSet._SERIAL += 1
        self.serial = FlatTableSet._SERIAL                   # (ids of dead objects are reused; serial numbers are not)

    @property
    def num_tables(self) -> int:
        return self.parent.n * self.parent.F

    _LIVE: "_weakref.WeakSet" = None        # every image alive (refresh_all)
    _SERIAL = 0
    _ALL_JOBS: dict = {}                    # device -> (ids, concatenated job table, count, max_elems)
    _KEEP: list = []                        # every job table ever launched (a few KB each)

    @classmethod
    def refresh_all(cls) -> None:
        """One pack launch per device for every live image (the start of a training step: prebuilt_H)."""
        live = [f for f in (cls._LIVE or ()) if f.parent_key == f.parent.key]
        by_dev = {}
        for f in live:
            by_dev.setdefault(f.dev, []).append(f)
        for dev, fs in by_dev.items():
            fs.sort(key=lambda f: f.serial)
            ids = tuple(f.serial for f in fs)
            ent = cls._ALL_JOBS.get(dev)
            if (ent is None or ent[0] != ids) and torch.cuda.is_current_stream_capturing():
                continue                                     # no uploads inside a capture: every image refreshes itself at first use
            if ent is None or ent[0] != ids:
                jobs = []
                for f in fs:                                 # absolute destinations: the images sit in different buffers
                    h = f.k
                    for c in range(f.parent.n):
                        for fe in range(f.parent.F):
                            off = f.parent.offsets[fe]
                            jobs.append(L.Copy2dJob(f.parent.key[c * f.parent.F + fe], f.image.data_ptr(), 0, off * h + c * f.parent.k,
                                                    f.parent.k, h, f.parent.vocab[fe], f.parent.k))
                arr = (L.Copy2dJob * len(jobs))(*jobs)
                ent = cls._ALL_JOBS[dev] = (ids, torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev), len(jobs),
                                            max(f.max_elems for f in fs))
                cls._KEEP.append(ent[1])                     # a captured step may still hold the address of an older table
            L.check(L.load().phc_copy2d_batched(L.ptr(ent[1]), ent[2], ent[3], None, None, L.stream_ptr(dev)),
                    "phc_copy2d_batched(pack all tables)")
            packed = getattr(_FLAT_STEP, "packed", None)
            if packed is not None:
                packed.update(f.serial for f in fs)

    def refresh(self, tables):              # (TableSet.current: the image has no tables of its own to follow)
        raise L.PhcError("FlatTableSet is derived from its parent TableSet")

    def refresh_image(self) -> None:
        packed = getattr(_FLAT_STEP, "packed", None)
        if packed is not None:
            # (serial numbers, not id(): an image that was marked at the step's start and is collected DURING the step — a dead
            # model's, freed by the cyclic collector at any moment — leaves its id() to the next object Python creates; a new
            # model's image that got it looked "already refreshed" and stayed at its initial zeros: an encoder that embeds
            # everything as 0 for one step, the loss a few percent off, once in a few hundred test runs)
            if self.serial in packed:
                return                      # already refreshed in this training step (the tables change only at its end)
            packed.add(self.serial)
        L.check(L.load().phc_copy2d_batched(L.ptr(self._pack_jobs), self.num_jobs, self.max_elems, None, L.ptr(self.image),
                                            L.stream_ptr(self.dev)), "phc_copy2d_batched(pack tables)")

    def grad_views(self, flat_grad: torch.Tensor):
        """Per-table [V_f, k] gradients of the PARENT's tables from the image's gradient ``[R, n*k]`` (one launch)."""
        p = self.parent
        real = torch.empty(p.n * p.R * p.k, dtype=torch.float32, device=flat_grad.device)
        L.check(L.load().phc_copy2d_batched(L.ptr(self._unpack_jobs), self.num_jobs, self.max_elems, L.ptr(flat_grad), L.ptr(real),
                                            L.stream_ptr(flat_grad.device)), "phc_copy2d_batched(unpack table gradients)")
        return p.grad_views(real)


_TABLESETS: "_weakref.WeakKeyDictionary" = _weakref.WeakKeyDictionary()


def tableset_for(owner, tables) -> TableSet:
    """TableSet cached per owning module (weakly: nothing is stored on the module, so
    whole-module pickling keeps working)."""
    ts = _TABLESETS.get(owner)
    if ts is None:
        ts = TableSet(tables)
        _TABLESETS[owner] = ts
        return ts
    return ts.current(tables)


def _idx64(idx: torch.Tensor) -> torch.Tensor:
    if idx.dim() == 1:
        idx = idx.unsqueeze(1)
    if idx.dtype != torch.int64:
        idx = idx.to(torch.int64)
    return idx.contiguous()


class _EncodeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, idx, ts: TableSet, *tables):
        lib = L.load()
        dev = L.require_device(idx)
        idx_ = _idx64(idx)
        N = idx_.shape[0]
        if idx_.shape[1] != ts.F:
            raise L.PhcError(f"index tensor has {idx_.shape[1]} features, encoder has {ts.F}")
        pre = _PREBUILT.encoded
        if pre is not None and pre[0] == idx_.data_ptr() and pre[1] == tuple(idx_.shape) and pre[2] == ts.ptrs.data_ptr():
            out = pre[3]                                     # evaluated by the step's operand-build launch (prebuilt_H(encode=...))
            _PREBUILT.encoded = None                         # (one consumer: a second call would alias the first one's output)
        else:
            out = torch.empty((N, ts.n * ts.k), dtype=torch.float32, device=dev)
            L.check(lib.phc_encode_fwd(L.ptr(idx_), L.ptr(ts.ptrs), ts.vocab_c, N, ts.F, ts.n, ts.k, L.ptr(out), L.stream_ptr(dev)),
                    "phc_encode_fwd")
            if L.PROF_ON:
                L.account("encode_fwd", 0.0, N * (8.0 * ts.F + 4.0 * ts.n * ts.k) + 4.0 * ts.R * ts.n * ts.k)
        L.maybe_check_indices(dev)
        ctx.ts = ts
        ctx.save_for_backward(idx_)
        ctx.defer_tables = tuple(tables) if (_DEFER.active and N > 0 and all(t.is_leaf and t.grad_fn is None for t in tables)) else None
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = L.load()
        ts: TableSet = ctx.ts
        (idx_,) = ctx.saved_tensors
        dout_ = L.f32c(dout)
        dev = dout_.device
        N = idx_.shape[0]
        if ctx.defer_tables is not None and _DEFER.active and not torch.is_grad_enabled():
            # the table gradients join the step's tail (phc_step_tail / phc_encode_bwd at the end of the backward pass)
            params = [t if ng else None for t, ng in zip(ctx.defer_tables, ctx.needs_input_grad[2:])]
            with _DEFER.lock:
                _DEFER.atom_jobs.append(dict(dout=dout_, idx=idx_, ts=ts, params=params))
            return (None, None, *([None] * ts.num_tables))
        flat = torch.empty(ts.n * ts.R * ts.k, dtype=torch.float32, device=dev)
        ws_bytes = lib.phc_encode_bwd_ws_bytes(N, ts.F, ts.vocab_c, ts.n, ts.k)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        L.check(lib.phc_encode_bwd(L.ptr(dout_), L.ptr(idx_), N, ts.F, ts.vocab_c, ts.n, ts.k, L.ptr(flat), L.ptr(ws),
                                   ws_bytes, L.stream_ptr(dev)), "phc_encode_bwd")
        if L.PROF_ON:
            L.account("table_grad", 1.0 * N * ts.F * ts.n * ts.k, N * (8.0 * ts.F + 4.0 * ts.n * ts.k))
        return (None, None, *ts.grad_views(flat))


def encode(idx: torch.Tensor, ts: TableSet, tables) -> torch.Tensor:
    """out[i, c*k+κ] = Σ_f tables[c][f][idx[i, f], κ]   ->  [N, n*k] component-major."""
    return _EncodeFn.apply(idx, ts.vector_form(tables), *[t for comp in tables for t in comp])


class _AggregateEncFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bond_idx, st: EdgeStructure, act: int, add_self: bool, ts: TableSet, dx_handoff, pending, lazy_box, dx_link, *tables):
        lib = L.load()
        ctx.dx_handoff = dx_handoff
        ctx.dx_link = dx_link
        if dx_handoff is not None:
            dx_handoff.accepted = True
        dev = L.require_device(x, bond_idx)
        # `pending`: x = act(norm(y)) (+ skip) has not been evaluated yet (PendingNorm): this launch does it on the fly for every row
        # it gathers and writes x itself once — or, outside the fused kernel's range, the stand-alone pass runs right here
        # `lazy_box` (a dict): the caller feeds this aggregate straight into the fused MLP + norm node — when the shapes allow it
        # nothing is launched here: the node's first GEMM builds the rows itself (phc_gemm_aggregate_stats) and writes `out` on the
        # way; anybody else who reads `out` first calls PendingAggregate.materialize (the launch that would have run here).
        # Hidden 128: up to NORM_INTO_AGGREGATE_MAX_WORK row elements; the wide layers (256, 180, 200: gemm_hs_agw_device.h, where the
        # gather hides under the MFMAs of the step in front) at any size the entry point takes.
        hx = x.shape[1]
        wide = hx != 128
        lazy_shape = lazy_box is not None and GEMM_AGGREGATE and act == 0 and st.num_edges > 0 and x.is_contiguous() and \
            x.dtype == torch.float32 and x.shape[0] == st.num_src == st.num_dst and x.data_ptr() % 16 == 0 and ts.k % 4 == 0 and \
            (wide or x.shape[0] * hx <= NORM_INTO_AGGREGATE_MAX_WORK) and \
            bool(lib.phc_gemm_aggregate_supported(st.num_dst, hx, hx, ts.F, ts.R))
        fuse_norm = pending is not None and not pending.done and act == 0 and ts.F <= 4 and ts.R <= 256 and x.is_contiguous() and \
            x.shape[0] == st.num_src == st.num_dst and \
            (x.shape[0] * hx <= NORM_INTO_AGGREGATE_MAX_WORK or (wide and lazy_shape and GEMM_AGGREGATE_WIDE_NORM)) and \
            bool(lib.phc_aggregate_norm_supported(st.num_dst, hx, ts.F, ts.R))
        if pending is not None and not pending.done and not fuse_norm:
            pending.materialize()
        x_ = L.f32c(x)
        bond_ = _idx64(bond_idx)
        h = x_.shape[1]
        if h != ts.n * ts.k:
            raise L.PhcError(f"x has {h} features, bond encoder produces {ts.n * ts.k}")
        if tuple(bond_.shape) != (st.num_edges, ts.F):
            raise L.PhcError(f"edge_attr indices must be [{st.num_edges}, {ts.F}], got {tuple(bond_.shape)}")
        if x_.shape[0] < st.num_src:
            raise L.PhcError(f"x has {x_.shape[0]} rows, edge structure expects {st.num_src} source rows")
        rowptr, col, perm = st.csr
        out = torch.empty((st.num_dst, h), dtype=torch.float32, device=dev)
        ctx.codes = None
        lazy = lazy_shape and x_.data_ptr() == x.data_ptr() and (pending is None or pending.done or fuse_norm)
        if lazy:
            codes = st.packed_codes(bond_, ts.vocab_c, ts.F)
            ctx.codes = codes
            lazy_box["pa"] = PendingAggregate(x_, codes, ts, st, add_self, out, pending if fuse_norm else None)
        elif fuse_norm:
            codes = st.packed_codes(bond_, ts.vocab_c, ts.F)
            ctx.codes = codes
            pending.run_fused(x_, codes, ts, rowptr, col, st.num_dst, add_self, out)
        elif ts.F <= 4 and ts.R <= 256:
            codes = st.packed_codes(bond_, ts.vocab_c, ts.F)        # 4 bytes per edge, in sorted order, built once per batch
            ctx.codes = codes
            L.check(lib.phc_aggregate_encp_fwd(L.ptr(x_), L.ptr(codes), L.ptr(ts.ptrs), ts.vocab_c, L.ptr(rowptr), L.ptr(col),
                                               st.num_dst, ts.F, ts.n, ts.k, act, 1 if add_self else 0, L.ptr(out),
                                               L.stream_ptr(dev)), "phc_aggregate_encp_fwd")
        else:
            L.check(lib.phc_aggregate_enc_fwd(L.ptr(x_), L.ptr(bond_), L.ptr(ts.ptrs), ts.vocab_c, L.ptr(rowptr), L.ptr(col),
                                              L.ptr(perm), st.num_dst, ts.F, ts.n, ts.k, act, 1 if add_self else 0, L.ptr(out),
                                              L.stream_ptr(dev)), "phc_aggregate_enc_fwd")
        if L.PROF_ON and not lazy:
            E, N = st.num_edges, st.num_dst
            L.account("aggregate_enc_fwd", 2.0 * E * h,
                      E * (4.0 * h + 4 + 4 + 8.0 * ts.F) + N * 4.0 * h * (2 if add_self else 1) + 4.0 * (N + 1))
        ctx.st, ctx.act, ctx.add_self, ctx.ts = st, act, add_self, ts
        ctx.x_rows = x_.shape[0]
        ctx.save_for_backward(x_, bond_)
        # table gradients can wait for the end of the backward pass when the tables are plain leaf parameters
        ctx.defer_tables = tuple(tables) if (_DEFER.active and st.num_edges > 0 and
                                             all(t.is_leaf and t.grad_fn is None for t in tables)) else None
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = L.load()
        st: EdgeStructure = ctx.st
        ts: TableSet = ctx.ts
        x_, bond_ = ctx.saved_tensors
        dout_ = L.f32c(dout)
        dev = dout_.device
        h = dout_.shape[1]
        need_x = ctx.needs_input_grad[0]
        need_t = any(ctx.needs_input_grad[10:])
        # the skip connection's share of dx (DxHandoff, parked by the layer's fused MLP + norm node)
        extra = ctx.dx_handoff.take() if ctx.dx_handoff is not None else None
        extras = extra if isinstance(extra, list) else ([extra] if extra is not None else [])    # DxSumHandoff parks several
        extra = extras[0] if len(extras) == 1 else None
        if extras and (not need_x or any(tuple(e.shape) != (ctx.x_rows, h) for e in extras)):
            raise L.PhcError("DxHandoff: a parked gradient needs a consumer that computes dx of the same shape")
        rowptr_t = col_t = perm_t = None
        dx = None
        if need_x:
            rowptr_t, col_t, perm_t = st.csc
            dx = torch.empty((ctx.x_rows, h), dtype=torch.float32, device=dev)
            if ctx.x_rows > st.num_src:
                dx[st.num_src:].zero_()
        dx_done = False
        link = ctx.dx_link
        if need_x and link is not None and link.ready and MLP_BWD_FUSED_AGG and ctx.act == 0 and len(extras) <= 1 and \
                ctx.x_rows == st.num_src == st.num_dst and h == 128 and getattr(st, "_ell_t", None) is not None and \
                dout_.data_ptr() % 16 == 0 and (extra is None or (extra.is_contiguous() and extra.data_ptr() % 16 == 0)) and \
                not torch.is_grad_enabled():
            # nothing is launched here: the backward of the node that produced x builds these rows inside its own launch (PendingDx)
            link.pending = PendingDx(dout_, extra, st, ctx.add_self, dx)
            dx_done, extra, extras = True, None, []
        elif need_x and extra is not None and ctx.act == 0 and ctx.x_rows == st.num_src and h % 4 == 0 and extra.data_ptr() % 16 == 0 \
                and dout_.data_ptr() % 16 == 0:
            # identity message function: dx by the row walker, the parked gradient added on the way out (no add launch)
            L.check(lib.phc_aggregate_dx_rows(L.ptr(dout_), L.ptr(rowptr_t), L.ptr(col_t), st.num_src, st.num_dst, h,
                                              1 if ctx.add_self else 0, L.ptr(extra), L.ptr(dx), L.stream_ptr(dev)),
                    "phc_aggregate_dx_rows")
            dx_done, extra, extras = True, None, []
        elif need_x and len(extras) > 1 and len(extras) <= 8 and ctx.act == 0 and ctx.x_rows == st.num_src and h % 4 == 0 \
                and dout_.data_ptr() % 16 == 0 and all(e.is_contiguous() and e.data_ptr() % 16 == 0 for e in extras):
            arr = (_C.c_void_p * len(extras))(*[e.data_ptr() for e in extras])
            L.check(lib.phc_aggregate_dx_rows_multi(L.ptr(dout_), L.ptr(rowptr_t), L.ptr(col_t), st.num_src, st.num_dst, h,
                                                    1 if ctx.add_self else 0, arr, len(extras), L.ptr(dx), L.stream_ptr(dev)),
                    "phc_aggregate_dx_rows_multi")
            dx_done, extras = True, []
        flat = ws = None
        ws_bytes = 0
        deferred = need_t and ctx.defer_tables is not None and _DEFER.active and not torch.is_grad_enabled()
        if deferred:
            params = [t if ng else None for t, ng in zip(ctx.defer_tables, ctx.needs_input_grad[10:])]
            with _DEFER.lock:
                _DEFER.table_jobs.append(dict(
                    key=(bond_.data_ptr(), st._src.data_ptr(), st._dst.data_ptr(), st.num_edges, ts.F, ts.n, ts.k, tuple(ts.vocab),
                         ctx.act), dout=dout_, x=x_, ts=ts, st=st, bond=bond_, act=ctx.act, params=params,
                    # identity message function: the per-node counted form (phc_table_grads_counted) reads these instead
                    codes=ctx.codes if (ctx.act == 0 and TABLE_GRADS_COUNTED) else None))
        if need_t and not deferred:
            flat = torch.empty(ts.n * ts.R * ts.k, dtype=torch.float32, device=dev)
            ws_bytes = lib.phc_aggregate_enc_bwd_ws_bytes(st.num_edges, ts.F, ts.vocab_c, ts.n, ts.k)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        if not dx_done or flat is not None:
            L.check(lib.phc_aggregate_enc_bwd(L.ptr(dout_), L.ptr(x_), L.ptr(bond_), L.ptr(ts.ptrs), ts.vocab_c, L.ptr(st._src),
                                              L.ptr(st._dst), st.num_edges, L.ptr(rowptr_t), L.ptr(col_t), L.ptr(perm_t), st.num_src,
                                              st.num_dst, ts.F, ts.n, ts.k, ctx.act, 1 if ctx.add_self else 0,
                                              None if dx_done else L.ptr(dx), L.ptr(flat), L.ptr(ws), ws_bytes, L.stream_ptr(dev)),
                    "phc_aggregate_enc_bwd")
        for e in extras:                                     # (shapes the row walker does not take: plain adds)
            dx += e
        if L.PROF_ON:
            E, Ns, Nd = st.num_edges, st.num_src, st.num_dst
            if need_x and not (link is not None and link.pending is not None):
                L.account("aggregate_enc_dx", 1.0 * E * h, E * (4.0 * h + 4) + Ns * 4.0 * h * (2 if ctx.add_self else 1)
                          + 4.0 * (Ns + 1))
            if need_t and not deferred:
                L.account("table_grad", 1.0 * E * ts.F * h, Nd * 4.0 * h + E * (4.0 + 8.0 * ts.F) + 4.0 * (Nd + 1))
        grads = ts.grad_views(flat) if (need_t and not deferred) else [None] * ts.num_tables
        return (dx, None, None, None, None, None, None, None, None, None, *grads)


def aggregate_sum_encoded(x: torch.Tensor, edge_index, bond_idx: torch.Tensor, ts: TableSet, tables,
                          msg_encoder: str = "identity", add_self_loops: bool = False,
                          num_dst: Optional[int] = None, dx_handoff=None, pending=None, lazy: bool = False, dx_link=None) -> torch.Tensor:
    """aggregate_sum with edge_attr[e] = Σ_f tables[c][f][bond_idx[e, f]] looked up inside the
    kernel (the [E, h] bond embedding is never written).  ``lazy``: the result goes straight into ``mlp_norm_fused`` — the
    launch may then be left to that node's first GEMM (``PendingAggregate`` on the returned tensor)."""
    st = edge_index if isinstance(edge_index, EdgeStructure) else edge_structure(edge_index, x.shape[0], num_dst)
    ts = ts.vector_form(tables)
    packed = ts.fusable_into_aggregate and ts.F <= 4 and ts.R <= 256
    if packed:
        bond_idx = _idx64(bond_idx)
    # the packed edge codes the forward will ask for ride in the launches that build the sorted views
    st.prepare(for_backward=torch.is_grad_enabled() and x.requires_grad,
               codes_for=(bond_idx, ts.vocab_c, ts.F) if packed else None)
    flat_tables = [t for comp in tables for t in comp]
    if not ts.fusable_into_aggregate:
        if pending is not None:
            pending.materialize()
        ea = _EncodeFn.apply(bond_idx, ts, *flat_tables)
        return _AggregateFn.apply(x, ea, st, _act_id(msg_encoder), bool(add_self_loops))
    box = {} if (lazy and torch.is_grad_enabled()) else None
    res = _AggregateEncFn.apply(x, bond_idx, st, _act_id(msg_encoder), bool(add_self_loops), ts, dx_handoff, pending, box, dx_link, *flat_tables)
    if box and box.get("pa") is not None:
        res._phc_pending_agg = box["pa"]
    return res


# ---------------------------------------------------------------------------
# batch-norm over component-major rows, fused with activation + residual
# ---------------------------------------------------------------------------
def _ptr_array(tensors):
    """ctypes array of device pointers (host memory, consumed at call time)."""
    return (_C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])



# ---- dropout fused behind the norm's activation (include/phc_hip.h: phc_bn_fwd / phc_rng_advance) ----
_RNG_STATE: dict = {}


def rng_state(dev) -> torch.Tensor:
    """Device int64 [2] = {seed, step} behind the engine's fused dropout masks (seeded from torch's generator at first use)."""
    st = _RNG_STATE.get(dev)
    if st is None:
        st = _RNG_STATE[dev] = torch.tensor([torch.initial_seed() & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64, device=dev)
    return st


def rng_seed(dev, seed: int) -> None:
    rng_state(dev).copy_(torch.tensor([int(seed) & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64))


def rng_advance(dev) -> None:
    """One tiny launch per training step: the next forward / backward pair draws new dropout masks (capturable)."""
    L.check(L.load().phc_rng_advance(L.ptr(rng_state(dev)), L.stream_ptr(dev)), "phc_rng_advance")


def model_uses_dropout(model) -> bool:
    return bool(model.training and (any(p > 0.0 for p in getattr(model, "dropout_mpnn", []) or [])
                                    or any(p > 0.0 for p in (getattr(getattr(model, "downstream", None), "dropout", None) or []))))


class _BatchNormActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, n_valid, running_mean, running_var, nbt, training, momentum, eps, act, n_comp, pre_stats,
                dropout, *affine):
        lib = L.load()
        dev = L.require_device(x, residual)
        x_ = L.f32c(x)
        res_ = L.f32c(residual)
        N, h = x_.shape
        weights = list(affine[:n_comp]) if affine else None
        biases = list(affine[n_comp:]) if affine else None
        w_arr = _ptr_array(weights) if weights else None
        b_arr = _ptr_array(biases) if biases else None
        rm_arr = _ptr_array(running_mean) if running_mean else None
        rv_arr = _ptr_array(running_var) if running_var else None
        nbt_arr = _ptr_array(nbt) if nbt else None
        out = torch.empty_like(x_)
        save_stat = torch.empty(6 * h, dtype=torch.float32, device=dev)   # mean|invstd|scale|shift|w|b per column
        # dropout = (p, same, site) or None: phm_dropout behind the activation, mask regenerated in the backward
        drop = (0.0, 0, 0, None)
        if dropout is not None and training and dropout[0] > 0.0:
            drop = (float(dropout[0]), (h // n_comp) if dropout[1] else 0, int(dropout[2]), rng_state(dev))
        ctx.drop = drop
        if pre_stats is not None:               # statistics already emitted by the GEMM that produced x
            L.check(lib.phc_bn_fwd_from_partials(L.ptr(x_), L.ptr(pre_stats.partial), L.ptr(pre_stats.counts), pre_stats.num,
                                                 w_arr, b_arr, rm_arr, rv_arr, nbt_arr, n_comp, L.ptr(n_valid), N, h,
                                                 float(momentum), float(eps), act, L.ptr(res_), L.ptr(out), L.ptr(save_stat),
                                                 drop[0], drop[1], drop[2], L.ptr(drop[3]), L.stream_ptr(dev)),
                    "phc_bn_fwd_from_partials")
            if L.PROF_ON:
                L.account("bn_apply", 2.0 * N * h, 4.0 * N * h * (3 if residual is not None else 2))
            ctx.save_for_backward(x_, save_stat)
            ctx.meta = (n_comp, bool(training), act, bool(affine), residual is not None, n_valid)
            if training:
                _tap(x_, save_stat, n_valid, act)
            return out
        ws_bytes = lib.phc_bn_ws_bytes(N, h)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        L.check(lib.phc_bn_fwd(L.ptr(x_), w_arr, b_arr, rm_arr, rv_arr, nbt_arr, n_comp, L.ptr(n_valid), N, h,
                               1 if training else 0, float(momentum), float(eps), act, L.ptr(res_), L.ptr(out),
                               L.ptr(save_stat), L.ptr(ws), ws_bytes, drop[0], drop[1], drop[2], L.ptr(drop[3]), L.stream_ptr(dev)),
                "phc_bn_fwd")
        if L.PROF_ON:
            if training:
                L.account("bn_stats_partial", 3.0 * N * h, 4.0 * N * h)
            L.account("bn_apply", 2.0 * N * h, 4.0 * N * h * (3 if residual is not None else 2))
        ctx.save_for_backward(x_, save_stat)
        ctx.meta = (n_comp, bool(training), act, bool(affine), residual is not None, n_valid)
        if training:
            _tap(x_, save_stat, n_valid, act)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = L.load()
        n_comp, training, act, has_affine, has_res, n_valid = ctx.meta
        x_, save_stat = ctx.saved_tensors
        dout_ = L.f32c(dout)
        dev = dout_.device
        N, h = x_.shape
        need_x = ctx.needs_input_grad[0]
        need_aff = has_affine and any(ctx.needs_input_grad[13:])
        drop = ctx.drop
        dx = torch.empty_like(x_) if need_x else None
        dwb = torch.empty(2 * h, dtype=torch.float32, device=dev) if need_aff else None
        ws = stat_acc_slot(dev, h) if N > 0 else None        # accumulator form: no merge launch between the two passes
        ws_bytes = 0
        if ws is None:
            ws_bytes = lib.phc_bn_ws_bytes(N, h)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        L.check(lib.phc_bn_bwd(L.ptr(dout_), L.ptr(x_), L.ptr(save_stat), L.ptr(n_valid), N, h, 1 if training else 0, act,
                               L.ptr(dx), L.ptr(dwb), L.ptr(ws), ws_bytes, drop[0], drop[1], drop[2], L.ptr(drop[3]), L.stream_ptr(dev)),
                "phc_bn_bwd")
        if L.PROF_ON:
            L.account("bn_bwd_partial", 6.0 * N * h, 8.0 * N * h)
            L.account("bn_bwd_apply", 8.0 * N * h, 12.0 * N * h)
        k = h // n_comp
        aff_grads = []
        if has_affine:
            if need_aff:
                aff_grads = [dwb[c * k:(c + 1) * k] for c in range(n_comp)] + \
                            [dwb[h + c * k:h + (c + 1) * k] for c in range(n_comp)]
            else:
                aff_grads = [None] * (2 * n_comp)
        dres = dout if (has_res and ctx.needs_input_grad[1]) else None
        return (dx, dres, None, None, None, None, None, None, None, None, None, None, None, *aff_grads)


# the norm backward's column sums come out of the dz GEMM's epilogue (False: separate bn_bwd_partial pass)
BN_BWD_SUMS_IN_GEMM = __import__("os").environ.get("PHC_BN_BWD_SUMS_IN_GEMM", "1") != "0"


class _BNActLinearFn(torch.autograd.Function):
    """[training batch-norm -> activation -> PHM linear] with the normalised activation never written to HBM
    (phc_bn_finalize_from_partials + phc_gemm_prebuilt_xf).  x is the norm's input — the output of a GEMM that
    emitted its statistics partials (``pre``); the linear layer's operand matrices come from the step's batched
    build and its parameter gradients go to the deferred batched launch (job with the xf_* fields)."""

    @staticmethod
    def forward(ctx, x, n_valid, pre, running_mean, running_var, nbt, momentum, eps, act, n_comp, A2, W2, b2, H2, HT2, out_stats,
                k4, *affine):
        lib = L.load()
        dev = x.device
        N, h = x.shape
        n2, kin2, kout2 = W2.shape
        weights = list(affine[:n_comp]) if affine else None
        biases = list(affine[n_comp:]) if affine else None
        save_stat = torch.empty(6 * h, dtype=torch.float32, device=dev)
        from_acc = pre.counts is None                   # the statistics are accumulator totals: the GEMM below derives scale / shift
        if not from_acc:
            L.check(lib.phc_bn_finalize_from_partials(
                L.ptr(pre.partial), L.ptr(pre.counts), pre.num, _ptr_array(weights) if weights else None,
                _ptr_array(biases) if biases else None, _ptr_array(running_mean) if running_mean else None,
                _ptr_array(running_var) if running_var else None, _ptr_array(nbt) if nbt else None, n_comp, L.ptr(n_valid), N, h,
                float(momentum), float(eps), L.ptr(save_stat), L.stream_ptr(dev)), "phc_bn_finalize_from_partials")
        y = torch.empty((N, n2 * kout2), dtype=torch.float32, device=dev)
        num = _C.c_int(0)
        sp = sc = None
        want_num = True
        if out_stats is not None:
            sp, sc, want_num = out_stats.allocate(dev, n2 * kout2, lib.phc_gemm_stats_max_partials())
        if from_acc:
            L.check(lib.phc_gemm_prebuilt_xf_acc(
                L.ptr(x), L.ptr(pre.partial), _ptr_array(weights) if weights else None, _ptr_array(biases) if biases else None,
                _ptr_array(running_mean) if running_mean else None, _ptr_array(running_var) if running_var else None,
                _ptr_array(nbt) if nbt else None, n_comp, float(momentum), float(eps), L.ptr(save_stat), act, L.ptr(n_valid),
                L.ptr(H2), L.ptr(k4[0]), L.ptr(b2), N, h, n2 * kout2, L.ptr(y), L.ptr(sp), L.ptr(sc), _C.byref(num) if want_num else None,
                L.stream_ptr(dev)), "phc_gemm_prebuilt_xf_acc")
        else:
            L.check(lib.phc_gemm_prebuilt_xf(L.ptr(x), L.ptr(save_stat), act, L.ptr(n_valid), L.ptr(H2), L.ptr(k4[0]), L.ptr(b2), N, h,
                                             n2 * kout2, L.ptr(y), L.ptr(sp), L.ptr(sc), _C.byref(num) if want_num else None,
                                             L.stream_ptr(dev)),
                    "phc_gemm_prebuilt_xf")
        if out_stats is not None:
            out_stats.num = num.value
        if L.PROF_ON:
            L.account("phm_gemm_fwd", 2.0 * N * h * n2 * kout2, 4.0 * (N * (h + n2 * kout2) + h * n2 * kout2))
        ctx.save_for_backward(x, save_stat, A2, W2)
        ctx.HT2, ctx.HT2k4, ctx.params2 = HT2, k4[1], (A2, W2, b2)
        ctx.meta = (n_comp, act, bool(affine), n_valid)
        _tap(x, save_stat, n_valid, act)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = L.load()
        x, save_stat, A2, W2 = ctx.saved_tensors
        n_comp, act, has_affine, n_valid = ctx.meta
        dev = x.device
        N, h = x.shape
        n2, kin2, kout2 = W2.shape
        dy_ = L.f32c(dy)
        # d(normalised activation) = dy . H2^T; the same launch emits the norm backward's column sums from its registers
        dz = torch.empty((N, h), dtype=torch.float32, device=dev)
        fused_sums = BN_BWD_SUMS_IN_GEMM and h % 4 == 0 and dy_.data_ptr() % 16 == 0
        sums = None
        num = _C.c_int(0)
        sums_acc = False
        if fused_sums:
            sums = stat_acc_slot(dev, h)
            sums_acc = sums is not None
            if not sums_acc:
                sums = torch.empty((lib.phc_gemm_stats_max_partials(), 2 * h), dtype=torch.float32, device=dev)
            L.check(lib.phc_gemm_prebuilt_bwsums(L.ptr(dy_), L.ptr(ctx.HT2), L.ptr(ctx.HT2k4), N, n2 * kout2, h, L.ptr(dz), L.ptr(x), L.ptr(save_stat),
                                                 act, L.ptr(n_valid), L.ptr(sums), None if sums_acc else _C.byref(num),
                                                 L.stream_ptr(dev)),
                    "phc_gemm_prebuilt_bwsums")
        else:
            L.check(lib.phc_gemm_prebuilt(L.ptr(dy_), L.ptr(ctx.HT2), L.ptr(ctx.HT2k4), None, None, N, n2 * kout2, h, L.ptr(dz), 1,
                                          L.stream_ptr(dev)), "phc_gemm_prebuilt(dx)")
        if L.PROF_ON:
            L.account("phm_gemm_dx", 2.0 * N * h * n2 * kout2, 4.0 * (N * (h + n2 * kout2) + h * n2 * kout2))
        # second layer's parameter gradients: x stays the norm's input, the kernel re-evaluates act(x*scale+shift)
        pA, pW, pb = ctx.params2
        needs = ctx.needs_input_grad
        need_A, need_W, need_b = needs[10], needs[11], needs[12] and pb is not None
        grads2 = (None, None, None)
        if need_A or need_W or need_b:
            job = dict(x=x, dy=dy_, A=A2, W=W2, need=(need_A, need_W, need_b),
                       params=(pA if need_A else None, pW if need_W else None, pb if need_b else None),
                       xf=(save_stat, n_valid, act))
            if _DEFER.active and not torch.is_grad_enabled() and all(p is None or p.is_leaf for p in (pA, pW, pb)):
                with _DEFER.lock:
                    _DEFER.jobs.append(job)
            else:                                            # not deferring: same batched entry point, one job, results returned
                job["params"] = (None, None, None)
                out = _flush_param_grads([job], return_grads=True)[0]
                grads2 = (out[0], out[1], out[2])
        # batch-norm backward on (dz, x, save_stat)
        need_x = needs[0]
        need_aff = has_affine and any(needs[17:])
        dx = torch.empty_like(x) if need_x else None
        dwb = torch.empty(2 * h, dtype=torch.float32, device=dev) if need_aff else None
        if fused_sums:
            ws = None if sums_acc else torch.empty(2 * h, dtype=torch.float32, device=dev)
            L.check(lib.phc_bn_bwd_from_partials(L.ptr(dz), L.ptr(x), L.ptr(save_stat), L.ptr(sums), 0 if sums_acc else num.value,
                                                 L.ptr(n_valid), N, h, act, L.ptr(dx), L.ptr(dwb), L.ptr(ws), 0 if sums_acc else 8 * h,
                                                 L.stream_ptr(dev)),
                    "phc_bn_bwd_from_partials")
        else:
            ws_bytes = lib.phc_bn_ws_bytes(N, h)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            L.check(lib.phc_bn_bwd(L.ptr(dz), L.ptr(x), L.ptr(save_stat), L.ptr(n_valid), N, h, 1, act, L.ptr(dx), L.ptr(dwb),
                                   L.ptr(ws), ws_bytes, 0.0, 0, 0, None, L.stream_ptr(dev)), "phc_bn_bwd")
        if L.PROF_ON:
            if not fused_sums:
                L.account("bn_bwd_partial", 6.0 * N * h, 8.0 * N * h)
            L.account("bn_bwd_apply", 8.0 * N * h, 12.0 * N * h)
        k = h // n_comp
        aff_grads = []
        if has_affine:
            aff_grads = ([dwb[c * k:(c + 1) * k] for c in range(n_comp)] + [dwb[h + c * k:h + (c + 1) * k] for c in range(n_comp)]) \
                if need_aff else [None] * (2 * n_comp)
        return (dx, None, None, None, None, None, None, None, None, None, grads2[0], grads2[1], grads2[2], None, None, None, None,
                *aff_grads)


def bn_act_linear(x: torch.Tensor, weights, biases, running_mean, running_var, num_batches_tracked, momentum: float, eps: float,
                  activation: str, n_valid, phm_rule: torch.Tensor, W: torch.Tensor, bias, out_stats=None):
    """``phm_linear(act(batch_norm(x)), phm_rule, W, bias)`` in training mode without materialising the normalised
    activation, or ``None`` when the preconditions do not hold (the caller then runs the two ops separately):
    x must be the tensor a statistics-emitting GEMM just returned (``x._phc_bn``), inside ``prebuilt_H`` with an
    entry for (phm_rule, W)."""
    pre = getattr(x, "_phc_bn", None)
    if pre is None or pre.partial is None or pre.n_valid is not n_valid or x.dim() != 2 \
            or not x.is_contiguous() or pre.partial.shape[1] != 2 * x.shape[1] or not _PREBUILT.table or x.data_ptr() % 16:
        return None
    ent = _prebuilt_for(phm_rule, W)
    if ent is None or W.shape[0] * W.shape[1] != x.shape[1] or (bias is not None and not bias.is_contiguous()) \
            or max(x.shape[1], W.shape[0] * W.shape[2]) > FUSED_MAX_WIDTH:      # (the operand transform handles one column per thread)
        return None
    n_comp = len(running_mean) if running_mean else len(weights)
    affine = (list(weights) + list(biases)) if weights else []
    y = _BNActLinearFn.apply(x, n_valid, pre, list(running_mean) if running_mean else None,
                             list(running_var) if running_var else None,
                             list(num_batches_tracked) if num_batches_tracked else None, momentum, eps, _act_id(activation),
                             n_comp, phm_rule, W, bias, ent[0], ent[1], out_stats, (ent[2], ent[3]), *affine)
    if out_stats is not None and out_stats.partial is not None:
        y._phc_bn = out_stats
    return y


def batch_norm_act(x: torch.Tensor, weights, biases, running_mean, running_var, num_batches_tracked, training: bool,
                   momentum: float, eps: float, activation: str = "identity", residual: Optional[torch.Tensor] = None,
                   n_valid: Optional[torch.Tensor] = None, dropout=None) -> torch.Tensor:
    """out = act(batch_norm(x)) (+ residual) over the rows of component-major x [N, h].

    ``weights`` / ``biases`` / running statistics are lists with one [h/n] tensor per hypercomplex
    component (the reference's separate BatchNorm1d modules, norm.py:20); ``n_valid`` is an optional
    device int32 scalar marking the padding rows of a fixed-shape batch."""
    n_comp = len(running_mean) if running_mean else len(weights)
    affine = (list(weights) + list(biases)) if weights else []
    pre = getattr(x, "_phc_bn", None)
    if pre is not None and not (training and pre.partial is not None and pre.n_valid is n_valid and x.dim() == 2
                                and pre.partial.shape[1] == 2 * x.shape[1] and x.is_contiguous()):
        pre = None
    return _BatchNormActFn.apply(x, residual, n_valid, list(running_mean) if running_mean else None,
                                 list(running_var) if running_var else None,
                                 list(num_batches_tracked) if num_batches_tracked else None, training, momentum, eps,
                                 _act_id(activation), n_comp, pre, dropout, *affine)


# ---------------------------------------------------------------------------
# PHMMLP + model-level norm of one message-passing layer as ONE autograd node
# ---------------------------------------------------------------------------
# x_next = act(norm_o(linear2(act(norm_i(linear1(s)))))) + skip      (layers.py:349-355 inside messagepassing.py:132-146, then
# models.py:208-215).  Forward: the three launches the separate nodes would issue (GEMM + statistics, GEMM with the inner norm
# as operand transform + statistics, normalise + activation + skip).  Backward: the two batch-norm backwards' element-wise
# passes ride in the dx GEMMs behind them (phc_gemm_bwd_apply) — sums of the outer norm, GEMM (outer apply, inner sums), GEMM
# (inner apply): 3 launches instead of 6, and one [N, h] write + two reads per norm less.
MLP_NORM_FUSED = __import__("os").environ.get("PHC_MLP_NORM_FUSED", "1") != "0"
# A/B inside the replayed steps: hidden 256 at 23-27 k rows zinc_b1024 2.315 -> 2.247 ms, molpcba_b1024 2.885 -> 2.807 ms; at hidden
# 128 and 12.9 k rows the two folded GEMMs cost what the GEMMs + element-wise launches they replace cost (0.7168 vs 0.7172 ms,
# three alternating runs) with 8 launches per step fewer (65 -> 57), and at 32 graphs per batch the step is launch-bound: 0.492 ->
# 0.462 ms.  On everywhere the kernels serve; PHC_MLP_NORM_FUSED_MIN_WORK sets a rows x width threshold for A/B runs.
# ... also for layers that drop (the reference's default and published settings): the mask inside the node's norm passes
MLP_NORM_DROPOUT = __import__("os").environ.get("PHC_MLP_NORM_DROPOUT", "1") != "0"
MLP_NORM_FUSED_MIN_WORK = int(__import__("os").environ.get("PHC_MLP_NORM_FUSED_MIN_WORK", "0"))


def _zero_acc(dev, h: int):
    """A zeroed statistics accumulator of width h: a slot of the step's arena, or (outside ``stat_accumulators`` / arena
    exhausted) a fresh zero tensor."""
    slot = stat_acc_slot(dev, h)
    if slot is not None:
        return slot
    lib = L.load()
    return torch.zeros(lib.phc_stat_acc_words(h), dtype=torch.int64, device=dev)


# the model-level norm's element-wise pass folded into the next layer's aggregation (phc_aggregate_norm_encp_fwd); off: the pass
# runs as its own launch where it always did
NORM_INTO_AGGREGATE = __import__("os").environ.get("PHC_NORM_INTO_AGGREGATE", "1") != "0"
# ... up to this many row elements (rows x width): the fused kernel gathers two rows per edge (y and the skip) and transforms every one
# of them; at 12.9 k x 128 that is cheaper than the launch it saves (molhiv_b512 0.606 -> 0.588 ms), at 23.6 k x 256 it is not
# (zinc_b1024 1.973 -> 1.993 ms: alternating A/B on one box)
NORM_INTO_AGGREGATE_MAX_WORK = int(__import__("os").environ.get("PHC_NORM_INTO_AGGREGATE_MAX_WORK", "3000000"))


GEMM_AGGREGATE = __import__("os").environ.get("PHC_GEMM_AGGREGATE", "1") != "0"
# the wide layers' fused launch also applies a pending norm to every row it gathers (two loads per gathered row: y and the skip);
# off: the norm's element-wise pass stays a launch of its own there and only the plain aggregation rides in the GEMM
GEMM_AGGREGATE_WIDE_NORM = __import__("os").environ.get("PHC_GEMM_AGGREGATE_WIDE_NORM", "1") != "0"


class PendingAggregate:
    """s = sum aggregation of a layer input that has NOT been launched: ``out`` (= s) is allocated, the fused MLP + norm node's first
    GEMM builds the rows itself and writes them on the way (``run_with_gemm``: phc_gemm_aggregate_stats, csrc/gemm_hs_ag_device.h).
    ``materialize`` is the launch that would have run in the aggregation's forward — identical rows.  ``pending``: the layer
    input's own un-applied norm (PendingNorm) rides along in either form."""

    def __init__(self, x_, codes, ts, st, add_self, out, pending):
        self.x_, self.codes, self.ts, self.st, self.add_self, self.out, self.pending = x_, codes, ts, st, add_self, out, pending
        self.done = False

    def _account(self):
        if L.PROF_ON:
            E, N, h, ts = self.st.num_edges, self.st.num_dst, self.out.shape[1], self.ts
            L.account("aggregate_enc_fwd", 2.0 * E * h,
                      E * (4.0 * h + 4 + 4 + 8.0 * ts.F) + N * 4.0 * h * (2 if self.add_self else 1) + 4.0 * (N + 1))

    def materialize(self):
        if self.done:
            return
        lib = L.load()
        rowptr, col, _ = self.st.csr
        ts = self.ts
        if self.pending is not None and not self.pending.done:
            self.pending.run_fused(self.x_, self.codes, ts, rowptr, col, self.st.num_dst, self.add_self, self.out)
        else:
            L.check(lib.phc_aggregate_encp_fwd(L.ptr(self.x_), L.ptr(self.codes), L.ptr(ts.ptrs), ts.vocab_c, L.ptr(rowptr), L.ptr(col),
                                               self.st.num_dst, ts.F, ts.n, ts.k, 0, 1 if self.add_self else 0, L.ptr(self.out),
                                               L.stream_ptr(self.out.device)), "phc_aggregate_encp_fwd")
        self._account()
        self.done = True

    def run_with_gemm(self, Mk4, b, y, n_valid, acc) -> bool:
        """The aggregate + ``y = s . M + b`` + the statistics of y in one launch; False = outside the fused form (nothing launched)."""
        lib = L.load()
        N, h = self.out.shape
        ts = self.ts
        if self.done or y.shape != (N, h) or not lib.phc_gemm_aggregate_supported(N, h, y.shape[1], ts.F, ts.R):
            return False
        rowptr, col, _ = self.st.csr
        d = L.GemmAgDesc()
        d.codes, d.tables, d.vocab = L.ptr(self.codes), L.ptr(ts.ptrs), _C.cast(ts.vocab_c, _C.c_void_p)
        d.rowptr, d.col, d.N, d.F, d.n, d.k, d.add_self = L.ptr(rowptr), L.ptr(col), N, ts.F, ts.n, ts.k, 1 if self.add_self else 0
        d.s_out = L.ptr(self.out)
        keep = None
        pn = self.pending if (self.pending is not None and not self.pending.done) else None
        if pn is not None:
            keep = pn._norm_args()
            w, bb, rm, rv, nbt, no = keep
            cast = lambda a: _C.cast(a, _C.c_void_p) if a is not None else None
            d.x, d.norm_acc = L.ptr(pn.y2), L.ptr(pn.acc2)
            d.weight, d.bias, d.running_mean, d.running_var, d.num_batches_tracked = cast(w), cast(bb), cast(rm), cast(rv), cast(nbt)
            d.n_comp, d.norm_act, d.momentum, d.eps = no, pn.act_o, float(pn.outer["momentum"]), float(pn.outer["eps"])
            d.skip, d.save_stat, d.x_out = L.ptr(pn.skip), L.ptr(pn.save_o), L.ptr(pn.out)
            assert self.x_.data_ptr() == pn.out.data_ptr()
        else:
            d.x = L.ptr(self.x_)
        d.Mk4, d.b, d.Nout, d.y, d.n_valid_dev, d.stat_acc = L.ptr(Mk4), L.ptr(b), y.shape[1], L.ptr(y), L.ptr(n_valid), L.ptr(acc)
        L.check(lib.phc_gemm_aggregate_stats(_C.byref(d), L.stream_ptr(self.out.device)), "phc_gemm_aggregate_stats")
        if pn is not None:
            pn.done = True
        if L.PROF_ON:
            E = self.st.num_edges
            # (flops: the GEMM's only — the sums of the aggregation are not matrix work)
            L.account("phm_gemm_ag_fwd", 2.0 * N * h * h,
                      4.0 * (2 * N * h + h * h + h) + E * (4.0 * h + 8 + 8.0 * ts.F) + N * 4.0 * h * (2 if self.add_self else 1))
        self.done = True
        return True


# the softmax aggregation's walk inside the one-launch MLP forward at hidden 128 (mlp_fused.hip AGG = 2); 0 = its own launch
SOFTMAX_INTO_MLP = __import__("os").environ.get("PHC_SOFTMAX_INTO_MLP", "1") != "0"


class PendingSoftmaxAggregate(PendingAggregate):
    """The SOFTMAX aggregation of a layer input that has NOT been launched (messagepassing.py:211-214, 297-300): ``agg`` (the plain
    aggregate), ``lse`` and ``out`` (= agg, + x with the layer's self loop) are allocated — the backward walk reads the first two —,
    the fused MLP + norm node's one-launch forward walks the rows itself and writes all three on the way.  ``materialize`` is the launch
    that would have run in the aggregation's forward: identical rows."""
    softmax = True

    def __init__(self, x_, codes, ts, st, add_self, out, pending, beta, agg, lse, act):
        super().__init__(x_, codes, ts, st, add_self, out, pending)
        self.beta, self.agg, self.lse, self.act = beta, agg, lse, act

    def _account(self):
        if L.PROF_ON:
            E, N, h = self.st.num_edges, self.st.num_dst, self.out.shape[1]
            L.account("aggregate_softmax_fwd", 8.0 * E * h, E * (4.0 * h + 12) + N * 8.0 * h)

    def materialize(self):
        if self.done:
            return
        lib = L.load()
        rowptr, col, _ = self.st.csr
        ts, st = self.ts, self.st
        out_self = self.out if self.add_self else None
        dev = self.out.device
        pn = self.pending if (self.pending is not None and not self.pending.done) else None
        if pn is not None:
            w, bb, rm, rv, nbt, no = pn._norm_args()
            L.check(lib.phc_aggregate_softmax_norm_enc_fwd(
                L.ptr(pn.y2), L.ptr(pn.acc2), w, bb, rm, rv, nbt, no, float(pn.outer["momentum"]), float(pn.outer["eps"]), pn.act_o,
                L.ptr(pn.n_valid), L.ptr(pn.skip), L.ptr(pn.save_o), L.ptr(pn.out), L.ptr(self.codes), L.ptr(ts.ptrs), ts.vocab_c, ts.F, ts.n,
                ts.k, L.ptr(rowptr), L.ptr(col), L.ptr(self.beta), st.num_dst, L.ptr(self.agg), L.ptr(out_self), L.ptr(self.lse),
                L.stream_ptr(dev)), "phc_aggregate_softmax_norm_enc_fwd")
            pn.done = True
        else:
            L.check(lib.phc_aggregate_softmax_enc_fwd(L.ptr(self.x_), L.ptr(self.codes), L.ptr(ts.ptrs), ts.vocab_c, ts.F, ts.n, ts.k,
                                                      L.ptr(rowptr), L.ptr(col), L.ptr(self.beta), st.num_dst, self.act, L.ptr(self.agg),
                                                      L.ptr(out_self), L.ptr(self.lse), L.stream_ptr(dev)), "phc_aggregate_softmax_enc_fwd")
        self._account()
        self.done = True

    def run_with_gemm(self, Mk4, b, y, n_valid, acc) -> bool:
        return False                                          # (the aggregation-inside-the-GEMM launches are the sum's)


def materialize_pending_aggregate(t: torch.Tensor) -> torch.Tensor:
    """Launch the parked aggregation of ``t`` (if it has one) before a consumer other than the fused MLP + norm node reads it."""
    pa = getattr(t, "_phc_pending_agg", None)
    if pa is not None:
        pa.materialize()
        t._phc_pending_agg = None
    return t


class PendingNorm:
    """x = act(norm(y2)) (+ skip) that has NOT been evaluated yet: the fused MLP + norm node allocated ``out`` (= x) and ``save_stat``
    and stopped in front of its last launch.  The consumer is the next layer's encoded sum aggregation (``run_fused``: one launch
    does both and writes x on the way); anybody else calls ``materialize`` first — the stand-alone pass, identical results."""

    def __init__(self, y2, acc2, outer, aff_o, act_o, skip, n_valid, out, save_o, drop=(0.0, 0, 0, None)):
        self.y2, self.acc2, self.outer, self.aff_o, self.act_o, self.skip, self.n_valid = y2, acc2, outer, aff_o, act_o, skip, n_valid
        self.out, self.save_o = out, save_o
        self.drop = drop                                      # (p, same_k, site, rng state): phm_dropout behind the activation
        self.done = False

    def _norm_args(self):
        outer, aff_o = self.outer, self.aff_o
        no = outer["n_comp"]
        return (_ptr_array(aff_o[:no]) if aff_o else None, _ptr_array(aff_o[no:]) if aff_o else None, _ptr_array(outer["running_mean"]),
                _ptr_array(outer["running_var"]), _ptr_array(outer["nbt"]), no)

    def materialize(self):
        if self.done:
            return
        lib = L.load()
        N, h = self.y2.shape
        w, b, rm, rv, nbt, no = self._norm_args()
        L.check(lib.phc_bn_fwd_from_partials(
            L.ptr(self.y2), L.ptr(self.acc2), None, 0, w, b, rm, rv, nbt, no, L.ptr(self.n_valid), N, h,
            float(self.outer["momentum"]), float(self.outer["eps"]), self.act_o, L.ptr(self.skip), L.ptr(self.out), L.ptr(self.save_o),
            self.drop[0], self.drop[1], self.drop[2], L.ptr(self.drop[3]), L.stream_ptr(self.y2.device)), "phc_bn_fwd_from_partials")
        if L.PROF_ON:
            L.account("bn_apply", 2.0 * N * h, 4.0 * N * h * (3 if self.skip is not None else 2))
        self.done = True

    def run_fused(self, x_, codes, ts, rowptr, col, n_dst, add_self, agg_out):
        lib = L.load()
        assert x_.data_ptr() == self.out.data_ptr()
        w, b, rm, rv, nbt, no = self._norm_args()
        L.check(lib.phc_aggregate_norm_encp_fwd(
            L.ptr(self.y2), L.ptr(self.acc2), w, b, rm, rv, nbt, no, float(self.outer["momentum"]), float(self.outer["eps"]), self.act_o,
            L.ptr(self.n_valid), L.ptr(self.skip), L.ptr(self.save_o), L.ptr(self.out), L.ptr(codes), L.ptr(ts.ptrs), ts.vocab_c,
            L.ptr(rowptr), L.ptr(col), n_dst, ts.F, ts.n, ts.k, 1 if add_self else 0, L.ptr(agg_out), L.stream_ptr(self.y2.device)),
            "phc_aggregate_norm_encp_fwd")
        self.done = True


def materialize_pending(t: torch.Tensor) -> torch.Tensor:
    """Run the parked norm pass of ``t`` (if it has one) before a consumer that is not the fused aggregation reads it."""
    pend = getattr(t, "_phc_pending", None)
    if pend is not None:
        pend.materialize()
        t._phc_pending = None
    return t


# Test hook: a list here collects, in forward order, (norm input, save_stat, n_valid, activation id) of every act(norm(.)) site of
# the fused nodes (the two sites of a PHMMLP + norm node, the two of the fused head).  tests/test_gpu_fullwidth.py rebuilds the
# ReLU gates the backward kernels take (sign of fma((z - mean) * invstd, w, b)) from them and imposes them on the float64 oracle.
# The list keeps the tensors alive — under hipGraph capture that also keeps their pool blocks from being reused.
RELU_TAP: Optional[list] = None


def _tap(z, save_stat, n_valid, act):
    if RELU_TAP is not None:
        RELU_TAP.append((z, save_stat, n_valid, act))


# the MLP + norm node's backward as ONE launch at hidden 128 (csrc/mlp_fused.hip, round 6); 0 = the three launches
MLP_BWD_FUSED = __import__("os").environ.get("PHC_MLP_BWD_FUSED", "1") != "0"
# ... and its forward behind the aggregation (both GEMMs + the inner norm); 0 = the two launches (the first carrying the aggregation)
MLP_FWD_FUSED = __import__("os").environ.get("PHC_MLP_FWD_FUSED", "1") != "0"
# ... with the layer's aggregation (and the un-applied norm of the layer in front) building the rows inside that launch; 0 = its own launch
MLP_FWD_FUSED_AGG = __import__("os").environ.get("PHC_MLP_FWD_FUSED_AGG", "1") != "0"
_GRID_BARRIERS: dict = {}


def _grid_barrier_words(dev) -> torch.Tensor:
    """{arrivals, generation} of the in-launch grid barriers, per device and stream: zero before the first use, owned by the kernels."""
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    w = _GRID_BARRIERS.get(key)
    if w is None:
        w = _GRID_BARRIERS[key] = torch.zeros(L.load().phc_mlp_norm_bwd_fused_barrier_words(), dtype=torch.int32, device=dev)
    return w


_SUPPORTED_CACHE: dict = {}


def _supported(name: str, *args) -> bool:
    """A library capability query, asked once per argument tuple (pure functions of the shape and the device)."""
    key = (name, torch.cuda.current_device()) + args
    r = _SUPPORTED_CACHE.get(key)
    if r is None:
        r = _SUPPORTED_CACHE[key] = bool(getattr(L.load(), name)(*args))
    return r


class _MLPNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, s, skip, n_valid, meta, A1, W1, b1, A2, W2, b2, *affine):
        lib = L.load()
        dev = s.device
        N, h = s.shape
        pre1, pre2, inner, outer = meta["pre1"], meta["pre2"], meta["inner"], meta["outer"]
        act_i, act_o = meta["act_inner"], meta["act_outer"]
        ni, no = inner["n_comp"], outer["n_comp"]
        aff_i = affine[:2 * ni] if inner["affine"] else ()
        aff_o = affine[len(aff_i):] if outer["affine"] else ()
        sp = L.stream_ptr(dev)
        z1 = torch.empty((N, h), dtype=torch.float32, device=dev)
        y2 = torch.empty((N, h), dtype=torch.float32, device=dev)
        out = torch.empty((N, h), dtype=torch.float32, device=dev)
        save_i = torch.empty(6 * h, dtype=torch.float32, device=dev)
        save_o = torch.empty(6 * h, dtype=torch.float32, device=dev)
        acc1, acc2 = _zero_acc(dev, h), _zero_acc(dev, h)
        pa = meta.pop("pending_agg", None)
        if MLP_FWD_FUSED and h == 128 and pre1[2] is not None and pre2[2] is not None and s.data_ptr() % 16 == 0 and \
                all(t is None or t.data_ptr() % 16 == 0 for t in (b1, b2)) and _supported("phc_mlp_norm_fwd_fused_supported", N, h):
            # both GEMMs and the inner norm in ONE launch (csrc/mlp_fused.hip): z1's rows stay in registers across the one grid-wide
            # dependency; the aggregation runs as its own (row-walker) launch in front
            fd = L.MlpNormFwdDesc()
            keep_o = None
            if pa is not None and not pa.done and MLP_FWD_FUSED_AGG and tuple(pa.out.shape) == (N, h) and \
                    getattr(pa.st, "_ell", None) is not None and len(pa.ts.key) <= 16 and \
                    _supported("phc_mlp_norm_fwd_fused_agg_supported", N, h, pa.ts.n, pa.ts.F, pa.ts.R):
                # ... and the layer's aggregation builds the rows of s inside it (with the un-applied norm of the layer in front)
                ts_, (rowptr_, col_, _) = pa.ts, pa.st.csr
                fd.codes, fd.tables, fd.vocab = L.ptr(pa.codes), L.ptr(ts_.ptrs), _C.cast(ts_.vocab_c, _C.c_void_p)
                fd.rowptr, fd.col, fd.F, fd.n, fd.add_self, fd.s_out = L.ptr(rowptr_), L.ptr(col_), ts_.F, ts_.n, 1 if pa.add_self else 0, L.ptr(pa.out)
                fd.ell = L.ptr(pa.st._ell)
                if getattr(pa, "softmax", False):             # the softmax walk instead of the sum (PendingSoftmaxAggregate)
                    fd.softmax_beta, fd.agg_out, fd.lse = L.ptr(pa.beta), L.ptr(pa.agg), L.ptr(pa.lse)
                tabs_host = (_C.c_void_p * len(ts_.key))(*ts_.key)          # the table pointers by value (no pointer hop in the kernel)
                fd.tables_host = _C.cast(tabs_host, _C.c_void_p)
                pn = pa.pending if (pa.pending is not None and not pa.pending.done) else None
                if pn is not None:
                    keep_o = pn._norm_args()
                    w_o, b_o, rm_o, rv_o, nbt_o, no_ = keep_o
                    cast_o = lambda arr: _C.cast(arr, _C.c_void_p) if arr is not None else None
                    fd.x, fd.norm_acc = L.ptr(pn.y2), L.ptr(pn.acc2)
                    fd.weight_o, fd.bias_o, fd.running_mean_o, fd.running_var_o, fd.num_batches_tracked_o = \
                        cast_o(w_o), cast_o(b_o), cast_o(rm_o), cast_o(rv_o), cast_o(nbt_o)
                    fd.n_comp_o, fd.norm_act, fd.momentum_o, fd.eps_o = no_, pn.act_o, float(pn.outer["momentum"]), float(pn.outer["eps"])
                    fd.skip, fd.save_stat_o, fd.x_out = L.ptr(pn.skip), L.ptr(pn.save_o), L.ptr(pn.out)
                    assert pa.x_.data_ptr() == pn.out.data_ptr()
                else:
                    fd.x = L.ptr(pa.x_)
                    pn = False
            else:
                pn = None
                if pa is not None:
                    pa.materialize()
            fd.s, fd.H1k4, fd.b1, fd.H2k4, fd.b2 = L.ptr(s), L.ptr(pre1[2]), L.ptr(b1), L.ptr(pre2[2]), L.ptr(b2)
            fd.z1, fd.y2, fd.acc1, fd.acc2 = L.ptr(z1), L.ptr(y2), L.ptr(acc1), L.ptr(acc2)
            keep = (_ptr_array(aff_i[:ni]) if aff_i else None, _ptr_array(aff_i[ni:]) if aff_i else None,
                    _ptr_array(inner["running_mean"]), _ptr_array(inner["running_var"]), _ptr_array(inner["nbt"]))
            cast = lambda arr: _C.cast(arr, _C.c_void_p) if arr is not None else None
            fd.weight, fd.bias, fd.running_mean, fd.running_var, fd.num_batches_tracked = (cast(k_) for k_ in keep)
            fd.n_comp, fd.momentum, fd.eps, fd.save_stat, fd.act = ni, float(inner["momentum"]), float(inner["eps"]), L.ptr(save_i), act_i
            fd.n_valid_dev, fd.N, fd.h, fd.barrier = L.ptr(n_valid), N, h, L.ptr(_grid_barrier_words(dev))
            L.check(lib.phc_mlp_norm_fwd_fused(_C.byref(fd), sp), "phc_mlp_norm_fwd_fused")
            if pn is not None:                               # the aggregation (and its pending norm) ran inside the launch
                if pn is not False:
                    pn.done = True
                pa.done = True
                if L.PROF_ON:
                    E_ = pa.st.num_edges
                    L.account("phm_gemm_ag_fwd", 4.0 * N * h * h,
                              4.0 * (3 * N * h + 2 * h * h + 2 * h) + E_ * (4.0 * h + 8 + 8.0 * pa.ts.F) + N * 4.0 * h * (2 if pa.add_self else 1))
                fused_agg = "mlp+agg"
            else:
                fused_agg = "mlp"                            # (accounting: ONE launch of the family with both GEMMs' work)
            return _MLPNormFn._finish_forward(ctx, s, skip, n_valid, meta, z1, y2, out, save_i, save_o, acc2, outer, aff_i, aff_o, act_i, act_o,
                                              A1, W1, b1, A2, W2, b2, fused_agg, N, h, dev, no)
        fused_agg = pa is not None and (b1 is None or b1.data_ptr() % 16 == 0) and pa.run_with_gemm(pre1[2], b1, z1, n_valid, acc1)
        if not fused_agg:
            if pa is not None:
                pa.materialize()
            L.check(lib.phc_gemm_prebuilt_stats(L.ptr(s), L.ptr(pre1[0]), L.ptr(pre1[2]), L.ptr(b1), N, h, h, L.ptr(z1), L.ptr(n_valid),
                                                L.ptr(acc1), None, None, sp), "phc_gemm_prebuilt_stats")
        L.check(lib.phc_gemm_prebuilt_xf_acc(
            L.ptr(z1), L.ptr(acc1), _ptr_array(aff_i[:ni]) if aff_i else None, _ptr_array(aff_i[ni:]) if aff_i else None,
            _ptr_array(inner["running_mean"]), _ptr_array(inner["running_var"]), _ptr_array(inner["nbt"]), ni,
            float(inner["momentum"]), float(inner["eps"]), L.ptr(save_i), act_i, L.ptr(n_valid), L.ptr(pre2[0]), L.ptr(pre2[2]),
            L.ptr(b2), N, h, h, L.ptr(y2), L.ptr(acc2), None, None, sp), "phc_gemm_prebuilt_xf_acc")
        return _MLPNormFn._finish_forward(ctx, s, skip, n_valid, meta, z1, y2, out, save_i, save_o, acc2, outer, aff_i, aff_o, act_i, act_o,
                                          A1, W1, b1, A2, W2, b2, fused_agg, N, h, dev, no)

    @staticmethod
    def _finish_forward(ctx, s, skip, n_valid, meta, z1, y2, out, save_i, save_o, acc2, outer, aff_i, aff_o, act_i, act_o,
                        A1, W1, b1, A2, W2, b2, fused_agg, N, h, dev, no):
        _tap(z1, save_i, n_valid, act_i)
        _tap(y2, save_o, n_valid, act_o)
        # dropout behind the outer norm's activation (models.py:211-213): the mask rides in the norm's apply pass here and in the norm
        # backward folded into the dx GEMM (HS_DR) — never deferred into another launch (those forms do not draw masks)
        drop = (0.0, 0, 0, None)
        dp = meta.get("dropout")
        if dp is not None and dp[0] > 0.0:
            drop = (float(dp[0]), (h // no) if dp[1] else 0, int(dp[2]), rng_state(dev))
        ctx.drop = drop
        pend = PendingNorm(y2, acc2, outer, tuple(aff_o), act_o, skip, n_valid, out, save_o, drop)
        if meta.get("defer_apply") and drop[0] == 0.0:
            meta["pending"] = pend                           # the caller hands it to the next layer's aggregation (or materialises it)
        else:
            pend.materialize()
        if L.PROF_ON:
            if fused_agg == "mlp":
                L.account("phm_gemm_fwd", 4.0 * N * h * h, 4.0 * (3 * N * h + 2 * h * h + 2 * h))
            elif fused_agg == "mlp+agg":
                pass                                         # (booked where it was launched)
            else:
                if not fused_agg:
                    L.account("phm_gemm_fwd", 2.0 * N * h * h, 4.0 * (2 * N * h + h * h + h))
                L.account("phm_gemm_fwd", 2.0 * N * h * h, 4.0 * (2 * N * h + h * h + h))
        # the aggregation that consumes `out` may leave its dx to this node's backward (DxLink: the caller attaches it to `out`)
        link = meta.get("dx_link")
        if link is not None:
            link.ready = bool(MLP_BWD_FUSED and MLP_BWD_FUSED_AGG and drop[0] == 0.0 and h == 128 and meta["pre2"][3] is not None and
                              meta["pre1"][3] is not None and _supported("phc_mlp_norm_bwd_fused_agg_supported", N, h))
        ctx.save_for_backward(s, z1, y2, save_i, save_o, A1, W1, A2, W2)
        ctx.meta, ctx.n_valid = meta, n_valid
        ctx.params = ((A1, W1, b1), (A2, W2, b2))
        ctx.has_skip = skip is not None
        ctx.n_aff = (len(aff_i), len(aff_o))
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = L.load()
        s, z1, y2, save_i, save_o, A1, W1, A2, W2 = ctx.saved_tensors
        meta, n_valid = ctx.meta, ctx.n_valid
        pre1, pre2, inner, outer = meta["pre1"], meta["pre2"], meta["inner"], meta["outer"]
        act_i, act_o = meta["act_inner"], meta["act_outer"]
        dev = s.device
        N, h = s.shape
        sp = L.stream_ptr(dev)
        dout_ = L.f32c(dout)
        needs = ctx.needs_input_grad
        # the gradient may be the un-launched dx of the aggregation above (PendingDx): built inside the one-launch form below, or by
        # the launch of its own right here
        link = meta.get("dx_link")
        pd = link.take() if link is not None else None
        if pd is not None and (dout_.data_ptr() != pd.dx.data_ptr() or tuple(dout_.shape) != tuple(pd.dx.shape)):
            raise L.PhcError("DxLink: the parked aggregation gradient is not the gradient this node received")
        acc_o, acc_i = _zero_acc(dev, h), _zero_acc(dev, h)
        dy2 = torch.empty((N, h), dtype=torch.float32, device=dev)       # d loss / d y2 (the outer norm's dx)
        da = torch.empty((N, h), dtype=torch.float32, device=dev)        # d loss / d act(norm_i(z1))
        dz1 = torch.empty((N, h), dtype=torch.float32, device=dev)       # d loss / d z1 (the inner norm's dx)
        ds = torch.empty((N, h), dtype=torch.float32, device=dev)
        dwb_o = torch.empty(2 * h, dtype=torch.float32, device=dev)
        dwb_i = torch.empty(2 * h, dtype=torch.float32, device=dev)
        drop = ctx.drop
        if MLP_BWD_FUSED and drop[0] == 0.0 and pre2[3] is not None and pre1[3] is not None and dout_.data_ptr() % 16 == 0 and \
                _supported("phc_mlp_norm_bwd_fused_supported", N, h):
            # the whole chain in ONE launch (csrc/mlp_fused.hip): rows stay in LDS, the two grid-wide dependencies are barriers inside it
            fd = L.MlpNormBwdDesc()
            fd.dout, fd.y2, fd.z1, fd.save_o, fd.save_i = L.ptr(dout_), L.ptr(y2), L.ptr(z1), L.ptr(save_o), L.ptr(save_i)
            fd.H2Tk4, fd.H1Tk4 = L.ptr(pre2[3]), L.ptr(pre1[3])
            fd.dy2, fd.dz1, fd.ds, fd.dwb_o, fd.dwb_i = L.ptr(dy2), L.ptr(dz1), L.ptr(ds), L.ptr(dwb_o), L.ptr(dwb_i)
            fd.acc_o, fd.acc_i, fd.n_valid_dev = L.ptr(acc_o), L.ptr(acc_i), L.ptr(n_valid)
            fd.N, fd.h, fd.act_o, fd.act_i = N, h, act_o, act_i
            fd.barrier = L.ptr(_grid_barrier_words(dev))
            if pd is not None and _supported("phc_mlp_norm_bwd_fused_agg_supported", N, h):
                rowptr_t_, col_t_, _ = pd.st.csc
                fd.agg_src, fd.agg_residual, fd.dx_out = L.ptr(pd.gsrc), L.ptr(pd.residual), L.ptr(pd.dx)
                fd.agg_rowptr_t, fd.agg_col_t, fd.agg_ell_t = L.ptr(rowptr_t_), L.ptr(col_t_), L.ptr(pd.st._ell_t)
                fd.agg_add_self = 1 if pd.add_self else 0
                fd.dout = None
                pd_fused, pd = pd, None
            elif pd is not None:
                pd.materialize()
                pd = None
            L.check(lib.phc_mlp_norm_bwd_fused(_C.byref(fd), sp), "phc_mlp_norm_bwd_fused")
            if L.PROF_ON:                                    # (both dx GEMMs of the node: booked as one launch of the family)
                L.account("phm_gemm_dx", 4.0 * N * h * h, 4.0 * (6 * N * h + 2 * h * h))
            return _MLPNormFn._finish_backward(ctx, dout, dout_, s, z1, y2, save_i, dy2, dz1, ds, dwb_o, dwb_i, N, h)
        if pd is not None:
            pd.materialize()
        if drop[0] > 0.0:
            L.check(lib.phc_bn_bwd_partial_acc_drop(L.ptr(dout_), L.ptr(y2), L.ptr(save_o), L.ptr(n_valid), N, h, act_o, L.ptr(acc_o),
                                                    drop[0], drop[1], drop[2], L.ptr(drop[3]), sp), "phc_bn_bwd_partial_acc_drop")
        else:
            L.check(lib.phc_bn_bwd_partial_acc(L.ptr(dout_), L.ptr(y2), L.ptr(save_o), L.ptr(n_valid), N, h, act_o, L.ptr(acc_o), sp),
                    "phc_bn_bwd_partial_acc")
        d = L.GemmBaDesc()
        d.dout, d.bn_x, d.bn_stat, d.sums_acc, d.n_valid_dev = L.ptr(dout_), L.ptr(y2), L.ptr(save_o), L.ptr(acc_o), L.ptr(n_valid)
        d.dz, d.dwb, d.Mk4, d.y = L.ptr(dy2), L.ptr(dwb_o), L.ptr(pre2[3]), L.ptr(da)
        d.N, d.K, d.Nout, d.act = N, h, h, act_o
        d.bw_act, d.bw_x, d.bw_stat, d.bw_sums_acc = act_i, L.ptr(z1), L.ptr(save_i), L.ptr(acc_i)
        d.drop_p, d.drop_same_k, d.drop_site, d.rng_state = drop[0], drop[1], drop[2], L.ptr(drop[3])
        L.check(lib.phc_gemm_bwd_apply(_C.byref(d), sp), "phc_gemm_bwd_apply(outer)")
        d2 = L.GemmBaDesc()
        d2.dout, d2.bn_x, d2.bn_stat, d2.sums_acc, d2.n_valid_dev = L.ptr(da), L.ptr(z1), L.ptr(save_i), L.ptr(acc_i), L.ptr(n_valid)
        d2.dz, d2.dwb, d2.Mk4, d2.y = L.ptr(dz1), L.ptr(dwb_i), L.ptr(pre1[3]), L.ptr(ds)
        d2.N, d2.K, d2.Nout, d2.act = N, h, h, act_i
        d2.bw_act, d2.bw_x, d2.bw_stat, d2.bw_sums_acc = 0, None, None, None
        L.check(lib.phc_gemm_bwd_apply(_C.byref(d2), sp), "phc_gemm_bwd_apply(inner)")
        if L.PROF_ON:
            L.account("bn_bwd_partial", 6.0 * N * h, 8.0 * N * h)
            L.account("phm_gemm_dx", 2.0 * N * h * h, 4.0 * (4 * N * h + h * h))
            L.account("phm_gemm_dx", 2.0 * N * h * h, 4.0 * (4 * N * h + h * h))
        return _MLPNormFn._finish_backward(ctx, dout, dout_, s, z1, y2, save_i, dy2, dz1, ds, dwb_o, dwb_i, N, h)

    @staticmethod
    def _finish_backward(ctx, dout, dout_, s, z1, y2, save_i, dy2, dz1, ds, dwb_o, dwb_i, N, h):
        meta, n_valid = ctx.meta, ctx.n_valid
        inner, outer = meta["inner"], meta["outer"]
        act_i = meta["act_inner"]
        needs = ctx.needs_input_grad
        # parameter gradients: (z1 through the inner norm, dy2) for linear2, (s, dz1) for linear1
        (pA1, pW1, pb1), (pA2, pW2, pb2) = ctx.params
        grads = [None] * 6
        jobs = []
        for li, (x_in, dy_in, A, W, b, xf, base) in enumerate(((s, dz1, pA1, pW1, pb1, None, 4), (z1, dy2, pA2, pW2, pb2, (save_i, n_valid, act_i), 7))):
            nA, nW, nb = needs[base], needs[base + 1], needs[base + 2] and b is not None
            if not (nA or nW or nb):
                continue
            job = dict(x=x_in, dy=dy_in, A=A, W=W, need=(nA, nW, nb), params=(A if nA else None, W if nW else None, b if nb else None))
            if xf is not None:
                job["xf"] = xf
            jobs.append((li, job))
        if jobs:
            deferred = _DEFER.active and not torch.is_grad_enabled() and all(p is None or p.is_leaf for _, jb in jobs for p in jb["params"])
            if deferred:
                with _DEFER.lock:
                    _DEFER.jobs.extend(jb for _, jb in jobs)
            else:
                for _, jb in jobs:
                    jb["params"] = (None, None, None)
                outs = _flush_param_grads([jb for _, jb in jobs], return_grads=True)
                for (li, _), o in zip(jobs, outs):
                    grads[3 * li:3 * li + 3] = list(o)
        aff = []
        for dwb, nm, cnt in ((dwb_i, inner, ctx.n_aff[0]), (dwb_o, outer, ctx.n_aff[1])):
            if cnt:
                n, k = nm["n_comp"], h // nm["n_comp"]
                aff.extend([dwb[c * k:(c + 1) * k] for c in range(n)])
                aff.extend([dwb[h + c * k:h + (c + 1) * k] for c in range(n)])
        dskip = dout if (ctx.has_skip and needs[1]) else None
        hd = meta.get("skip_handoff")
        if hd is not None and hd.accepted and dskip is not None:
            hd.park(dout_)                                    # the skip's share of dx: added inside the aggregation's dx kernel
            dskip = None
        return (ds if needs[0] else None, dskip, None, None, *grads, *aff)


def _norm_meta(nm):
    """What the fused kernels need of a NaivePHMNorm (training mode, running statistics, fixed momentum), or None."""
    mods = list(getattr(nm, "bn", []))
    if not mods or not getattr(nm, "track_running_stats", False) or nm.momentum is None:
        return None
    return dict(n_comp=len(mods), momentum=nm.momentum, eps=nm.eps, affine=bool(nm.affine),
                running_mean=[m.running_mean for m in mods], running_var=[m.running_var for m in mods],
                nbt=[m.num_batches_tracked for m in mods],
                weights=[m.weight for m in mods] if nm.affine else [], biases=[m.bias for m in mods] if nm.affine else [])


def mlp_norm_fused(s, skip, n_valid, mlp, outer_norm, activation: str, skip_handoff=None, defer_apply: bool = False, dropout=None):
    """``_mlp_norm_fused`` (below); an aggregation parked on ``s`` (PendingAggregate) is handed to the node's first GEMM, or launched
    here when the node does not apply.  ``dropout`` = (p, same, site): phm_dropout behind the outer norm's activation (round 5)."""
    pa = getattr(s, "_phc_pending_agg", None)
    if pa is not None:
        s._phc_pending_agg = None
    out = _mlp_norm_fused(s, skip, n_valid, mlp, outer_norm, activation, skip_handoff, defer_apply, pa, dropout)
    if out is None and pa is not None:
        pa.materialize()
    return out


def _mlp_norm_fused(s, skip, n_valid, mlp, outer_norm, activation: str, skip_handoff=None, defer_apply: bool = False, pending_agg=None,
                    dropout=None):
    """``act(outer_norm(mlp(s))) + skip`` for a PHMMLP with an inner naive batch-norm, as one autograd node (training mode),
    or ``None`` when the preconditions do not hold: both linears square of width 128 or 256 with prebuilt k4 operands (inside
    ``prebuilt_H``), both norms naive batch-norms with running statistics, fusable activations, 16-byte aligned dense rows."""
    if not (MLP_NORM_FUSED and _PREBUILT.table and s.is_cuda and s.dim() == 2 and s.dtype == torch.float32 and s.is_contiguous()
            and s.data_ptr() % 16 == 0 and getattr(mlp, "norm_flag", False) and torch.is_grad_enabled()):
        return None
    l1, l2 = mlp.linear1, mlp.linear2
    N, h = s.shape
    lib = L.load()
    if N < 1 or N * h < MLP_NORM_FUSED_MIN_WORK or tuple(l1.W.shape) != tuple(l2.W.shape) or l1.W.shape[0] * l1.W.shape[1] != h \
            or l1.W.shape[0] * l1.W.shape[2] != h or not lib.phc_gemm_bwd_apply_supported(h, h):
        return None
    try:
        act_i, act_o = L.ACT_IDS[mlp.activation_str.lower()], L.ACT_IDS[activation.lower()]
    except KeyError:
        return None
    inner = _norm_meta(getattr(mlp.norm, "bn", None))
    outer = _norm_meta(getattr(outer_norm, "bn", None))
    pre1, pre2 = _prebuilt_for(l1.phm_rule, l1.W), _prebuilt_for(l2.phm_rule, l2.W)
    if inner is None or outer is None or pre1 is None or pre2 is None or pre1[2] is None or pre2[2] is None:
        return None
    if skip is not None and not (skip.is_contiguous() and skip.shape == s.shape and skip.data_ptr() % 16 == 0):
        return None
    if any(b is not None and not b.is_contiguous() for b in (l1.b, l2.b)):
        return None
    dropping = dropout is not None and dropout[0] > 0.0
    if dropping and not (h in (128, 256, 180, 200) and h % 4 == 0 and (not dropout[1] or (h // outer["n_comp"]) % 4 == 0)):
        return None                                          # (the mask's column groups; the HS_DR instantiations)
    link = DxLink() if MLP_BWD_FUSED_AGG else None
    meta = dict(pre1=pre1, pre2=pre2, inner=inner, outer=outer, act_inner=act_i, act_outer=act_o, skip_handoff=skip_handoff,
                defer_apply=bool(defer_apply and NORM_INTO_AGGREGATE and not dropping), pending_agg=pending_agg,
                dropout=dropout if dropping else None, dx_link=link)
    affine = inner["weights"] + inner["biases"] + outer["weights"] + outer["biases"]
    out = _MLPNormFn.apply(s, skip, n_valid, meta, l1.phm_rule, l1.W, l1.b, l2.phm_rule, l2.W, l2.b, *affine)
    if link is not None and link.ready:
        out._phc_dx_link = link                              # (the model loop hands it to out's consumer when there is exactly one)
    pend = meta.pop("pending", None)
    if pend is not None:
        out._phc_pending = pend                              # (the model loop passes it on or materialises it: materialize_pending)
    return out


# ---------------------------------------------------------------------------
# the downstream head in one launch each way (csrc/downstream.hip)
# ---------------------------------------------------------------------------
DOWNSTREAM_FUSED = __import__("os").environ.get("PHC_DOWNSTREAM_FUSED", "1") != "0"
_DN_BUFFERS: dict = {}


def _dn_buffers(dev, which: str, B: int):
    """(sync words, partial buffer) of the fused head, per device, stream and direction: the three sync words must start
    at zero and are owned by the kernels from then on (generation / arrival counters, never reset)."""
    lib = L.load()
    # ... and per number of rows B (hence per number of workgroups: 32 rows each on the first kernels, 16 / 32 on the k4 ones): a
    # launch's granules carry its generation tag, one set of words serves one grid shape only
    key = (dev, torch.cuda.current_stream(dev).cuda_stream, which, B)
    need = lib.phc_downstream_partial_floats(B)
    ent = _DN_BUFFERS.get(key)
    if ent is None or ent[1].numel() < need:
        sync = ent[0] if ent is not None else torch.zeros(4, dtype=torch.int32, device=dev)
        ent = (sync, torch.zeros(max(need, 1), dtype=torch.float32, device=dev))     # tag 0 never matches a launch's tag (>= 1)
        _DN_BUFFERS[key] = ent
    return ent


def _dn_norm(norm_tensors, n_comp, momentum, eps, keep):
    """phc_downstream_norm of one NaivePHMNorm: (weights, biases, running_mean, running_var, nbt) lists of n_comp tensors."""
    w, b, rm, rv, nbt = norm_tensors
    arrs = [_ptr_array(t) if t else None for t in (w, b, rm, rv, nbt)]
    keep.extend(arrs)
    cast = lambda a: _C.cast(a, _C.c_void_p) if a is not None else None
    return L.DownstreamNorm(cast(arrs[0]), cast(arrs[1]), cast(arrs[2]), cast(arrs[3]), cast(arrs[4]), n_comp, float(momentum), float(eps))


def _dn_set_drop(d, drop):
    (p1, k1, s1), (p2, k2, s2), state = drop
    d.drop_p1, d.drop_same_k1, d.drop_site1 = float(p1), int(k1), int(s1)
    d.drop_p2, d.drop_same_k2, d.drop_site2 = float(p2), int(k2), int(s2)
    d.rng_state = L.ptr(state)


class _DownstreamFn(torch.autograd.Function):
    """logits = RealTransformer(PHMDownstreamNet(x)) for two hidden layers, training mode (downstream.py:90-120,
    layers.py:399-414): ONE engine launch forward, ONE backward (``phc_downstream_fwd`` / ``_bwd``).  The PHM layers'
    parameter gradients leave as three jobs of the deferred / batched gradient machinery (x, dz1), ((z1, stat1), dz2),
    ((z2, stat2), dz3); the norms' affine gradients and the dense head's come straight out of the backward launch."""

    @staticmethod
    def forward(ctx, x, n_valid, meta, A1, W1, b1, A2, W2, b2, A3, W3, b3, Wr, br, *affine):
        lib = L.load()
        dev = L.require_device(x, Wr)
        act, norms, pre = meta["act"], meta["norms"], meta["pre"]
        x_ = L.f32c(x)
        B, d0 = x_.shape
        d1, d2, d3 = W1.shape[0] * W1.shape[2], W2.shape[0] * W2.shape[2], W3.shape[0] * W3.shape[2]
        T = Wr.shape[0]
        z1 = torch.empty((B, d1), dtype=torch.float32, device=dev)
        z2 = torch.empty((B, d2), dtype=torch.float32, device=dev)
        z3 = torch.empty((B, d3), dtype=torch.float32, device=dev)
        logits = torch.empty((B, T), dtype=torch.float32, device=dev)
        stat1 = torch.empty(6 * d1, dtype=torch.float32, device=dev)
        stat2 = torch.empty(6 * d2, dtype=torch.float32, device=dev)
        sync, partial = _dn_buffers(dev, "fwd", B)
        keep = []
        d = L.DownstreamDesc()
        d.B, d.d0, d.d1, d.d2, d.d3, d.T, d.act = B, d0, d1, d2, d3, T, act
        d.n_valid_dev = L.ptr(n_valid)
        d.x, d.H1, d.b1, d.H2, d.b2, d.H3, d.b3 = L.ptr(x_), L.ptr(pre[0][0]), L.ptr(b1), L.ptr(pre[1][0]), L.ptr(b2), L.ptr(pre[2][0]), L.ptr(b3)
        d.Wr, d.br = L.ptr(Wr), L.ptr(br)
        d.norm1 = _dn_norm(norms[0]["tensors"], norms[0]["n_comp"], norms[0]["momentum"], norms[0]["eps"], keep)
        d.norm2 = _dn_norm(norms[1]["tensors"], norms[1]["n_comp"], norms[1]["momentum"], norms[1]["eps"], keep)
        d.z1, d.z2, d.z3, d.logits, d.stat1, d.stat2 = L.ptr(z1), L.ptr(z2), L.ptr(z3), L.ptr(logits), L.ptr(stat1), L.ptr(stat2)
        d.partial, d.sync = L.ptr(partial), L.ptr(sync)
        d.H1k4, d.H2k4 = L.ptr(pre[0][2]), L.ptr(pre[1][2])                        # k4 forms (None: the 32-row kernels)
        d.H1Tk4, d.H2Tk4 = L.ptr(pre[0][3]), L.ptr(pre[1][3])
        drop = meta.get("drop")                              # ((p1, same_k1, site1), (p2, same_k2, site2), rng state) or None
        a1 = a2 = None
        if drop is not None:
            a1 = torch.empty((B, d1), dtype=torch.float32, device=dev)
            a2 = torch.empty((B, d2), dtype=torch.float32, device=dev)
            _dn_set_drop(d, drop)
            d.a1, d.a2 = L.ptr(a1), L.ptr(a2)
        loss = meta.get("loss")                              # (labels, kind) when the step said which loss follows (expect_loss)
        if loss is not None:
            y_, kind = loss
            out3 = torch.empty(3, dtype=torch.float32, device=dev)
            dz_unit = torch.empty((B, T), dtype=torch.float32, device=dev)
            d.loss_y, d.loss_kind, d.loss_out, d.loss_dlogits = L.ptr(y_), kind, L.ptr(out3), L.ptr(dz_unit)
            d.loss_granules = L.ptr(_dn_loss_granules(dev, B, B * T))
            meta["loss_result"] = (out3, dz_unit, y_, kind, y_._version)
        ctx.fused_bwd = None
        if loss is not None and HEAD_FUSED_BWD and any(ctx.needs_input_grad) and pre[0][1] is not None and pre[1][1] is not None and \
                pre[2][1] is not None and _supported("phc_downstream_fused_bwd_supported", B, d0, d1, d2, d3, T):
            # ... and the BACKWARD in the same launch, for an upstream gradient of 1 (the step's loss.backward(): ops.register_unit_seed):
            # d loss / d logits of a workgroup's rows needs the labeled count only, which every workgroup counts itself.  The backward
            # node hands these results out when the gradient it receives IS dz_unit, and launches phc_downstream_bwd otherwise.
            need_x = ctx.needs_input_grad[0]
            fb = dict(dx=torch.empty_like(x_) if need_x else None, dz1=torch.empty_like(z1), dz2=torch.empty_like(z2), dz3=torch.empty_like(z3),
                      dwb1=torch.empty(2 * d1, dtype=torch.float32, device=dev), dwb2=torch.empty(2 * d2, dtype=torch.float32, device=dev),
                      dWr=torch.empty(T * d3 + T, dtype=torch.float32, device=dev), token=dz_unit)
            sync_b, partial_b = _dn_buffers(dev, "bwd", B)
            d.H1T, d.H2T, d.H3T = L.ptr(pre[0][1]), L.ptr(pre[1][1]), L.ptr(pre[2][1])
            d.dx, d.dz1, d.dz2, d.dz3 = L.ptr(fb["dx"]), L.ptr(fb["dz1"]), L.ptr(fb["dz2"]), L.ptr(fb["dz3"])
            d.dwb1, d.dwb2, d.dWr = L.ptr(fb["dwb1"]), L.ptr(fb["dwb2"]), L.ptr(fb["dWr"])
            d.fused_bwd, d.partial_bwd, d.sync_bwd = 1, L.ptr(partial_b), L.ptr(sync_b)
            ctx.fused_bwd = fb
        L.check(lib.phc_downstream_fwd(_C.byref(d), L.stream_ptr(dev)), "phc_downstream_fwd")
        if drop is None:
            _tap(z1, stat1, n_valid, act)
            _tap(z2, stat2, n_valid, act)
        if L.PROF_ON:
            L.account("downstream_fwd", 2.0 * B * (d0 * d1 + d1 * d2 + d2 * d3 + d3 * T), 4.0 * (B * (d0 + d1 + d2 + d3 + T) + d0 * d1 + d1 * d2))
        ctx.save_for_backward(x_, z1, z2, z3, stat1, stat2, Wr)
        ctx.meta = meta
        ctx.n_valid = n_valid
        ctx.phm = ((A1, W1, b1), (A2, W2, b2), (A3, W3, b3))
        ctx.n_affine = len(affine)
        ctx.dropped = (a1, a2)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        lib = L.load()
        x_, z1, z2, z3, stat1, stat2, Wr = ctx.saved_tensors
        meta, n_valid = ctx.meta, ctx.n_valid
        act, norms, pre = meta["act"], meta["norms"], meta["pre"]
        dev = x_.device
        B, d0 = x_.shape
        d1, d2, d3, T = z1.shape[1], z2.shape[1], z3.shape[1], Wr.shape[0]
        dl = L.f32c(dlogits)
        need_x = ctx.needs_input_grad[0]
        a1, a2 = ctx.dropped
        fb = ctx.fused_bwd
        if fb is not None and dl.data_ptr() == fb["token"].data_ptr() and tuple(dl.shape) == tuple(fb["token"].shape):
            # the forward launch ran this backward already (upstream gradient 1: the gradient received IS its d loss / d logits)
            dx, dz1, dz2, dz3, dwb1, dwb2, dWr = fb["dx"], fb["dz1"], fb["dz2"], fb["dz3"], fb["dwb1"], fb["dwb2"], fb["dWr"]
            return _DownstreamFn._finish_backward(ctx, x_, z1, z2, stat1, stat2, n_valid, act, norms, a1, a2, dx, dz1, dz2, dz3, dwb1, dwb2, dWr,
                                                  d1, d2, d3, T)
        dx = torch.empty_like(x_) if need_x else None
        dz1, dz2, dz3 = torch.empty_like(z1), torch.empty_like(z2), torch.empty_like(z3)
        dwb1 = torch.empty(2 * d1, dtype=torch.float32, device=dev)
        dwb2 = torch.empty(2 * d2, dtype=torch.float32, device=dev)
        dWr = torch.empty(T * d3 + T, dtype=torch.float32, device=dev)
        sync, partial = _dn_buffers(dev, "bwd", B)
        d = L.DownstreamDesc()
        d.B, d.d0, d.d1, d.d2, d.d3, d.T, d.act = B, d0, d1, d2, d3, T, act
        d.n_valid_dev = L.ptr(n_valid)
        d.H1T, d.H2T, d.H3T, d.Wr = L.ptr(pre[0][1]), L.ptr(pre[1][1]), L.ptr(pre[2][1]), L.ptr(Wr)
        d.z1, d.z2, d.z3, d.stat1, d.stat2 = L.ptr(z1), L.ptr(z2), L.ptr(z3), L.ptr(stat1), L.ptr(stat2)
        d.dlogits = L.ptr(dl)
        d.dx, d.dz1, d.dz2, d.dz3, d.dwb1, d.dwb2, d.dWr = L.ptr(dx), L.ptr(dz1), L.ptr(dz2), L.ptr(dz3), L.ptr(dwb1), L.ptr(dwb2), L.ptr(dWr)
        d.partial, d.sync = L.ptr(partial), L.ptr(sync)
        d.H1k4, d.H2k4 = L.ptr(pre[0][2]), L.ptr(pre[1][2])
        d.H1Tk4, d.H2Tk4 = L.ptr(pre[0][3]), L.ptr(pre[1][3])
        if meta.get("drop") is not None:
            _dn_set_drop(d, meta["drop"])                    # the forward's masks, re-derived from (seed, step, site, row, column)
        L.check(lib.phc_downstream_bwd(_C.byref(d), L.stream_ptr(dev)), "phc_downstream_bwd")
        if L.PROF_ON:
            L.account("downstream_bwd", 2.0 * B * (d0 * d1 + d1 * d2 + 2 * d2 * d3 + 2 * d3 * T), 4.0 * (B * (d0 + 2 * d1 + 2 * d2 + 2 * d3 + T) + d0 * d1 + d1 * d2))
        return _DownstreamFn._finish_backward(ctx, x_, z1, z2, stat1, stat2, n_valid, act, norms, a1, a2, dx, dz1, dz2, dz3, dwb1, dwb2, dWr,
                                              d1, d2, d3, T)

    @staticmethod
    def _finish_backward(ctx, x_, z1, z2, stat1, stat2, n_valid, act, norms, a1, a2, dx, dz1, dz2, dz3, dwb1, dwb2, dWr, d1, d2, d3, T):
        # the three PHM layers' parameter gradients: jobs for the batched gradient launches
        needs = ctx.needs_input_grad
        # (with dropout the layers' inputs are the dropped activations the forward wrote: the operand transform knows no mask)
        xs = ((x_, None), (z1, (stat1, n_valid, act)), (z2, (stat2, n_valid, act))) if a1 is None else ((x_, None), (a1, None), (a2, None))
        dys = (dz1, dz2, dz3)
        jobs, grads = [], []
        for i, (A, W, b) in enumerate(ctx.phm):
            nA, nW, nb = needs[3 + 3 * i], needs[4 + 3 * i], needs[5 + 3 * i] and b is not None
            if not (nA or nW or nb):
                grads.append((None, None, None))
                continue
            job = dict(x=xs[i][0], dy=dys[i], A=A, W=W, need=(nA, nW, nb), params=(A if nA else None, W if nW else None, b if nb else None),
                       late=True)
            if xs[i][1] is not None:
                job["xf"] = xs[i][1]
            jobs.append((i, job))
            grads.append(None)
        if jobs:
            deferred = _DEFER.active and not torch.is_grad_enabled() and \
                all(p is None or p.is_leaf for _, jb in jobs for p in jb["params"])
            if deferred:
                with _DEFER.lock:
                    _DEFER.jobs.extend(jb for _, jb in jobs)
                for i, _ in jobs:
                    grads[i] = (None, None, None)
            else:
                for _, jb in jobs:
                    jb["params"] = (None, None, None)
                outs = _flush_param_grads([jb for _, jb in jobs], return_grads=True)
                for (i, _), o in zip(jobs, outs):
                    grads[i] = o
        flat = []
        for gA, gW, gb in grads:
            flat.extend([gA, gW, gb])
        aff = []
        if ctx.n_affine:
            for dwb, nm, width in ((dwb1, norms[0], d1), (dwb2, norms[1], d2)):
                if not nm["affine"]:
                    continue
                n, k = nm["n_comp"], width // nm["n_comp"]
                aff.extend([dwb[c * k:(c + 1) * k] for c in range(n)])
                aff.extend([dwb[width + c * k:width + (c + 1) * k] for c in range(n)])
        return (dx, None, None, *flat, dWr[:T * d3].view(T, d3), dWr[T * d3:], *aff)


HEAD_CHAIN = __import__("os").environ.get("PHC_HEAD_CHAIN", "1") != "0"
_HC_BUFFERS: dict = {}


def _hc_buffers(dev, which: str, B: int, widths: tuple):
    """(sync words, work buffer) of a head-chain launch, per device, stream, direction and shape: zero before the first use and
    owned by the kernels from then on (generation-tagged granules, as ``_dn_buffers``)."""
    lib = L.load()
    key = (dev, torch.cuda.current_stream(dev).cuda_stream, which, B, widths)
    ent = _HC_BUFFERS.get(key)
    if ent is None:
        arr = (_C.c_int32 * len(widths))(*widths)
        need = lib.phc_head_chain_work_floats(B, len(widths), arr)
        ent = (torch.zeros(4, dtype=torch.int32, device=dev), torch.zeros(max(need, 4), dtype=torch.float32, device=dev))
        _HC_BUFFERS[key] = ent
    return ent


def _hc_stage(d, i, M, bias, K, N, post, out, z=None, stat=None, dwb=None, norm=None, drop=None, a_out=None):
    st = d.stage[i]
    st.M, st.bias, st.K, st.N, st.post, st.out = L.ptr(M), L.ptr(bias), K, N, post, L.ptr(out)
    st.z, st.stat, st.dwb = L.ptr(z), L.ptr(stat), L.ptr(dwb)
    if norm is not None:
        st.norm = norm
    if drop is not None:                                   # (p, same_k, site)
        st.drop_p, st.drop_same_k, st.drop_site = float(drop[0]), int(drop[1]), int(drop[2])
    st.a_out = L.ptr(a_out)


class _HeadChainFn(torch.autograd.Function):
    """logits = RealTransformer(PHMDownstreamNet(x)) for two hidden layers, training mode (downstream.py:90-120,
    layers.py:399-414) at widths beyond ``_DownstreamFn`` (molpcba: 256 -> 768 -> 256 -> 512 -> 128): ONE ``phc_head_chain`` launch
    forward — stages (H1^T, H2^T, H3^T, Wr) with norm + activation behind the first two — and ONE backward — stages
    (Wr^T, H3, H2, H1) with the norms' backward behind the second and third.  The four layers' parameter gradients leave as jobs of
    the deferred / batched gradient machinery: (x, dz1), ((z1, stat1), dz2), ((z2, stat2), dz3) and the dense (z3, dlogits)."""

    @staticmethod
    def forward(ctx, x, n_valid, meta, A1, W1, b1, A2, W2, b2, A3, W3, b3, Wr, br, *affine):
        lib = L.load()
        dev = L.require_device(x, Wr)
        act, norms, pre = meta["act"], meta["norms"], meta["pre"]
        x_ = L.f32c(x)
        B, d0 = x_.shape
        d1, d2, d3 = W1.shape[0] * W1.shape[2], W2.shape[0] * W2.shape[2], W3.shape[0] * W3.shape[2]
        T = Wr.shape[0]
        z1 = torch.empty((B, d1), dtype=torch.float32, device=dev)
        z2 = torch.empty((B, d2), dtype=torch.float32, device=dev)
        z3 = torch.empty((B, d3), dtype=torch.float32, device=dev)
        logits = torch.empty((B, T), dtype=torch.float32, device=dev)
        stat1 = torch.empty(6 * d1, dtype=torch.float32, device=dev)
        stat2 = torch.empty(6 * d2, dtype=torch.float32, device=dev)
        sync, work = _hc_buffers(dev, "fwd", B, (d1, d2, d3, T))
        keep = []
        d = L.HeadChainDesc()
        d.B, d.n_stages, d.act, d.n_valid_dev, d.in_ = B, 4, act, L.ptr(n_valid), L.ptr(x_)
        d.valid_rows_max = meta["valid_max"]
        drop = meta.get("drop")                              # ((p1, same_k1, site1), (p2, same_k2, site2), rng state) or None
        a1 = a2 = None
        if drop is not None:
            a1 = torch.empty((B, d1), dtype=torch.float32, device=dev)
            a2 = torch.empty((B, d2), dtype=torch.float32, device=dev)
            d.rng_state = L.ptr(drop[2])
        _hc_stage(d, 0, pre[0][1], b1, d0, d1, 1, z1, stat=stat1,
                  norm=_dn_norm(norms[0]["tensors"], norms[0]["n_comp"], norms[0]["momentum"], norms[0]["eps"], keep),
                  drop=drop[0] if drop else None, a_out=a1)
        _hc_stage(d, 1, pre[1][1], b2, d1, d2, 1, z2, stat=stat2,
                  norm=_dn_norm(norms[1]["tensors"], norms[1]["n_comp"], norms[1]["momentum"], norms[1]["eps"], keep),
                  drop=drop[1] if drop else None, a_out=a2)
        _hc_stage(d, 2, pre[2][1], b3, d2, d3, 0, z3)
        _hc_stage(d, 3, Wr, br, d3, T, 0, logits)
        d.work, d.sync = L.ptr(work), L.ptr(sync)
        L.check(lib.phc_head_chain(_C.byref(d), L.stream_ptr(dev)), "phc_head_chain(forward)")
        if drop is None:
            _tap(z1, stat1, n_valid, act)
            _tap(z2, stat2, n_valid, act)
        if L.PROF_ON:
            L.account("head_chain_fwd", 2.0 * B * (d0 * d1 + d1 * d2 + d2 * d3 + d3 * T),
                      4.0 * (B * (d0 + d1 + d2 + d3 + T) + d0 * d1 + d1 * d2 + d2 * d3 + d3 * T))
        ctx.save_for_backward(x_, z1, z2, z3, stat1, stat2, Wr)
        ctx.meta = meta
        ctx.n_valid = n_valid
        ctx.phm = ((A1, W1, b1), (A2, W2, b2), (A3, W3, b3))
        ctx.br = br
        ctx.n_affine = len(affine)
        ctx.dropped = (a1, a2)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        lib = L.load()
        x_, z1, z2, z3, stat1, stat2, Wr = ctx.saved_tensors
        meta, n_valid = ctx.meta, ctx.n_valid
        act, norms, pre, WrT = meta["act"], meta["norms"], meta["pre"], meta["WrT"]
        dev = x_.device
        B, d0 = x_.shape
        d1, d2, d3, T = z1.shape[1], z2.shape[1], z3.shape[1], Wr.shape[0]
        dl = L.f32c(dlogits)
        need_x = ctx.needs_input_grad[0]
        dx = torch.empty_like(x_) if need_x else None
        dz1, dz2, dz3 = torch.empty_like(z1), torch.empty_like(z2), torch.empty_like(z3)
        dwb1 = torch.empty(2 * d1, dtype=torch.float32, device=dev)
        dwb2 = torch.empty(2 * d2, dtype=torch.float32, device=dev)
        widths = (d3, d2, d1, d0) if need_x else (d3, d2, d1)
        sync, work = _hc_buffers(dev, "bwd", B, widths)
        d = L.HeadChainDesc()
        d.B, d.n_stages, d.act, d.n_valid_dev, d.in_ = B, len(widths), act, L.ptr(n_valid), L.ptr(dl)
        d.valid_rows_max = meta["valid_max"]
        _hc_stage(d, 0, WrT, None, T, d3, 0, dz3)                                  # dz3 = dlogits . Wr
        drop = meta.get("drop")
        a1, a2 = ctx.dropped
        if drop is not None:
            d.rng_state = L.ptr(drop[2])                     # the forward's masks, re-derived from (seed, step, site, row, column)
        _hc_stage(d, 1, pre[2][0], None, d3, d2, 2, dz2, z=z2, stat=stat2, dwb=dwb2, drop=drop[1] if drop else None)      # da2 = dz3 . H3^T, norm 2 backwards
        _hc_stage(d, 2, pre[1][0], None, d2, d1, 2, dz1, z=z1, stat=stat1, dwb=dwb1, drop=drop[0] if drop else None)      # da1 = dz2 . H2^T, norm 1 backwards
        if need_x:
            _hc_stage(d, 3, pre[0][0], None, d1, d0, 0, dx)                         # dx = dz1 . H1^T
        d.work, d.sync = L.ptr(work), L.ptr(sync)
        L.check(lib.phc_head_chain(_C.byref(d), L.stream_ptr(dev)), "phc_head_chain(backward)")
        if L.PROF_ON:
            L.account("head_chain_bwd", 2.0 * B * (d0 * d1 + d1 * d2 + d2 * d3 + d3 * T),
                      4.0 * (B * (d0 + 2 * d1 + 2 * d2 + d3 + T) + d0 * d1 + d1 * d2 + d2 * d3 + d3 * T))
        needs = ctx.needs_input_grad
        xs = ((x_, None), (z1, (stat1, n_valid, act)), (z2, (stat2, n_valid, act))) if a1 is None else ((x_, None), (a1, None), (a2, None))
        dys = (dz1, dz2, dz3)
        jobs, grads = [], []
        for i, (A, W, b) in enumerate(ctx.phm):
            nA, nW, nb = needs[3 + 3 * i], needs[4 + 3 * i], needs[5 + 3 * i] and b is not None
            if not (nA or nW or nb):
                grads.append((None, None, None))
                continue
            job = dict(x=xs[i][0], dy=dys[i], A=A, W=W, need=(nA, nW, nb), params=(A if nA else None, W if nW else None, b if nb else None),
                       late=True)
            if xs[i][1] is not None:
                job["xf"] = xs[i][1]
            jobs.append((i, job))
            grads.append(None)
        br = ctx.br
        nWr, nbr = needs[12], needs[13] and br is not None
        if nWr or nbr:                                        # the dense head in PHM form: n = 1, W = Wr^T, dW written as [T, d3]
            jobs.append((3, dict(x=z3, dy=dl, A=_one(dev), W=Wr, need=(False, nWr, nbr),
                                 params=(None, Wr if nWr else None, br if nbr else None), dense=True, late=True)))
        grads.append((None, None, None))
        if jobs:
            deferred = _DEFER.active and not torch.is_grad_enabled() and \
                all(p is None or p.is_leaf for _, jb in jobs for p in jb["params"])
            if deferred:
                with _DEFER.lock:
                    _DEFER.jobs.extend(jb for _, jb in jobs)
                for i, _ in jobs:
                    grads[i] = (None, None, None)
            else:
                for _, jb in jobs:
                    jb["params"] = (None, None, None)
                outs = _flush_param_grads([jb for _, jb in jobs], return_grads=True)
                for (i, _), o in zip(jobs, outs):
                    grads[i] = o
        flat = []
        for gA, gW, gb in grads[:3]:
            flat.extend([gA, gW, gb])
        aff = []
        if ctx.n_affine:
            for dwb, nm, width in ((dwb1, norms[0], d1), (dwb2, norms[1], d2)):
                if not nm["affine"]:
                    continue
                n, k = nm["n_comp"], width // nm["n_comp"]
                aff.extend([dwb[c * k:(c + 1) * k] for c in range(n)])
                aff.extend([dwb[width + c * k:width + (c + 1) * k] for c in range(n)])
        return (dx, None, None, *flat, grads[3][1], grads[3][2], *aff)


HEAD_DROPOUT = __import__("os").environ.get("PHC_HEAD_DROPOUT", "1") != "0"


def downstream_head(x, n_valid, affine_layers, norm_modules, real_affine, activation: str, dropout=None):
    """The fused head, or ``None`` when its preconditions do not hold (the caller then runs layer by layer): three PHM
    layers with prebuilt operand matrices (inside ``prebuilt_H``), two naive batch-norms with running statistics, a dense
    nn.Linear head, widths the kernel supports (``phc_downstream_supported``)."""
    if not (DOWNSTREAM_FUSED and _PREBUILT.table and x.is_cuda and x.dim() == 2 and len(affine_layers) == 3 and len(norm_modules) == 2):
        return None
    lib = L.load()
    pre = []
    for lin in affine_layers:
        ent = _prebuilt_for(lin.phm_rule, lin.W)
        if ent is None or ent[1] is None:
            return None
        pre.append(ent)
    W1, W2, W3 = (lin.W for lin in affine_layers)
    d0, d1, d2, d3 = W1.shape[0] * W1.shape[1], W1.shape[0] * W1.shape[2], W2.shape[0] * W2.shape[2], W3.shape[0] * W3.shape[2]
    Wr, br = real_affine.weight, real_affine.bias
    T = Wr.shape[0]
    has_k4 = all(pre[i][2] is not None and pre[i][3] is not None for i in (0, 1))
    supported = lib.phc_downstream_supported_k4 if has_k4 else lib.phc_downstream_supported
    if x.shape[1] != d0 or W2.shape[0] * W2.shape[1] != d1 or W3.shape[0] * W3.shape[1] != d2 or Wr.shape[1] != d3 or \
            x.dtype != torch.float32 or not Wr.is_contiguous():
        return None
    chain = False
    WrT, valid_max = None, x.shape[0]
    drop = None
    if dropout is not None and any(p > 0.0 for p in dropout[0]):
        # phm_dropout behind the two activations (``dropout`` = ((p1, p2), same, (site1, site2))): the 16-row kernels draw the masks
        # of the stand-alone fused passes; widths and component widths multiples of 4
        (p1, p2), same, (s1, s2) = dropout
        n1, n2 = len(list(norm_modules[0].bn)), len(list(norm_modules[1].bn))
        k1, k2 = (d1 // n1 if same else 0), (d2 // n2 if same else 0)
        if not (HEAD_DROPOUT and d1 % 4 == 0 and d2 % 4 == 0 and k1 % 4 == 0 and k2 % 4 == 0):
            return None
        # (served by the 16-row kernels or by the chain kernel; the 32-row kernels know no mask)
        if supported(x.shape[0], d0, d1, d2, d3, T) and not (has_k4 and lib.phc_downstream_dropout_supported(x.shape[0], d0, d1, d2, d3, T)):
            return None
        drop = ((float(p1), k1, int(s1)), (float(p2), k2, int(s2)), rng_state(x.device))
    if not supported(x.shape[0], d0, d1, d2, d3, T):
        # beyond the 16-row kernels (molpcba: 768-wide hidden layer, 128 targets): the 2-D chain kernel, when its shapes and the
        # device allow it and the dense head's transpose is among the step's prebuilt operands
        dense = _prebuilt_for(None, Wr) if HEAD_CHAIN else None
        if dense is None or real_affine.bias is None:
            return None
        # (a fixed-shape batch says how many of its graphs can be real: the rows behind them are padding in every replay)
        valid_max = min(int(getattr(n_valid, "_phc_max_valid", x.shape[0])), x.shape[0]) if n_valid is not None else x.shape[0]
        Ks, Ns = (_C.c_int32 * 4)(d0, d1, d2, d3), (_C.c_int32 * 4)(d1, d2, d3, T)
        if not lib.phc_head_chain_supported(x.shape[0], valid_max, 4, Ks, Ns):
            return None
        Kb, Nb = (_C.c_int32 * 4)(T, d3, d2, d1), (_C.c_int32 * 4)(d3, d2, d1, d0)
        if not lib.phc_head_chain_supported(x.shape[0], valid_max, 4, Kb, Nb):
            return None
        chain, WrT = True, dense[0]
    norms, affine = [], []
    for nm in norm_modules:
        mods = list(nm.bn)
        if not nm.track_running_stats or nm.momentum is None:
            return None
        w = [m.weight for m in mods] if nm.affine else None
        b = [m.bias for m in mods] if nm.affine else None
        norms.append(dict(tensors=(w, b, [m.running_mean for m in mods], [m.running_var for m in mods],
                                   [m.num_batches_tracked for m in mods]), n_comp=len(mods), momentum=nm.momentum, eps=nm.eps,
                          affine=bool(nm.affine)))
        if nm.affine:
            affine.extend(w + b)
    meta = dict(act=_act_id(activation), norms=norms, pre=pre, WrT=WrT, valid_max=valid_max, drop=drop)
    l1, l2, l3 = affine_layers
    if chain:
        return _HeadChainFn.apply(x, n_valid, meta, l1.phm_rule, l1.W, l1.b, l2.phm_rule, l2.W, l2.b, l3.phm_rule, l3.W, l3.b, Wr, br, *affine)
    hint = getattr(_LOSS_HINT, "spec", None)
    if hint is not None and has_k4 and torch.is_grad_enabled() and tuple(hint[0].shape) == (x.shape[0], T) and hint[0].device == x.device \
            and lib.phc_downstream_loss_supported(x.shape[0], d0, d1, d2, d3, T):
        meta["loss"] = hint
    out = _DownstreamFn.apply(x, n_valid, meta, l1.phm_rule, l1.W, l1.b, l2.phm_rule, l2.W, l2.b, l3.phm_rule, l3.W, l3.b, Wr, br, *affine)
    if meta.get("loss_result") is not None:
        out._phc_loss = meta.pop("loss_result")              # what masked_bce_with_logits / masked_l1 on these logits return
    return out



# ---------------------------------------------------------------------------
# The step's contexts entered by the MODEL when nobody else did: the reference's loop body
# (benchmarks/train_hiv.py:174-202) calls ``model(data)``, ``loss.backward()`` and knows nothing of ``prebuilt_H`` /
# ``stat_accumulators`` / ``deferred_param_grads``.  A training-mode forward on a ROCm device outside those contexts opens them
# itself and closes them at the END of the backward pass that reaches its output (an autograd-engine callback queued from a hook
# on the logits): the operand matrices of all layers are built by one launch, the statistics travel through accumulators, and the
# parameter gradients of all layers come out of the step tail's batched launches — the launches of ``harness.train_step``, in the
# unchanged loop.  A backward that does NOT accumulate into ``.grad`` (``torch.autograd.grad``) is recognised in that hook (the
# parameters' AccumulateGrad nodes are not part of the running graph task) and takes the per-layer gradient path instead.  A forward
# whose backward never comes leaves the block open; the next forward (or ``close_auto_step``) closes it without flushing anything.
# ---------------------------------------------------------------------------
AUTO_STEP = __import__("os").environ.get("PHC_AUTO_STEP", "1") != "0"


# (_AUTO: defined beside _DEFER)


def step_context_active() -> bool:
    return bool(_PREBUILT.table) or _DEFER.active


def auto_step_open() -> bool:
    return _AUTO.stack is not None


def close_auto_step(flush: bool = False) -> None:
    """Close a step block a forward opened by itself (no-op when none is open).  ``flush`` False: drop whatever was deferred."""
    stack = _AUTO.stack
    if stack is None:
        return
    _AUTO.stack = None
    if flush and _DEFER.active:
        _deferred_flush()
    else:
        _DEFER.active = False
        _deferred_reset()
    stack.close()


def _accumulate_nodes(model, all_params=None):
    """The AccumulateGrad nodes of the model's leaf parameters (cached on the model; they live as long as the parameters).
    ``all_params``: the caller's ``list(model.parameters())`` of this step (one module-tree walk per step instead of two: the walk is
    0.3 ms of host time at 205 parameter tensors, scripts/dbg/refloop_profile.py)."""
    cache = model.__dict__.get("_phc_acc_nodes")
    params = [p for p in (model.parameters() if all_params is None else all_params) if p.requires_grad]
    if cache is None or cache[0] != tuple(id(p) for p in params):
        with torch.enable_grad():
            nodes = [p.view_as(p).grad_fn.next_functions[0][0] for p in params]
        cache = model.__dict__["_phc_acc_nodes"] = (tuple(id(p) for p in params), nodes)
    return cache[1]


def auto_step_forward(model, data, run):
    """``run()`` = the model's forward.  Opens the step's contexts around it and arranges for them to close behind the backward.
    Parameter-gradient deferral is live only INSIDE the forward (the layers note that their gradients may be deferred) and between
    the start and the end of the backward pass that reaches the output: code that runs between the two — or a backward that never
    comes — sees no open deferral block."""
    close_auto_step(flush=False)                             # a block whose backward never came
    dev = data.x.device
    specs = model.__dict__.get("_phc_auto_specs")
    all_params = list(model.parameters())
    key = tuple(id(p) for p in all_params)
    if specs is None or specs[0] != key:
        specs = model.__dict__["_phc_auto_specs"] = (key, phm_pairs(model), dense_weights(model), gate_specs(model))
    stack = contextlib.ExitStack()
    _AUTO.in_forward = True
    try:
        stack.enter_context(stat_accumulators(dev))
        stack.enter_context(prebuilt_H(specs[1], specs[2], specs[3], encode=atom_encode_spec(model, data)))
        _DEFER.active = True
        _deferred_reset()
        out = run()
    except BaseException:
        _DEFER.active = False
        _deferred_reset()
        stack.close()
        raise
    finally:
        _AUTO.in_forward = False
    _DEFER.active = False
    if not (torch.is_tensor(out) and out.requires_grad):
        stack.close()
        return out
    _AUTO.stack = stack
    nodes = _accumulate_nodes(model, all_params)

    def at_backward_start(grad):
        if _AUTO.stack is not stack:
            return grad                                      # (closed meanwhile: a second backward over a retained graph runs per layer)
        try:
            accumulating = all(torch._C._will_engine_execute_node(nd) for nd in nodes)
        except Exception:
            accumulating = False
        # torch.autograd.grad / inputs=...: no AccumulateGrad runs — every layer returns its own gradients (no deferral)
        _DEFER.active = accumulating
        torch.autograd.Variable._execution_engine.queue_callback(lambda: close_auto_step(flush=True) if _AUTO.stack is stack else None)
        return grad
    out.register_hook(at_backward_start)
    return out


class _WeightRegFn(torch.autograd.Function):
    """sum_W mean_{kappa, pi} ||W[:, kappa, pi]||_p (regularization.py:15-23) for ROCm weights: the value and d / dW from ONE engine
    launch in the forward; the backward scales the stored gradients by the upstream gradient."""

    @staticmethod
    def forward(ctx, p, *Ws):
        lib = L.load()
        dev = L.require_device(*Ws)
        k = len(Ws)
        flat = torch.zeros(sum(w.numel() for w in Ws), dtype=torch.float32, device=dev)
        grads, off = [], 0
        for w in Ws:
            grads.append(flat[off:off + w.numel()].view_as(w))
            off += w.numel()
        W_arr = (_C.c_void_p * k)(*[w.data_ptr() for w in Ws])
        dW_arr = (_C.c_void_p * k)(*[g.data_ptr() for g in grads])
        n_arr = (_C.c_int32 * k)(*[w.shape[0] for w in Ws])
        c_arr = (_C.c_int32 * k)(*[w.shape[1] * w.shape[2] for w in Ws])
        reg = torch.empty((), dtype=torch.float32, device=dev)
        key = (dev, torch.cuda.current_stream(dev).cuda_stream)
        ws = _REG_WS.get(key)
        if ws is None:
            ws = _REG_WS[key] = torch.zeros(lib.phc_phm_weight_reg_ws_bytes(), dtype=torch.uint8, device=dev)
        L.check(lib.phc_phm_weight_reg(W_arr, dW_arr, n_arr, c_arr, k, 1.0, int(p), L.ptr(reg), None, L.ptr(ws), ws.numel(), L.stream_ptr(dev)),
                "phc_phm_weight_reg")
        ctx.grads = grads
        return reg

    @staticmethod
    def backward(ctx, g):
        return (None, *torch._foreach_mul(ctx.grads, g))


def weight_regularization(Ws, p: int):
    """The reference's ``phm_weight_regularization`` value over ``Ws`` as one engine launch (differentiable), or None when the
    weights are outside the kernel's range (the caller evaluates the PyTorch expression)."""
    Ws = list(Ws)
    if not Ws or p not in (1, 2) or not all(w.is_cuda and w.dtype == torch.float32 and w.dim() == 3 and w.is_contiguous() and w.shape[0] <= 16
                                           for w in Ws):
        return None
    lib = L.load()
    blocks = sum((w.shape[1] * w.shape[2] + 255) // 256 for w in Ws)
    if len(Ws) > lib.phc_phm_weight_reg_max_jobs() or blocks > lib.phc_phm_weight_reg_max_blocks():
        return None
    return _WeightRegFn.apply(p, *Ws)


# ---------------------------------------------------------------------------
# explicit weight regulariser (regularization.py:15-23, train_hiv.py:181-196) as one launch
# ---------------------------------------------------------------------------
def regularised_weights(model) -> list:
    """The weights the reference's regulariser sees: any sub-module with a non-None attribute ``W`` (regularization.py:15-23)."""
    return [m.W for _, m in model.named_modules() if getattr(m, "W", None) is not None]


def _weight_reg_launch(Ws, coef: float, p: int, loss: Optional[torch.Tensor]):
    class _M:                                                # (phm_weight_regularization_ walks named_modules)
        def __init__(self, ws):
            self._ws = ws

        def named_modules(self):
            for i, w in enumerate(self._ws):
                holder = type("H", (), {})()
                holder.W = w
                yield str(i), holder
    return phm_weight_regularization_(_M(Ws), coef, p, loss)


def phm_weight_regularization_(model, coef: float, p: int = 2, loss: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``coef * phm_weight_regularization(model, p)`` of the reference, evaluated AFTER the backward pass: the value comes
    back as a device scalar (and is added to ``loss`` in place when given, the reference's ``loss += reg_loss``), its
    gradient is added into the ``.grad`` of every ``W`` — what autograd would have accumulated had the term been part of
    the loss.  Modules are found like the reference finds them: any sub-module with a non-None attribute ``W``."""
    lib = L.load()
    Ws = [m.W for _, m in model.named_modules() if getattr(m, "W", None) is not None]
    if not Ws:
        raise L.PhcError("phm_weight_regularization_: the model has no module with a `W` attribute")
    dev = L.require_device(*Ws)
    blocks = sum((w.shape[1] * w.shape[2] + 255) // 256 for w in Ws if w.dim() == 3)       # the kernel's workgroup count
    if p not in (1, 2) or len(Ws) > lib.phc_phm_weight_reg_max_jobs() or any(w.shape[0] > 16 for w in Ws) or \
            blocks > lib.phc_phm_weight_reg_max_blocks():
        # outside the kernel's range: the reference's own expression through autograd
        reg = coef * sum(w.norm(p=p, dim=0).mean() for w in Ws)
        grads = torch.autograd.grad(reg, Ws)
        for w, g in zip(Ws, grads):
            w.grad = g if w.grad is None else w.grad + g
        if loss is not None:
            loss.data.add_(reg.detach())
        return reg.detach()
    for w in Ws:
        if w.dtype != torch.float32 or not w.is_contiguous() or w.dim() != 3:
            raise L.PhcError("phm_weight_regularization_: W must be contiguous fp32 [n, k_in, k_out]")
        if w.grad is None:
            w.grad = torch.zeros_like(w)
    k = len(Ws)
    W_arr = (_C.c_void_p * k)(*[w.data_ptr() for w in Ws])
    dW_arr = (_C.c_void_p * k)(*[w.grad.data_ptr() for w in Ws])
    n_arr = (_C.c_int32 * k)(*[w.shape[0] for w in Ws])
    c_arr = (_C.c_int32 * k)(*[w.shape[1] * w.shape[2] for w in Ws])
    reg = torch.empty((), dtype=torch.float32, device=dev)
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    ws = _REG_WS.get(key)
    if ws is None:
        ws = _REG_WS[key] = torch.zeros(lib.phc_phm_weight_reg_ws_bytes(), dtype=torch.uint8, device=dev)
    L.check(lib.phc_phm_weight_reg(W_arr, dW_arr, n_arr, c_arr, k, float(coef), int(p), L.ptr(reg),
                                   L.ptr(loss) if loss is not None else None, L.ptr(ws), ws.numel(), L.stream_ptr(dev)),
            "phc_phm_weight_reg")
    return reg


_REG_WS: dict = {}
