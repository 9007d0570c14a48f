"""Load-time self check of the H-stationary GEMM (csrc/gemm_hs_device.h).

That kernel's correctness rests on things hipcc cannot see: LDS-DMA transfers and prologue loads issued from inline asm whose
registers / LDS lines are only valid behind hand-counted ``s_waitcnt vmcnt(N)``.  The build audits the generated code
(scripts/audit_hs_asm_loads.py, run by phc_gnn_amd.build), but a compiler or runtime upgrade could still change behaviour in a
way a static audit misses.  So once per process and device, before the first engine op, every variant family is run on a small
problem and compared with the LDS-panel kernel (``phm_stream_kernel``: same entry points without the k4 operand) or, for the
folded norm backward (which has no LDS-panel form), with the same formula (norm.py:30-35 backward) in plain tensor arithmetic
around an LDS-panel GEMM.  On a mismatch the H-stationary kernel is switched off for the process
(``phc_gemm_hs_set_enabled(0)``: every layer then runs on the LDS-panel kernel, the behaviour of ``PHC_GEMM_NO_HS=1``) and a
RuntimeWarning says so.  ``PHC_SELFCHECK=0`` skips the check."""
from __future__ import annotations

import ctypes as C
import os
import warnings

import torch

from . import lib as L

_DONE: dict = {}
TOL = 2e-5                     # max |a - b| / max |b| per tensor: both sides are fp32 sums in different orders


def _k4(M: torch.Tensor) -> torch.Tensor:
    K, N = M.shape
    return M.view(K // 4, 4, N).permute(0, 2, 1).contiguous()


def _rel(a: torch.Tensor, b: torch.Tensor) -> float:
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def _acc_totals(acc: torch.Tensor, h: int, rep: int):
    w = acc.view(rep, 2, 2, h).to(torch.float64)
    tot = (w[:, :, 0, :] * 2.0 ** -12 + w[:, :, 1, :] * 2.0 ** -62).sum(0)
    return tot[0], tot[1]


def _run(dev: torch.device, K: int) -> list:
    """Worst relative deviation per check: [(name, err)]."""
    lib = L.load()
    sp = L.stream_ptr(dev)
    N, valid, Nout = 333, 301, K
    g = torch.Generator(device=dev).manual_seed(1234 + K)
    x = torch.randn(N, K, generator=g, device=dev)
    M = torch.randn(K, Nout, generator=g, device=dev) / K ** 0.5
    b = torch.randn(Nout, generator=g, device=dev)
    M4, MT = _k4(M), M.t().contiguous()
    MT4 = _k4(MT)
    nv = torch.tensor([valid], dtype=torch.int32, device=dev)
    rep = lib.phc_stat_acc_replicas()
    words = lib.phc_stat_acc_words(Nout)
    out = []

    def zeros_acc():
        return torch.zeros(words, dtype=torch.int64, device=dev)

    # 1. plain + bias
    y_hs, y_ref = torch.empty(N, Nout, device=dev), torch.empty(N, Nout, device=dev)
    L.check(lib.phc_gemm_prebuilt(L.ptr(x), L.ptr(M), L.ptr(M4), L.ptr(b), None, N, K, Nout, L.ptr(y_hs), 0, sp), "selfcheck gemm hs")
    L.check(lib.phc_gemm_prebuilt(L.ptr(x), L.ptr(M), None, L.ptr(b), None, N, K, Nout, L.ptr(y_ref), 0, sp), "selfcheck gemm ref")
    out.append(("plain", _rel(y_hs, y_ref)))
    # 2. statistics epilogue (accumulator form)
    a_hs, a_ref = zeros_acc(), zeros_acc()
    L.check(lib.phc_gemm_prebuilt_stats(L.ptr(x), L.ptr(M), L.ptr(M4), L.ptr(b), N, K, Nout, L.ptr(y_hs), L.ptr(nv), L.ptr(a_hs), None, None, sp),
            "selfcheck stats hs")
    L.check(lib.phc_gemm_prebuilt_stats(L.ptr(x), L.ptr(M), None, L.ptr(b), N, K, Nout, L.ptr(y_ref), L.ptr(nv), L.ptr(a_ref), None, None, sp),
            "selfcheck stats ref")
    (s0, q0), (s1, q1) = _acc_totals(a_hs, Nout, rep), _acc_totals(a_ref, Nout, rep)
    out.append(("stats y", _rel(y_hs, y_ref)))
    out.append(("stats sums", max(_rel(s0, s1), _rel(q0, q1))))
    # 3. normalising operand from an accumulator (+ statistics of the output): z = y_ref with its totals a_ref
    n_comp = 4
    k = K // n_comp
    w = [0.5 + torch.rand(k, generator=g, device=dev) for _ in range(n_comp)]
    bb = [torch.randn(k, generator=g, device=dev) for _ in range(n_comp)]
    arr = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    res = {}
    for tag, m4 in (("hs", M4), ("ref", None)):
        rm = [torch.zeros(k, device=dev) for _ in range(n_comp)]
        rv = [torch.ones(k, device=dev) for _ in range(n_comp)]
        nbt = [torch.zeros((), dtype=torch.int64, device=dev) for _ in range(n_comp)]
        save = torch.empty(6 * K, device=dev)
        y2 = torch.empty(N, Nout, device=dev)
        acc2 = zeros_acc()
        L.check(lib.phc_gemm_prebuilt_xf_acc(L.ptr(y_ref), L.ptr(a_ref), arr(w), arr(bb), arr(rm), arr(rv), arr(nbt), n_comp, 0.1, 1e-5,
                                             L.ptr(save), L.ACT_IDS["relu"], L.ptr(nv), L.ptr(M), L.ptr(m4), L.ptr(b), N, K, Nout, L.ptr(y2),
                                             L.ptr(acc2), None, None, sp), "selfcheck xf " + tag)
        res[tag] = (y2, save, torch.cat(rm + rv), acc2)
    out.append(("xf y", _rel(res["hs"][0], res["ref"][0])))
    out.append(("xf save_stat", _rel(res["hs"][1], res["ref"][1])))
    out.append(("xf running", _rel(res["hs"][2], res["ref"][2])))
    t_hs, t_ref = _acc_totals(res["hs"][3], Nout, rep), _acc_totals(res["ref"][3], Nout, rep)
    out.append(("xf sums", max(_rel(t_hs[0], t_ref[0]), _rel(t_hs[1], t_ref[1]))))
    # 4. folded norm backward (BA) + next norm's sums (BW) against the formula in tensor arithmetic
    if lib.phc_gemm_bwd_apply_supported(K, Nout):
        save = res["ref"][1]
        dout = torch.randn(N, K, generator=g, device=dev)
        sums = zeros_acc()
        L.check(lib.phc_bn_bwd_partial_acc(L.ptr(dout), L.ptr(y_ref), L.ptr(save), L.ptr(nv), N, K, L.ACT_IDS["relu"], L.ptr(sums), sp),
                "selfcheck bn_bwd_partial_acc")
        mean, invstd, wv, bv = save[:K], save[K:2 * K], save[4 * K:5 * K], save[5 * K:6 * K]
        xh = (y_ref - mean) * invstd
        gq = torch.where(xh * wv + bv > 0, dout, torch.zeros_like(dout))
        gq[valid:] = 0
        mg, mgx = gq[:valid].double().sum(0) / valid, (gq[:valid].double() * xh[:valid].double()).sum(0) / valid
        dz_ref = (wv * invstd) * (gq - mg.float() - xh * mgx.float())
        dz_ref[valid:] = 0
        ds_ref = torch.empty(N, Nout, device=dev)
        L.check(lib.phc_gemm_prebuilt(L.ptr(dz_ref), L.ptr(MT), None, None, None, N, K, Nout, L.ptr(ds_ref), 1, sp), "selfcheck dx ref")
        # second stage: the sums of a norm in front (its input x, statistics save2 = the second norm of check 3)
        save2 = res["ref"][1]
        d = L.GemmBaDesc()
        dz, ds, dwb, acc_bw = torch.empty(N, K, device=dev), torch.empty(N, Nout, device=dev), torch.empty(2 * K, device=dev), zeros_acc()
        d.dout, d.bn_x, d.bn_stat, d.sums_acc, d.n_valid_dev = L.ptr(dout), L.ptr(y_ref), L.ptr(save), L.ptr(sums), L.ptr(nv)
        d.dz, d.dwb, d.Mk4, d.y = L.ptr(dz), L.ptr(dwb), L.ptr(MT4), L.ptr(ds)
        d.N, d.K, d.Nout, d.act = N, K, Nout, L.ACT_IDS["relu"]
        d.bw_act, d.bw_x, d.bw_stat, d.bw_sums_acc = L.ACT_IDS["relu"], L.ptr(x), L.ptr(save2), L.ptr(acc_bw)
        L.check(lib.phc_gemm_bwd_apply(C.byref(d), sp), "selfcheck bwd_apply")
        out.append(("ba dz", _rel(dz, dz_ref)))
        out.append(("ba dx", _rel(ds, ds_ref)))
        out.append(("ba dweight|dbias", _rel(dwb, torch.cat([mgx * valid, mg * valid]).float())))
        mean2, inv2, w2, b2 = save2[:K], save2[K:2 * K], save2[4 * K:5 * K], save2[5 * K:6 * K]
        xh2 = (x - mean2) * inv2
        g2 = torch.where(xh2 * w2 + b2 > 0, ds_ref, torch.zeros_like(ds_ref))[:valid].double()
        sg, sgx = _acc_totals(acc_bw, Nout, rep)
        out.append(("bw sums", max(_rel(sg, g2.sum(0)), _rel(sgx, (g2 * xh2[:valid].double()).sum(0)))))
    return out


def hs_self_check(dev: torch.device) -> bool:
    """True = the H-stationary kernel stays on.  Runs once per process and device; never while a stream capture is open (the
    check then waits for the next call)."""
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key in _DONE:
        return _DONE[key]
    if os.environ.get("PHC_SELFCHECK", "1") == "0":
        _DONE[key] = True
        return True
    lib = L.load()
    if not lib.phc_gemm_hs_enabled():
        _DONE[key] = False
        return False
    if torch.cuda.is_current_stream_capturing():
        return True
    _DONE[key] = True                          # (the check's own ops come back through require_device)
    worst = []
    try:
        with torch.no_grad():
            for K in (128, 256):
                worst += [(f"K={K} {name}", err) for name, err in _run(dev, K)]
        torch.cuda.synchronize(dev)
    except L.PhcError as exc:                  # a launch the device refused: the LDS-panel kernel takes over
        worst.append((f"raised {exc}", float("inf")))
    bad = [(n, e) for n, e in worst if not (e <= TOL)]
    if bad:
        lib.phc_gemm_hs_set_enabled(0)
        _DONE[key] = False
        warnings.warn("phc_gnn_amd: the H-stationary GEMM failed its load-time self check against the LDS-panel kernel ("
                      + ", ".join(f"{n}: {e:.2e}" for n, e in bad[:6]) + "); it is switched OFF for this process (every PHM layer "
                      "runs on phm_stream_kernel, as with PHC_GEMM_NO_HS=1).  Rebuild with the compiler the asm audit was validated "
                      "on (scripts/audit_hs_asm_loads.py).", RuntimeWarning)
    return _DONE[key]
