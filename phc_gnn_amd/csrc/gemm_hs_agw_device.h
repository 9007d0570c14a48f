// Aggregation INSIDE the first GEMM of a message-passing layer for the WIDE layers (round 5): hidden 256 (BASELINE configs[2], [3])
// and the reference's published widths 180 / 200 on the 192 / 208 register geometry — what gemm_hs_ag_device.h does at hidden 128:
//     s[d] = sum_{e: dst_e = d} (x[src_e] + enc_e) (+ x[d])        messagepassing.py:132-146 (PHMGINEConv.propagate, GINE sum)
//     z    = s . H (+ b),  column statistics of z                   layers.py:198-219, 349-355 (PHMMLP.linear1 + its batch-norm's sums)
// optionally with the model-level norm of the layer in FRONT still un-applied:  x := act(norm(y)) (+ skip)   (models.py:208-215).
//
// What differs from the 128-wide kernel, and why:
//   * the operand matrix does not fit the LDS (256 x 256 x 4 B): it lives in the wavefronts' REGISTERS exactly as in
//     gemm_hs16_kernel<K, 2, 8> (8 wavefronts x 2 column tiles, K / 16 float4 per tile and lane: 128 registers at K = 256), one
//     workgroup per CU, 256 registers per lane;
//   * a row is 720 B - 1 KiB: ONE WAVEFRONT gathers a row (lane = 16-byte chunk; lanes past the row's end idle), so everything about
//     a row — its edge range, its sources, its packed codes — is wave-uniform and travels in SGPRs; a wavefront owns rows 2 w and
//     2 w + 1 of every 16-row step;
//   * a workgroup runs 3 - 7 steps (23.6 k rows / 256 CUs = 5.8) and a gathered row takes 1 - 2 us to arrive under load: the
//     gather of step i + 1 must be in flight under the MFMAs of step i.  Two LDS images alternate; per wavefront the dependent chain
//     row pointers -> sources / codes -> rows is a rolling three-stage pipeline of ordinary loads (metadata of steps i + 2 .. i + 4
//     in flight on a few lanes, decoded into SGPRs by v_readlane).  FULL (no norm pending: 20 registers of pieces per row): both
//     rows' pieces of step i + 1 travel a whole step ahead.  With a norm pending every gathered row is two loads (y and the skip: 32
//     registers per row at NE = 3) and only ONE row's pieces fit beside the 128 operand registers: row A travels under the second
//     half of the previous step's MFMAs, row B under the first half of this step's.  (The last k-groups of the operand in LDS
//     instead — 64 - 96 KB, 32 - 48 registers back — was measured: the fragment reads make the MFMA phase 3.8 instead of 2.0 us.)
// Same edge order, same additions as aggregate_encp_direct_fwd_kernel<64, 0, F> / aggregate_norm_direct_fwd_kernel<64, F> (pair
// first: acc0 = m0 + m2 + m3 + ..., acc1 = m1, s = (acc0 + acc1) + self), same MFMA order, row ownership and statistics epilogue
// as gemm_hs16_kernel<K, 2, 8, 1, R, HS_ST> with the same number of workgroups: s, x, save_stat, running statistics, z and the
// integer statistics totals are bit-identical to the separate launches.
#pragma once
#include "gemm_hs_ag_device.h"

namespace phc {

constexpr int HS_AGW_THREADS = 512;

template <int K, bool NORMED, bool RELU, int NE, bool FULL>
__global__ __launch_bounds__(HS_AGW_THREADS) void gemm_hs_agw_kernel(GemmHsArgs a, HsAgArgs ag) {
    static_assert(K > 128 && K <= 256 && K % 16 == 0, "one row per 1-KiB image instruction");
    static_assert(NE >= 2 && NE <= 4, "edges of a row fetched ahead");
    constexpr int G = K / 16, TPW = 2, ISTRIDE = HS_ISTRIDE, IMG_F4 = 16 * ISTRIDE;
    __shared__ f32x4_t image[2][IMG_F4];                          // two 16-row operand images (one row per padded 1-KiB line)
    __shared__ __attribute__((aligned(16))) float s_scale[K], s_shift[K], s_bias[K];   // (the bias lives in LDS: 8 registers fewer)
    extern __shared__ __attribute__((aligned(16))) float agw_dyn[];   // [R][KR] bond tables: row = packed code byte, columns as in x
    const int KR = a.K, H4 = KR >> 2, Nout = a.Nout;
    const int F = ag.F;                                           // (a run-time bound, 1..4: one instantiation per geometry and norm form)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, kq = lane >> 4;
    const int col0 = wave * 16 * TPW;
    const int group = (int)blockIdx.x;
    const int M = (int)a.M;
    const int num_steps = (M + 15) / 16;
    // (a contiguous range of steps per XCD instead of group + i * groups — neighbouring rows' gathers meeting in one L2 — measured:
    // 50.36 vs 50.47 us per launch at 24 k rows x 256; the mapping stays the statistics GEMM's, whose totals this launch reproduces)
    const int gfirst = group, gstride = a.groups;
    const int my_steps = group < num_steps ? (num_steps - group + a.groups - 1) / a.groups : 0;
    if (my_steps == 0) return;
    // diagnostics (phc_debug_hs_stamps; null in the product): 100 MHz ticks at entry | loop start | loop done | end, and per phase
    // summed over the steps: image barrier | MFMAs first half | finish row A | MFMAs second half | finish row B + metadata |
    // epilogue + stores + next rows' issue   (scripts/dbg/agw_phases.py)
    unsigned long long tk[4] = {0, 0, 0, 0}, ph[6] = {0, 0, 0, 0, 0, 0}, t_chain = 0;
    if (a.stamps) tk[0] = __builtin_amdgcn_s_memrealtime();
    const int count = a.n_valid ? min(*a.n_valid, M) : M;
    // Steps that hold padding rows ONLY (every row >= *n_valid: the tail of a fixed-shape batch; the LAST steps of a workgroup) take no
    // gather and no matrix work (gemm_hs16_kernel's XF / BA launches do the same): their rows are written as s = 0, x = 0, z = bias —
    // finite values nobody weighs (every consumer masks rows >= *n_valid or multiplies them with a zero gradient).  Whether a
    // bucket's padded row count is a whole number of rounds of this launch then no longer decides whether it pays a nearly empty
    // extra pass (24 832 rows: seven passes against six).  The workgroup's first step is always computed.
    const int live_steps = (count + 15) / 16;
    const int my_live = max(1, min(my_steps, group < live_steps ? (live_steps - group + a.groups - 1) / a.groups : 0));
    const bool live = lane < H4;                                  // this lane's 16-byte chunk exists in a row
    const int c4 = live ? lane : 0;
    auto lds_barrier = []() PHC_HS_INLINE {                                       // LDS traffic only: __syncthreads() would also wait for stores / prefetches
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    const float4 *Y4 = reinterpret_cast<const float4 *>(ag.y) + c4;
    const float4 *S4 = reinterpret_cast<const float4 *>(ag.skip ? ag.skip : ag.y) + c4;
    const bool has_skip = ag.skip != nullptr;
    const float4 zf4 = make_float4(0.f, 0.f, 0.f, 0.f);

    // ---- per-step metadata of this wavefront's two rows (A = 2 w, B = 2 w + 1 of the step), as vector loads on a few lanes -------
    auto rowA_of = [&](int sl) PHC_HS_INLINE { return (gfirst + sl * gstride) * 16 + 2 * wave; };
    // row pointers of rows A, A + 1, A + 2 (lanes 0..2; clamped to the table's last entry: a row past M is empty)
    auto load_ptrs = [&](int sl) PHC_HS_INLINE {
        const int r = min(rowA_of(min(sl, my_steps - 1)) + min(lane, 2), M);
        return ag.rowptr[r];
    };
    struct Meta {
        int e0[2], e1[2];
    };
    auto ptrs_of = [&](int v) PHC_HS_INLINE {
        Meta m;
        m.e0[0] = __builtin_amdgcn_readlane(v, 0);
        m.e1[0] = m.e0[1] = __builtin_amdgcn_readlane(v, 1);
        m.e1[1] = __builtin_amdgcn_readlane(v, 2);
        return m;
    };
    // sources and codes of the first NE edges of both rows: lanes 0..NE-1 row A, lanes 8..8+NE-1 row B (an empty row reads row pointers)
    auto load_edges = [&](const Meta &m, int &srcs, unsigned &cods) PHC_HS_INLINE {
        const int rsel = (lane >> 3) & 1, j = lane & 7;
        const int e0 = rsel ? m.e0[1] : m.e0[0], e1 = rsel ? m.e1[1] : m.e1[0];
        const bool any = e0 < e1;
        const int idx = any ? min(e0 + min(j, NE - 1), e1 - 1) : 0;
        srcs = (any ? ag.col : ag.rowptr)[idx];
        cods = (any ? ag.codes : reinterpret_cast<const unsigned *>(ag.rowptr))[idx];
    };

    float *agw_tbl = agw_dyn;
    // ---- one row's gathered pieces ----
    struct Raw {
        float4 v[NE + 1], rr[NORMED ? NE + 1 : 1];               // edges 0..NE-1, then the row itself
    };
    auto load_row = [&](int r, float4 &v, float4 &rr) PHC_HS_INLINE {
        v = Y4[(size_t)r * H4];
        if constexpr (NORMED) rr = S4[(size_t)r * H4];
    };
    auto issue_raw = [&](Raw &w, int d, int e0, int e1, int srcs, int which) PHC_HS_INLINE {
        const int deg = e1 - e0;
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            w.v[j] = zf4;
            if constexpr (NORMED) w.rr[j] = zf4;
            if (deg > j) load_row(__builtin_amdgcn_readlane(srcs, 8 * which + j), w.v[j], w.rr[NORMED ? j : 0]);
        }
        w.v[NE] = zf4;
        if constexpr (NORMED) w.rr[NE] = zf4;
        if ((NORMED || ag.add_self) && d < M) load_row(d, w.v[NE], w.rr[NORMED ? NE : 0]);
    };
    const float4 *tb = reinterpret_cast<const float4 *>(agw_tbl) + c4;   // this lane's chunk of every table row
    auto xval = [&](const float4 v, const float4 rr, int r) PHC_HS_INLINE {
        if constexpr (!NORMED) {
            return v;
        } else {
            // (scale / shift of this lane's four columns are re-read from LDS at every use: 8 registers the K = 256 geometry lacks)
            const float4 sc4 = *reinterpret_cast<const float4 *>(s_scale + 4 * c4), sh4 = *reinterpret_cast<const float4 *>(s_shift + 4 * c4);
            float4 o;
            if constexpr (RELU) {
                o.x = fmaxf(fmaf(v.x, sc4.x, sh4.x), 0.f); o.y = fmaxf(fmaf(v.y, sc4.y, sh4.y), 0.f);
                o.z = fmaxf(fmaf(v.z, sc4.z, sh4.z), 0.f); o.w = fmaxf(fmaf(v.w, sc4.w, sh4.w), 0.f);
            } else {
                o.x = act_fwd_rt(ag.norm_act, fmaf(v.x, sc4.x, sh4.x)); o.y = act_fwd_rt(ag.norm_act, fmaf(v.y, sc4.y, sh4.y));
                o.z = act_fwd_rt(ag.norm_act, fmaf(v.z, sc4.z, sh4.z)); o.w = act_fwd_rt(ag.norm_act, fmaf(v.w, sc4.w, sh4.w));
            }
            o.x = has_skip ? o.x + rr.x : o.x; o.y = has_skip ? o.y + rr.y : o.y;
            o.z = has_skip ? o.z + rr.z : o.z; o.w = has_skip ? o.w + rr.w : o.w;
            if (r >= count) o = zf4;
            return o;
        }
    };
    auto message = [&](const float4 v0, const float4 rr, int s, unsigned code) PHC_HS_INLINE {
        float4 v = xval(v0, rr, s);
#pragma unroll
        for (int f = 0; f < HS_AG_MAX_F; ++f) {
            if (f < F) {
                const float4 t = tb[((code >> (8 * f)) & 255u) * H4];
                v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
            }
        }
        return v;
    };
    // add one row up from its pieces -> the image line of row `rs` (0..15) of buffer `buf`; returns s and x of the row for the stores
    auto finish = [&](const Raw &w, int d, int e0, int e1, int srcs, unsigned cods, int which, int buf, int rs) PHC_HS_INLINE {
        const bool valid = d < M;
        const int deg = e1 - e0;
        const float4 selfv = xval(w.v[NE], w.rr[NORMED ? NE : 0], valid ? d : 0);
        // (x of the row goes out at once: older than every load issued from here on, and 8 registers fewer across the rotation)
        if constexpr (NORMED) {
            if (valid && live) reinterpret_cast<float4 *>(ag.x_out)[(size_t)d * H4 + c4] = selfv;
        }
        float4 acc0 = zf4, acc1 = zf4;
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            if (deg > j) {
                const float4 m = message(w.v[j], w.rr[NORMED ? j : 0], __builtin_amdgcn_readlane(srcs, 8 * which + j),
                                         (unsigned)__builtin_amdgcn_readlane((int)cods, 8 * which + j));
                if (j == 1) { acc1.x += m.x; acc1.y += m.y; acc1.z += m.z; acc1.w += m.w; }
                else { acc0.x += m.x; acc0.y += m.y; acc0.z += m.z; acc0.w += m.w; }
            }
        }
        for (int e = e0 + NE; e < e1; ++e) {                      // (past the window: one round trip each; rare on molecule graphs)
            const int t0 = ag.col[e];
            const unsigned c0 = ag.codes[e];
            float4 v, rr = zf4;
            load_row(t0, v, rr);
            const float4 m = message(v, rr, t0, c0);
            acc0.x += m.x; acc0.y += m.y; acc0.z += m.z; acc0.w += m.w;
        }
        float4 tot = make_float4(acc0.x + acc1.x, acc0.y + acc1.y, acc0.z + acc1.z, acc0.w + acc1.w);
        if (ag.add_self) { tot.x += selfv.x; tot.y += selfv.y; tot.z += selfv.z; tot.w += selfv.w; }
        if (!valid || !live) tot = zf4;
        image[buf][rs * ISTRIDE + lane] = f32x4_t{tot.x, tot.y, tot.z, tot.w};
    };
    // (s of the row is re-read from the image line it was written to: 8 registers fewer across the metadata rotation)
    auto store_row = [&](int d, int buf, int rs) PHC_HS_INLINE {
        if (d < M && live) {
            const f32x4_t t4 = image[buf][rs * ISTRIDE + lane];
            reinterpret_cast<float4 *>(ag.s_out)[(size_t)d * H4 + c4] = make_float4(t4[0], t4[1], t4[2], t4[3]);
        }
    };

    // ================= prologue =================
    // the dependent chain of the first steps goes out first; everything else of the prologue travels behind it
    int pv0 = load_ptrs(0), pv1 = load_ptrs(1), pv2 = load_ptrs(2), pv3 = load_ptrs(3);
    // pending norm: (column, replica pair) per thread; the four replicas' parts meet in replica order ((p0 + p1) + p2) + p3 — the
    // order of bn_apply_acc_kernel / aggregate_norm_direct_fwd_kernel: the same bits
    const int ncol = tid >> 1, nhalf = tid & 1;
    const bool ncol_ok = NORMED && ncol < KR;
    const int ncomp = NORMED ? min(ncol, KR - 1) / ag.p.k : 0, nkk = NORMED ? min(ncol, KR - 1) - ncomp * ag.p.k : 0;
    const bool owner = blockIdx.x == 0;
    unsigned long long araw[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};    // [replica of this half][sum hi | sum lo | squares hi | squares lo]
    float nw = 1.f, nb = 0.f, rm_old = 0.f, rv_old = 1.f;
    Meta m0 = ptrs_of(pv0), m1 = ptrs_of(pv1), m2 = ptrs_of(pv2);
    int srcs0, srcs1, srcs2;
    unsigned cods0, cods1, cods2;
    load_edges(m0, srcs0, cods0);
    load_edges(m1, srcs1, cods1);
    load_edges(m2, srcs2, cods2);
    if constexpr (NORMED) {
        if (ncol_ok) {
#pragma unroll
            for (int r2 = 0; r2 < 2; ++r2) {
                const unsigned long long *ap = ag.acc + (size_t)((2 * nhalf + r2) * 2) * 2 * KR + ncol;   // (layout of stat_acc_part)
                araw[r2][0] = ap[0]; araw[r2][1] = ap[KR]; araw[r2][2] = ap[2 * KR]; araw[r2][3] = ap[3 * KR];
            }
            if (nhalf == 0) {
                if (ag.p.has_weight) nw = *bn_col_ptr(ag.p.weight, ag.p.flat, ncomp, nkk, ncol);
                if (ag.p.has_bias) nb = *bn_col_ptr(ag.p.bias, ag.p.flat, ncomp, nkk, ncol);
                if (ag.p.has_running && owner) {
                    rm_old = *bn_col_ptr(ag.p.running_mean, ag.p.flat, ncomp, nkk, ncol);
                    rv_old = *bn_col_ptr(ag.p.running_var, ag.p.flat, ncomp, nkk, ncol);
                }
            }
        }
    }
    // the rows of step 0 go in flight in front of everything bulky (FULL: both of this wavefront's rows; otherwise row A)
    Raw rawA, rawB;                                               // (rawB: FULL only — without it one set of pieces is live at a time)
    int dA = rowA_of(0), dB = dA + 1;
    issue_raw(rawA, dA, m0.e0[0], m0.e1[0], srcs0, 0);
    if constexpr (FULL) issue_raw(rawB, dB, m0.e0[1], m0.e1[1], srcs0, 1);
    // bond tables -> LDS (R x H4 float4)
    int tblR = ag.off[1];
#pragma unroll
    for (int t = 2; t <= HS_AG_MAX_F; ++t) tblR = t <= F ? ag.off[t] : tblR;
    for (int e = tid; e < tblR * H4; e += HS_AGW_THREADS) {
        const int r = e / H4, q4 = e - r * H4;
        const int comp = (q4 * 4) / ag.k, kk = q4 * 4 - comp * ag.k;
        int f = 0, fo = 0;
#pragma unroll
        for (int t = 1; t < HS_AG_MAX_F; ++t) {
            const bool past = t < F && r >= ag.off[t];
            f = past ? t : f;
            fo = past ? ag.off[t] : fo;
        }
        reinterpret_cast<float4 *>(agw_tbl)[e] = *reinterpret_cast<const float4 *>(ag.tables[comp * F + f] + (size_t)(r - fo) * ag.k + kk);
    }
    if constexpr (NORMED) {
        double pS[2], pQ[2];
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) {
            pS[r2] = (double)(long long)araw[r2][0] * (1.0 / 4096.0) + (double)(long long)araw[r2][1] * (1.0 / 4611686018427387904.0);
            pQ[r2] = (double)(long long)araw[r2][2] * (1.0 / 4096.0) + (double)(long long)araw[r2][3] * (1.0 / 4611686018427387904.0);
        }
        // even lane: replicas 0, 1; odd lane: replicas 2, 3 -> ((p0 + p1) + p2) + p3 on the even lane
        const double oS0 = __shfl(pS[0], lane | 1, 64), oS1 = __shfl(pS[1], lane | 1, 64);
        const double oQ0 = __shfl(pQ[0], lane | 1, 64), oQ1 = __shfl(pQ[1], lane | 1, 64);
        if (nhalf == 0 && ncol_ok) {
            const double S = ((pS[0] + pS[1]) + oS0) + oS1, Q = ((pQ[0] + pQ[1]) + oQ0) + oQ1;
            const BnFromTotals bt = bn_from_totals(S, Q, count, nw, nb, ag.eps);
            s_scale[ncol] = bt.scale;
            s_shift[ncol] = bt.shift;
            if (owner) {
                float *stat = ag.stat;
                stat[ncol] = bt.mean;
                stat[KR + ncol] = bt.invstd;
                stat[2 * KR + ncol] = bt.scale;
                stat[3 * KR + ncol] = bt.shift;
                stat[4 * KR + ncol] = nw;
                stat[5 * KR + ncol] = nb;
                if (ag.p.has_running) {
                    *bn_col_ptr(ag.p.running_mean, ag.p.flat, ncomp, nkk, ncol) = bn_running_update(rm_old, ag.momentum, bt.mean);
                    *bn_col_ptr(ag.p.running_var, ag.p.flat, ncomp, nkk, ncol) = bn_running_update(rv_old, ag.momentum, bt.unbiased);
                }
            }
        }
        if (owner && ag.p.has_nbt && tid < ag.p.n_comp)
            for (int q = 0; q < ag.p.n_comp; ++q)
                if (q == tid) *ag.p.num_batches_tracked[q] += 1;
    }
    const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
    if (tid < K) s_bias[tid] = (a.bias && tid < Nout) ? a.bias[tid] : 0.f;
    // ---- the operand -> registers (gemm_hs16_kernel's layout and order) ----
    f32x4_t hreg[TPW][G];
    {
        const f32x4_t *src = reinterpret_cast<const f32x4_t *>(a.hk4);
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const bool ok = 4 * (4 * g + kq) < KR && col0 + c + 16 * t < Nout;
                const f32x4_t v = src[ok ? (size_t)(4 * g + kq) * Nout + col0 + c + 16 * t : 0];
                hreg[t][g] = ok ? v : zero4;
            }
    }
    lds_barrier();                                                // tables, bias (and scale / shift) are in LDS
    if (a.stamps) t_chain = __builtin_amdgcn_s_memrealtime();
    finish(rawA, dA, m0.e0[0], m0.e1[0], srcs0, cods0, 0, 0, 2 * wave);
    if constexpr (FULL) {
        finish(rawB, dB, m0.e0[1], m0.e1[1], srcs0, cods0, 1, 0, 2 * wave + 1);
    } else {
        issue_raw(rawA, dB, m0.e0[1], m0.e1[1], srcs0, 1);
        finish(rawA, dB, m0.e0[1], m0.e1[1], srcs0, cods0, 1, 0, 2 * wave + 1);
    }
    // pipeline state at the top of iteration i: rows of step i + 1 in flight (FULL: both, otherwise row A); (mn, srcs_n, cods_n) =
    // step i + 1, arrived; (m2, srcs2, cods2) = step i + 2, sources / codes in flight; pv3 = row pointers of step i + 3 in flight
    Meta mn = m1;
    int srcs_n = srcs1;
    unsigned cods_n = cods1;
    if (my_live > 1) {
        issue_raw(rawA, rowA_of(1), mn.e0[0], mn.e1[0], srcs_n, 0);
        if constexpr (FULL) issue_raw(rawB, rowA_of(1) + 1, mn.e0[1], mn.e1[1], srcs_n, 1);
    }
    store_row(dA, 0, 2 * wave);
    store_row(dB, 0, 2 * wave + 1);

    const int fbase = c * ISTRIDE + kq;                           // fragment base of this lane: row c, chunk 4 g + kq of k-group g
    f32x4_t st_p[TPW], st_s1[TPW], st_s2[TPW];
    float st_n = 0.f;
#pragma unroll
    for (int t = 0; t < TPW; ++t) st_p[t] = st_s1[t] = st_s2[t] = zero4;
    // The operand slice must be COMPLETE in hipcc's own bookkeeping before the loop: a load it still counts as pending at the loop
    // header is waited for with vmcnt(0) at its first use in EVERY iteration — in front of the MFMAs, where that drain also waits
    // for the rows issued a moment ago (seen in the generated code and in the clocks: the gather latency stood in front of the
    // matrix work instead of under it).  An empty asm that reads the registers places that wait here, once (gemm_hs16_kernel: settle).
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int g = 0; g < G; g += 4)
            asm volatile("" ::"v"(hreg[t][g]), "v"(hreg[t][g + 1]), "v"(hreg[t][g + 2]), "v"(hreg[t][g + 3]));
    if (a.stamps) tk[1] = __builtin_amdgcn_s_memrealtime();
    // ================= steps =================
    for (int i = 0; i < my_live; ++i) {
        const int step = gfirst + i * gstride;
        const int buf = i & 1;
        const bool more = i + 1 < my_live;
        unsigned long long p0 = 0, p1 = 0, p5 = 0;
        if (a.stamps) p0 = __builtin_amdgcn_s_memrealtime();
        lds_barrier();                                            // image[buf] complete; image[buf ^ 1] free (read by step i - 1)
        if (a.stamps) p1 = __builtin_amdgcn_s_memrealtime();
        const f32x4_t *sl = image[buf];
        f32x4_t acc[TPW];
#pragma unroll
        for (int t = 0; t < TPW; ++t) acc[t] = *reinterpret_cast<const f32x4_t *>(s_bias + col0 + 16 * t + 4 * kq);
        f32x4_t fr[2];
        auto mfma_range = [&](auto g0c, auto g1c) PHC_HS_INLINE {
            constexpr int g0 = decltype(g0c)::value, g1 = decltype(g1c)::value;
            unsigned long long q0 = 0;
            if (a.stamps) q0 = __builtin_amdgcn_s_memrealtime();
            fr[g0 & 1] = sl[fbase + 4 * g0];
#pragma unroll
            for (int g = g0; g < g1; ++g) {
                if (g + 1 < g1) fr[(g + 1) & 1] = sl[fbase + 4 * (g + 1)];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int t = 0; t < TPW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(hreg[t][g][q], fr[g & 1][q], acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (a.stamps) { asm volatile("s_nop 0" ::"v"(acc[0])); ph[1] += __builtin_amdgcn_s_memrealtime() - q0; }
        };
        const int nA = rowA_of(min(i + 1, my_steps - 1)), nB = nA + 1;
        // rotate: step i + 2's sources / codes (issued an iteration ago) become "next"; step i + 3's row pointers (ditto) are decoded
        // and its sources / codes go out; step i + 4's row pointers go out
        auto rotate = [&]() PHC_HS_INLINE {
            mn = m2; srcs_n = srcs2; cods_n = cods2;
            m2 = ptrs_of(pv3);
            load_edges(m2, srcs2, cods2);
            pv3 = load_ptrs(i + 4);
        };
        // the gather work of step i + 1.  FULL: both rows at once, then the rows of step i + 2 go out.  Otherwise in two halves around
        // the MFMAs: (row A, issue row B) and (row B, rotate, issue row A of step i + 2).
        auto gather_a = [&]() PHC_HS_INLINE {
            if (!more) return;
            unsigned long long q0 = 0;
            if (a.stamps) q0 = __builtin_amdgcn_s_memrealtime();
            finish(rawA, nA, mn.e0[0], mn.e1[0], srcs_n, cods_n, 0, buf ^ 1, 2 * wave);
            if constexpr (!FULL) issue_raw(rawA, nB, mn.e0[1], mn.e1[1], srcs_n, 1);
            if (a.stamps) ph[2] += __builtin_amdgcn_s_memrealtime() - q0;
        };
        auto gather_b = [&]() PHC_HS_INLINE {
            if (!more) return;
            unsigned long long q0 = 0;
            if (a.stamps) q0 = __builtin_amdgcn_s_memrealtime();
            if constexpr (FULL) finish(rawB, nB, mn.e0[1], mn.e1[1], srcs_n, cods_n, 1, buf ^ 1, 2 * wave + 1);
            else finish(rawA, nB, mn.e0[1], mn.e1[1], srcs_n, cods_n, 1, buf ^ 1, 2 * wave + 1);
            if (a.stamps) { ph[3] += __builtin_amdgcn_s_memrealtime() - q0; q0 = __builtin_amdgcn_s_memrealtime(); }
            rotate();
            if (i + 2 < my_live) {
                issue_raw(rawA, rowA_of(i + 2), mn.e0[0], mn.e1[0], srcs_n, 0);
                if constexpr (FULL) issue_raw(rawB, rowA_of(i + 2) + 1, mn.e0[1], mn.e1[1], srcs_n, 1);
            }
            if (a.stamps) ph[4] += __builtin_amdgcn_s_memrealtime() - q0;
        };
        using I0 = std::integral_constant<int, 0>;
        using IH = std::integral_constant<int, G / 2>;
        using IG = std::integral_constant<int, G>;
        // (Wavefronts w and w + 4 share a SIMD.  Running them in ANTI-PHASE — 4 - 7 gathering under the MFMAs of 0 - 3 and multiplying
        // afterwards — was measured (scripts/dbg/agw_phases.py, 24 k rows x 180): the gather's VALU work takes 1.5 - 1.6 us beside a
        // partner that issues MFMAs against 0.3 us beside an idle one, the MFMAs then run uncontended at 1.45 instead of 2.9 us, and
        // the step takes what it took (28.7 vs 29.3 us per launch): per SIMD the two wavefronts' matrix and vector work ADD UP
        // whatever the order.  What shortens the step is fewer vector instructions in the gather, not their placement.)
        if constexpr (FULL) {
            mfma_range(I0{}, IG{});
            gather_a();
            gather_b();
        } else {
            mfma_range(I0{}, IH{});
            gather_a();
            mfma_range(IH{}, IG{});
            gather_b();
        }
        if (a.stamps) p5 = __builtin_amdgcn_s_memrealtime();
        // ---- epilogue of step i: lane (c, kq) holds z[row][col0 + 16 t + 4 kq .. +3]; stores LAST in the iteration ----
        const int row = step * 16 + c;
        if (row < count) {
            if (st_n == 0.f) {
#pragma unroll
                for (int t = 0; t < TPW; ++t) st_p[t] = acc[t];
            } else {
#pragma unroll
                for (int t = 0; t < TPW; ++t) {
                    const f32x4_t dlt = acc[t] - st_p[t];
                    st_s1[t] += dlt;
                    st_s2[t] += dlt * dlt;
                }
            }
            st_n += 1.f;
        }
        if (row < M) {
            float *yr = a.y + (size_t)row * Nout + col0 + 4 * kq;
#pragma unroll
            for (int t = 0; t < TPW; ++t)
                if (col0 + 4 * kq + 16 * t < Nout) *reinterpret_cast<f32x4_t *>(yr + 16 * t) = acc[t];
        }
        if (more) {
            store_row(nA, buf ^ 1, 2 * wave);
            store_row(nB, buf ^ 1, 2 * wave + 1);
        }
        if (a.stamps) {
            const unsigned long long p6 = __builtin_amdgcn_s_memrealtime();
            ph[0] += p1 - p0; ph[5] += p6 - p5;
        }
    }
    // padding-only steps: s = 0 (x = 0), z = bias
    for (int i = my_live; i < my_steps; ++i) {
        const int step = gfirst + i * gstride;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int d = step * 16 + 2 * wave + u;
            if (d < M && live) {
                reinterpret_cast<float4 *>(ag.s_out)[(size_t)d * H4 + c4] = zf4;
                if constexpr (NORMED) reinterpret_cast<float4 *>(ag.x_out)[(size_t)d * H4 + c4] = zf4;
            }
        }
        const int row = step * 16 + c;
        if (row < M) {
            float *yr = a.y + (size_t)row * Nout + col0 + 4 * kq;
#pragma unroll
            for (int t = 0; t < TPW; ++t)
                if (col0 + 4 * kq + 16 * t < Nout) *reinterpret_cast<f32x4_t *>(yr + 16 * t) = *reinterpret_cast<const f32x4_t *>(s_bias + col0 + 16 * t + 4 * kq);
        }
    }
    if (a.stamps) tk[2] = __builtin_amdgcn_s_memrealtime();
    // ---- statistics of the output columns (gemm_hs16_kernel's HS_ST finale) ----
    {
        double S[TPW][4], Q[TPW][4];
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double n = (double)st_n, p = (double)st_p[t][q], s1 = (double)st_s1[t][q], s2 = (double)st_s2[t][q];
                S[t][q] = hs_sum16(n * p + s1);
                Q[t][q] = hs_sum16(n * p * p + 2.0 * p * s1 + s2);
            }
        hs_publish<TPW>(a.stat_acc, Nout, (int)(group % STAT_ACC_REP), col0, kq, c, S, Q);
    }
    if (a.stamps) {
        tk[3] = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) {
            unsigned long long *o = a.stamps + ((size_t)blockIdx.x * 8 + wave) * 12;
            o[0] = tk[0]; o[1] = tk[1]; o[2] = tk[2]; o[3] = tk[3];
#pragma unroll
            for (int q = 0; q < 6; ++q) o[4 + q] = ph[q];
            o[10] = (unsigned long long)my_steps;
            o[11] = t_chain;
        }
    }
}

}  // namespace phc
