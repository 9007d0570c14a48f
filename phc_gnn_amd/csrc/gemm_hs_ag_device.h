// Aggregation INSIDE the first GEMM of a message-passing layer on a molecule batch (round 4):
//     s[d] = sum_{e: dst_e = d} (x[src_e] + enc_e) (+ x[d])        messagepassing.py:132-146 (PHMGINEConv.propagate, GINE sum)
//     z    = s . H (+ b),  column statistics of z                   layers.py:198-219, 349-355 (PHMMLP.linear1 + its batch-norm's sums)
// with, optionally, the model-level norm of the layer in FRONT still un-applied:  x := act(norm(y)) (+ skip)   (models.py:208-215).
// Separately these were `aggregate_norm_direct_fwd_kernel` / `aggregate_encp_direct_fwd_kernel` (encoder.hip) followed by the
// H-stationary GEMM with the statistics epilogue (gemm_hs_device.h, HS_ST): two launches whose second streams straight back in what
// the first wrote.  Here the workgroup that owns a 16-row step BUILDS the step's operand image itself: every wavefront gathers two
// rows (one per half-wavefront: the row-per-lane-group walk of the aggregation kernels, same edge order, same additions — the rows
// are bit-identical), writes them into the LDS image in the layout the MFMA fragment reads expect, and also out to memory once (s is
// the input of the layer's parameter-gradient GEMM; x, when a norm was pending, is what every later reader of the layer input sees).
// Then the 16 x 128 x 128 product on the operand slice the wavefront holds in registers, exactly as gemm_hs16_kernel does it
// (v_mfma_f32_16x16x4_f32, swapped roles, k-permutation k = 16 g + 4 kq + q), and that kernel's statistics epilogue.
//
// Register budget decides the shape: 805 steps at 12.9 k rows want 512 workgroups resident (two per CU, 16 wavefronts per CU:
// 128 registers per lane), and a lane group's walk with a norm pending keeps 6 row loads + the norm constants in flight.  So
//   * 8 wavefronts x ONE 16-column tile (the 4 x 2 split of the streaming kernel would have a wavefront gather four rows);
//   * the operand matrix (64 KB, k4 form) sits in LDS — brought in by LDS-DMA at entry, no register round trip — instead of in the
//     wavefronts' registers: the fragment reads of both MFMA operands are ds_read_b128 (the operand's are conflict-free in the
//     k4 layout as it stands: the 16 lanes of a kq group read 256 contiguous bytes);
//   * the bond tables (13 rows on molecule data) are staged in LDS too: as global reads inside the walk they were 2 F more
//     float4 loads in flight per lane;
//   * rows with more than 16 in-edges (the aggregation kernels hand those to the whole wavefront, the lane groups taking alternate
//     edges) are walked by their own lane group in two strided passes whose partial sums are added in the same order: the same
//     bits, without a second inlined copy of the walk.
// The only hand-counted wait is the vmcnt(0) behind the operand's DMA, in front of the first barrier.
#pragma once
#include "gemm_hs_device.h"

namespace phc {

constexpr int HS_AG_MAX_F = 4;
constexpr int HS_AG_MAX_ROWS = 64;                             // bond-table rows staged in LDS (32 KB)
struct HsAgArgs {
    const float *y;                    // rows that are gathered: x itself (acc == null) or the input of the un-applied norm
    const float *skip;                 // residual behind act(norm(.)), or null (only with acc)
    const unsigned *codes;             // packed bond codes, CSR order (one byte per feature: table row id incl. the feature's offset)
    const float *const *tables;        // [n_comp * F] table pointers
    const int *rowptr, *col;           // destination-sorted view
    float *s_out;                      // [M, 128] the aggregate
    float *x_out;                      // [M, 128] the normalised rows (acc != null)
    float *stat;                       // [6 * 128] save_stat of the pending norm (acc != null)
    const unsigned long long *acc;     // the pending norm's statistics accumulator, or null
    BnParams p;
    float momentum, eps;
    int norm_act;
    int F, off[HS_AG_MAX_F + 1];       // features per edge, row offset of feature f in the packed row space
    int n, k, add_self;
};

constexpr int HS_AG_THREADS = 512;
#ifndef PHC_AG_OCC
#define PHC_AG_OCC __attribute__((amdgpu_waves_per_eu(4, 4)))
#endif

template <int F, bool NORMED>
__global__ __launch_bounds__(HS_AG_THREADS) PHC_AG_OCC void gemm_hs_ag_kernel(GemmHsArgs a, HsAgArgs ag) {
    constexpr int K = 128, G = K / 16, ISTRIDE = HS_ISTRIDE, INSTR = 8, H4 = K / 4, NOUT = 128;
    __shared__ f32x4_t image[INSTR * ISTRIDE];                   // one 16-row operand image (8 two-row "instructions", padded)
    __shared__ __attribute__((aligned(16))) float s_scale[K], s_shift[K];
    extern __shared__ __attribute__((aligned(16))) float ag_dyn[];       // [K / 4][128] float4 operand (k4 form) | [R][128] bond tables
    const f32x4_t *opnd = reinterpret_cast<const f32x4_t *>(ag_dyn);
    float *ag_tbl = ag_dyn + (size_t)K * NOUT;                   // row = packed code byte, columns component-major
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, kq = lane >> 4;
    const int col0 = wave * 16;
    const int group = (int)blockIdx.x;
    const int num_steps = (int)((a.M + 15) / 16);
    const int my_steps = group < num_steps ? (num_steps - group + a.groups - 1) / a.groups : 0;
    if (my_steps == 0) return;
    const int M = (int)a.M;
    const int count = a.n_valid ? min(*a.n_valid, M) : M;
    auto lds_barrier = [] {                                       // LDS traffic only: __syncthreads() would also wait for the s / x
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // stores of the step (vmcnt(0))
        __builtin_amdgcn_s_barrier();
    };

    // ---- the operand matrix -> LDS by LDS-DMA: 64 instructions of 1 KiB, eight per wavefront, a straight copy of the k4 form ----
    {
        const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(hs_lds_void *)ag_dyn);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int ii = wave * 8 + u;
            hs_dma16(a.hk4 + (size_t)ii * 256 + 4 * lane, lds0 + 1024u * (unsigned)ii);
        }
    }
    // ---- this half-wavefront's row of a step: image instruction `wave`, row 2 wave + dsub; its 16-byte chunk (odd rows XOR 1) ----
    const int dsub = lane >> 5, li = lane & 31;
    const int c4 = li ^ dsub;
    auto row_of = [&](int sl) { return (group + sl * a.groups) * 16 + 2 * wave + dsub; };
    // row pointers of the first two steps go in flight before anything else (clamped rows)
    int d_[2], eb_[2], ee_[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        d_[q] = row_of(min(q, my_steps - 1));
        const int dc = d_[q] < M ? d_[q] : 0;
        eb_[q] = ag.rowptr[dc];
        ee_[q] = ag.rowptr[dc + 1];
    }
    // ---- bond tables -> LDS (one float4 per thread on molecule data; pointer hop + rows overlap the row-pointer chain) ----
    {
        const int R = ag.off[F];
        for (int e = tid; e < R * H4; e += HS_AG_THREADS) {
            const int r = e >> 5, q4 = e & 31;
            const int comp = (q4 * 4) / ag.k, kk = q4 * 4 - comp * ag.k;
            int f = 0;
#pragma unroll
            for (int t = 1; t < F; ++t) f = r >= ag.off[t] ? t : f;
            reinterpret_cast<float4 *>(ag_tbl)[e] = *reinterpret_cast<const float4 *>(ag.tables[comp * F + f] + (size_t)(r - ag.off[f]) * ag.k + kk);
        }
    }
    // ---- pending norm: scale / shift of every column (the arithmetic of aggregate_norm_direct_fwd_kernel's prologue) ----
    if constexpr (NORMED) {
        const bool owner = blockIdx.x == 0;
        const int h = K;
        for (int cc = tid; cc < h; cc += HS_AG_THREADS) {
            double S = 0.0, Q = 0.0;
#pragma unroll
            for (int rp = 0; rp < STAT_ACC_REP; ++rp) {
                S += stat_acc_part(ag.acc, h, rp, 0, cc);
                Q += stat_acc_part(ag.acc, h, rp, 1, cc);
            }
            const int comp = cc / ag.p.k, kk = cc - comp * ag.p.k;
            float w = 1.f, bq = 0.f, rm_old = 0.f, rv_old = 1.f;
            for (int q = 0; q < ag.p.n_comp; ++q)
                if (q == comp) {
                    if (ag.p.has_weight) w = ag.p.weight[q][kk];
                    if (ag.p.has_bias) bq = ag.p.bias[q][kk];
                    if (ag.p.has_running && owner) { rm_old = ag.p.running_mean[q][kk]; rv_old = ag.p.running_var[q][kk]; }
                }
            const BnFromTotals bt = bn_from_totals(S, Q, count, w, bq, ag.eps);
            s_scale[cc] = bt.scale;
            s_shift[cc] = bt.shift;
            if (owner) {
                float *stat = ag.stat;
                stat[cc] = bt.mean;
                stat[h + cc] = bt.invstd;
                stat[2 * h + cc] = bt.scale;
                stat[3 * h + cc] = bt.shift;
                stat[4 * h + cc] = w;
                stat[5 * h + cc] = bq;
                if (ag.p.has_running) {
                    for (int q = 0; q < ag.p.n_comp; ++q)
                        if (q == comp) {
                            ag.p.running_mean[q][kk] = bn_running_update(rm_old, ag.momentum, bt.mean);
                            ag.p.running_var[q][kk] = bn_running_update(rv_old, ag.momentum, bt.unbiased);
                        }
                }
            }
        }
        if (owner && ag.p.has_nbt && tid < ag.p.n_comp)
            for (int q = 0; q < ag.p.n_comp; ++q)
                if (q == tid) *ag.p.num_batches_tracked[q] += 1;
    }
    hs_wait_vm<0>();                                             // the operand's DMA (hipcc does not see it) has landed
    lds_barrier();                                               // operand, tables (and scale / shift) are in LDS
    const bool relu = ag.norm_act == PHC_ACT_RELU;
    const float4 *tb = reinterpret_cast<const float4 *>(ag_tbl) + c4;    // this lane's chunk of every table row
    const float *skp = ag.skip ? ag.skip : ag.y;                 // (both loads unconditional: a missing skip re-reads y and is not added)
    const bool has_skip = ag.skip != nullptr;
    // one row slice as the aggregation sees it: x, or act(y * scale + shift) (+ skip), 0 for a padding row of a normalised tensor
    auto xrow = [&](int r) {
        const float4 v = reinterpret_cast<const float4 *>(ag.y)[(size_t)r * H4 + c4];
        if constexpr (!NORMED) {
            return v;
        } else {
            const float4 rr = reinterpret_cast<const float4 *>(skp)[(size_t)r * H4 + c4];
            const float4 sc = *reinterpret_cast<const float4 *>(s_scale + 4 * c4), sh = *reinterpret_cast<const float4 *>(s_shift + 4 * c4);
            float4 o;
            if (relu) {
                o.x = fmaxf(fmaf(v.x, sc.x, sh.x), 0.f); o.y = fmaxf(fmaf(v.y, sc.y, sh.y), 0.f);
                o.z = fmaxf(fmaf(v.z, sc.z, sh.z), 0.f); o.w = fmaxf(fmaf(v.w, sc.w, sh.w), 0.f);
            } else {
                o.x = act_fwd_rt(ag.norm_act, fmaf(v.x, sc.x, sh.x)); o.y = act_fwd_rt(ag.norm_act, fmaf(v.y, sc.y, sh.y));
                o.z = act_fwd_rt(ag.norm_act, fmaf(v.z, sc.z, sh.z)); o.w = act_fwd_rt(ag.norm_act, fmaf(v.w, sc.w, sh.w));
            }
            o.x = has_skip ? o.x + rr.x : o.x; o.y = has_skip ? o.y + rr.y : o.y;
            o.z = has_skip ? o.z + rr.z : o.z; o.w = has_skip ? o.w + rr.w : o.w;
            if (r >= count) o = make_float4(0.f, 0.f, 0.f, 0.f);
            return o;
        }
    };
    auto message = [&](int s, unsigned code) {
        float4 v = xrow(s);
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const float4 t = tb[((code >> (8 * f)) & 255u) * H4];
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        return v;
    };
    // Build this wavefront's two rows of a step: LDS image + s (+ x) in memory.  The walk is the aggregation kernels' `row` body:
    // edges e0 + first, + stride, ... of [e0, e1), the first two in one round trip, partial sums acc0 / acc1.  A row with more
    // than 16 in-edges takes two passes (first = 0, 1; stride 2) whose results are added — what the two lane groups of a wavefront
    // compute and exchange there.
    auto fill = [&](int d, int eb_raw, int ee_raw) {
        const bool valid = d < M;
        const int e0 = valid ? eb_raw : 0, e1 = valid ? ee_raw : 0;
        const bool heavy = e1 - e0 > 16;                         // (AGG_HEAVY of encoder.hip)
        const int stride = heavy ? 2 : 1;
        const float4 selfv = xrow((NORMED || ag.add_self) && valid ? d : 0);
        float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int first = 0; first < stride; ++first) {
            float4 acc0 = make_float4(0.f, 0.f, 0.f, 0.f), acc1 = acc0;
            int e = e0 + first;
            if (e0 < e1) {                                       // the first two edges travel together (clamped indices)
                const int i0 = min(e, e1 - 1), i1 = min(e + stride, e1 - 1);
                const int s0 = ag.col[i0], s1 = ag.col[i1];
                const unsigned k0 = ag.codes[i0], k1 = ag.codes[i1];
                const float4 v0 = message(s0, k0), v1 = message(s1, k1);
                if (e < e1) { acc0.x += v0.x; acc0.y += v0.y; acc0.z += v0.z; acc0.w += v0.w; }
                if (e + stride < e1) { acc1.x += v1.x; acc1.y += v1.y; acc1.z += v1.z; acc1.w += v1.w; }
                e += 2 * stride;
            }
            for (; e < e1; e += stride) {
                const float4 v = message(ag.col[e], ag.codes[e]);
                acc0.x += v.x; acc0.y += v.y; acc0.z += v.z; acc0.w += v.w;
            }
            const float4 a4 = make_float4(acc0.x + acc1.x, acc0.y + acc1.y, acc0.z + acc1.z, acc0.w + acc1.w);
            // (one pass: tot = a4.  two passes: a4(first 0) + a4(first 1), the shuffle-add of the shared walk)
            tot = first == 0 ? a4 : make_float4(tot.x + a4.x, tot.y + a4.y, tot.z + a4.z, tot.w + a4.w);
        }
        if (ag.add_self) { tot.x += selfv.x; tot.y += selfv.y; tot.z += selfv.z; tot.w += selfv.w; }
        if (!valid) tot = make_float4(0.f, 0.f, 0.f, 0.f);
        image[wave * ISTRIDE + lane] = f32x4_t{tot.x, tot.y, tot.z, tot.w};
        if (valid) {
            reinterpret_cast<float4 *>(ag.s_out)[(size_t)d * H4 + c4] = tot;
            if constexpr (NORMED) reinterpret_cast<float4 *>(ag.x_out)[(size_t)d * H4 + c4] = selfv;
        }
    };

    const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4_t bias4 = zero4;
    if (a.bias) bias4 = *reinterpret_cast<const f32x4_t *>(a.bias + col0 + 4 * kq);
    const int fbase = (c >> 1) * ISTRIDE + (c & 1) * H4 + (kq ^ (c & 1));
    const f32x4_t *hfrag = opnd + kq * NOUT + col0 + c;          // operand fragment of k-group g: hfrag[4 g * NOUT]
    f32x4_t st_p = zero4, st_s1 = zero4, st_s2 = zero4;
    float st_n = 0.f;
    for (int i = 0; i < my_steps; ++i) {
        const int step = group + i * a.groups;
        {
            int d = i == 0 ? d_[0] : d_[1], eb = i == 0 ? eb_[0] : eb_[1], ee = i == 0 ? ee_[0] : ee_[1];
            if (i >= 2) {                                        // (third and later steps of a workgroup: launches past 512 x 32 rows)
                d = row_of(i);
                const int dc = d < M ? d : 0;
                eb = ag.rowptr[dc];
                ee = ag.rowptr[dc + 1];
            }
            fill(d, eb, ee);
        }
        lds_barrier();                                           // the step's image is complete
        f32x4_t acc = bias4;
        f32x4_t fr[2], hr[2];
        fr[0] = image[fbase];
        hr[0] = hfrag[0];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (g + 1 < G) {
                fr[(g + 1) & 1] = image[fbase + 4 * (g + 1)];
                hr[(g + 1) & 1] = hfrag[4 * (g + 1) * NOUT];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(hr[g & 1][q], fr[g & 1][q], acc, 0, 0, 0);
        }
        const int row = step * 16 + c;
        if (row < count) {
            if (st_n == 0.f) {
                st_p = acc;
            } else {
                const f32x4_t dlt = acc - st_p;
                st_s1 += dlt;
                st_s2 += dlt * dlt;
            }
            st_n += 1.f;
        }
        if (row < M) *reinterpret_cast<f32x4_t *>(a.y + (size_t)row * NOUT + col0 + 4 * kq) = acc;
        if (i + 1 < my_steps) lds_barrier();                     // every wavefront has read the image
    }
    // ---- statistics of the output columns (gemm_hs16_kernel's HS_ST finale) ----
    {
        double S[1][4], Q[1][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double n = (double)st_n, p = (double)st_p[q], s1 = (double)st_s1[q], s2 = (double)st_s2[q];
            S[0][q] = hs_sum16(n * p + s1);
            Q[0][q] = hs_sum16(n * p * p + 2.0 * p * s1 + s2);
        }
        hs_publish<1>(a.stat_acc, NOUT, (int)(group % STAT_ACC_REP), col0, kq, c, S, Q);
    }
}

}  // namespace phc
