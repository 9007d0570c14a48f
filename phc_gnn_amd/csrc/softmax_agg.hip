// Softmax neighbour aggregation — the reference's default / paper-best molhiv setting.
//
// Reference: PHMConvSoftmax / PHMGINEConvSoftmax.aggregate
// (phc/hypercomplex/undirectional/messagepassing.py:211-214, 297-300):
//     w   = scatter_softmax(m * beta, index)          per destination AND per feature column
//     out = scatter_sum(m * w, index)                 m = msg_encoder(x_j + edge_attr)
// with torch-scatter 2.0.5's scatter_softmax (max-shifted, eps = 1e-12 added to the segment sum).
// The reference runs 2 gathers + 6 scatter/elementwise passes over [E, h]; here one wavefront walks one
// destination segment with an online (running max / running sum) softmax per column, all in registers.
// Backward walks the source-sorted structure and rebuilds the weights from the saved log-sum-exp:
//     w_e = exp(beta*m_e - lse[d]);  d out/d m_e = w_e * (1 + beta*(m_e - out[d]));  d out/d beta = sum_e w_e m_e (m_e - out[d])
#include <string.h>
#include "phc_common.h"

namespace phc {

constexpr int SM_THREADS = 256;
// (SM_EPS, sm_update: phc_common.h — the aggregating one-launch MLP forward of mlp_fused.hip runs the same online softmax)

// Bond embedding read in place (SURVEY §8f N1 on the softmax operators): instead of an [E, h] edge tensor the kernels
// take the packed table rows of every edge (phc_pack_edge_codes: byte f = off_f + value) and the n*F table pointers,
// and add the F table rows to x[src] themselves (rows come from L2: a bond encoder has 13 rows).
struct SmResiduals {                         // further gradients of the rows a backward walk writes (softmax_agg_bwd_kernel)
    const float *p[8];
    int n;
};
struct SmEnc {
    const unsigned *codes;                   // NULL: use the edge tensor
    const float *const *tables;              // [n * F] device pointers, table (c, f) is [V_f, k]
    int F, k;
    int off[4];
};

// the F <= 4 table rows of one edge: all four loads go out together (a feature the encoder does not have re-reads
// feature 0's row and is masked out of the sum) — a runtime-F loop would wait for each row before asking for the next
__device__ __forceinline__ float4 sm_enc_rows(const SmEnc &enc, const float *const *tp, unsigned code) {
    float4 t[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const int sh = f < enc.F ? 8 * f : 0;            // tp[f] of a missing feature is feature 0's pointer (sm_table_ptrs)
        t[f] = *reinterpret_cast<const float4 *>(tp[f] + (size_t)((code >> sh) & 255u) * enc.k);
    }
    float4 a = t[0];
#pragma unroll
    for (int f = 1; f < 4; ++f) {
        const float on = f < enc.F ? 1.f : 0.f;
        a.x = fmaf(on, t[f].x, a.x); a.y = fmaf(on, t[f].y, a.y); a.z = fmaf(on, t[f].z, a.z); a.w = fmaf(on, t[f].w, a.w);
    }
    return a;
}

// this lane's slice (column group c4) of its component's F tables; slots past F repeat feature 0
__device__ __forceinline__ void sm_table_ptrs(const SmEnc &enc, int c4, const float *(&tp)[4]) {
    const int c = (c4 * 4) / enc.k, kk = c4 * 4 - c * enc.k;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const int ff = f < enc.F ? f : 0;
        tp[f] = enc.tables[c * enc.F + ff] + kk - (size_t)enc.off[ff] * enc.k;
    }
}

struct Sm4 {            // online softmax state for 4 columns
    float4 mx, s, w;
};

__device__ __forceinline__ void sm_merge(float &mx, float &s, float &w, float mx2, float s2, float w2) {
    const float M = fmaxf(mx, mx2);
    const float a = (mx == -INFINITY) ? 0.f : __expf(mx - M);
    const float b = (mx2 == -INFINITY) ? 0.f : __expf(mx2 - M);
    s = s * a + s2 * b;
    w = w * a + w2 * b;
    mx = M;
}

// Work split.  With h/4 <= 32 a wavefront holds G = 64/LPR lane groups, and every lane group owns a ROW of its own: it
// walks that row's edges itself, the first two in one round trip (clamped indices), with the row's own reads (the self row,
// dout) in flight from the start.  At a molecule's average degree of two that is half (a quarter, an eighth) the wavefronts
// of one row per wavefront — one resident set instead of two or three — and no cross-group merge.  A row with more than
// SM_HEAVY edges is left to the whole wavefront afterwards: its edges are dealt over the G groups and the partial states
// merged by shuffles (hubs of a general graph; a group walking 10^4 edges alone would hold its wavefront for milliseconds).
constexpr int SM_HEAVY = 16;

template <int LPR, int ACT>
__global__ __launch_bounds__(SM_THREADS) void softmax_agg_fwd_kernel(const float *__restrict__ x, const float *__restrict__ ea,
                                                                  const int *__restrict__ rowptr, const int *__restrict__ col,
                                                                  const int *__restrict__ perm, const float *__restrict__ beta_p,
                                                                  long long n_dst, int h, float *__restrict__ out,
                                                                  float *__restrict__ lse, float *__restrict__ out_self,
                                                                  SmEnc enc) {
    // out_self != NULL: also writes out + x[d] (the layer's self loop, messagepassing.py:222-231 / :303-309); `out`
    // stays the plain aggregate, which the backward needs
    constexpr int G = WAVE / LPR;
    const int lane = threadIdx.x & 63;
    const int grp = lane / LPR, li = lane - grp * LPR;
    const long long wave_id = ((long long)xcd_contiguous_block(blockIdx.x, gridDim.x) * SM_THREADS + threadIdx.x) >> 6;
    const long long n_waves = ((long long)gridDim.x * SM_THREADS) >> 6;
    const int h4 = h >> 2;
    const float beta = *beta_p;
    // one row: edges e0 + first, e0 + first + step, ... of [e0, e1); coop = the groups of the wavefront share the row
    auto row = [&](long long d, int e0, int e1, int first, int step, bool coop) {
        for (int c4 = li; c4 < h4; c4 += LPR) {
            float4 xd = make_float4(0.f, 0.f, 0.f, 0.f);
            if (out_self) xd = reinterpret_cast<const float4 *>(x + (size_t)d * h)[c4];
            float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, s[4] = {0.f, 0.f, 0.f, 0.f}, w[4] = {0.f, 0.f, 0.f, 0.f};
            const float *tp[4] = {nullptr, nullptr, nullptr, nullptr};
            if (enc.codes) sm_table_ptrs(enc, c4, tp);
            auto message = [&](int e) {                      // codes in destination-sorted order: indexed by e
                float4 v = reinterpret_cast<const float4 *>(x + (size_t)col[e] * h)[c4];
                if (enc.codes) {
                    const float4 a = sm_enc_rows(enc, tp, enc.codes[e]);
                    v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
                } else if (ea) {
                    const float4 a = reinterpret_cast<const float4 *>(ea + (size_t)perm[e] * h)[c4];
                    v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
                }
                return make_float4(act_fwd<ACT>(v.x), act_fwd<ACT>(v.y), act_fwd<ACT>(v.z), act_fwd<ACT>(v.w));
            };
            auto update = [&](const float4 &m) {
                sm_update(mx[0], s[0], w[0], m.x, beta);
                sm_update(mx[1], s[1], w[1], m.y, beta);
                sm_update(mx[2], s[2], w[2], m.z, beta);
                sm_update(mx[3], s[3], w[3], m.w, beta);
            };
            int e = e0 + first;
            if (e0 < e1) {                                   // the first two edges travel together (clamped indices)
                const float4 m0 = message(min(e, e1 - 1)), m1 = message(min(e + step, e1 - 1));
                if (e < e1) update(m0);
                if (e + step < e1) update(m1);
                e += 2 * step;
            }
            for (; e < e1; e += step) update(message(e));
            if (coop) {
#pragma unroll
                for (int off = LPR; off < WAVE; off <<= 1) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float m2 = __shfl_xor(mx[q], off, WAVE), s2 = __shfl_xor(s[q], off, WAVE), w2 = __shfl_xor(w[q], off, WAVE);
                        sm_merge(mx[q], s[q], w[q], m2, s2, w2);
                    }
                }
            }
            if (!coop || grp == 0) {
                float4 o, l;
                float *op = &o.x, *lp = &l.x;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float z = s[q] + SM_EPS;
                    op[q] = (e1 > e0) ? w[q] / z : 0.f;
                    lp[q] = (e1 > e0) ? mx[q] + __logf(z) : 0.f;
                }
                reinterpret_cast<float4 *>(out + (size_t)d * h)[c4] = o;
                reinterpret_cast<float4 *>(lse + (size_t)d * h)[c4] = l;
                if (out_self)
                    reinterpret_cast<float4 *>(out_self + (size_t)d * h)[c4] = make_float4(o.x + xd.x, o.y + xd.y, o.z + xd.z, o.w + xd.w);
            }
        }
    };
    for (long long d0 = wave_id * G; d0 < n_dst; d0 += n_waves * G) {           // wave-uniform bounds
        const long long d = d0 + grp;
        const bool valid = d < n_dst;
        const int e_begin = valid ? rowptr[d] : 0, e_end = valid ? rowptr[d + 1] : 0;
        const bool heavy = G > 1 && e_end - e_begin > SM_HEAVY;
        if (valid && !heavy) row(d, e_begin, e_end, 0, 1, false);
        if (G > 1) {
            const unsigned long long hb = __ballot(heavy);
            if (hb) {
                for (int g = 0; g < G; ++g) {
                    const int eb = __shfl(e_begin, g * LPR, WAVE), ee = __shfl(e_end, g * LPR, WAVE);
                    if ((hb >> (g * LPR)) & 1ull) row(d0 + g, eb, ee, grp, G, true);
                }
            }
        }
    }
}

// ---- the model-level norm folded into the NEXT layer's softmax aggregation (round 5) ------------------------------------------------
// models.py:208-215 ends a layer with  x = act(norm(y)) (+ skip);  with the reference's default aggregation the next layer starts with
// the softmax walk above over x.  As in aggregate_norm_direct_fwd_kernel (encoder.hip) the element-wise pass between the two is not a
// launch of its own: the prologue derives scale / shift from the statistics accumulator the producing GEMM filled (the arithmetic of
// bn_apply_acc_kernel: replica totals in double, in replica order), every gathered row and the row itself are normalised on the
// fly, x is written once by the row's owner, block 0 leaves save_stat and the running statistics behind.  Identity message function,
// packed bond codes; rows >= *n_valid are padding and read as 0.  Bit-identical to phc_bn_fwd_from_partials + the walk above.
constexpr int SMN_MAX_H = 512;
template <int LPR>
__global__ __launch_bounds__(SM_THREADS) void softmax_agg_norm_fwd_kernel(
    const float *__restrict__ y, const float *__restrict__ skip, BnParams p, const int *__restrict__ n_valid, float momentum, float eps,
    const unsigned long long *__restrict__ acc, int norm_act, float *__restrict__ stat, float *__restrict__ xout,
    const int *__restrict__ rowptr, const int *__restrict__ col, const float *__restrict__ beta_p, long long n_dst, int h,
    float *__restrict__ out, float *__restrict__ lse, float *__restrict__ out_self, SmEnc enc) {
    __shared__ __attribute__((aligned(16))) float s_scale[SMN_MAX_H], s_shift[SMN_MAX_H];
    constexpr int G = WAVE / LPR;
    const int tid = threadIdx.x, lane = tid & 63;
    const int grp = lane / LPR, li = lane - grp * LPR;
    const long long wave_id = ((long long)xcd_contiguous_block(blockIdx.x, gridDim.x) * SM_THREADS + tid) >> 6;
    const long long n_waves = ((long long)gridDim.x * SM_THREADS) >> 6;
    const int h4 = h >> 2;
    const float beta = *beta_p;
    const long long count = n_valid ? min((long long)*n_valid, n_dst) : n_dst;
    const bool owner = blockIdx.x == 0;
    for (int c = tid; c < h; c += SM_THREADS) {
        double S = 0.0, Q = 0.0;
#pragma unroll
        for (int rp = 0; rp < STAT_ACC_REP; ++rp) {
            S += stat_acc_part(acc, h, rp, 0, c);
            Q += stat_acc_part(acc, h, rp, 1, c);
        }
        const int comp = c / p.k, kk = c - comp * p.k;
        float w = 1.f, bq = 0.f, rm_old = 0.f, rv_old = 1.f;
        if (p.has_weight) w = *bn_col_ptr(p.weight, p.flat, comp, kk, c);
        if (p.has_bias) bq = *bn_col_ptr(p.bias, p.flat, comp, kk, c);
        if (p.has_running && owner) {
            rm_old = *bn_col_ptr(p.running_mean, p.flat, comp, kk, c);
            rv_old = *bn_col_ptr(p.running_var, p.flat, comp, kk, c);
        }
        const BnFromTotals bt = bn_from_totals(S, Q, count, w, bq, eps);
        s_scale[c] = bt.scale;
        s_shift[c] = bt.shift;
        if (owner) {
            stat[c] = bt.mean;
            stat[h + c] = bt.invstd;
            stat[2 * h + c] = bt.scale;
            stat[3 * h + c] = bt.shift;
            stat[4 * h + c] = w;
            stat[5 * h + c] = bq;
            if (p.has_running) {
                *bn_col_ptr(p.running_mean, p.flat, comp, kk, c) = bn_running_update(rm_old, momentum, bt.mean);
                *bn_col_ptr(p.running_var, p.flat, comp, kk, c) = bn_running_update(rv_old, momentum, bt.unbiased);
            }
        }
    }
    if (owner && p.has_nbt && tid < p.n_comp)
        for (int q = 0; q < p.n_comp; ++q)
            if (q == tid) *p.num_batches_tracked[q] += 1;
    __syncthreads();
    const bool relu = norm_act == PHC_ACT_RELU;
    auto xrow = [&](long long r, int c4, const float4 sc, const float4 sh) {
        const float4 v = reinterpret_cast<const float4 *>(y + (size_t)r * h)[c4];
        const float4 rr = reinterpret_cast<const float4 *>(skip ? skip + (size_t)r * h : y)[c4];
        float4 o;
        if (relu) {
            o.x = fmaxf(fmaf(v.x, sc.x, sh.x), 0.f); o.y = fmaxf(fmaf(v.y, sc.y, sh.y), 0.f);
            o.z = fmaxf(fmaf(v.z, sc.z, sh.z), 0.f); o.w = fmaxf(fmaf(v.w, sc.w, sh.w), 0.f);
        } else {
            o.x = act_fwd_rt(norm_act, fmaf(v.x, sc.x, sh.x)); o.y = act_fwd_rt(norm_act, fmaf(v.y, sc.y, sh.y));
            o.z = act_fwd_rt(norm_act, fmaf(v.z, sc.z, sh.z)); o.w = act_fwd_rt(norm_act, fmaf(v.w, sc.w, sh.w));
        }
        if (skip) { o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w; }
        if (r >= count) o = make_float4(0.f, 0.f, 0.f, 0.f);
        return o;
    };
    auto row = [&](long long d, int e0, int e1, int first, int step, bool coop) {
        for (int c4 = li; c4 < h4; c4 += LPR) {
            const float4 sc = *reinterpret_cast<const float4 *>(s_scale + 4 * c4), sh = *reinterpret_cast<const float4 *>(s_shift + 4 * c4);
            const float4 xd = xrow(d, c4, sc, sh);
            float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, s[4] = {0.f, 0.f, 0.f, 0.f}, w[4] = {0.f, 0.f, 0.f, 0.f};
            const float *tp[4] = {nullptr, nullptr, nullptr, nullptr};
            sm_table_ptrs(enc, c4, tp);
            auto message = [&](int e) {
                float4 v = xrow(col[e], c4, sc, sh);
                const float4 a = sm_enc_rows(enc, tp, enc.codes[e]);
                v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
                return v;
            };
            auto update = [&](const float4 &m) {
                sm_update(mx[0], s[0], w[0], m.x, beta);
                sm_update(mx[1], s[1], w[1], m.y, beta);
                sm_update(mx[2], s[2], w[2], m.z, beta);
                sm_update(mx[3], s[3], w[3], m.w, beta);
            };
            int e = e0 + first;
            if (e0 < e1) {
                const float4 m0 = message(min(e, e1 - 1)), m1 = message(min(e + step, e1 - 1));
                if (e < e1) update(m0);
                if (e + step < e1) update(m1);
                e += 2 * step;
            }
            for (; e < e1; e += step) update(message(e));
            if (coop) {
#pragma unroll
                for (int off = LPR; off < WAVE; off <<= 1) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float m2 = __shfl_xor(mx[q], off, WAVE), s2 = __shfl_xor(s[q], off, WAVE), w2 = __shfl_xor(w[q], off, WAVE);
                        sm_merge(mx[q], s[q], w[q], m2, s2, w2);
                    }
                }
            }
            if (!coop || grp == 0) {
                float4 o, l;
                float *op = &o.x, *lp = &l.x;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float z = s[q] + SM_EPS;
                    op[q] = (e1 > e0) ? w[q] / z : 0.f;
                    lp[q] = (e1 > e0) ? mx[q] + __logf(z) : 0.f;
                }
                reinterpret_cast<float4 *>(xout + (size_t)d * h)[c4] = xd;                  // x itself, once
                reinterpret_cast<float4 *>(out + (size_t)d * h)[c4] = o;
                reinterpret_cast<float4 *>(lse + (size_t)d * h)[c4] = l;
                if (out_self)
                    reinterpret_cast<float4 *>(out_self + (size_t)d * h)[c4] = make_float4(o.x + xd.x, o.y + xd.y, o.z + xd.z, o.w + xd.w);
            }
        }
    };
    for (long long d0 = wave_id * G; d0 < n_dst; d0 += n_waves * G) {           // wave-uniform bounds
        const long long d = d0 + grp;
        const bool valid = d < n_dst;
        const int e_begin = valid ? rowptr[d] : 0, e_end = valid ? rowptr[d + 1] : 0;
        const bool heavy = G > 1 && e_end - e_begin > SM_HEAVY;
        if (valid && !heavy) row(d, e_begin, e_end, 0, 1, false);
        if (G > 1) {
            const unsigned long long hb = __ballot(heavy);
            if (hb) {
                for (int g = 0; g < G; ++g) {
                    const int eb = __shfl(e_begin, g * LPR, WAVE), ee = __shfl(e_end, g * LPR, WAVE);
                    if ((hb >> (g * LPR)) & 1ull) row(d0 + g, eb, ee, grp, G, true);
                }
            }
        }
    }
}

// source-sorted walk; dbeta partial sums per workgroup (fixed order => reproducible).  dbeta_partial has n_partials slots
// (phc_softmax_bwd_num_partials: the one-row-per-wavefront bound); the grid may be smaller and zeroes the rest.
template <int LPR, int ACT>
__global__ __launch_bounds__(SM_THREADS) void softmax_agg_bwd_kernel(
    const float *__restrict__ dout, const float *__restrict__ out, const float *__restrict__ lse, const float *__restrict__ x,
    const float *__restrict__ ea, const float *__restrict__ beta_p, const int *__restrict__ rowptr_t, const int *__restrict__ col_t,
    const int *__restrict__ perm_t, long long n_src, int h, float *__restrict__ dx, float *__restrict__ dea,
    float *__restrict__ dbeta_partial, int n_partials, long long self_n_dst, SmEnc enc, SmResiduals res) {
    // res: up to 8 further gradients of the SAME rows (the skip connections' shares of dx, ops.DxHandoff / DxSumHandoff) added on the
    // way out in slot order — the n-way sum launch in front of the atom-table gradient disappears (as aggregate_dx_rows_kernel<.., NRES>)
    __shared__ float red[SM_THREADS];
    constexpr int G = WAVE / LPR;
    const int lane = threadIdx.x & 63;
    const int grp = lane / LPR, li = lane - grp * LPR;
    const long long wave_id = ((long long)xcd_contiguous_block(blockIdx.x, gridDim.x) * SM_THREADS + threadIdx.x) >> 6;
    const long long n_waves = ((long long)gridDim.x * SM_THREADS) >> 6;
    const int h4 = h >> 2;
    const float beta = *beta_p;
    float db = 0.f;
    auto row = [&](long long sidx, int e0, int e1, int first, int step, bool coop) {
        for (int c4 = li; c4 < h4; c4 += LPR) {
            const float4 xs = reinterpret_cast<const float4 *>(x + (size_t)sidx * h)[c4];
            float4 gs = make_float4(0.f, 0.f, 0.f, 0.f);    // the self loop's gradient: dx[s] += dout[s]
            if (sidx < self_n_dst) gs = reinterpret_cast<const float4 *>(dout + (size_t)sidx * h)[c4];
            float4 rv[8];
            if (res.n > 0) {                                 // (uniform; unconditional loads: an unused slot re-reads x's row)
#pragma unroll
                for (int q = 0; q < 8; ++q) rv[q] = reinterpret_cast<const float4 *>((q < res.n ? res.p[q] : x) + (size_t)sidx * h)[c4];
            }
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            const float *tp[4] = {nullptr, nullptr, nullptr, nullptr};
            if (enc.codes) sm_table_ptrs(enc, c4, tp);
            struct EdgeRows { float4 pre, go, o, l; size_t p; };
            auto fetch = [&](int e) {                        // every row of one out-edge, all loads in flight together
                EdgeRows r;
                const size_t d = (size_t)col_t[e];
                r.p = (size_t)perm_t[e];
                r.pre = xs;
                if (enc.codes) {                             // codes in SOURCE-sorted order: indexed by e
                    const float4 a = sm_enc_rows(enc, tp, enc.codes[e]);
                    r.pre.x += a.x; r.pre.y += a.y; r.pre.z += a.z; r.pre.w += a.w;
                } else if (ea) {
                    const float4 a = reinterpret_cast<const float4 *>(ea + r.p * h)[c4];
                    r.pre.x += a.x; r.pre.y += a.y; r.pre.z += a.z; r.pre.w += a.w;
                }
                r.go = reinterpret_cast<const float4 *>(dout + d * h)[c4];
                r.o = reinterpret_cast<const float4 *>(out + d * h)[c4];
                r.l = reinterpret_cast<const float4 *>(lse + d * h)[c4];
                return r;
            };
            auto apply = [&](const EdgeRows &r) {
                const float pre[4] = {r.pre.x, r.pre.y, r.pre.z, r.pre.w};
                const float go[4] = {r.go.x, r.go.y, r.go.z, r.go.w}, o[4] = {r.o.x, r.o.y, r.o.z, r.o.w}, l[4] = {r.l.x, r.l.y, r.l.z, r.l.w};
                float g[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float m = act_fwd<ACT>(pre[q]);
                    const float wgt = __expf(beta * m - l[q]);
                    const float t = wgt * (m - o[q]);
                    db = fmaf(go[q] * t, m, db);
                    g[q] = go[q] * (wgt + beta * t) * act_grad<ACT>(pre[q]);
                    acc[q] += g[q];
                }
                if (dea) reinterpret_cast<float4 *>(dea + r.p * h)[c4] = make_float4(g[0], g[1], g[2], g[3]);
            };
            int e = e0 + first;
            if (e0 < e1) {                                   // the first two out-edges travel together (clamped indices)
                const EdgeRows r0 = fetch(min(e, e1 - 1)), r1 = fetch(min(e + step, e1 - 1));
                if (e < e1) apply(r0);
                if (e + step < e1) apply(r1);
                e += 2 * step;
            }
            for (; e < e1; e += step) apply(fetch(e));
            if (coop) {
#pragma unroll
                for (int off = LPR; off < WAVE; off <<= 1)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[q] += __shfl_xor(acc[q], off, WAVE);
            }
            if (!coop || grp == 0) {
                float4 o4 = make_float4(acc[0] + gs.x, acc[1] + gs.y, acc[2] + gs.z, acc[3] + gs.w);
                if (res.n > 0) {
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        if (q < res.n) { o4.x += rv[q].x; o4.y += rv[q].y; o4.z += rv[q].z; o4.w += rv[q].w; }     // fixed order: slot 0, 1, ...
                }
                reinterpret_cast<float4 *>(dx + (size_t)sidx * h)[c4] = o4;
            }
        }
    };
    for (long long s0 = wave_id * G; s0 < n_src; s0 += n_waves * G) {           // wave-uniform bounds
        const long long sidx = s0 + grp;
        const bool valid = sidx < n_src;
        const int e_begin = valid ? rowptr_t[sidx] : 0, e_end = valid ? rowptr_t[sidx + 1] : 0;
        const bool heavy = G > 1 && e_end - e_begin > SM_HEAVY;
        if (valid && !heavy) row(sidx, e_begin, e_end, 0, 1, false);
        if (G > 1) {
            const unsigned long long hb = __ballot(heavy);
            if (hb) {
                for (int g = 0; g < G; ++g) {
                    const int eb = __shfl(e_begin, g * LPR, WAVE), ee = __shfl(e_end, g * LPR, WAVE);
                    if ((hb >> (g * LPR)) & 1ull) row(s0 + g, eb, ee, grp, G, true);
                }
            }
        }
    }
    red[threadIdx.x] = db;
    __syncthreads();
    for (int st = SM_THREADS / 2; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        dbeta_partial[blockIdx.x] = red[0];
        for (int i = (int)(gridDim.x + blockIdx.x); i < n_partials; i += (int)gridDim.x) dbeta_partial[i] = 0.f;
    }
}

static inline int sm_lpr(int h4) { return h4 <= 8 ? 8 : h4 <= 16 ? 16 : h4 <= 32 ? 32 : 64; }
static inline unsigned sm_grid(long long rows, int rows_per_wave = 1) {
    long long b = (rows + 4 * rows_per_wave - 1) / (4 * rows_per_wave);
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (unsigned)xcd_grid(b);                 // a multiple of 8: xcd_contiguous_block
}
static inline bool sm_al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

#define PHC_SM_ACT(act, MACRO)                            \
    switch (act) {                                        \
        case PHC_ACT_IDENTITY: MACRO(PHC_ACT_IDENTITY); break; \
        case PHC_ACT_RELU: MACRO(PHC_ACT_RELU); break;         \
        case PHC_ACT_LRELU: MACRO(PHC_ACT_LRELU); break;       \
        case PHC_ACT_ELU: MACRO(PHC_ACT_ELU); break;           \
        case PHC_ACT_SWISH: MACRO(PHC_ACT_SWISH); break;       \
        default: MACRO(PHC_ACT_SELU); break;                   \
    }

}  // namespace phc

using namespace phc;

extern "C" {

int phc_softmax_bwd_num_partials(int64_t n_src) { return (int)sm_grid(n_src); }

int phc_aggregate_softmax_fwd_self(const float *x, const float *ea, const int32_t *rowptr, const int32_t *col, const int32_t *perm,
                                   const float *beta_dev, int64_t n_dst, int h, int act, float *out, float *out_self, float *lse,
                                   void *stream);

int phc_aggregate_softmax_fwd(const float *x, const float *ea, const int32_t *rowptr, const int32_t *col, const int32_t *perm,
                              const float *beta_dev, int64_t n_dst, int h, int act, float *out, float *lse, void *stream) {
    return phc_aggregate_softmax_fwd_self(x, ea, rowptr, col, perm, beta_dev, n_dst, h, act, out, nullptr, lse, stream);
}

static int sm_make_enc(const char *who, const uint32_t *codes, const float *const *tables, const int32_t *vocab, int F, int n, int k, int h,
                       SmEnc *enc) {
    PHC_REQUIRE(codes && tables && vocab && F >= 1 && F <= 4 && n >= 1 && k >= 1 && n * k == h && k % 4 == 0, PHC_ERR_INVALID_ARG,
                "%s: needs codes, tables, 1 <= F <= 4, h = n*k, k %% 4 == 0", who);
    enc->codes = codes; enc->tables = tables; enc->F = F; enc->k = k;
    int off = 0;
    for (int f = 0; f < 4; ++f) {
        enc->off[f] = off;
        if (f < F) {
            PHC_REQUIRE(vocab[f] >= 1, PHC_ERR_INVALID_ARG, "%s: vocab[%d] = %d", who, f, vocab[f]);
            off += vocab[f];
        }
    }
    PHC_REQUIRE(off <= 256, PHC_ERR_UNSUPPORTED, "%s: packed codes need sum(vocab) <= 256", who);
    return PHC_OK;
}

int phc_aggregate_softmax_norm_supported(int64_t N, int h, int F, int vocab_rows) {
    return (N >= 1 && h >= 4 && h % 4 == 0 && h <= SMN_MAX_H && F >= 1 && F <= 4 && vocab_rows >= 1 && vocab_rows <= 256) ? 1 : 0;
}

int phc_aggregate_softmax_norm_enc_fwd(const float *y, const uint64_t *norm_acc, const float *const *weight, const float *const *bias,
                                       float *const *running_mean, float *const *running_var, int64_t *const *num_batches_tracked,
                                       int n_comp, float momentum, float eps, int norm_act, const int32_t *n_valid_dev, const float *skip,
                                       float *save_stat, float *x_out, const uint32_t *codes_sorted, const float *const *tables,
                                       const int32_t *vocab, int F, int n, int k, const int32_t *rowptr, const int32_t *col,
                                       const float *beta_dev, int64_t n_dst, float *out, float *out_self, float *lse, void *stream) {
    const int h = n * k;
    SmEnc enc;
    int rc = sm_make_enc("phc_aggregate_softmax_norm_enc_fwd", codes_sorted, tables, vocab, F, n, k, h, &enc);
    if (rc) return rc;
    PHC_REQUIRE(phc_aggregate_softmax_norm_supported(n_dst, h, F, enc.off[F - 1] + vocab[F - 1]), PHC_ERR_UNSUPPORTED,
                "phc_aggregate_softmax_norm_enc_fwd: %lld rows, h = %d, F = %d outside the fused form's range", (long long)n_dst, h, F);
    PHC_REQUIRE(y && norm_acc && save_stat && x_out && rowptr && col && beta_dev && out && lse && codes_sorted, PHC_ERR_INVALID_ARG,
                "phc_aggregate_softmax_norm_enc_fwd: null pointer");
    PHC_REQUIRE(norm_act >= PHC_ACT_IDENTITY && norm_act <= PHC_ACT_SELU, PHC_ERR_INVALID_ARG,
                "phc_aggregate_softmax_norm_enc_fwd: unknown activation %d", norm_act);
    PHC_REQUIRE(sm_al16(y) && sm_al16(x_out) && sm_al16(out) && sm_al16(lse) && (!skip || sm_al16(skip)) && (!out_self || sm_al16(out_self)) &&
                    (reinterpret_cast<uintptr_t>(norm_acc) & 7) == 0,
                PHC_ERR_UNSUPPORTED, "phc_aggregate_softmax_norm_enc_fwd: 16-byte aligned rows, 8-byte aligned accumulator");
    BnParams p;
    rc = fill_params("phc_aggregate_softmax_norm_enc_fwd", &p, weight, bias, running_mean, running_var, num_batches_tracked, n_comp, h);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int lpr = sm_lpr(h >> 2);
    const unsigned long long *acc = reinterpret_cast<const unsigned long long *>(norm_acc);
    ProfScope prof(K_SOFTMAX_FWD, s);
#define PHC_SMN(LPRC)                                                                                                                 \
    PHC_LAUNCH((softmax_agg_norm_fwd_kernel<LPRC>), dim3(sm_grid(n_dst, 64 / lpr)), dim3(SM_THREADS), 0, s, y, skip, p, n_valid_dev, momentum, \
               eps, acc, norm_act, save_stat, x_out, rowptr, col, beta_dev, (long long)n_dst, h, out, lse, out_self, enc)
    switch (lpr) {
        case 8: PHC_SMN(8); break;
        case 16: PHC_SMN(16); break;
        case 32: PHC_SMN(32); break;
        default: PHC_SMN(64); break;
    }
#undef PHC_SMN
    return check_launch("aggregate_softmax_norm_enc_fwd");
}

static int sm_fwd_impl(const float *x, const float *ea, const int32_t *rowptr, const int32_t *col, const int32_t *perm,
                       const float *beta_dev, int64_t n_dst, int h, int act, float *out, float *out_self, float *lse, SmEnc enc,
                       void *stream);

int phc_aggregate_softmax_fwd_self(const float *x, const float *ea, const int32_t *rowptr, const int32_t *col, const int32_t *perm,
                                   const float *beta_dev, int64_t n_dst, int h, int act, float *out, float *out_self, float *lse,
                                   void *stream) {
    SmEnc enc;
    SmEnc zero = {};
    enc = zero;
    return sm_fwd_impl(x, ea, rowptr, col, perm, beta_dev, n_dst, h, act, out, out_self, lse, enc, stream);
}

int phc_aggregate_softmax_enc_fwd(const float *x, const uint32_t *codes_sorted, const float *const *tables, const int32_t *vocab, int F,
                                  int n, int k, const int32_t *rowptr, const int32_t *col, const float *beta_dev, int64_t n_dst,
                                  int act, float *out, float *out_self, float *lse, void *stream) {
    SmEnc enc;
    int rc = sm_make_enc("phc_aggregate_softmax_enc_fwd", codes_sorted, tables, vocab, F, n, k, n * k, &enc);
    if (rc) return rc;
    return sm_fwd_impl(x, nullptr, rowptr, col, nullptr, beta_dev, n_dst, n * k, act, out, out_self, lse, enc, stream);
}

static int sm_fwd_impl(const float *x, const float *ea, const int32_t *rowptr, const int32_t *col, const int32_t *perm,
                       const float *beta_dev, int64_t n_dst, int h, int act, float *out, float *out_self, float *lse, SmEnc enc,
                       void *stream) {
    PHC_REQUIRE(!out_self || sm_al16(out_self), PHC_ERR_UNSUPPORTED, "phc_aggregate_softmax_fwd_self: out_self must be 16-byte aligned");
    PHC_REQUIRE(n_dst >= 0 && h >= 1, PHC_ERR_INVALID_ARG, "phc_aggregate_softmax_fwd: bad sizes");
    PHC_REQUIRE(act >= PHC_ACT_IDENTITY && act <= PHC_ACT_SELU, PHC_ERR_INVALID_ARG, "phc_aggregate_softmax_fwd: unknown activation %d", act);
    if (n_dst == 0) return PHC_OK;
    PHC_REQUIRE(x && rowptr && col && beta_dev && out && lse && (!ea || perm), PHC_ERR_INVALID_ARG, "phc_aggregate_softmax_fwd: null pointer");
    PHC_REQUIRE(h % 4 == 0 && sm_al16(x) && sm_al16(out) && sm_al16(lse) && (!ea || sm_al16(ea)), PHC_ERR_UNSUPPORTED,
                "phc_aggregate_softmax_fwd: needs h %% 4 == 0 and 16-byte aligned rows");
    hipStream_t s = (hipStream_t)stream;
    const int lpr = sm_lpr(h >> 2);
    ProfScope prof(K_SOFTMAX_FWD, s);
#define PHC_SM_FWD(ACTC)                                                                                                 \
    switch (lpr) {                                                                                                       \
        case 8: PHC_LAUNCH((softmax_agg_fwd_kernel<8, ACTC>), dim3(sm_grid(n_dst, 64 / lpr)), dim3(SM_THREADS), 0, s, x, ea, rowptr, col, perm, beta_dev, (long long)n_dst, h, out, lse, out_self, enc); break;   \
        case 16: PHC_LAUNCH((softmax_agg_fwd_kernel<16, ACTC>), dim3(sm_grid(n_dst, 64 / lpr)), dim3(SM_THREADS), 0, s, x, ea, rowptr, col, perm, beta_dev, (long long)n_dst, h, out, lse, out_self, enc); break; \
        case 32: PHC_LAUNCH((softmax_agg_fwd_kernel<32, ACTC>), dim3(sm_grid(n_dst, 64 / lpr)), dim3(SM_THREADS), 0, s, x, ea, rowptr, col, perm, beta_dev, (long long)n_dst, h, out, lse, out_self, enc); break; \
        default: PHC_LAUNCH((softmax_agg_fwd_kernel<64, ACTC>), dim3(sm_grid(n_dst, 64 / lpr)), dim3(SM_THREADS), 0, s, x, ea, rowptr, col, perm, beta_dev, (long long)n_dst, h, out, lse, out_self, enc); break; \
    }
    PHC_SM_ACT(act, PHC_SM_FWD)
#undef PHC_SM_FWD
    return check_launch("aggregate_softmax_fwd");
}

int phc_aggregate_softmax_bwd_self(const float *dout, const float *out, const float *lse, const float *x, const float *ea,
                                   const float *beta_dev, const int32_t *rowptr_t, const int32_t *col_t, const int32_t *perm_t,
                                   int64_t n_src, int h, int act, int64_t self_n_dst, float *dx, float *dea, float *dbeta_partial,
                                   void *stream);

int phc_aggregate_softmax_bwd(const float *dout, const float *out, const float *lse, const float *x, const float *ea,
                              const float *beta_dev, const int32_t *rowptr_t, const int32_t *col_t, const int32_t *perm_t,
                              int64_t n_src, int h, int act, float *dx, float *dea, float *dbeta_partial, void *stream) {
    return phc_aggregate_softmax_bwd_self(dout, out, lse, x, ea, beta_dev, rowptr_t, col_t, perm_t, n_src, h, act, 0, dx, dea,
                                          dbeta_partial, stream);
}

static int sm_bwd_impl(const float *dout, const float *out, const float *lse, const float *x, const float *ea, const float *beta_dev,
                       const int32_t *rowptr_t, const int32_t *col_t, const int32_t *perm_t, int64_t n_src, int h, int act,
                       int64_t self_n_dst, float *dx, float *dea, float *dbeta_partial, SmEnc enc, void *stream,
                       const float *const *residuals = nullptr, int num_residuals = 0);

int phc_aggregate_softmax_bwd_self(const float *dout, const float *out, const float *lse, const float *x, const float *ea,
                                   const float *beta_dev, const int32_t *rowptr_t, const int32_t *col_t, const int32_t *perm_t,
                                   int64_t n_src, int h, int act, int64_t self_n_dst, float *dx, float *dea, float *dbeta_partial,
                                   void *stream) {
    SmEnc enc;
    SmEnc zero = {};
    enc = zero;
    return sm_bwd_impl(dout, out, lse, x, ea, beta_dev, rowptr_t, col_t, perm_t, n_src, h, act, self_n_dst, dx, dea, dbeta_partial,
                       enc, stream);
}

int phc_aggregate_softmax_enc_bwd(const float *dout, const float *out, const float *lse, const float *x, const uint32_t *codes_src,
                                  const float *const *tables, const int32_t *vocab, int F, int n, int k, const float *beta_dev,
                                  const int32_t *rowptr_t, const int32_t *col_t, const int32_t *perm_t, int64_t n_src, int act,
                                  int64_t self_n_dst, float *dx, float *dea, float *dbeta_partial, void *stream) {
    SmEnc enc;
    int rc = sm_make_enc("phc_aggregate_softmax_enc_bwd", codes_src, tables, vocab, F, n, k, n * k, &enc);
    if (rc) return rc;
    return sm_bwd_impl(dout, out, lse, x, nullptr, beta_dev, rowptr_t, col_t, perm_t, n_src, n * k, act, self_n_dst, dx, dea,
                       dbeta_partial, enc, stream);
}

/* ... with up to 8 further gradients of the same rows added on the way out (the skip connections' shares of dx: models.py:230-234 with
 * sc_type "first", the skip tensor itself with "last") — what an n-way sum launch did behind this one */
int phc_aggregate_softmax_enc_bwd_res(const float *dout, const float *out, const float *lse, const float *x, const uint32_t *codes_src,
                                      const float *const *tables, const int32_t *vocab, int F, int n, int k, const float *beta_dev,
                                      const int32_t *rowptr_t, const int32_t *col_t, const int32_t *perm_t, int64_t n_src, int act,
                                      int64_t self_n_dst, const float *const *residuals, int num_residuals, float *dx, float *dea,
                                      float *dbeta_partial, void *stream) {
    SmEnc enc;
    int rc = sm_make_enc("phc_aggregate_softmax_enc_bwd_res", codes_src, tables, vocab, F, n, k, n * k, &enc);
    if (rc) return rc;
    return sm_bwd_impl(dout, out, lse, x, nullptr, beta_dev, rowptr_t, col_t, perm_t, n_src, n * k, act, self_n_dst, dx, dea,
                       dbeta_partial, enc, stream, residuals, num_residuals);
}

static int sm_bwd_impl(const float *dout, const float *out, const float *lse, const float *x, const float *ea, const float *beta_dev,
                       const int32_t *rowptr_t, const int32_t *col_t, const int32_t *perm_t, int64_t n_src, int h, int act,
                       int64_t self_n_dst, float *dx, float *dea, float *dbeta_partial, SmEnc enc, void *stream,
                       const float *const *residuals, int num_residuals) {
    SmResiduals res = {};
    PHC_REQUIRE(num_residuals >= 0 && num_residuals <= 8 && (num_residuals == 0 || residuals), PHC_ERR_INVALID_ARG,
                "phc_aggregate_softmax_bwd: %d residual tensors (0..8)", num_residuals);
    for (int q = 0; q < num_residuals; ++q) {
        PHC_REQUIRE(residuals[q] && sm_al16(residuals[q]), PHC_ERR_INVALID_ARG, "phc_aggregate_softmax_bwd: residual %d is null / not 16-byte aligned", q);
        res.p[q] = residuals[q];
    }
    res.n = num_residuals;
    PHC_REQUIRE(self_n_dst >= 0, PHC_ERR_INVALID_ARG, "phc_aggregate_softmax_bwd_self: bad self_n_dst");
    PHC_REQUIRE(n_src >= 0 && h >= 1, PHC_ERR_INVALID_ARG, "phc_aggregate_softmax_bwd: bad sizes");
    PHC_REQUIRE(act >= PHC_ACT_IDENTITY && act <= PHC_ACT_SELU, PHC_ERR_INVALID_ARG, "phc_aggregate_softmax_bwd: unknown activation %d", act);
    if (n_src == 0) return PHC_OK;
    PHC_REQUIRE(dout && out && lse && x && beta_dev && rowptr_t && col_t && perm_t && dx && dbeta_partial, PHC_ERR_INVALID_ARG,
                "phc_aggregate_softmax_bwd: null pointer");
    PHC_REQUIRE(h % 4 == 0 && sm_al16(x) && sm_al16(dout) && sm_al16(dx) && (!ea || sm_al16(ea)) && (!dea || sm_al16(dea)),
                PHC_ERR_UNSUPPORTED, "phc_aggregate_softmax_bwd: needs h %% 4 == 0 and 16-byte aligned rows");
    hipStream_t s = (hipStream_t)stream;
    const int lpr = sm_lpr(h >> 2);
    ProfScope prof(K_SOFTMAX_BWD, s);
#define PHC_SM_BWD(ACTC)                                                                                                 \
    switch (lpr) {                                                                                                       \
        case 8: PHC_LAUNCH((softmax_agg_bwd_kernel<8, ACTC>), dim3(sm_grid(n_src, 64 / lpr)), dim3(SM_THREADS), 0, s, dout, out, lse, x, ea, beta_dev, rowptr_t, col_t, perm_t, (long long)n_src, h, dx, dea, dbeta_partial, (int)sm_grid(n_src), (long long)self_n_dst, enc, res); break;   \
        case 16: PHC_LAUNCH((softmax_agg_bwd_kernel<16, ACTC>), dim3(sm_grid(n_src, 64 / lpr)), dim3(SM_THREADS), 0, s, dout, out, lse, x, ea, beta_dev, rowptr_t, col_t, perm_t, (long long)n_src, h, dx, dea, dbeta_partial, (int)sm_grid(n_src), (long long)self_n_dst, enc, res); break; \
        case 32: PHC_LAUNCH((softmax_agg_bwd_kernel<32, ACTC>), dim3(sm_grid(n_src, 64 / lpr)), dim3(SM_THREADS), 0, s, dout, out, lse, x, ea, beta_dev, rowptr_t, col_t, perm_t, (long long)n_src, h, dx, dea, dbeta_partial, (int)sm_grid(n_src), (long long)self_n_dst, enc, res); break; \
        default: PHC_LAUNCH((softmax_agg_bwd_kernel<64, ACTC>), dim3(sm_grid(n_src, 64 / lpr)), dim3(SM_THREADS), 0, s, dout, out, lse, x, ea, beta_dev, rowptr_t, col_t, perm_t, (long long)n_src, h, dx, dea, dbeta_partial, (int)sm_grid(n_src), (long long)self_n_dst, enc, res); break; \
    }
    PHC_SM_ACT(act, PHC_SM_BWD)
#undef PHC_SM_BWD
    return check_launch("aggregate_softmax_bwd");
}

}  // extern "C"
