// Integer-feature encoders fused into the message-passing path.
//
// Reference: PHMEncoder = phm_dim independent IntegerEncoders, each a sum of
// nn.Embedding lookups (phc/hypercomplex/encoder.py:31-34,
// phc/quaternion/encoder.py:43-51); the model materialises the [E, h] bond
// embedding of every layer (models.py:238-240) and autograd runs one
// embedding_dense_backward (a device sort!) per table — 84 of them per step.
//
// Here the n*F tiny tables are read in place through a device pointer array:
//   * encode_fwd           : out[i] = sum_f T_f[idx[i,f]]            (atoms)
//   * aggregate_enc_fwd    : the aggregation kernel with edge features looked up
//                            from LDS-staged tables — the [E, h] tensor never exists
//   * table_grad + reduce  : dT = onehot^T . g as a GEMM on the matrix cores (one-hot tiles
//                            generated in LDS), fixed-order split slabs, then a slab sum
//                            into ONE flat gradient buffer laid out [c][f][V_f][k]
//                            (the per-table grads are views).  No atomics, reproducible.
#include "phc_common.h"
#include <string.h>
#include "phm_grad_device.h"

namespace phc {

constexpr int ENC_MAX_F = PHC_MAX_ENC_FEATS;
constexpr int ENC_THREADS = 256;

struct EncMeta {
    int F;
    int R;                       // total table rows = sum_f V_f
    int off[ENC_MAX_F + 1];      // row offset of feature f in the packed row space
};

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void acc4(float4 &a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }

// ---- atoms: out[i, c*k+kap] = sum_f T_{c,f}[idx[i,f]][kap] --------------------------------
// LPR lanes cover one output row with float4 columns; 64/LPR nodes per wavefront.
template <int LPR, int FMAX>
__global__ __launch_bounds__(ENC_THREADS) void encode_fwd_kernel(const long long *__restrict__ idx,
                                                                const float *const *__restrict__ tables, long long N,
                                                                int F, int n, int k, float *__restrict__ out, EncMeta m,
                                                                int *__restrict__ err) {
    __shared__ const float *tptr[PHC_MAX_COMPONENTS * ENC_MAX_F];   // table base pointers: no pointer chase per lookup
    for (int e = threadIdx.x; e < n * F; e += ENC_THREADS) tptr[e] = tables[e];
    __syncthreads();
    constexpr int G = WAVE / LPR;
    const int lane = threadIdx.x & 63;
    const int grp = lane / LPR, li = lane - grp * LPR;
    const long long wave_id = ((long long)blockIdx.x * ENC_THREADS + threadIdx.x) >> 6;
    const long long n_waves = ((long long)gridDim.x * ENC_THREADS) >> 6;
    const int h = n * k, h4 = h >> 2;
    for (long long base = wave_id * G; base < N; base += n_waves * G) {
        const long long i = base + grp;
        if (i >= N) continue;
        // all F indices of the row first (one round of loads), then all F table rows (a second round)
        long long iv[FMAX];
#pragma unroll
        for (int f = 0; f < FMAX; ++f) {
            const int ff = min(f, F - 1);
            iv[f] = checked_index(idx[i * F + ff], m.off[ff + 1] - m.off[ff], err, PHC_IDX_ERR_ATOM);
        }
        for (int c4 = li; c4 < h4; c4 += LPR) {
            const int col = c4 << 2;
            const int c = col / k, kap = col - c * k;
            float4 rows[FMAX];
#pragma unroll
            for (int f = 0; f < FMAX; ++f) rows[f] = ld4(tptr[c * F + min(f, F - 1)] + iv[f] * k + kap);
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int f = 0; f < FMAX; ++f)
                if (f < F) acc4(acc, rows[f]);
            reinterpret_cast<float4 *>(out + (size_t)i * h)[c4] = acc;
        }
    }
}

__global__ __launch_bounds__(ENC_THREADS) void encode_fwd_scalar_kernel(const long long *__restrict__ idx,
                                                                       const float *const *__restrict__ tables,
                                                                       long long N, int F, int n, int k,
                                                                       float *__restrict__ out, EncMeta m, int *__restrict__ err) {
    const int h = n * k;
    const long long total = N * h;
    for (long long e = (long long)blockIdx.x * ENC_THREADS + threadIdx.x; e < total; e += (long long)gridDim.x * ENC_THREADS) {
        const long long i = e / h;
        const int col = (int)(e - i * h);
        const int c = col / k, kap = col - c * k;
        float acc = 0.f;
        for (int f = 0; f < F; ++f)
            acc += tables[c * F + f][checked_index(idx[i * F + f], m.off[f + 1] - m.off[f], err, PHC_IDX_ERR_ATOM) * k + kap];
        out[e] = acc;
    }
}

// ---- fused aggregation with table-encoded edge features -------------------------------------
// out[d] = sum_{e in row d} phi( x[col[e]] + sum_f T_f[bond[perm[e], f]] ) (+ x[d])
template <int LPR, int ACT>
__global__ __launch_bounds__(ENC_THREADS) void aggregate_enc_fwd_kernel(
    const float *__restrict__ x, const long long *__restrict__ bond, const float *const *__restrict__ tables, EncMeta m,
    const int *__restrict__ rowptr, const int *__restrict__ col, const int *__restrict__ perm, long long n_dst, int n, int k,
    int add_self, float *__restrict__ out, int *__restrict__ err) {
    extern __shared__ __attribute__((aligned(16))) float tbl[];      // [R][h], component-major columns
    const int h = n * k, h4 = h >> 2;
    for (int e = threadIdx.x; e < m.R * h; e += ENC_THREADS) {
        const int r = e / h, cidx = e - r * h;
        const int c = cidx / k, kap = cidx - c * k;
        int f = 0;
        while (f + 1 < m.F && r >= m.off[f + 1]) ++f;
        tbl[e] = tables[c * m.F + f][(size_t)(r - m.off[f]) * k + kap];
    }
    __syncthreads();

    constexpr int G = WAVE / LPR;
    const int lane = threadIdx.x & 63;
    const int grp = lane / LPR, li = lane - grp * LPR;
    const long long wave_id = ((long long)blockIdx.x * ENC_THREADS + threadIdx.x) >> 6;
    const long long n_waves = ((long long)gridDim.x * ENC_THREADS) >> 6;
    const int F = m.F;

    for (long long d = wave_id; d < n_dst; d += n_waves) {
        const int e_begin = rowptr[d], e_end = rowptr[d + 1];
        for (int c4 = li; c4 < h4; c4 += LPR) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int e = e_begin + grp; e < e_end; e += G) {
                const int s = col[e];
                const long long p = perm[e];
                float4 v = reinterpret_cast<const float4 *>(x + (size_t)s * h)[c4];
                for (int f = 0; f < F; ++f) {
                    const int r = m.off[f] + (int)checked_index(bond[p * F + f], m.off[f + 1] - m.off[f], err, PHC_IDX_ERR_BOND);
                    acc4(v, reinterpret_cast<const float4 *>(tbl + r * h)[c4]);
                }
                acc.x += act_fwd<ACT>(v.x); acc.y += act_fwd<ACT>(v.y);
                acc.z += act_fwd<ACT>(v.z); acc.w += act_fwd<ACT>(v.w);
            }
#pragma unroll
            for (int off = LPR; off < WAVE; off <<= 1) {
                acc.x += __shfl_xor(acc.x, off, WAVE); acc.y += __shfl_xor(acc.y, off, WAVE);
                acc.z += __shfl_xor(acc.z, off, WAVE); acc.w += __shfl_xor(acc.w, off, WAVE);
            }
            if (grp == 0) {
                if (add_self) acc4(acc, reinterpret_cast<const float4 *>(x + (size_t)d * h)[c4]);
                reinterpret_cast<float4 *>(out + (size_t)d * h)[c4] = acc;
            }
        }
    }
}

// ---- packed edge codes --------------------------------------------------------------------------
// codes[e] = table rows (off_f + bond[perm[e], f]) of the e-th edge of a SORTED edge list, one byte per feature
// (F <= 4, sum(vocab) <= 256).  Built once per (edge structure, bond tensor); the aggregation then reads
// 4 bytes per edge next to col[e] instead of chasing perm[e] -> F int64 loads.
__global__ void pack_edge_codes_kernel(const long long *__restrict__ bond, const int *__restrict__ perm, EncMeta m, long long E,
                                       unsigned *__restrict__ codes, int *__restrict__ err) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (long long)gridDim.x * blockDim.x) {
        const long long p = perm ? perm[e] : e;
        unsigned c = 0;
        for (int f = 0; f < m.F; ++f)
            c |= (unsigned)(m.off[f] + (int)checked_index(bond[p * m.F + f], m.off[f + 1] - m.off[f], err, PHC_IDX_ERR_BOND)) << (8 * f);
        codes[e] = c;
    }
}

template <int F>
__device__ __forceinline__ void add_code_rows(float4 &v, const float *__restrict__ tbl, unsigned code, int h, int c4) {
#pragma unroll
    for (int f = 0; f < F; ++f) acc4(v, reinterpret_cast<const float4 *>(tbl + ((code >> (8 * f)) & 255u) * h)[c4]);
}

// out[d] = sum_{e in row d} phi( x[col[e]] + sum_f tbl[code_f(e)] ) (+ x[d]);  two edges per lane group in flight
template <int LPR, int ACT, int F>
__global__ __launch_bounds__(ENC_THREADS) void aggregate_encp_fwd_kernel(
    const float *__restrict__ x, const unsigned *__restrict__ codes, const float *const *__restrict__ tables, EncMeta m,
    const int *__restrict__ rowptr, const int *__restrict__ col, long long n_dst, int n, int k, int add_self,
    float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float tbl[];      // [R][h], component-major columns
    __shared__ const float *tptr[PHC_MAX_COMPONENTS * 4];             // table base pointers (n * F <= 64)
    const int h = n * k, h4 = h >> 2;
    if ((int)threadIdx.x < n * m.F) tptr[threadIdx.x] = tables[threadIdx.x];
    __syncthreads();
    // stage the tables: k/4 float4 per (row, component); loads are independent => batched 8 deep
    {
        const int k4 = k >> 2, per_row = n * k4, total = m.R * per_row;
#pragma unroll 8
        for (int e = threadIdx.x; e < total; e += ENC_THREADS) {
            const int r = e / per_row, q = e - r * per_row;
            const int c = q / k4, kk = q - c * k4;
            int f = 0;
            while (f + 1 < m.F && r >= m.off[f + 1]) ++f;
            reinterpret_cast<float4 *>(tbl + r * h + c * k)[kk] =
                reinterpret_cast<const float4 *>(tptr[c * m.F + f] + (size_t)(r - m.off[f]) * k)[kk];
        }
    }
    __syncthreads();
    constexpr int G = WAVE / LPR;
    const int lane = threadIdx.x & 63;
    const int grp = lane / LPR, li = lane - grp * LPR;
    const long long wave_id = ((long long)blockIdx.x * ENC_THREADS + threadIdx.x) >> 6;
    const long long n_waves = ((long long)gridDim.x * ENC_THREADS) >> 6;
    for (long long d = wave_id; d < n_dst; d += n_waves) {
        const int e_begin = rowptr[d], e_end = rowptr[d + 1];
        for (int c4 = li; c4 < h4; c4 += LPR) {
            float4 acc0 = make_float4(0.f, 0.f, 0.f, 0.f), acc1 = acc0;
            // the self row goes in flight first (pointer select, no branch): it used to be a serial round trip at the end
            const float4 selfv = reinterpret_cast<const float4 *>(x + (add_self ? (size_t)d * h : 0))[c4];
            int e = e_begin + grp;
            if (e_begin < e_end) {                           // wave-uniform
                // the first two edges of every lane group with clamped indices: col / codes of both in one round trip,
                // then both row gathers in one (molecule graphs rarely have more: in-degree 2-4)
                const int i0 = min(e, e_end - 1), i1 = min(e + G, e_end - 1);
                const int s0 = col[i0], s1 = col[i1];
                const unsigned k0 = codes[i0], k1 = codes[i1];
                float4 v0 = reinterpret_cast<const float4 *>(x + (size_t)s0 * h)[c4];
                float4 v1 = reinterpret_cast<const float4 *>(x + (size_t)s1 * h)[c4];
                add_code_rows<F>(v0, tbl, k0, h, c4);
                add_code_rows<F>(v1, tbl, k1, h, c4);
                const bool h0 = e < e_end, h1 = e + G < e_end;
                acc0.x += h0 ? act_fwd<ACT>(v0.x) : 0.f; acc0.y += h0 ? act_fwd<ACT>(v0.y) : 0.f;
                acc0.z += h0 ? act_fwd<ACT>(v0.z) : 0.f; acc0.w += h0 ? act_fwd<ACT>(v0.w) : 0.f;
                acc1.x += h1 ? act_fwd<ACT>(v1.x) : 0.f; acc1.y += h1 ? act_fwd<ACT>(v1.y) : 0.f;
                acc1.z += h1 ? act_fwd<ACT>(v1.z) : 0.f; acc1.w += h1 ? act_fwd<ACT>(v1.w) : 0.f;
                e += 2 * G;
            }
            for (; e + G < e_end; e += 2 * G) {
                const int s0 = col[e], s1 = col[e + G];
                const unsigned k0 = codes[e], k1 = codes[e + G];
                float4 v0 = reinterpret_cast<const float4 *>(x + (size_t)s0 * h)[c4];
                float4 v1 = reinterpret_cast<const float4 *>(x + (size_t)s1 * h)[c4];
                add_code_rows<F>(v0, tbl, k0, h, c4);
                add_code_rows<F>(v1, tbl, k1, h, c4);
                acc0.x += act_fwd<ACT>(v0.x); acc0.y += act_fwd<ACT>(v0.y); acc0.z += act_fwd<ACT>(v0.z); acc0.w += act_fwd<ACT>(v0.w);
                acc1.x += act_fwd<ACT>(v1.x); acc1.y += act_fwd<ACT>(v1.y); acc1.z += act_fwd<ACT>(v1.z); acc1.w += act_fwd<ACT>(v1.w);
            }
            if (e < e_end) {
                float4 v0 = reinterpret_cast<const float4 *>(x + (size_t)col[e] * h)[c4];
                add_code_rows<F>(v0, tbl, codes[e], h, c4);
                acc0.x += act_fwd<ACT>(v0.x); acc0.y += act_fwd<ACT>(v0.y); acc0.z += act_fwd<ACT>(v0.z); acc0.w += act_fwd<ACT>(v0.w);
            }
            float4 acc = make_float4(acc0.x + acc1.x, acc0.y + acc1.y, acc0.z + acc1.z, acc0.w + acc1.w);
#pragma unroll
            for (int off = LPR; off < WAVE; off <<= 1) {
                acc.x += __shfl_xor(acc.x, off, WAVE); acc.y += __shfl_xor(acc.y, off, WAVE);
                acc.z += __shfl_xor(acc.z, off, WAVE); acc.w += __shfl_xor(acc.w, off, WAVE);
            }
            if (grp == 0) {
                if (add_self) acc4(acc, selfv);
                reinterpret_cast<float4 *>(out + (size_t)d * h)[c4] = acc;
            }
        }
    }
}

// The same aggregation for small destination counts (molecule batches): no LDS copy of the tables.  Staging them costs
// every workgroup a pointer-table hop, a table round trip and two barriers before its first row (about half of an 11 us
// launch when a workgroup owns only a handful of rows); here every wavefront owns ONE row and reads the F table rows of an
// edge from L2 in the same round trip as the edge's source row: rowptr | table pointers -> col, codes -> rows -> store.
// Work split: a wavefront holds G = 64/LPR lane groups and every lane group owns a row of its own (at a molecule's degree of
// two a row split over the groups is one edge per group, a cross-group sum, and twice the wavefronts — two resident sets
// instead of one).  A row with more than AGG_HEAVY edges is left to the whole wavefront afterwards, its edges dealt over the
// groups (a hub walked by one group alone would hold its wavefront for as long as the rest of the launch takes).
constexpr int AGG_HEAVY = 16;

template <int LPR, int ACT, int F>
__global__ __launch_bounds__(ENC_THREADS) void aggregate_encp_direct_fwd_kernel(
    const float *__restrict__ x, const unsigned *__restrict__ codes, const float *const *__restrict__ tables, EncMeta m,
    const int *__restrict__ rowptr, const int *__restrict__ col, long long n_dst, int n, int k, int add_self,
    float *__restrict__ out) {
    const int h = n * k, h4 = h >> 2;
    constexpr int G = WAVE / LPR;
    const int lane = threadIdx.x & 63;
    const int grp = lane / LPR, li = lane - grp * LPR;
    const long long d0 = (((long long)xcd_contiguous_block(blockIdx.x, gridDim.x) * ENC_THREADS + threadIdx.x) >> 6) * G;
    if (d0 >= n_dst) return;                                 // wave-uniform
    // one row: edges e0 + first, e0 + first + step, ... of [e0, e1); coop = the groups of the wavefront share the row
    auto row = [&](long long d, int e0, int e1, int first, int step, bool coop) {
        for (int c4 = li; c4 < h4; c4 += LPR) {
            const int c = (c4 * 4) / k, kk = c4 * 4 - c * k;
            const float *tp[F];                              // this lane's slice of its component's F tables
#pragma unroll
            for (int f = 0; f < F; ++f) tp[f] = tables[c * F + f] + kk - (size_t)m.off[f] * k;  // codes hold off_f + value
            const float4 selfv = reinterpret_cast<const float4 *>(x + (add_self ? (size_t)d * h : 0))[c4];
            float4 acc0 = make_float4(0.f, 0.f, 0.f, 0.f), acc1 = acc0;
            auto message = [&](int s, unsigned code) {
                float4 v = reinterpret_cast<const float4 *>(x + (size_t)s * h)[c4];
#pragma unroll
                for (int f = 0; f < F; ++f) acc4(v, *reinterpret_cast<const float4 *>(tp[f] + (size_t)((code >> (8 * f)) & 255u) * k));
                v.x = act_fwd<ACT>(v.x); v.y = act_fwd<ACT>(v.y); v.z = act_fwd<ACT>(v.z); v.w = act_fwd<ACT>(v.w);
                return v;
            };
            int e = e0 + first;
            if (e0 < e1) {                                   // the first two edges travel together (clamped indices)
                const int i0 = min(e, e1 - 1), i1 = min(e + step, e1 - 1);
                const int s0 = col[i0], s1 = col[i1];
                const unsigned k0 = codes[i0], k1 = codes[i1];
                const float4 v0 = message(s0, k0), v1 = message(s1, k1);
                if (e < e1) acc4(acc0, v0);
                if (e + step < e1) acc4(acc1, v1);
                e += 2 * step;
            }
            for (; e < e1; e += step) acc4(acc0, message(col[e], codes[e]));
            float4 acc = make_float4(acc0.x + acc1.x, acc0.y + acc1.y, acc0.z + acc1.z, acc0.w + acc1.w);
            if (coop) {
#pragma unroll
                for (int off = LPR; off < WAVE; off <<= 1) {
                    acc.x += __shfl_xor(acc.x, off, WAVE); acc.y += __shfl_xor(acc.y, off, WAVE);
                    acc.z += __shfl_xor(acc.z, off, WAVE); acc.w += __shfl_xor(acc.w, off, WAVE);
                }
            }
            if (!coop || grp == 0) {
                if (add_self) acc4(acc, selfv);
                reinterpret_cast<float4 *>(out + (size_t)d * h)[c4] = acc;
            }
        }
    };
    const long long d = d0 + grp;
    const bool valid = d < n_dst;
    const int e_begin = valid ? rowptr[d] : 0, e_end = valid ? rowptr[d + 1] : 0;
    const bool heavy = G > 1 && e_end - e_begin > AGG_HEAVY;
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

// ---- the model-level norm folded into the NEXT layer's aggregation (round 4) -------------------------------------------------
// models.py:208-215 ends a layer with  x = act(norm(y)) (+ skip);  the next layer starts with  s[d] = sum_e (x[src_e] + enc_e) + x[d]
// (messagepassing.py:132-146).  With the norm's batch statistics in an accumulator (the producing GEMM's epilogue) the element-wise
// pass between the two is not a launch of its own: this kernel derives scale / shift in its prologue (the arithmetic of
// bn_apply_acc_kernel: replica totals summed in double, in replica order), applies  act(y * scale + shift) + skip  to every row
// it gathers AND to its own row, which it also writes out once as x (later readers: the skip of an sc_type "last" model, the
// gradient kernels through save_stat, nothing else).  One row per lane group as above; identity message function; rows >= *n_valid
// are padding and read as 0, exactly as the stand-alone pass leaves them.  Bit-identical to the two launches it replaces.
constexpr int AGN_MAX_H = 512;
template <int LPR, int F>
__global__ __launch_bounds__(ENC_THREADS) void aggregate_norm_direct_fwd_kernel(
    const float *__restrict__ y, const float *__restrict__ skip, BnParams p, const int *__restrict__ n_valid, float momentum, float eps,
    const unsigned long long *__restrict__ acc, int norm_act, float *__restrict__ stat, float *__restrict__ xout,
    const unsigned *__restrict__ codes, const float *const *__restrict__ tables, EncMeta m, const int *__restrict__ rowptr,
    const int *__restrict__ col, long long n_dst, int n, int k, int add_self, float *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) float s_scale[AGN_MAX_H], s_shift[AGN_MAX_H];
    const int h = n * k, h4 = h >> 2;
    constexpr int G = WAVE / LPR;
    const int tid = threadIdx.x, lane = tid & 63;
    const int grp = lane / LPR, li = lane - grp * LPR;
    const long long count = n_valid ? min((long long)*n_valid, n_dst) : n_dst;
    const long long d0 = (((long long)xcd_contiguous_block(blockIdx.x, gridDim.x) * ENC_THREADS + tid) >> 6) * G;
    // this lane group's row pointers go in flight before the totals prologue (clamped: a workgroup past the rows still helps nobody,
    // but it must reach the barrier)
    const long long d = d0 + grp;
    const bool valid = d < n_dst;
    const long long dc = valid ? d : 0;
    const int e_begin_raw = rowptr[dc], e_end_raw = rowptr[dc + 1];
    // ---- prologue: scale / shift of every column (one column per thread and round) ----
    const bool owner = blockIdx.x == 0;
    for (int c = tid; c < h; c += ENC_THREADS) {
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
    if (d0 >= n_dst) return;                                 // wave-uniform
    const bool relu = norm_act == PHC_ACT_RELU;
    // one normalised row slice: act(y * scale + shift) (+ skip), 0 for a padding row
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
    auto row = [&](long long dd, int e0, int e1, int first, int step, bool coop) {
        for (int c4 = li; c4 < h4; c4 += LPR) {
            const int c = (c4 * 4) / k, kk = c4 * 4 - c * k;
            const float *tp[F];
#pragma unroll
            for (int f = 0; f < F; ++f) tp[f] = tables[c * F + f] + kk - (size_t)m.off[f] * k;
            const float4 sc = *reinterpret_cast<const float4 *>(s_scale + 4 * c4), sh = *reinterpret_cast<const float4 *>(s_shift + 4 * c4);
            const float4 selfv = xrow(dd, c4, sc, sh);
            float4 acc0 = make_float4(0.f, 0.f, 0.f, 0.f), acc1 = acc0;
            auto message = [&](int s, unsigned code) {
                float4 v = xrow(s, c4, sc, sh);
#pragma unroll
                for (int f = 0; f < F; ++f) acc4(v, *reinterpret_cast<const float4 *>(tp[f] + (size_t)((code >> (8 * f)) & 255u) * k));
                return v;
            };
            int e = e0 + first;
            if (e0 < e1) {
                const int i0 = min(e, e1 - 1), i1 = min(e + step, e1 - 1);
                const int s0 = col[i0], s1 = col[i1];
                const unsigned k0 = codes[i0], k1 = codes[i1];
                const float4 v0 = message(s0, k0), v1 = message(s1, k1);
                if (e < e1) acc4(acc0, v0);
                if (e + step < e1) acc4(acc1, v1);
                e += 2 * step;
            }
            for (; e < e1; e += step) acc4(acc0, message(col[e], codes[e]));
            float4 a4 = make_float4(acc0.x + acc1.x, acc0.y + acc1.y, acc0.z + acc1.z, acc0.w + acc1.w);
            if (coop) {
#pragma unroll
                for (int off = LPR; off < WAVE; off <<= 1) {
                    a4.x += __shfl_xor(a4.x, off, WAVE); a4.y += __shfl_xor(a4.y, off, WAVE);
                    a4.z += __shfl_xor(a4.z, off, WAVE); a4.w += __shfl_xor(a4.w, off, WAVE);
                }
            }
            if (!coop || grp == 0) {
                reinterpret_cast<float4 *>(xout + (size_t)dd * h)[c4] = selfv;        // x itself, once
                if (add_self) acc4(a4, selfv);
                reinterpret_cast<float4 *>(out + (size_t)dd * h)[c4] = a4;
            }
        }
    };
    const int e_begin = valid ? e_begin_raw : 0, e_end = valid ? e_end_raw : 0;
    const bool heavy = G > 1 && e_end - e_begin > AGG_HEAVY;
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

// dx of the same aggregation for an identity message function on a molecule batch (dx[s] = sum of dout over the out-edges
// of s, + dout[s] for the self loop): one source row per lane group as above, no tables, no x.
// NRES > 1: up to NRES further gradients of the same rows (the shares of a skip tensor that feeds every layer, sc_type "first":
// models.py:230-234) are added on the way out — the n-way sum launch in front of the atom-table gradient disappears.
struct DxResiduals {
    const float *p[8];
    int n;
};
template <int LPR, int NRES = 1>
__global__ __launch_bounds__(ENC_THREADS) void aggregate_dx_rows_kernel(
    const float *__restrict__ dout, const int *__restrict__ rowptr_t, const int *__restrict__ col_t, long long n_src,
    long long n_dst, int h, int add_self, float *__restrict__ dx, const float *__restrict__ residual = nullptr, DxResiduals more = {}) {
    const int h4 = h >> 2;
    constexpr int G = WAVE / LPR;
    const int lane = threadIdx.x & 63;
    const int grp = lane / LPR, li = lane - grp * LPR;
    const long long s0 = (((long long)xcd_contiguous_block(blockIdx.x, gridDim.x) * ENC_THREADS + threadIdx.x) >> 6) * G;
    if (s0 >= n_src) return;                                 // wave-uniform
    auto row = [&](long long s, int e0, int e1, int first, int step, bool coop) {
        for (int c4 = li; c4 < h4; c4 += LPR) {
            const bool self_on = add_self && s < n_dst;      // the self row goes in flight first (pointer select, no branch)
            const float4 selfv = reinterpret_cast<const float4 *>(dout + (self_on ? (size_t)s * h : 0))[c4];
            // residual: another gradient of the same rows (the skip connection's share of dx) added on the way out
            const float4 resv = reinterpret_cast<const float4 *>(residual ? residual + (size_t)s * h : dout)[c4];
            float4 morev[NRES > 1 ? NRES : 1];
            if constexpr (NRES > 1) {                        // unconditional loads (an unused slot re-reads the dout row), selected below
#pragma unroll
                for (int q = 0; q < NRES; ++q)
                    morev[q] = reinterpret_cast<const float4 *>(q < more.n ? more.p[q] + (size_t)s * h : dout)[c4];
            }
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            int e = e0 + first;
            if (e0 < e1) {                                   // both column indices in one round trip, both rows in the next
                const int i0 = min(e, e1 - 1), i1 = min(e + step, e1 - 1);
                const int t0 = col_t[i0], t1 = col_t[i1];
                const float4 g0 = reinterpret_cast<const float4 *>(dout + (size_t)t0 * h)[c4];
                const float4 g1 = reinterpret_cast<const float4 *>(dout + (size_t)t1 * h)[c4];
                const bool h0 = e < e1, h1 = e + step < e1;
                acc.x += (h0 ? g0.x : 0.f) + (h1 ? g1.x : 0.f); acc.y += (h0 ? g0.y : 0.f) + (h1 ? g1.y : 0.f);
                acc.z += (h0 ? g0.z : 0.f) + (h1 ? g1.z : 0.f); acc.w += (h0 ? g0.w : 0.f) + (h1 ? g1.w : 0.f);
                e += 2 * step;
            }
            for (; e < e1; e += step) acc4(acc, reinterpret_cast<const float4 *>(dout + (size_t)col_t[e] * h)[c4]);
            if (coop) {
#pragma unroll
                for (int off = LPR; off < WAVE; off <<= 1) {
                    acc.x += __shfl_xor(acc.x, off, WAVE); acc.y += __shfl_xor(acc.y, off, WAVE);
                    acc.z += __shfl_xor(acc.z, off, WAVE); acc.w += __shfl_xor(acc.w, off, WAVE);
                }
            }
            if (!coop || grp == 0) {
                if (self_on) acc4(acc, selfv);
                if (residual) acc4(acc, resv);
                if constexpr (NRES > 1) {
#pragma unroll
                    for (int q = 0; q < NRES; ++q)
                        if (q < more.n) acc4(acc, morev[q]);     // fixed order: slot 0, 1, ...
                }
                reinterpret_cast<float4 *>(dx + (size_t)s * h)[c4] = acc;
            }
        }
    };
    const long long s = s0 + grp;
    const bool valid = s < n_src;
    const int e_begin = valid ? rowptr_t[s] : 0, e_end = valid ? rowptr_t[s + 1] : 0;
    const bool heavy = G > 1 && e_end - e_begin > AGG_HEAVY;
    if (valid && !heavy) row(s, e_begin, e_end, 0, 1, false);
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

// dx of the fused aggregation (source-sorted walk): dx[s] = sum_{e out of s} phi'(x[s] + enc_e) * dout[dst_e] (+ dout[s])
template <int LPR, int ACT>
__global__ __launch_bounds__(ENC_THREADS) void aggregate_enc_dx_kernel(
    const float *__restrict__ dout, const float *__restrict__ x, const long long *__restrict__ bond,
    const float *const *__restrict__ tables, EncMeta m, const int *__restrict__ rowptr_t, const int *__restrict__ col_t,
    const int *__restrict__ perm_t, long long n_src, long long n_dst, int n, int k, int add_self, float *__restrict__ dx) {
    extern __shared__ __attribute__((aligned(16))) float tbl[];
    const int h = n * k, h4 = h >> 2;
    if (ACT != PHC_ACT_IDENTITY) {
        for (int e = threadIdx.x; e < m.R * h; e += ENC_THREADS) {
            const int r = e / h, cidx = e - r * h;
            const int c = cidx / k, kap = cidx - c * k;
            int f = 0;
            while (f + 1 < m.F && r >= m.off[f + 1]) ++f;
            tbl[e] = tables[c * m.F + f][(size_t)(r - m.off[f]) * k + kap];
        }
        __syncthreads();
    }
    constexpr int G = WAVE / LPR;
    const int lane = threadIdx.x & 63;
    const int grp = lane / LPR, li = lane - grp * LPR;
    const long long wave_id = ((long long)xcd_contiguous_block(blockIdx.x, gridDim.x) * ENC_THREADS + threadIdx.x) >> 6;
    const long long n_waves = ((long long)gridDim.x * ENC_THREADS) >> 6;
    const int F = m.F;
    for (long long s = wave_id; s < n_src; s += n_waves) {
        const int e_begin = rowptr_t[s], e_end = rowptr_t[s + 1];
        for (int c4 = li; c4 < h4; c4 += LPR) {
            float4 xs = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ACT != PHC_ACT_IDENTITY) xs = reinterpret_cast<const float4 *>(x + (size_t)s * h)[c4];
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            // the self row goes in flight first (pointer select, no branch) instead of a serial round trip at the end
            const bool self_on = add_self && s < n_dst;
            const float4 selfv = reinterpret_cast<const float4 *>(dout + (self_on ? (size_t)s * h : 0))[c4];
            int e = e_begin + grp;
            if (ACT == PHC_ACT_IDENTITY && e_begin < e_end) {                // wave-uniform
                // identity message function: the first two out-edges of every lane group with clamped indices — both
                // column indices in one round trip, both row gathers in the next
                const int i0 = min(e, e_end - 1), i1 = min(e + G, e_end - 1);
                const int d0 = col_t[i0], d1 = col_t[i1];
                const float4 g0 = reinterpret_cast<const float4 *>(dout + (size_t)d0 * h)[c4];
                const float4 g1 = reinterpret_cast<const float4 *>(dout + (size_t)d1 * h)[c4];
                const bool h0 = e < e_end, h1 = e + G < e_end;
                acc.x += (h0 ? g0.x : 0.f) + (h1 ? g1.x : 0.f); acc.y += (h0 ? g0.y : 0.f) + (h1 ? g1.y : 0.f);
                acc.z += (h0 ? g0.z : 0.f) + (h1 ? g1.z : 0.f); acc.w += (h0 ? g0.w : 0.f) + (h1 ? g1.w : 0.f);
                e += 2 * G;
            }
            for (; e < e_end; e += G) {
                float4 g = reinterpret_cast<const float4 *>(dout + (size_t)col_t[e] * h)[c4];
                if (ACT != PHC_ACT_IDENTITY) {
                    const long long p = perm_t[e];
                    float4 pre = xs;
                    for (int f = 0; f < F; ++f)
                        acc4(pre, reinterpret_cast<const float4 *>(tbl + (m.off[f] + (int)safe_index(bond[p * F + f], m.off[f + 1] - m.off[f])) * h)[c4]);
                    g.x *= act_grad<ACT>(pre.x); g.y *= act_grad<ACT>(pre.y);
                    g.z *= act_grad<ACT>(pre.z); g.w *= act_grad<ACT>(pre.w);
                }
                acc4(acc, g);
            }
#pragma unroll
            for (int off = LPR; off < WAVE; off <<= 1) {
                acc.x += __shfl_xor(acc.x, off, WAVE); acc.y += __shfl_xor(acc.y, off, WAVE);
                acc.z += __shfl_xor(acc.z, off, WAVE); acc.w += __shfl_xor(acc.w, off, WAVE);
            }
            if (grp == 0) {
                if (self_on) acc4(acc, selfv);
                reinterpret_cast<float4 *>(dx + (size_t)s * h)[c4] = acc;
            }
        }
    }
}

// ---- table gradients on the matrix cores -------------------------------------------------------
// dT[r][j] = sum over items t of onehot[t][r] * g[t][j]  ==  (onehot^T . g), a GEMM whose K dimension is
// the item index (nodes for the atom encoder, edges for the bond encoders).  The one-hot operand is never
// stored: each 64-item step scatters F ones per item into an LDS tile; g rows are gathered into a second
// tile; v_mfma_f32_32x32x2_f32 accumulates the 64 x 64 (table rows x columns) tile.  Split over item
// ranges into fixed-order slabs => no atomics, bitwise reproducible (the reference runs one
// embedding_dense_backward — a device sort — per table).
//   MODE 0: item = node i,  rows off_f + idx[i,f],     g = dout[i]
//   MODE 1: item = edge e,  rows off_f + bond[e,f],    g = phi'(x[src_e] + enc_e) * dout[dst_e]
constexpr int TG_T = 64;          // tile edge (table rows x feature columns)
constexpr int TG_IB = 64;         // items per step

template <int MODE, int ACT>
__device__ __forceinline__ void table_grad_body(
    const float *__restrict__ dout, const long long *__restrict__ idx, const EncMeta &m, long long n_items, long long items_per_split,
    int h, int tiles_j, const long long *__restrict__ src, const long long *__restrict__ dst, const float *__restrict__ x,
    const float *const *__restrict__ tables, int n, int k, int vec4, float *__restrict__ slabs, int tile, long long split,
    float *__restrict__ smem) {
    float *oh = smem;                                                   // one-hot tile  [item][table row]
    float *gs = smem + TG_IB * TG_T;                                    // gradient tile [item][column]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int r = lane & 31, hh = lane >> 5;
    const int ti = tile / tiles_j, tj = tile - ti * tiles_j;
    const int r0 = ti * TG_T, j0 = tj * TG_T;
    const long long t_begin = split * items_per_split;
    const long long t_end = min(n_items, t_begin + items_per_split);
    const int F = m.F;

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    for (long long tb = t_begin; tb < t_end; tb += TG_IB) {
        __syncthreads();
        // zero the one-hot tile (16-byte stores)
        for (int e = tid; e < TG_IB * TG_T / 4; e += ENC_THREADS) reinterpret_cast<float4 *>(oh)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        // gather the gradient rows: thread -> (item = e / 16, 4 columns)
        if (!vec4) {                                         // widths that are not a multiple of 4 floats
            for (int e = tid; e < TG_IB * TG_T; e += ENC_THREADS) {
                const int item = e / TG_T, cc = e - item * TG_T;
                const long long t = tb + item;
                const int col = j0 + cc;
                float g = 0.f;
                if (t < t_end && col < h) {
                    g = dout[(size_t)(MODE == 0 ? t : dst[t]) * h + col];
                    if (MODE == 1 && ACT != PHC_ACT_IDENTITY) {
                        float pre = x[(size_t)src[t] * h + col];
                        const int c = col / k, kap = col - c * k;
                        for (int f = 0; f < F; ++f)
                            pre += tables[c * F + f][(size_t)safe_index(idx[t * F + f], m.off[f + 1] - m.off[f]) * k + kap];
                        g *= act_grad<ACT>(pre);
                    }
                }
                gs[e] = g;
            }
        } else
        for (int e = tid; e < TG_IB * (TG_T / 4); e += ENC_THREADS) {
            const int item = e / (TG_T / 4), c4 = e - item * (TG_T / 4);
            const long long t = tb + item;
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
            const int col = j0 + c4 * 4;
            if (t < t_end && col < h) {
                const long long row = MODE == 0 ? t : dst[t];
                g = *reinterpret_cast<const float4 *>(dout + (size_t)row * h + col);
                if (MODE == 1 && ACT != PHC_ACT_IDENTITY) {
                    float4 pre = *reinterpret_cast<const float4 *>(x + (size_t)src[t] * h + col);
                    const int c = col / k, kap = col - c * k;
                    for (int f = 0; f < F; ++f) {
                        const float4 tv = *reinterpret_cast<const float4 *>(
                            tables[c * F + f] + (size_t)safe_index(idx[t * F + f], m.off[f + 1] - m.off[f]) * k + kap);
                        pre.x += tv.x; pre.y += tv.y; pre.z += tv.z; pre.w += tv.w;
                    }
                    g.x *= act_grad<ACT>(pre.x); g.y *= act_grad<ACT>(pre.y);
                    g.z *= act_grad<ACT>(pre.z); g.w *= act_grad<ACT>(pre.w);
                }
            }
            *reinterpret_cast<float4 *>(gs + item * TG_T + c4 * 4) = g;
        }
        __syncthreads();
        // scatter the ones: thread -> (item, feature); the F rows of one item are distinct
        for (int e = tid; e < TG_IB * F; e += ENC_THREADS) {
            const int item = e / F, f = e - item * F;
            const long long t = tb + item;
            if (t < t_end) {
                const int rr = m.off[f] + (int)idx[t * F + f] - r0;
                if (rr >= 0 && rr < TG_T) oh[item * TG_T + rr] = 1.f;
            }
        }
        __syncthreads();
        const float *pa = oh + hh * TG_T + wi * 32 + r;
        const float *pb = gs + hh * TG_T + wj * 32 + r;
#pragma unroll 8
        for (int it = 0; it < TG_IB; it += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[it * TG_T], pb[it * TG_T], acc, 0, 0, 0);
    }
    float *slab = slabs + (size_t)split * m.R * h;
    const int col = j0 + wj * 32 + r;
    if (col < h) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = r0 + wi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
            if (row < m.R) slab[(size_t)row * h + col] = acc[reg];
        }
    }
}

// Row-contiguous forms (the gradient operand is read in place, no gather):
//   MODE 0: item = node i, ones at rows off_f + idx[i, f], g = dout[i]                       (atom encoder)
//   MODE 2: item = destination node d, COUNTS of the table rows among d's in-edges, g = dout[d]
//           (bond encoders with an identity message function: d_enc[e] = dout[dst_e] for every in-edge, so
//           dT = C^T . dout with C[d][r] = #{in-edges of d whose code has row r}; C is rebuilt per 64-node block from the
//           destination-sorted rowptr / packed codes and never stored.  Reads n_dst rows instead of gathering E.)
// The next block's operands are loaded into registers before the current block's MFMAs, clamped instead of branched, so
// one workgroup keeps its own loads in flight behind the matrix work.  MODE 2: the in-edges of 64 consecutive nodes are
// one contiguous span of `codes`; it is staged in LDS (first TG_CODES of it; longer spans fall back to global reads) and
// thread (node, table row) counts its matches — plain stores, no read-modify-write, no per-node serial loop.
constexpr int TG_CODES = 1024;

template <int MODE>
__device__ __forceinline__ void table_grad_rows_body(
    const float *__restrict__ dout, const long long *__restrict__ idx, const int *__restrict__ rowptr,
    const unsigned *__restrict__ codes, long long n_edges, const EncMeta &m, long long n_items, long long items_per_split, int h,
    int tiles_j, int vec4, float *__restrict__ slabs, int tile, long long split, float *__restrict__ smem) {
    static_assert(MODE == 0 || MODE == 2, "row-contiguous modes");
    __shared__ unsigned cbuf[MODE == 2 ? TG_CODES : 1];                 // packed codes of the block's in-edge span
    __shared__ int rp[MODE == 2 ? TG_IB + 1 : 1];                       // row pointers of the block's nodes
    float *oh = smem;                                                   // one-hot / count tile [item][table row]
    float *gs = smem + TG_IB * TG_T;                                    // gradient tile        [item][column]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int r = lane & 31, hh = lane >> 5;
    const int ti = tile / tiles_j, tj = tile - ti * tiles_j;
    const int r0 = ti * TG_T, j0 = tj * TG_T;
    const long long t_begin = split * items_per_split;
    const long long t_end = min(n_items, t_begin + items_per_split);
    const int F = m.F;

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    if (t_begin < t_end) {
        const int c4 = tid & 15, row0 = tid >> 4;
        const int cj = min(j0 + c4 * 4, h - 4);
        const bool col_ok = j0 + c4 * 4 < h;
        const int RT = min(TG_T, m.R - r0);                  // table rows of this tile that exist
        f32x4 gv[4];
        int rpv = 0;                                         // MODE 2: rowptr[min(tb + tid, t_end)], tid <= 64
        unsigned cv[4] = {0u, 0u, 0u, 0u};                   // MODE 2: codes[span start + tid + 256 q]
        int rows[4] = {-1, -1, -1, -1};                      // MODE 0: table rows of (item, feature) pairs tid + 256 q
        auto prefetch_rows = [&](long long tb) {
            if (vec4) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const long long t = min(tb + row0 + 16 * q, t_end - 1);
                    gv[q] = *reinterpret_cast<const f32x4 *>(dout + (size_t)t * h + cj);
                }
            }
            if (MODE == 2) {
                rpv = rowptr[min(tb + min(tid, TG_IB), t_end)];
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int e = tid + ENC_THREADS * q;
                    const int item = e / F, f = e - item * F;
                    const bool live = e < TG_IB * F && tb + item < t_end;
                    const long long t = min(tb + item, t_end - 1);
                    const int v = (int)idx[t * F + (live ? f : 0)];
                    rows[q] = live ? m.off[f] + v - r0 : -1;
                }
            }
        };
        auto prefetch_codes = [&](long long span0) {         // MODE 2
#pragma unroll
            for (int q = 0; q < 4; ++q) cv[q] = codes[min(span0 + tid + ENC_THREADS * q, n_edges - 1)];
        };
        // rows of the tile beyond RT (and, MODE 0, everything) start from zero; MODE 2 rewrites [0, RT) every block
#pragma unroll
        for (int q = 0; q < 4; ++q) reinterpret_cast<float4 *>(oh)[tid + ENC_THREADS * q] = make_float4(0.f, 0.f, 0.f, 0.f);
        prefetch_rows(t_begin);
        if (MODE == 2) prefetch_codes(rowptr[t_begin]);
        for (long long tb = t_begin; tb < t_end; tb += TG_IB) {
            __syncthreads();
            if (MODE == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) reinterpret_cast<float4 *>(oh)[tid + ENC_THREADS * q] = make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) cbuf[tid + ENC_THREADS * q] = cv[q];
                if (tid <= TG_IB) rp[tid] = rpv;
            }
            if (vec4) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int item = row0 + 16 * q;
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    *reinterpret_cast<f32x4 *>(gs + item * TG_T + c4 * 4) = (tb + item < t_end && col_ok) ? gv[q] : z;
                }
            } else {
                for (int e = tid; e < TG_IB * TG_T; e += ENC_THREADS) {
                    const int item = e / TG_T, cc = e - item * TG_T;
                    gs[e] = (tb + item < t_end && j0 + cc < h) ? dout[(size_t)(tb + item) * h + j0 + cc] : 0.f;
                }
            }
            __syncthreads();
            if (MODE == 2) {
                const int span0 = rp[0];
                for (int p = tid; p < TG_IB * RT; p += ENC_THREADS) {
                    const int item = p / RT, rr = p - item * RT;
                    const unsigned row = (unsigned)(r0 + rr);
                    int f = 0;
                    for (int q = 1; q < F; ++q) f += (int)row >= m.off[q];
                    const int sh = 8 * f;
                    float cnt = 0.f;
                    const int a = rp[item], b = rp[item + 1];
                    for (int e = a; e < b; ++e) {
                        const unsigned code = e - span0 < TG_CODES ? cbuf[e - span0] : codes[e];
                        cnt += ((code >> sh) & 255u) == row ? 1.f : 0.f;
                    }
                    oh[item * TG_T + rr] = cnt;
                }
                const long long next0 = rp[TG_IB];           // = rowptr[min(tb + 64, t_end)]: where the next block's span starts
                prefetch_rows(tb + TG_IB);                   // past the end: clamped re-reads, never used
                prefetch_codes(next0);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {                // the F rows of one item are distinct
                    const int e = tid + ENC_THREADS * q;
                    if (rows[q] >= 0 && rows[q] < TG_T) oh[(e / F) * TG_T + rows[q]] = 1.f;
                }
                prefetch_rows(tb + TG_IB);
            }
            __syncthreads();
            const float *pa = oh + hh * TG_T + wi * 32 + r;
            const float *pb = gs + hh * TG_T + wj * 32 + r;
#pragma unroll 8
            for (int it = 0; it < TG_IB; it += 2)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[it * TG_T], pb[it * TG_T], acc, 0, 0, 0);
        }
    }
    float *slab = slabs + (size_t)split * m.R * h;
    const int col = j0 + wj * 32 + r;
    if (col < h) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = r0 + wi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
            if (row < m.R) slab[(size_t)row * h + col] = acc[reg];
        }
    }
}

// ---- atom-table gradients, register-resident (the form of dH_direct_body, phm_grad_device.h) ---------------------------------
// dT = O^T . g with O[i][t] = 1 iff table row t = off_f + idx[i, f] for the feature f that owns row t: the contraction index is
// the node, so v_mfma_f32_32x32x2_f32 takes both operands straight from global memory — half-wave hh handles node i + hh, lane r
// holds g[i + hh][j0 + 2r, 2r + 1] (one coalesced 256-byte row per half-wave) and, per 32-row M block q, the ONE index
// idx[i + hh, f(t)] of the feature that owns its table row t = r0 + 32 q + r (a 4-byte load, same cache line for the whole
// half-wave): A = (index == t - off_f) ? 1 : 0.  No LDS tile, no barrier per 64 items (the LDS form spends four), nothing scattered.
// A workgroup = 4 wavefronts on 4 node quarters of one (M group of TD_MB blocks, 64-column tile, split); they meet once through LDS.
constexpr int TD_MB = 3;                                 // M blocks per tile: 96 table rows
// node pairs per load batch (two batches alternate).  Deeper batches (8 for atoms, 6 for bonds, so that a batch's MFMAs outlast a
// memory round trip) cost registers (228 -> 256) and made the launch 8 us slower at molecule scale: 4 and 4.
#ifndef PHC_TD_PFA
#define PHC_TD_PFA 4
#endif
#ifndef PHC_TD_PFB
#define PHC_TD_PFB 4
#endif
constexpr int TD_PFA = PHC_TD_PFA, TD_PFB = PHC_TD_PFB;

__device__ __forceinline__ void atom_direct_body(const float *__restrict__ dout, const long long *__restrict__ idx, const EncMeta &m,
                                                 long long n_items, long long items_per_split, int h, int tiles_j, float *__restrict__ slabs,
                                                 int tile, long long split, float *__restrict__ smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int mg = tile / tiles_j, tj = tile - mg * tiles_j;
    const int r0 = mg * 32 * TD_MB, j0 = tj * 64;
    const int F = m.F;
    const long long t_begin = split * items_per_split, t_end = min(n_items, t_begin + items_per_split);
    const long long rq = items_per_split >> 2;           // items_per_split % 8 == 0: quarters start on even nodes
    const long long w_begin = min(t_begin + wave * rq, t_end), w_end = min(w_begin + rq, t_end);
    // the feature that owns this lane's table row in each M block, as a byte offset into an index row, and the value to match
    unsigned foff[TD_MB];
    int val[TD_MB];
#pragma unroll
    for (int q = 0; q < TD_MB; ++q) {
        const int t = r0 + 32 * q + r;
        int f = 0;
        for (int u = 1; u < F; ++u) f += t >= m.off[u];
        int base = 0;
        for (int u = 0; u < F; ++u) base = u == f ? m.off[u] : base;            // uniform walk of the by-value table
        foff[q] = (unsigned)f * 8u;
        val[q] = t < m.R ? t - base : -1;                // rows past the tables match nothing (indices are >= 0)
    }
    f32x16 acc[TD_MB][2];
#pragma unroll
    for (int q = 0; q < TD_MB; ++q)
#pragma unroll
        for (int p2 = 0; p2 < 2; ++p2)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[q][p2][i] = 0.f;
    const long long last = max(t_end - 1, 0LL);
    // (h need not be a multiple of 64 — the published 180 / 200-wide models: a lane whose two columns lie past the row reads the
    // tile's first columns instead, in bounds, and stores nothing; h is even, a pair never straddles the end)
    const bool clive = j0 + 2 * r < h;
    const unsigned goff = (unsigned)(clive ? j0 + 2 * r : j0) * 4u;
    f32x2 ga[TD_PFA], gb[TD_PFA];
    int ia[TD_PFA][TD_MB], ib[TD_PFA][TD_MB];
    auto load = [&](f32x2 (&G)[TD_PFA], int (&I)[TD_PFA][TD_MB], long long row0) {
#pragma unroll
        for (int t = 0; t < TD_PFA; ++t) {
            const long long g = min(row0 + 2 * t + hh, last);               // clamped: nodes past the range are read and masked
            G[t] = *reinterpret_cast<const f32x2 *>(reinterpret_cast<const char *>(dout + (size_t)g * h) + goff);
            const char *irow = reinterpret_cast<const char *>(idx + (size_t)g * F);
#pragma unroll
            for (int q = 0; q < TD_MB; ++q) I[t][q] = *reinterpret_cast<const int *>(irow + foff[q]);
        }
    };
    auto pairs = [&](const f32x2 (&G)[TD_PFA], const int (&I)[TD_PFA][TD_MB], long long row0) {
#pragma unroll
        for (int t = 0; t < TD_PFA; ++t) {
            const bool live = row0 + 2 * t + hh < w_end;
            f32x2 gv = G[t];
            gv[0] = live ? gv[0] : 0.f;
            gv[1] = live ? gv[1] : 0.f;
#pragma unroll
            for (int q = 0; q < TD_MB; ++q) {
                const float a = I[t][q] == val[q] ? 1.f : 0.f;
                acc[q][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, gv[0], acc[q][0], 0, 0, 0);
                acc[q][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, gv[1], acc[q][1], 0, 0, 0);
            }
        }
    };
    if (w_begin < w_end) {
        load(ga, ia, w_begin);
        for (long long row = w_begin; row < w_end; row += 4 * TD_PFA) {
            load(gb, ib, row + 2 * TD_PFA);
            __builtin_amdgcn_sched_barrier(0);
            pairs(ga, ia, row);
            __builtin_amdgcn_sched_barrier(0);
            load(ga, ia, row + 4 * TD_PFA);
            __builtin_amdgcn_sched_barrier(0);
            pairs(gb, ib, row + 2 * TD_PFA);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // the four wavefronts meet: per M block, every wavefront parks it in LDS ([wave][reg][lane] float2 = 32 KB), then wavefront
    // w sums registers 4w .. 4w+3 of all four in the fixed order 0, 1, 2, 3 and writes those table rows of the slab
    float *slab = slabs + (size_t)split * m.R * h;
    f32x2 *slots = reinterpret_cast<f32x2 *>(smem);
#pragma unroll
    for (int q = 0; q < TD_MB; ++q) {
        if (q) __syncthreads();
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const f32x2 v = {acc[q][0][reg], acc[q][1][reg]};
            slots[(wave * 16 + reg) * 64 + lane] = v;
        }
        __syncthreads();
        f32x2 part[4][4];
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2)
#pragma unroll
            for (int e = 0; e < 4; ++e) part[w2][e] = slots[(w2 * 16 + 4 * wave + e) * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const f32x2 v = ((part[0][e] + part[1][e]) + part[2][e]) + part[3][e];
            const int row = r0 + 32 * q + e + 8 * wave + 4 * hh;            // reg = 4 wave + e
            if (row < m.R && clive) *reinterpret_cast<f32x2 *>(slab + (size_t)row * h + j0 + 2 * r) = v;
        }
    }
}

template <int MODE, int ACT>
__global__ __launch_bounds__(ENC_THREADS) void table_grad_mfma_kernel(
    const float *__restrict__ dout, const long long *__restrict__ idx, EncMeta m, long long n_items, long long items_per_split,
    int h, int tiles_j, const long long *__restrict__ src, const long long *__restrict__ dst, const float *__restrict__ x,
    const float *const *__restrict__ tables, int n, int k, int vec4, float *__restrict__ slabs) {
    __shared__ __attribute__((aligned(16))) float smem[2 * TG_IB * TG_T];
    if (MODE == 0)
        table_grad_rows_body<0>(dout, idx, nullptr, nullptr, 0, m, n_items, items_per_split, h, tiles_j, vec4, slabs, blockIdx.x,
                                blockIdx.y, smem);
    else
        table_grad_body<MODE, ACT>(dout, idx, m, n_items, items_per_split, h, tiles_j, src, dst, x, tables, n, k, vec4, slabs, blockIdx.x,
                                   blockIdx.y, smem);
}

// Several layers' bond-table gradients in one launch (phc_table_grads_batched): the layers share the edge list and
// the integer bond features, only dout / x / tables / output differ; blockIdx.z picks the layer.
constexpr int TG_JOBS_PER_LAUNCH = 8;
struct TgJobDev {
    const float *dout, *x;
    const float *const *tables;
    float *slabs, *flat;
};
struct TgBatch {
    TgJobDev j[TG_JOBS_PER_LAUNCH];
};

template <int ACT>
__global__ __launch_bounds__(ENC_THREADS) void table_grad_mfma_batched_kernel(
    TgBatch b, const long long *__restrict__ idx, EncMeta m, long long n_items, long long items_per_split, int h, int tiles_j,
    const long long *__restrict__ src, const long long *__restrict__ dst, int n, int k, int vec4) {
    const TgJobDev &jb = b.j[blockIdx.z];
    __shared__ __attribute__((aligned(16))) float smem[2 * TG_IB * TG_T];
    table_grad_body<1, ACT>(jb.dout, idx, m, n_items, items_per_split, h, tiles_j, src, dst, jb.x, jb.tables, n, k, vec4, jb.slabs,
                            blockIdx.x, blockIdx.y, smem);
}

// several integer encoders over ONE index tensor (e.g. the bond encoders of all layers on the [E, h] tensor path):
// blockIdx.z picks the encoder's gradient rows / output; MODE 0 body
__global__ __launch_bounds__(ENC_THREADS) void table_grad_rows_batched_kernel(
    TgBatch b, const long long *__restrict__ idx, EncMeta m, long long n_items, long long items_per_split, int h, int tiles_j,
    int vec4) {
    const TgJobDev &jb = b.j[blockIdx.z];
    __shared__ __attribute__((aligned(16))) float smem[2 * TG_IB * TG_T];
    table_grad_rows_body<0>(jb.dout, idx, nullptr, nullptr, 0, m, n_items, items_per_split, h, tiles_j, vec4, jb.slabs, blockIdx.x,
                            blockIdx.y, smem);
}

// identity message function: per destination node with in-edge row counts (MODE 2), n_items = n_dst
__global__ __launch_bounds__(ENC_THREADS) void table_grad_counted_batched_kernel(
    TgBatch b, const int *__restrict__ rowptr, const unsigned *__restrict__ codes, long long n_edges, EncMeta m, long long n_items,
    long long items_per_split, int h, int tiles_j, int vec4) {
    const TgJobDev &jb = b.j[blockIdx.z];
    __shared__ __attribute__((aligned(16))) float smem[2 * TG_IB * TG_T];
    table_grad_rows_body<2>(jb.dout, nullptr, rowptr, codes, n_edges, m, n_items, items_per_split, h, tiles_j, vec4, jb.slabs, blockIdx.x,
                            blockIdx.y, smem);
}

__device__ __forceinline__ void table_grad_reduce_body(const float *__restrict__ slabs, int num_slabs, int R, int n, int k,
                                                       float *__restrict__ flat, int bx);

__global__ __launch_bounds__(256) void table_grad_reduce_batched_kernel(TgBatch b, int num_slabs, int R, int n, int k) {
    const TgJobDev &jb = b.j[blockIdx.y];
    table_grad_reduce_body(jb.slabs, num_slabs, R, n, k, jb.flat, blockIdx.x);
}

// flat[c][r][kap] = sum_b slab[b][r][c*k+kap]    (flat is [c][f][V_f][k] because r = off_f + v)
// 64 elements x 4 slab lanes per workgroup, fixed summation tree.
__global__ __launch_bounds__(256) void table_grad_reduce_kernel(const float *__restrict__ slabs, int num_slabs, int R, int n, int k,
                                                               float *__restrict__ flat) {
    table_grad_reduce_body(slabs, num_slabs, R, n, k, flat, blockIdx.x);
}

__device__ __forceinline__ void table_grad_reduce_body(const float *__restrict__ slabs, int num_slabs, int R, int n, int k,
                                                       float *__restrict__ flat, int bx) {
    __shared__ float part[4][64];
    const int h = n * k;
    const int el = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int e = bx * 64 + el;
    float s = 0.f;
    if (e < R * h) {
        // eight of this lane's slabs per round trip (clamped, selected; ascending order)
        for (int b0 = sl; b0 < num_slabs; b0 += 32) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = slabs[(size_t)min(b0 + 4 * q, num_slabs - 1) * R * h + e];
#pragma unroll
            for (int q = 0; q < 8; ++q) s += b0 + 4 * q < num_slabs ? v[q] : 0.f;
        }
    }
    part[sl][el] = s;
    __syncthreads();
    if (sl == 0 && e < R * h) {
        const int r = e / h, j = e - r * h;
        const int c = j / k, kap = j - c * k;
        flat[((size_t)c * R + r) * k + kap] = ((part[0][el] + part[1][el]) + part[2][el]) + part[3][el];
    }
}

// ---- bond-table gradients (identity message function), register-resident ------------------------------------------------------
// dT_l = C^T . dout_l for every message-passing layer l, C[d][t] = number of in-edges of node d whose code names table row t
// (code_counts of phc_csr_build_pair_codes: built once per batch by the structure build, [n_dst][CR]).  One M block (<= 32 table
// rows); the layers of a group of TD_JG share the A operand: per node pair one 4-byte load of C and one 8-byte load of every
// layer's dout feed 2 * TD_JG MFMAs.  Rows past the range are masked in A only (a zero row of C annihilates whatever dout holds).
constexpr int TD_JG = 4;                                 // layers per tile

__device__ __forceinline__ void bond_direct_body(const TgBatch &jobs, int job0, int njobs, const float *__restrict__ counts, int CR, int R,
                                                 long long n_items, long long items_per_split, int h, int tj, long long split,
                                                 float *__restrict__ smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int j0 = tj * 64;
    const long long t_begin = split * items_per_split, t_end = min(n_items, t_begin + items_per_split);
    const long long rq = items_per_split >> 2;
    const long long w_begin = min(t_begin + wave * rq, t_end), w_end = min(w_begin + rq, t_end);
    const float *dptr[TD_JG];
#pragma unroll
    for (int u = 0; u < TD_JG; ++u) dptr[u] = jobs.j[min(job0 + u, job0 + njobs - 1)].dout;       // (a short group re-reads its last layer)
    f32x16 acc[TD_JG][2];
#pragma unroll
    for (int u = 0; u < TD_JG; ++u)
#pragma unroll
        for (int p2 = 0; p2 < 2; ++p2)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[u][p2][i] = 0.f;
    const long long last = max(t_end - 1, 0LL);
    // (h need not be a multiple of 64 — the published 180 / 200-wide models: a lane whose two columns lie past the row reads the
    // tile's first columns instead, in bounds, and stores nothing; h is even, a pair never straddles the end)
    const bool clive = j0 + 2 * r < h;
    const unsigned goff = (unsigned)(clive ? j0 + 2 * r : j0) * 4u;
    const bool ccol = r < CR;
    const int cr = ccol ? r : 0;
    float ca[TD_PFB], cb[TD_PFB];
    f32x2 ga[TD_PFB][TD_JG], gb[TD_PFB][TD_JG];
    auto load = [&](float (&Cv)[TD_PFB], f32x2 (&G)[TD_PFB][TD_JG], long long row0) {
#pragma unroll
        for (int t = 0; t < TD_PFB; ++t) {
            const long long g = min(row0 + 2 * t + hh, last);
            Cv[t] = counts[(size_t)g * CR + cr];
#pragma unroll
            for (int u = 0; u < TD_JG; ++u)
                G[t][u] = *reinterpret_cast<const f32x2 *>(reinterpret_cast<const char *>(dptr[u] + (size_t)g * h) + goff);
        }
    };
    auto pairs = [&](const float (&Cv)[TD_PFB], const f32x2 (&G)[TD_PFB][TD_JG], long long row0) {
#pragma unroll
        for (int t = 0; t < TD_PFB; ++t) {
            const float a = (ccol && row0 + 2 * t + hh < w_end) ? Cv[t] : 0.f;
#pragma unroll
            for (int u = 0; u < TD_JG; ++u) {
                acc[u][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, G[t][u][0], acc[u][0], 0, 0, 0);
                acc[u][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, G[t][u][1], acc[u][1], 0, 0, 0);
            }
        }
    };
    if (w_begin < w_end) {
        load(ca, ga, w_begin);
        for (long long row = w_begin; row < w_end; row += 4 * TD_PFB) {
            load(cb, gb, row + 2 * TD_PFB);
            __builtin_amdgcn_sched_barrier(0);
            pairs(ca, ga, row);
            __builtin_amdgcn_sched_barrier(0);
            load(ca, ga, row + 4 * TD_PFB);
            __builtin_amdgcn_sched_barrier(0);
            pairs(cb, gb, row + 2 * TD_PFB);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    f32x2 *slots = reinterpret_cast<f32x2 *>(smem);
#pragma unroll
    for (int u = 0; u < TD_JG; ++u) {
        if (u) __syncthreads();
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const f32x2 v = {acc[u][0][reg], acc[u][1][reg]};
            slots[(wave * 16 + reg) * 64 + lane] = v;
        }
        __syncthreads();
        f32x2 part[4][4];
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2)
#pragma unroll
            for (int e = 0; e < 4; ++e) part[w2][e] = slots[(w2 * 16 + 4 * wave + e) * 64 + lane];
        if (u < njobs) {                                 // uniform
            float *slab = jobs.j[job0 + u].slabs + (size_t)split * R * h;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const f32x2 v = ((part[0][e] + part[1][e]) + part[2][e]) + part[3][e];
                const int row = e + 8 * wave + 4 * hh;
                if (row < R && clive) *reinterpret_cast<f32x2 *>(slab + (size_t)row * h + j0 + 2 * r) = v;
            }
        }
    }
}

// ---- several encoders over ONE index tensor (phc_encode_bwd_batched), <= 32 table rows, register-resident ----------------------
// dT_j = O^T . dout_j for every encoder j of a group of TD_JG: O[i][t] = 1 iff t = off_f + idx[i, f] for the feature f owning row t
// (generated per item from ONE 4-byte index load, as in atom_direct_body); the group shares it (bond_direct_body's B side).
__device__ __forceinline__ void enc_direct_body(const TgBatch &jobs, int job0, int njobs, const long long *__restrict__ idx, const EncMeta &m,
                                                long long n_items, long long items_per_split, int h, int tj, long long split,
                                                float *__restrict__ smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int j0 = tj * 64, F = m.F, R = m.R;
    const long long t_begin = split * items_per_split, t_end = min(n_items, t_begin + items_per_split);
    const long long rq = items_per_split >> 2;
    const long long w_begin = min(t_begin + wave * rq, t_end), w_end = min(w_begin + rq, t_end);
    int f = 0;
    for (int u = 1; u < F; ++u) f += r >= m.off[u];
    int base = 0;
    for (int u = 0; u < F; ++u) base = u == f ? m.off[u] : base;
    const unsigned foff = (unsigned)f * 8u;
    const int val = r < R ? r - base : -1;
    const float *dptr[TD_JG];
#pragma unroll
    for (int u = 0; u < TD_JG; ++u) dptr[u] = jobs.j[min(job0 + u, job0 + njobs - 1)].dout;
    f32x16 acc[TD_JG][2];
#pragma unroll
    for (int u = 0; u < TD_JG; ++u)
#pragma unroll
        for (int p2 = 0; p2 < 2; ++p2)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[u][p2][i] = 0.f;
    const long long last = max(t_end - 1, 0LL);
    // (h need not be a multiple of 64 — the published 180 / 200-wide models: a lane whose two columns lie past the row reads the
    // tile's first columns instead, in bounds, and stores nothing; h is even, a pair never straddles the end)
    const bool clive = j0 + 2 * r < h;
    const unsigned goff = (unsigned)(clive ? j0 + 2 * r : j0) * 4u;
    int ia[TD_PFB], ib[TD_PFB];
    f32x2 ga[TD_PFB][TD_JG], gb[TD_PFB][TD_JG];
    auto load = [&](int (&I)[TD_PFB], f32x2 (&G)[TD_PFB][TD_JG], long long row0) {
#pragma unroll
        for (int t = 0; t < TD_PFB; ++t) {
            const long long g = min(row0 + 2 * t + hh, last);
            I[t] = *reinterpret_cast<const int *>(reinterpret_cast<const char *>(idx + (size_t)g * F) + foff);
#pragma unroll
            for (int u = 0; u < TD_JG; ++u)
                G[t][u] = *reinterpret_cast<const f32x2 *>(reinterpret_cast<const char *>(dptr[u] + (size_t)g * h) + goff);
        }
    };
    auto pairs = [&](const int (&I)[TD_PFB], const f32x2 (&G)[TD_PFB][TD_JG], long long row0) {
#pragma unroll
        for (int t = 0; t < TD_PFB; ++t) {
            const float a = (I[t] == val && row0 + 2 * t + hh < w_end) ? 1.f : 0.f;
#pragma unroll
            for (int u = 0; u < TD_JG; ++u) {
                acc[u][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, G[t][u][0], acc[u][0], 0, 0, 0);
                acc[u][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, G[t][u][1], acc[u][1], 0, 0, 0);
            }
        }
    };
    if (w_begin < w_end) {
        load(ia, ga, w_begin);
        for (long long row = w_begin; row < w_end; row += 4 * TD_PFB) {
            load(ib, gb, row + 2 * TD_PFB);
            __builtin_amdgcn_sched_barrier(0);
            pairs(ia, ga, row);
            __builtin_amdgcn_sched_barrier(0);
            load(ia, ga, row + 4 * TD_PFB);
            __builtin_amdgcn_sched_barrier(0);
            pairs(ib, gb, row + 2 * TD_PFB);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    f32x2 *slots = reinterpret_cast<f32x2 *>(smem);
#pragma unroll
    for (int u = 0; u < TD_JG; ++u) {
        if (u) __syncthreads();
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const f32x2 v = {acc[u][0][reg], acc[u][1][reg]};
            slots[(wave * 16 + reg) * 64 + lane] = v;
        }
        __syncthreads();
        f32x2 part[4][4];
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2)
#pragma unroll
            for (int e = 0; e < 4; ++e) part[w2][e] = slots[(w2 * 16 + 4 * wave + e) * 64 + lane];
        if (u < njobs) {                                 // uniform
            float *slab = jobs.j[job0 + u].slabs + (size_t)split * R * h;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const f32x2 v = ((part[0][e] + part[1][e]) + part[2][e]) + part[3][e];
                const int row = e + 8 * wave + 4 * hh;
                if (row < R && clive) *reinterpret_cast<f32x2 *>(slab + (size_t)row * h + j0 + 2 * r) = v;
            }
        }
    }
}

__global__ __launch_bounds__(ENC_THREADS) void enc_direct_batched_kernel(TgBatch b, int num_jobs, const long long *__restrict__ idx, EncMeta m,
                                                                        long long n_items, long long items_per_split, int h,
                                                                        int tiles_j, int num_slabs) {
    __shared__ __attribute__((aligned(16))) float smem[4 * 16 * 64 * 2];           // exchange slots, 32 KB
    const unsigned per_group = (unsigned)tiles_j * (unsigned)num_slabs;
    const int jg = blockIdx.x / per_group;
    const unsigned local = blockIdx.x - jg * per_group;
    const int job0 = jg * TD_JG;
    enc_direct_body(b, job0, min(TD_JG, num_jobs - job0), idx, m, n_items, items_per_split, h, local % tiles_j, local / tiles_j, smem);
}

// ---- the step's tail in one pass ----------------------------------------------------------------------------
// All parameter gradients of a model are due only when the optimizer runs.  phc_step_tail launches their first
// stages SIDE BY SIDE: workgroups [0, nb_atom) build atom-table slabs, [nb_atom, nb_atom + nb_bond) the bond-table
// slabs of every message-passing layer, the rest the PHM dH = x^T.dy slabs of every linear layer.  The table workgroups
// are latency-bound gathers, the dH workgroups MFMA/bandwidth-bound; dispatched first, the former run in the slots the
// latter leave free instead of owning the chip for 50 us on their own.  Second launch: all slab reductions.
struct TailTables {
    // bond tables (MODE 1): jobs share idx / src / dst
    TgBatch bond;
    EncMeta mb;
    const long long *bond_idx, *src, *dst;
    const int *rowptr_b;                                     // counted form (identity message function): destination-sorted
    const unsigned *codes_b;                                 // row pointers + packed codes, E = number of destination nodes
    long long n_edges_b;
    long long E, ips_b;
    int tiles_j_b, tiles_b, slabs_b, jobs_b, vec4_b;
    // atom tables (MODE 0): one job
    const float *atom_dout;
    const long long *atom_idx;
    float *atom_slabs, *atom_flat;
    EncMeta ma;
    long long N_a, ips_a;
    int tiles_j_a, tiles_a, slabs_a, vec4_a;
    int n, k, h;
    int dh_first;                                            // block order: the dH family before the table families
    int atom_direct;                                         // atom tables in the register-resident form (atom_direct_body)
    const float *counts_b;                                   // != NULL: bond tables in the register-resident form (bond_direct_body)
    int cr_b;                                                // columns of counts_b (16 | 32)
    unsigned nb_atom, nb_bond;                               // workgroup counts of the first launch
    unsigned nr_phm, nr_bond, nr_atom;                       // ... of the reduce launch
};

#ifdef PHC_TAIL_STAMPS
// experiment build only (scripts/dbg/tail_wg_timeline.py): entry / exit wall clock (100 MHz), family and hardware id of EVERY workgroup
static __device__ unsigned long long g_tail_ts[4096 * 4];
struct TailStamp {
    unsigned long long t0;
    __device__ TailStamp() : t0(__builtin_amdgcn_s_memrealtime()) {}
    __device__ void done(unsigned family) const {
        __syncthreads();
        if (threadIdx.x == 0 && blockIdx.x < 4096) {
            unsigned long long *o = g_tail_ts + 4 * (size_t)blockIdx.x;
            o[0] = t0; o[1] = __builtin_amdgcn_s_memrealtime(); o[2] = family;
            o[3] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 11) | (0 << 6) | 31) << 32) |     // XCC_ID
                   __builtin_amdgcn_s_getreg((4 << 11) | (0 << 6) | 31);                                     // HW_ID
        }
    }
};
#define TAIL_DONE(f) stamp_.done(f)
#else
#define TAIL_DONE(f) do { } while (0)
#endif

template <int ACT>
__global__ __launch_bounds__(256, 2) void tail_partial_kernel(GradBatch gb, TailTables t) {
    __shared__ __attribute__((aligned(16))) float smem[2 * TG_IB * TG_T];
#ifdef PHC_TAIL_STAMPS
    const TailStamp stamp_;
#endif
    static_assert(2 * TG_IB * TG_T == 2 * DH_RB * DH_STRIDE, "table and dH workgroups stage the same 32 KB");
    // Dispatch order: workgroups start in block order, two per CU: table families, then the dH family.  The long dH workgroups
    // first (t.dh_first, PHC_TAIL_DH_FIRST=1) measured within noise of it; interleaving the families in proportion cost 15 us.
    unsigned bx = blockIdx.x;
    if (t.dh_first) {
        const unsigned nT = t.nb_atom + t.nb_bond, nD = gridDim.x - nT;
        bx = bx < nD ? nT + bx : bx - nD;
    }
    if (bx < t.nb_atom) {
        const int tile = bx % t.tiles_a;
        if (t.atom_direct) {
            atom_direct_body(t.atom_dout, t.atom_idx, t.ma, t.N_a, t.ips_a, t.h, t.tiles_j_a, t.atom_slabs, tile, bx / t.tiles_a, smem);
            TAIL_DONE(0);
            return;
        }
        table_grad_rows_body<0>(t.atom_dout, t.atom_idx, nullptr, nullptr, 0, t.ma, t.N_a, t.ips_a, t.h, t.tiles_j_a, t.vec4_a, t.atom_slabs,
                                tile, bx / t.tiles_a, smem);
        TAIL_DONE(0);
        return;
    }
    bx -= t.nb_atom;
    if (bx < t.nb_bond && t.counts_b) {
        // register-resident form: (layer group, 64-column tile, split)
        const unsigned per_group = (unsigned)t.tiles_b * (unsigned)t.slabs_b;
        const int jg = bx / per_group;
        const unsigned local = bx - jg * per_group;
        const int job0 = jg * TD_JG;
        bond_direct_body(t.bond, job0, min(TD_JG, t.jobs_b - job0), t.counts_b, t.cr_b, t.mb.R, t.E, t.ips_b, t.h, local % t.tiles_b,
                         local / t.tiles_b, smem);
        TAIL_DONE(1);
        return;
    }
    if (bx < t.nb_bond) {
        const unsigned per_job = (unsigned)t.tiles_b * (unsigned)t.slabs_b;
        const int job = bx / per_job;
        const unsigned local = bx - job * per_job;
        const TgJobDev &jb = t.bond.j[job];
        if (ACT == PHC_ACT_IDENTITY && t.codes_b)
            table_grad_rows_body<2>(jb.dout, nullptr, t.rowptr_b, t.codes_b, t.n_edges_b, t.mb, t.E, t.ips_b, t.h, t.tiles_j_b, t.vec4_b, jb.slabs,
                                    local % t.tiles_b, local / t.tiles_b, smem);
        else
            table_grad_body<1, ACT>(jb.dout, t.bond_idx, t.mb, t.E, t.ips_b, t.h, t.tiles_j_b, t.src, t.dst, jb.x, jb.tables, t.n, t.k,
                                    t.vec4_b, jb.slabs, local % t.tiles_b, local / t.tiles_b, smem);
        TAIL_DONE(1);
        return;
    }
    bx -= t.nb_bond;
    int k = 0;
    for (int q = 0; q < gb.num; ++q)
        if (bx >= gb.j[q].blk0_partial) k = q;
    const GradJobDev &j = gb.j[k];
    const unsigned local = bx - j.blk0_partial;
    const long long xf_count = j.xf_stat ? (j.xf_n_valid ? min((long long)*j.xf_n_valid, j.N) : j.N) : 0;
    int tile;
    long long split;
    // Even shares (GradJobDev.pair > 1, form 3): the job's rows are cut into splits - 1 full splits and a shorter last one, and the
    // last splits of `pair` tiles are ONE workgroup's work, one after the other — every workgroup of the family then has (about)
    // rows_per_split rows whatever N / rows_per_split leaves over, instead of the family's split count being rounded to what fits the
    // slots (88 tiles x 5 splits = 440 of 512 at hidden 256).  Which workgroup computes a (tile, split) never changes what is computed.
    int tasks = 1, tile_step = 0;
    if (j.t128 == 3 && j.pair > 1) {
        const unsigned full = (unsigned)j.tiles * (unsigned)(j.splits - 1);
        if (local < full) {
            xcd_tile_split(blockIdx.x, local, (unsigned)j.tiles, (unsigned)(j.splits - 1), tile, split);
        } else {
            tile = (int)(local - full) * j.pair;
            split = j.splits - 1;
            tasks = min(j.pair, j.tiles - tile);
            tile_step = 1;
        }
    } else {
        xcd_tile_split(blockIdx.x, local, (unsigned)j.tiles, (unsigned)j.splits, tile, split);
    }
    if (j.t128 == 3) {
        for (int q = 0; q < tasks; ++q, tile += tile_step) {
            if (q) __syncthreads();                          // (the staging area of the task in front)
            if (dh3_np(j.out) == 2)
                dH_direct_body<2>(j.x, j.dy, j.slabs, j.N, j.in, j.out, j.tiles_j, j.rows_per_split, tile, split, smem, j.xf_stat, j.xf_act,
                                  xf_count);
            else                                             // out an odd multiple of 32: 128 x 32 tiles
                dH_direct_body<1>(j.x, j.dy, j.slabs, j.N, j.in, j.out, j.tiles_j, j.rows_per_split, tile, split, smem, j.xf_stat, j.xf_act,
                                  xf_count);
        }
    } else {
        dH_partial_body(j.x, j.dy, j.slabs, j.N, j.in, j.out, j.tiles_j, j.rows_per_split, j.vec4, tile, split, smem, j.xf_stat, j.xf_act,
                        xf_count);
    }
    TAIL_DONE(2u + 16u * (unsigned)k + 4096u * (unsigned)tasks + 65536u * (unsigned)split);
}

__global__ __launch_bounds__(256) void tail_reduce_kernel(GradBatch gb, TailTables t, TailReg reg) {
    unsigned bx = blockIdx.x;
    const unsigned nr_tail = t.nr_phm + t.nr_bond + t.nr_atom;
    if (bx >= nr_tail) {                                     // the weight regulariser's workgroups ride behind the reductions
        tail_reg_body(reg, bx - nr_tail);
        return;
    }
    if (bx < t.nr_phm) {
        const int k = grad_job_of_block<1>(gb, bx);
        const GradJobDev &j = gb.j[k];
        dH_reduce_body(j.slabs, j.dH, (size_t)j.in * j.out + j.out, j.splits, bx - j.blk0_reduce);
        return;
    }
    bx -= t.nr_phm;
    if (bx < t.nr_bond) {
        const unsigned per_job = (unsigned)((t.mb.R * t.h + 63) / 64);
        const int job = bx / per_job;
        table_grad_reduce_body(t.bond.j[job].slabs, t.slabs_b, t.mb.R, t.n, t.k, t.bond.j[job].flat, (int)(bx - job * per_job));
        return;
    }
    bx -= t.nr_bond;
    table_grad_reduce_body(t.atom_slabs, t.slabs_a, t.ma.R, t.n, t.k, t.atom_flat, (int)bx);
}

// ---- host helpers ---------------------------------------------------------------------------------
static int make_meta(const char *who, int F, const int32_t *vocab, EncMeta *m) {
    PHC_REQUIRE(F >= 1 && F <= ENC_MAX_F, PHC_ERR_UNSUPPORTED, "%s: %d features (max %d)", who, F, ENC_MAX_F);
    PHC_REQUIRE(vocab, PHC_ERR_INVALID_ARG, "%s: vocab is null", who);
    m->F = F;
    m->off[0] = 0;
    for (int f = 0; f < F; ++f) {
        PHC_REQUIRE(vocab[f] >= 1, PHC_ERR_INVALID_ARG, "%s: vocab[%d] = %d", who, f, vocab[f]);
        m->off[f + 1] = m->off[f] + vocab[f];
    }
    m->R = m->off[F];
    return PHC_OK;
}

struct TgPlan {
    int tiles_i, tiles_j, num_slabs;
    long long items_per_split;
    size_t slab_floats;
};

// register-resident atom form: tiles = (M groups of 96 table rows) x (64-column tiles), one workgroup per (tile, split), about 256
// workgroups, every split the same number of nodes (a multiple of 8: four wavefront quarters starting on even nodes)
static bool tg_direct_shape_ok(long long n_items, int h) {
    constexpr bool off = false;
    return !off && h % 2 == 0 && h >= 32 && n_items >= 8;  // (64-column tiles, the last one partial; rows of dout 8-byte aligned)
}
constexpr long long TG_DIRECT_TARGET = 256;              // workgroups of a register-resident table family on its own
static TgPlan tg_plan_direct(long long n_items, int R, int h, long long target = TG_DIRECT_TARGET) {
    TgPlan p;
    p.tiles_i = (R + 32 * TD_MB - 1) / (32 * TD_MB);
    p.tiles_j = (h + 63) / 64;
    const long long tiles = (long long)p.tiles_i * p.tiles_j;
    long long ips = ((n_items * tiles + target - 1) / target + 7) / 8 * 8;
    if (ips < 8) ips = 8;
    const size_t cap_slabs = ((size_t)16 << 20) / (sizeof(float) * (size_t)R * h);       // <= 16 MB of slabs
    if ((size_t)((n_items + ips - 1) / ips) > cap_slabs && cap_slabs > 0) ips = ((n_items + (long long)cap_slabs - 1) / (long long)cap_slabs + 7) / 8 * 8;
    p.items_per_split = ips;
    p.num_slabs = (int)((n_items + ips - 1) / ips);
    if (p.num_slabs < 1) p.num_slabs = 1;
    p.slab_floats = (size_t)p.num_slabs * R * h;
    return p;
}

// register-resident bond form: tiles = (layer groups of TD_JG) x (64-column tiles); slab_floats is PER JOB
static TgPlan tg_plan_direct_bond(long long n_items, int R, int h, int jobs, long long target = TG_DIRECT_TARGET) {
    TgPlan p;
    p.tiles_i = (jobs + TD_JG - 1) / TD_JG;                  // layer groups
    p.tiles_j = (h + 63) / 64;
    const long long tiles = (long long)p.tiles_i * p.tiles_j;
    long long ips = ((n_items * tiles + target - 1) / target + 7) / 8 * 8;
    if (ips < 8) ips = 8;
    p.items_per_split = ips;
    p.num_slabs = (int)((n_items + ips - 1) / ips);
    if (p.num_slabs < 1) p.num_slabs = 1;
    p.slab_floats = (size_t)p.num_slabs * R * h;
    return p;
}

static TgPlan tg_plan(long long n_items, int R, int h, int jobs = 1) {
    TgPlan p;
    p.tiles_i = (R + TG_T - 1) / TG_T;
    p.tiles_j = (h + TG_T - 1) / TG_T;
    const int tiles = p.tiles_i * p.tiles_j;
    // workgroups of the whole launch: ~2-4 per CU for molecule batches; for large inputs one workgroup per ~8k items
    // (a fixed 512 left each workgroup with thousands of dependent gather -> MFMA rounds: 19 ms for 10 M atoms)
    constexpr long long knob = 0;
    long long target = knob > 0 ? knob : (jobs > 1 ? 1024 : 512);
    const long long by_items = (n_items / 8192) * tiles * jobs;
    if (by_items > target) target = by_items;
    long long splits = (target + tiles * jobs - 1) / (tiles * jobs);
    // slabs: 16 MB for small inputs, up to 256 MB when the input itself is gigabytes
    const size_t slab_budget = n_items > (1 << 20) ? ((size_t)256 << 20) : ((size_t)16 << 20);
    const long long cap = (long long)(slab_budget / (sizeof(float) * (size_t)R * h * (size_t)jobs));
    if (splits > cap) splits = cap;
    if (splits < 1) splits = 1;
    long long ips = (n_items + splits - 1) / splits;
    ips = (ips + TG_IB - 1) / TG_IB * TG_IB;
    if (ips < TG_IB) ips = TG_IB;
    p.items_per_split = ips;
    p.num_slabs = (int)((n_items + ips - 1) / ips);
    if (p.num_slabs < 1) p.num_slabs = 1;
    p.slab_floats = (size_t)p.num_slabs * R * h;
    return p;
}

static inline bool al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline int lpr_for(int h4) { return h4 <= 8 ? 8 : h4 <= 16 ? 16 : h4 <= 32 ? 32 : 64; }

template <int MODE>
static int launch_table_grad(const float *dout, const long long *idx, const EncMeta &m, long long n_items, int h, int n, int k,
                             int act, const long long *src, const long long *dst, const float *x, const float *const *tables,
                             float *flat, void *ws, size_t ws_bytes, hipStream_t s, const char *who) {
    const int vec4 = (k % 4 == 0) && al16(dout) && (!x || al16(x));
    TgPlan p = tg_plan(n_items, m.R, h);
    PHC_REQUIRE(ws && ws_bytes >= sizeof(float) * p.slab_floats, PHC_ERR_WORKSPACE, "%s: workspace %zu B < required %zu B",
                who, ws_bytes, sizeof(float) * p.slab_floats);
    float *slabs = static_cast<float *>(ws);
    dim3 grid(p.tiles_i * p.tiles_j, p.num_slabs);
#define PHC_TG_LAUNCH(ACTC)                                                                                               \
    PHC_LAUNCH((table_grad_mfma_kernel<MODE, ACTC>), grid, dim3(ENC_THREADS), 0, s, dout, idx, m, n_items,          \
                       p.items_per_split, h, p.tiles_j, src, dst, x, tables, n, k, vec4, slabs)
    {
        ProfScope prof(K_TABLE_GRAD, s);
        if (MODE == 0 || act == PHC_ACT_IDENTITY) {
            PHC_TG_LAUNCH(PHC_ACT_IDENTITY);
        } else {
            switch (act) {
                case PHC_ACT_RELU: PHC_TG_LAUNCH(PHC_ACT_RELU); break;
                case PHC_ACT_LRELU: PHC_TG_LAUNCH(PHC_ACT_LRELU); break;
                case PHC_ACT_ELU: PHC_TG_LAUNCH(PHC_ACT_ELU); break;
                case PHC_ACT_SWISH: PHC_TG_LAUNCH(PHC_ACT_SWISH); break;
                default: PHC_TG_LAUNCH(PHC_ACT_SELU); break;
            }
        }
        int rc = check_launch(who);
        if (rc) return rc;
    }
#undef PHC_TG_LAUNCH
    ProfScope prof(K_TABLE_GRAD_REDUCE, s);
    PHC_LAUNCH(table_grad_reduce_kernel, dim3((m.R * h + 63) / 64), dim3(256), 0, s, slabs, p.num_slabs, m.R, n, k,
                       flat);
    return check_launch(who);
}

}  // namespace phc

using namespace phc;

extern "C" {

#ifdef PHC_TAIL_STAMPS
int phc_debug_tail_stamps(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tail_ts), sizeof(unsigned long long) * 4096 * 4) == hipSuccess ? 0 : -1;
}
#endif


int phc_encode_fwd(const int64_t *idx_i64, const float *const *tables, const int32_t *vocab, int64_t N, int F, int n, int k,
                   float *out, void *stream) {
    PHC_REQUIRE(N >= 0 && F >= 1 && n >= 1 && k >= 1, PHC_ERR_INVALID_ARG, "phc_encode_fwd: N=%lld F=%d n=%d k=%d", (long long)N,
                F, n, k);
    EncMeta m;
    {
        int rc = make_meta("phc_encode_fwd", F, vocab, &m);
        if (rc) return rc;
    }
    int *err = err_word();
    if (N == 0) return PHC_OK;
    PHC_REQUIRE(idx_i64 && tables && out, PHC_ERR_INVALID_ARG, "phc_encode_fwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int h = n * k;
    ProfScope prof(K_ENCODE_FWD, s);
    if (k % 4 == 0 && al16(out)) {
        const int lpr = lpr_for(h >> 2);
        const int G = 64 / lpr;
        long long blocks = (N + 4 * G - 1) / (4 * G);
        if (blocks > (N > 64 * 1024 ? 32768 : 4096)) blocks = N > 64 * 1024 ? 32768 : 4096;
        const long long *idx = reinterpret_cast<const long long *>(idx_i64);
#define PHC_ENC_FWD(LPRC, FM)                                                                                               \
    PHC_LAUNCH((encode_fwd_kernel<LPRC, FM>), dim3((unsigned)blocks), dim3(ENC_THREADS), 0, s, idx, tables, (long long)N, F, n, k, out, \
               m, err)
#define PHC_ENC_FWD_F(LPRC)             \
    if (F <= 4) {                       \
        PHC_ENC_FWD(LPRC, 4);           \
    } else if (F <= 10) {               \
        PHC_ENC_FWD(LPRC, 10);          \
    } else {                            \
        PHC_ENC_FWD(LPRC, ENC_MAX_F);   \
    }
        switch (lpr) {
            case 8: PHC_ENC_FWD_F(8); break;
            case 16: PHC_ENC_FWD_F(16); break;
            case 32: PHC_ENC_FWD_F(32); break;
            default: PHC_ENC_FWD_F(64); break;
        }
#undef PHC_ENC_FWD_F
#undef PHC_ENC_FWD
    } else {
        long long blocks = (N * h + ENC_THREADS - 1) / ENC_THREADS;
        if (blocks > (N > 64 * 1024 ? 32768 : 4096)) blocks = N > 64 * 1024 ? 32768 : 4096;
        PHC_LAUNCH(encode_fwd_scalar_kernel, dim3((unsigned)blocks), dim3(ENC_THREADS), 0, s,
                           reinterpret_cast<const long long *>(idx_i64), tables, (long long)N, F, n, k, out, m, err);
    }
    return check_launch("encode_fwd");
}

size_t phc_encode_bwd_ws_bytes(int64_t N, int F, const int32_t *vocab, int n, int k) {
    EncMeta m;
    if (N < 0 || n < 1 || k < 1 || make_meta("phc_encode_bwd_ws_bytes", F, vocab, &m)) return 0;
    return sizeof(float) * tg_plan(N, m.R, n * k).slab_floats;
}

int phc_encode_bwd(const float *dout, const int64_t *idx_i64, int64_t N, int F, const int32_t *vocab, int n, int k,
                   float *dtables_flat, void *ws, size_t ws_bytes, void *stream) {
    EncMeta m;
    int rc = make_meta("phc_encode_bwd", F, vocab, &m);
    if (rc) return rc;
    PHC_REQUIRE(N >= 0 && n >= 1 && k >= 1, PHC_ERR_INVALID_ARG, "phc_encode_bwd: N=%lld n=%d k=%d", (long long)N, n, k);
    PHC_REQUIRE(dtables_flat && (N == 0 || (dout && idx_i64)), PHC_ERR_INVALID_ARG, "phc_encode_bwd: null pointer");
    return launch_table_grad<0>(dout, reinterpret_cast<const long long *>(idx_i64), m, N, n * k, n, k, PHC_ACT_IDENTITY, nullptr,
                                nullptr, nullptr, nullptr, dtables_flat, ws, ws_bytes, (hipStream_t)stream, "phc_encode_bwd");
}

static bool enc_direct_batched_ok(const EncMeta &m, int64_t N, int h) { return m.R <= 32 && tg_direct_shape_ok(N, h); }

size_t phc_encode_bwd_batched_ws_bytes(int64_t N, int F, const int32_t *vocab, int n, int k, int num_jobs) {
    size_t need = phc_table_grads_ws_bytes(N, F, vocab, n, k, num_jobs);          // same plan, items = encoded rows
    EncMeta m;
    if (num_jobs >= 1 && !make_meta("phc_encode_bwd_batched_ws_bytes", F, vocab, &m) && enc_direct_batched_ok(m, N, n * k)) {
        size_t direct = 0;                                                     // the register-resident form's slabs (either may run)
        for (int c0 = 0; c0 < num_jobs; c0 += TG_JOBS_PER_LAUNCH) {
            const int cn = num_jobs - c0 < TG_JOBS_PER_LAUNCH ? num_jobs - c0 : TG_JOBS_PER_LAUNCH;
            direct += sizeof(float) * tg_plan_direct_bond(N, m.R, n * k, cn).slab_floats * cn;
        }
        if (direct > need) need = direct;
    }
    return need;
}

int phc_encode_bwd_batched(const phc_table_grad_job *jobs, int num_jobs, const int64_t *idx_i64, int64_t N, int F,
                           const int32_t *vocab, int n, int k, void *ws, size_t ws_bytes, void *stream) {
    EncMeta m;
    int rc = make_meta("phc_encode_bwd_batched", F, vocab, &m);
    if (rc) return rc;
    PHC_REQUIRE(num_jobs >= 1 && jobs && N >= 1 && n >= 1 && k >= 1 && idx_i64, PHC_ERR_INVALID_ARG, "phc_encode_bwd_batched: bad arguments");
    PHC_REQUIRE(ws && ws_bytes >= phc_encode_bwd_batched_ws_bytes(N, F, vocab, n, k, num_jobs), PHC_ERR_WORKSPACE,
                "phc_encode_bwd_batched: workspace too small");
    const int h = n * k;
    hipStream_t s = (hipStream_t)stream;
    float *wsf = static_cast<float *>(ws);
    for (int c0 = 0; c0 < num_jobs; c0 += TG_JOBS_PER_LAUNCH) {
        const int cn = num_jobs - c0 < TG_JOBS_PER_LAUNCH ? num_jobs - c0 : TG_JOBS_PER_LAUNCH;
        bool direct = enc_direct_batched_ok(m, N, h);
        for (int q = 0; q < cn && direct; ++q) direct = jobs[c0 + q].dout && (reinterpret_cast<uintptr_t>(jobs[c0 + q].dout) & 7) == 0;
        const TgPlan p = direct ? tg_plan_direct_bond(N, m.R, h, cn) : tg_plan(N, m.R, h, cn);
        TgBatch b;
        int vec4 = (k % 4 == 0);
        for (int q = 0; q < cn; ++q) {
            const phc_table_grad_job &jb = jobs[c0 + q];
            PHC_REQUIRE(jb.dout && jb.dtables_flat, PHC_ERR_INVALID_ARG, "phc_encode_bwd_batched: job %d: null pointer", c0 + q);
            b.j[q].dout = jb.dout; b.j[q].x = nullptr; b.j[q].tables = nullptr; b.j[q].flat = jb.dtables_flat;
            b.j[q].slabs = wsf;
            wsf += p.slab_floats;
            vec4 = vec4 && al16(jb.dout);
        }
        if (direct) {
            // register-resident form: (encoder group, 64-column tile, split) workgroups
            ProfScope prof(K_TABLE_GRAD, s);
            PHC_LAUNCH(enc_direct_batched_kernel, dim3((unsigned)(p.tiles_i * p.tiles_j * p.num_slabs)), dim3(ENC_THREADS), 0, s, b, cn,
                       reinterpret_cast<const long long *>(idx_i64), m, (long long)N, p.items_per_split, h, p.tiles_j, p.num_slabs);
            rc = check_launch("encode_bwd_batched(direct)");
            if (rc) return rc;
        } else {
            ProfScope prof(K_TABLE_GRAD, s);
            PHC_LAUNCH(table_grad_rows_batched_kernel, dim3(p.tiles_i * p.tiles_j, p.num_slabs, cn), dim3(ENC_THREADS), 0, s, b,
                       reinterpret_cast<const long long *>(idx_i64), m, (long long)N, p.items_per_split, h, p.tiles_j, vec4);
            rc = check_launch("encode_bwd_batched");
            if (rc) return rc;
        }
        ProfScope prof(K_TABLE_GRAD_REDUCE, s);
        PHC_LAUNCH(table_grad_reduce_batched_kernel, dim3((m.R * h + 63) / 64, cn), dim3(256), 0, s, b, p.num_slabs, m.R, n, k);
        rc = check_launch("encode_bwd_batched(reduce)");
        if (rc) return rc;
    }
    return PHC_OK;
}

int phc_aggregate_enc_fwd(const float *x, const int64_t *bond_i64, const float *const *tables, const int32_t *vocab,
                          const int32_t *rowptr, const int32_t *col, const int32_t *perm, int64_t n_dst, int F, int n, int k,
                          int act, int add_self, float *out, void *stream) {
    EncMeta m;
    int rc = make_meta("phc_aggregate_enc_fwd", F, vocab, &m);
    if (rc) return rc;
    PHC_REQUIRE(n_dst >= 0 && n >= 1 && k >= 1, PHC_ERR_INVALID_ARG, "phc_aggregate_enc_fwd: bad sizes");
    PHC_REQUIRE(act >= PHC_ACT_IDENTITY && act <= PHC_ACT_SELU, PHC_ERR_INVALID_ARG, "phc_aggregate_enc_fwd: unknown activation %d", act);
    if (n_dst == 0) return PHC_OK;
    PHC_REQUIRE(x && bond_i64 && tables && rowptr && col && perm && out, PHC_ERR_INVALID_ARG, "phc_aggregate_enc_fwd: null pointer");
    const int h = n * k;
    const size_t lds = sizeof(float) * (size_t)m.R * h;
    PHC_REQUIRE(k % 4 == 0 && al16(x) && al16(out), PHC_ERR_UNSUPPORTED, "phc_aggregate_enc_fwd: needs k %% 4 == 0 and 16-byte aligned rows");
    PHC_REQUIRE(lds <= 48 * 1024, PHC_ERR_UNSUPPORTED, "phc_aggregate_enc_fwd: tables (%zu B) exceed the LDS budget", lds);
    hipStream_t s = (hipStream_t)stream;
    long long blocks = (n_dst + 3) / 4;
    if (blocks > 1024) blocks = 1024;                       // persistent-ish: amortise the LDS table fill
    const long long *bond = reinterpret_cast<const long long *>(bond_i64);
    const int lpr = lpr_for(h >> 2);
    ProfScope prof(K_AGG_ENC_FWD, s);
#define PHC_AE_LAUNCH(LPRC, ACTC)                                                                                          \
    PHC_LAUNCH((aggregate_enc_fwd_kernel<LPRC, ACTC>), dim3((unsigned)blocks), dim3(ENC_THREADS), lds, s, x, bond,      \
                       tables, m, rowptr, col, perm, (long long)n_dst, n, k, add_self, out, err_word())
#define PHC_AE_ACT(LPRC)                                            \
    switch (act) {                                                  \
        case PHC_ACT_IDENTITY: PHC_AE_LAUNCH(LPRC, PHC_ACT_IDENTITY); break; \
        case PHC_ACT_RELU: PHC_AE_LAUNCH(LPRC, PHC_ACT_RELU); break;         \
        case PHC_ACT_LRELU: PHC_AE_LAUNCH(LPRC, PHC_ACT_LRELU); break;       \
        case PHC_ACT_ELU: PHC_AE_LAUNCH(LPRC, PHC_ACT_ELU); break;           \
        case PHC_ACT_SWISH: PHC_AE_LAUNCH(LPRC, PHC_ACT_SWISH); break;       \
        default: PHC_AE_LAUNCH(LPRC, PHC_ACT_SELU); break;                   \
    }
    switch (lpr) {
        case 8: PHC_AE_ACT(8); break;
        case 16: PHC_AE_ACT(16); break;
        case 32: PHC_AE_ACT(32); break;
        default: PHC_AE_ACT(64); break;
    }
#undef PHC_AE_ACT
#undef PHC_AE_LAUNCH
    return check_launch("aggregate_enc_fwd");
}

int phc_pack_edge_codes(const int64_t *bond_i64, const int32_t *perm, const int32_t *vocab, int64_t E, int F, uint32_t *codes,
                        void *stream) {
    EncMeta m;
    int rc = make_meta("phc_pack_edge_codes", F, vocab, &m);
    if (rc) return rc;
    PHC_REQUIRE(F <= 4 && m.R <= 256, PHC_ERR_UNSUPPORTED, "phc_pack_edge_codes: needs F <= 4 and sum(vocab) <= 256 (got %d, %d)", F, m.R);
    PHC_REQUIRE(E >= 0 && (E == 0 || (bond_i64 && codes)), PHC_ERR_INVALID_ARG, "phc_pack_edge_codes: bad arguments");
    if (E == 0) return PHC_OK;
    hipStream_t s = (hipStream_t)stream;
    long long blocks = (E + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    ProfScope prof(K_PACK_CODES, s);
    PHC_LAUNCH(pack_edge_codes_kernel, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const long long *>(bond_i64),
                       perm, m, (long long)E, codes, err_word());
    return check_launch("pack_edge_codes");
}

int phc_aggregate_encp_fwd(const float *x, const uint32_t *codes, const float *const *tables, const int32_t *vocab,
                           const int32_t *rowptr, const int32_t *col, int64_t n_dst, int F, int n, int k, int act, int add_self,
                           float *out, void *stream) {
    EncMeta m;
    int rc = make_meta("phc_aggregate_encp_fwd", F, vocab, &m);
    if (rc) return rc;
    PHC_REQUIRE(n_dst >= 0 && n >= 1 && k >= 1, PHC_ERR_INVALID_ARG, "phc_aggregate_encp_fwd: bad sizes");
    PHC_REQUIRE(act >= PHC_ACT_IDENTITY && act <= PHC_ACT_SELU, PHC_ERR_INVALID_ARG, "phc_aggregate_encp_fwd: unknown activation %d", act);
    if (n_dst == 0) return PHC_OK;
    PHC_REQUIRE(x && codes && tables && rowptr && col && out, PHC_ERR_INVALID_ARG, "phc_aggregate_encp_fwd: null pointer");
    const int h = n * k;
    const size_t lds = sizeof(float) * (size_t)m.R * h;
    PHC_REQUIRE(F <= 4 && m.R <= 256 && n * F <= PHC_MAX_COMPONENTS * 4, PHC_ERR_UNSUPPORTED,
                "phc_aggregate_encp_fwd: needs F <= 4, sum(vocab) <= 256, n*F <= 64");
    PHC_REQUIRE(k % 4 == 0 && al16(x) && al16(out), PHC_ERR_UNSUPPORTED, "phc_aggregate_encp_fwd: needs k %% 4 == 0 and 16-byte aligned rows");
    PHC_REQUIRE(lds <= 48 * 1024, PHC_ERR_UNSUPPORTED, "phc_aggregate_encp_fwd: tables (%zu B) exceed the LDS budget", lds);
    hipStream_t s = (hipStream_t)stream;
    long long blocks = (n_dst + 3) / 4;
    // persistent-ish workgroups amortise the LDS table fill; large graphs take 8x more of them so that rows of very
    // different degree balance out (10 M rows: 44.7 % -> 59.7 % of the HBM peak), molecule batches stay at one resident set
    const long long cap = n_dst > 64 * 1024 ? 16384 : 2048;
    if (blocks > cap) blocks = cap;
    const int lpr = lpr_for(h >> 2);
    const int rows_per_wg = 4 * (WAVE / lpr);                 // direct form: one row per lane group
    const long long dblocks = xcd_grid((n_dst + rows_per_wg - 1) / rows_per_wg);      // a multiple of 8: one contiguous row range per XCD
    // molecule batches: one wavefront per row, tables read from L2 (no LDS staging); large graphs amortise the staging
    constexpr long long direct_max = 32768;
    const bool direct = n_dst <= direct_max;
    ProfScope prof(K_AGG_ENC_FWD, s);
#define PHC_AP_LAUNCH(LPRC, ACTC, FC)                                                                                     \
    if (direct)                                                                                                           \
        PHC_LAUNCH((aggregate_encp_direct_fwd_kernel<LPRC, ACTC, FC>), dim3((unsigned)dblocks), dim3(ENC_THREADS), 0, s, \
                   x, codes, tables, m, rowptr, col, (long long)n_dst, n, k, add_self, out);                              \
    else                                                                                                                  \
        PHC_LAUNCH((aggregate_encp_fwd_kernel<LPRC, ACTC, FC>), dim3((unsigned)blocks), dim3(ENC_THREADS), lds, s, x,    \
                   codes, tables, m, rowptr, col, (long long)n_dst, n, k, add_self, out)
#define PHC_AP_F(LPRC, ACTC)                                    \
    switch (F) {                                                \
        case 1: PHC_AP_LAUNCH(LPRC, ACTC, 1); break;            \
        case 2: PHC_AP_LAUNCH(LPRC, ACTC, 2); break;            \
        case 3: PHC_AP_LAUNCH(LPRC, ACTC, 3); break;            \
        default: PHC_AP_LAUNCH(LPRC, ACTC, 4); break;           \
    }
#define PHC_AP_ACT(LPRC)                                            \
    switch (act) {                                                  \
        case PHC_ACT_IDENTITY: PHC_AP_F(LPRC, PHC_ACT_IDENTITY); break; \
        case PHC_ACT_RELU: PHC_AP_F(LPRC, PHC_ACT_RELU); break;         \
        case PHC_ACT_LRELU: PHC_AP_F(LPRC, PHC_ACT_LRELU); break;       \
        case PHC_ACT_ELU: PHC_AP_F(LPRC, PHC_ACT_ELU); break;           \
        case PHC_ACT_SWISH: PHC_AP_F(LPRC, PHC_ACT_SWISH); break;       \
        default: PHC_AP_F(LPRC, PHC_ACT_SELU); break;                   \
    }
    switch (lpr) {
        case 8: PHC_AP_ACT(8); break;
        case 16: PHC_AP_ACT(16); break;
        case 32: PHC_AP_ACT(32); break;
        default: PHC_AP_ACT(64); break;
    }
#undef PHC_AP_ACT
#undef PHC_AP_F
#undef PHC_AP_LAUNCH
    return check_launch("aggregate_encp_fwd");
}

static inline long long encp_direct_max() {
    constexpr long long v = 32768;
    return v;
}
int phc_aggregate_norm_supported(int64_t n_dst, int h, int F, int R) {
    return (n_dst >= 1 && n_dst <= encp_direct_max() && h >= 4 && h % 4 == 0 && h <= AGN_MAX_H && F >= 1 && F <= 4 && R <= 256) ? 1 : 0;
}

int phc_aggregate_norm_encp_fwd(const float *y, const uint64_t *acc, const float *const *weight, const float *const *bias,
                                float *const *running_mean, float *const *running_var, int64_t *const *num_batches_tracked, int n_comp,
                                float momentum, float eps, int norm_act, const int32_t *n_valid_dev, const float *skip, float *save_stat,
                                float *x_out, const uint32_t *codes, const float *const *tables, const int32_t *vocab,
                                const int32_t *rowptr, const int32_t *col, int64_t n_dst, int F, int n, int k, int add_self, float *out,
                                void *stream) {
    EncMeta m;
    int rc = make_meta("phc_aggregate_norm_encp_fwd", F, vocab, &m);
    if (rc) return rc;
    const int h = n * k;
    PHC_REQUIRE(phc_aggregate_norm_supported(n_dst, h, F, m.R) && n * F <= PHC_MAX_COMPONENTS * 4 && k % 4 == 0, PHC_ERR_UNSUPPORTED,
                "phc_aggregate_norm_encp_fwd: %lld rows, h = %d, F = %d, sum(vocab) = %d outside the fused form's range", (long long)n_dst,
                h, F, m.R);
    PHC_REQUIRE(norm_act >= PHC_ACT_IDENTITY && norm_act <= PHC_ACT_SELU, PHC_ERR_INVALID_ARG, "phc_aggregate_norm_encp_fwd: unknown activation %d",
                norm_act);
    PHC_REQUIRE(y && acc && save_stat && x_out && codes && tables && rowptr && col && out, PHC_ERR_INVALID_ARG,
                "phc_aggregate_norm_encp_fwd: null pointer");
    PHC_REQUIRE(al16(y) && al16(x_out) && al16(out) && (!skip || al16(skip)) && (reinterpret_cast<uintptr_t>(acc) & 7) == 0, PHC_ERR_UNSUPPORTED,
                "phc_aggregate_norm_encp_fwd: 16-byte aligned rows / 8-byte aligned accumulator");
    BnParams p;
    rc = fill_params("phc_aggregate_norm_encp_fwd", &p, weight, bias, running_mean, running_var, num_batches_tracked, n_comp, h);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int lpr = lpr_for(h >> 2);
    const int rows_per_wg = 4 * (WAVE / lpr);
    const long long dblocks = xcd_grid((n_dst + rows_per_wg - 1) / rows_per_wg);
    const unsigned long long *acc64 = reinterpret_cast<const unsigned long long *>(acc);
    ProfScope prof(K_AGG_ENC_FWD, s);
#define PHC_AGN_LAUNCH(LPRC, FC)                                                                                                      \
    PHC_LAUNCH((aggregate_norm_direct_fwd_kernel<LPRC, FC>), dim3((unsigned)dblocks), dim3(ENC_THREADS), 0, s, y, skip, p, n_valid_dev,  \
               momentum, eps, acc64, norm_act, save_stat, x_out, codes, tables, m, rowptr, col, (long long)n_dst, n, k, add_self, out)
#define PHC_AGN_F(LPRC)                                 \
    switch (F) {                                        \
        case 1: PHC_AGN_LAUNCH(LPRC, 1); break;         \
        case 2: PHC_AGN_LAUNCH(LPRC, 2); break;         \
        case 3: PHC_AGN_LAUNCH(LPRC, 3); break;         \
        default: PHC_AGN_LAUNCH(LPRC, 4); break;        \
    }
    switch (lpr) {
        case 8: PHC_AGN_F(8); break;
        case 16: PHC_AGN_F(16); break;
        case 32: PHC_AGN_F(32); break;
        default: PHC_AGN_F(64); break;
    }
#undef PHC_AGN_F
#undef PHC_AGN_LAUNCH
    return check_launch("aggregate_norm_encp_fwd");
}

int phc_aggregate_dx_rows_multi(const float *dout, const int32_t *rowptr_t, const int32_t *col_t, int64_t n_src, int64_t n_dst, int h,
                                int add_self, const float *const *residuals, int num_residuals, float *dx, void *stream) {
    PHC_REQUIRE(n_src >= 0 && n_dst >= 0 && h >= 4 && h % 4 == 0, PHC_ERR_INVALID_ARG, "phc_aggregate_dx_rows_multi: bad sizes (h %% 4 == 0)");
    PHC_REQUIRE(num_residuals >= 0 && num_residuals <= 8 && (num_residuals == 0 || residuals), PHC_ERR_UNSUPPORTED,
                "phc_aggregate_dx_rows_multi: %d residual tensors (0..8)", num_residuals);
    if (num_residuals <= 1) return phc_aggregate_dx_rows(dout, rowptr_t, col_t, n_src, n_dst, h, add_self, num_residuals ? residuals[0] : nullptr, dx, stream);
    if (n_src == 0) return PHC_OK;
    PHC_REQUIRE(dout && rowptr_t && col_t && dx, PHC_ERR_INVALID_ARG, "phc_aggregate_dx_rows_multi: null pointer");
    DxResiduals more = {};
    more.n = num_residuals;
    bool aligned = al16(dout) && al16(dx);
    for (int q = 0; q < num_residuals; ++q) {
        PHC_REQUIRE(residuals[q], PHC_ERR_INVALID_ARG, "phc_aggregate_dx_rows_multi: residual %d is null", q);
        more.p[q] = residuals[q];
        aligned = aligned && al16(residuals[q]);
    }
    PHC_REQUIRE(aligned, PHC_ERR_UNSUPPORTED, "phc_aggregate_dx_rows_multi: 16-byte aligned rows");
    hipStream_t s = (hipStream_t)stream;
    const int lpr = lpr_for(h >> 2);
    const int rows_per_wg = 4 * (WAVE / lpr);
    const unsigned rb = (unsigned)xcd_grid((n_src + rows_per_wg - 1) / rows_per_wg);
    ProfScope prof(K_AGG_ENC_DX, s);
#define PHC_DXM(LPRC, NR)                                                                                                            \
    PHC_LAUNCH((aggregate_dx_rows_kernel<LPRC, NR>), dim3(rb), dim3(ENC_THREADS), 0, s, dout, rowptr_t, col_t, (long long)n_src, \
               (long long)n_dst, h, add_self, dx, (const float *)nullptr, more)
#define PHC_DXM_N(LPRC)            \
    if (num_residuals <= 4)        \
        PHC_DXM(LPRC, 4);          \
    else                           \
        PHC_DXM(LPRC, 8)
    switch (lpr) {
        case 8: PHC_DXM_N(8); break;
        case 16: PHC_DXM_N(16); break;
        case 32: PHC_DXM_N(32); break;
        default: PHC_DXM_N(64); break;
    }
#undef PHC_DXM_N
#undef PHC_DXM
    return check_launch("aggregate_dx_rows_multi");
}

int phc_aggregate_dx_rows(const float *dout, const int32_t *rowptr_t, const int32_t *col_t, int64_t n_src, int64_t n_dst, int h,
                          int add_self, const float *residual, float *dx, void *stream) {
    PHC_REQUIRE(n_src >= 0 && n_dst >= 0 && h >= 4 && h % 4 == 0, PHC_ERR_INVALID_ARG, "phc_aggregate_dx_rows: bad sizes (h %% 4 == 0)");
    if (n_src == 0) return PHC_OK;
    PHC_REQUIRE(dout && rowptr_t && col_t && dx, PHC_ERR_INVALID_ARG, "phc_aggregate_dx_rows: null pointer");
    PHC_REQUIRE(al16(dout) && al16(dx) && (!residual || al16(residual)), PHC_ERR_UNSUPPORTED, "phc_aggregate_dx_rows: 16-byte aligned rows");
    hipStream_t s = (hipStream_t)stream;
    const int lpr = lpr_for(h >> 2);
    const int rows_per_wg = 4 * (WAVE / lpr);
    const unsigned rb = (unsigned)xcd_grid((n_src + rows_per_wg - 1) / rows_per_wg);
    ProfScope prof(K_AGG_ENC_DX, s);
    switch (lpr) {
        case 8: PHC_LAUNCH((aggregate_dx_rows_kernel<8>), dim3(rb), dim3(ENC_THREADS), 0, s, dout, rowptr_t, col_t, (long long)n_src, (long long)n_dst, h, add_self, dx, residual); break;
        case 16: PHC_LAUNCH((aggregate_dx_rows_kernel<16>), dim3(rb), dim3(ENC_THREADS), 0, s, dout, rowptr_t, col_t, (long long)n_src, (long long)n_dst, h, add_self, dx, residual); break;
        case 32: PHC_LAUNCH((aggregate_dx_rows_kernel<32>), dim3(rb), dim3(ENC_THREADS), 0, s, dout, rowptr_t, col_t, (long long)n_src, (long long)n_dst, h, add_self, dx, residual); break;
        default: PHC_LAUNCH((aggregate_dx_rows_kernel<64>), dim3(rb), dim3(ENC_THREADS), 0, s, dout, rowptr_t, col_t, (long long)n_src, (long long)n_dst, h, add_self, dx, residual); break;
    }
    return check_launch("aggregate_dx_rows");
}

size_t phc_aggregate_enc_bwd_ws_bytes(int64_t E, int F, const int32_t *vocab, int n, int k) {
    EncMeta m;
    if (E < 0 || n < 1 || k < 1 || make_meta("phc_aggregate_enc_bwd_ws_bytes", F, vocab, &m)) return 0;
    return sizeof(float) * tg_plan(E, m.R, n * k).slab_floats;
}

int phc_aggregate_enc_bwd(const float *dout, const float *x, const int64_t *bond_i64, const float *const *tables,
                          const int32_t *vocab, const int64_t *edge_src_i64, const int64_t *edge_dst_i64, int64_t E,
                          const int32_t *rowptr_t, const int32_t *col_t, const int32_t *perm_t, int64_t n_src, int64_t n_dst,
                          int F, int n, int k, int act, int add_self, float *dx, float *dtables_flat, void *ws, size_t ws_bytes,
                          void *stream) {
    EncMeta m;
    int rc = make_meta("phc_aggregate_enc_bwd", F, vocab, &m);
    if (rc) return rc;
    PHC_REQUIRE(n_src >= 0 && n_dst >= 0 && n >= 1 && k >= 1, PHC_ERR_INVALID_ARG, "phc_aggregate_enc_bwd: bad sizes");
    PHC_REQUIRE(act >= PHC_ACT_IDENTITY && act <= PHC_ACT_SELU, PHC_ERR_INVALID_ARG, "phc_aggregate_enc_bwd: unknown activation %d", act);
    PHC_REQUIRE(dout && bond_i64 && tables, PHC_ERR_INVALID_ARG, "phc_aggregate_enc_bwd: null pointer");
    PHC_REQUIRE(act == PHC_ACT_IDENTITY || x, PHC_ERR_INVALID_ARG, "phc_aggregate_enc_bwd: x is required when act != identity");
    const int h = n * k;
    hipStream_t s = (hipStream_t)stream;
    const long long *bond = reinterpret_cast<const long long *>(bond_i64);
    if (dx && n_src > 0) {
        PHC_REQUIRE(rowptr_t && col_t && perm_t, PHC_ERR_INVALID_ARG, "phc_aggregate_enc_bwd: source-sorted structure missing");
        PHC_REQUIRE(k % 4 == 0 && al16(dout) && al16(dx) && (!x || al16(x)), PHC_ERR_UNSUPPORTED,
                    "phc_aggregate_enc_bwd: needs k %% 4 == 0 and 16-byte aligned rows");
        const size_t lds = act == PHC_ACT_IDENTITY ? 0 : sizeof(float) * (size_t)m.R * h;
        PHC_REQUIRE(lds <= 48 * 1024, PHC_ERR_UNSUPPORTED, "phc_aggregate_enc_bwd: tables exceed the LDS budget");
        long long blocks = (n_src + 3) / 4;
        const long long cap = n_src > 64 * 1024 ? 16384 : (act == PHC_ACT_IDENTITY ? 8192 : 1024);   // large graphs: see encp_fwd
        if (blocks > cap) blocks = cap;
        blocks = xcd_grid(blocks);
        const int lpr = lpr_for(h >> 2);
        ProfScope prof(K_AGG_ENC_DX, s);
        constexpr long long rows_max = 32768;
        if (act == PHC_ACT_IDENTITY && n_src <= rows_max) {                   // molecule batches: one source row per lane group
            const int rows_per_wg = 4 * (WAVE / lpr);
            const unsigned rb = (unsigned)xcd_grid((n_src + rows_per_wg - 1) / rows_per_wg);
            switch (lpr) {
                case 8: PHC_LAUNCH((aggregate_dx_rows_kernel<8>), dim3(rb), dim3(ENC_THREADS), 0, s, dout, rowptr_t, col_t, (long long)n_src, (long long)n_dst, h, add_self, dx, (const float *)nullptr); break;
                case 16: PHC_LAUNCH((aggregate_dx_rows_kernel<16>), dim3(rb), dim3(ENC_THREADS), 0, s, dout, rowptr_t, col_t, (long long)n_src, (long long)n_dst, h, add_self, dx, (const float *)nullptr); break;
                case 32: PHC_LAUNCH((aggregate_dx_rows_kernel<32>), dim3(rb), dim3(ENC_THREADS), 0, s, dout, rowptr_t, col_t, (long long)n_src, (long long)n_dst, h, add_self, dx, (const float *)nullptr); break;
                default: PHC_LAUNCH((aggregate_dx_rows_kernel<64>), dim3(rb), dim3(ENC_THREADS), 0, s, dout, rowptr_t, col_t, (long long)n_src, (long long)n_dst, h, add_self, dx, (const float *)nullptr); break;
            }
        } else {
#define PHC_AD_LAUNCH(LPRC, ACTC)                                                                                         \
    PHC_LAUNCH((aggregate_enc_dx_kernel<LPRC, ACTC>), dim3((unsigned)blocks), dim3(ENC_THREADS), lds, s, dout, x, bond, \
                       tables, m, rowptr_t, col_t, perm_t, (long long)n_src, (long long)n_dst, n, k, add_self, dx)
#define PHC_AD_ACT(LPRC)                                            \
    switch (act) {                                                  \
        case PHC_ACT_IDENTITY: PHC_AD_LAUNCH(LPRC, PHC_ACT_IDENTITY); break; \
        case PHC_ACT_RELU: PHC_AD_LAUNCH(LPRC, PHC_ACT_RELU); break;         \
        case PHC_ACT_LRELU: PHC_AD_LAUNCH(LPRC, PHC_ACT_LRELU); break;       \
        case PHC_ACT_ELU: PHC_AD_LAUNCH(LPRC, PHC_ACT_ELU); break;           \
        case PHC_ACT_SWISH: PHC_AD_LAUNCH(LPRC, PHC_ACT_SWISH); break;       \
        default: PHC_AD_LAUNCH(LPRC, PHC_ACT_SELU); break;                   \
    }
        switch (lpr) {
            case 8: PHC_AD_ACT(8); break;
            case 16: PHC_AD_ACT(16); break;
            case 32: PHC_AD_ACT(32); break;
            default: PHC_AD_ACT(64); break;
        }
#undef PHC_AD_ACT
#undef PHC_AD_LAUNCH
        }
        rc = check_launch("aggregate_enc_dx");
        if (rc) return rc;
    }
    if (dtables_flat) {
        PHC_REQUIRE(E >= 0 && (E == 0 || (edge_src_i64 && edge_dst_i64)), PHC_ERR_INVALID_ARG, "phc_aggregate_enc_bwd: edge list missing");
        return launch_table_grad<1>(dout, bond, m, E, h, n, k, act, reinterpret_cast<const long long *>(edge_src_i64),
                                    reinterpret_cast<const long long *>(edge_dst_i64), x, tables, dtables_flat, ws, ws_bytes, s,
                                    "phc_aggregate_enc_bwd");
    }
    return PHC_OK;
}

size_t phc_table_grads_ws_bytes(int64_t E, int F, const int32_t *vocab, int n, int k, int num_jobs) {
    EncMeta m;
    if (num_jobs < 1 || make_meta("phc_table_grads_ws_bytes", F, vocab, &m)) return 0;
    size_t total = 0;
    for (int c0 = 0; c0 < num_jobs; c0 += TG_JOBS_PER_LAUNCH) {
        const int cn = num_jobs - c0 < TG_JOBS_PER_LAUNCH ? num_jobs - c0 : TG_JOBS_PER_LAUNCH;
        total += sizeof(float) * tg_plan(E, m.R, n * k, cn).slab_floats * cn;
    }
    return total;
}

int phc_table_grads_batched(const phc_table_grad_job *jobs, int num_jobs, const int64_t *bond_i64, const int32_t *vocab,
                            const int64_t *edge_src_i64, const int64_t *edge_dst_i64, int64_t E, int F, int n, int k, int act,
                            void *ws, size_t ws_bytes, void *stream) {
    EncMeta m;
    int rc = make_meta("phc_table_grads_batched", F, vocab, &m);
    if (rc) return rc;
    PHC_REQUIRE(num_jobs >= 1 && jobs && E >= 1 && n >= 1 && k >= 1, PHC_ERR_INVALID_ARG, "phc_table_grads_batched: bad arguments");
    PHC_REQUIRE(act >= PHC_ACT_IDENTITY && act <= PHC_ACT_SELU, PHC_ERR_INVALID_ARG, "phc_table_grads_batched: unknown activation %d", act);
    PHC_REQUIRE(bond_i64 && edge_src_i64 && edge_dst_i64, PHC_ERR_INVALID_ARG, "phc_table_grads_batched: null pointer");
    PHC_REQUIRE(ws && ws_bytes >= phc_table_grads_ws_bytes(E, F, vocab, n, k, num_jobs), PHC_ERR_WORKSPACE,
                "phc_table_grads_batched: workspace too small");
    const int h = n * k;
    hipStream_t s = (hipStream_t)stream;
    const long long *idx = reinterpret_cast<const long long *>(bond_i64);
    const long long *src = reinterpret_cast<const long long *>(edge_src_i64);
    const long long *dst = reinterpret_cast<const long long *>(edge_dst_i64);
    float *wsf = static_cast<float *>(ws);
    for (int c0 = 0; c0 < num_jobs; c0 += TG_JOBS_PER_LAUNCH) {
        const int cn = num_jobs - c0 < TG_JOBS_PER_LAUNCH ? num_jobs - c0 : TG_JOBS_PER_LAUNCH;
        const TgPlan p = tg_plan(E, m.R, h, cn);
        TgBatch b;
        int vec4 = (k % 4 == 0);
        for (int q = 0; q < cn; ++q) {
            const phc_table_grad_job &jb = jobs[c0 + q];
            PHC_REQUIRE(jb.dout && jb.tables && jb.dtables_flat && (act == PHC_ACT_IDENTITY || jb.x), PHC_ERR_INVALID_ARG,
                        "phc_table_grads_batched: job %d: null pointer", c0 + q);
            b.j[q].dout = jb.dout; b.j[q].x = jb.x; b.j[q].tables = jb.tables; b.j[q].flat = jb.dtables_flat;
            b.j[q].slabs = wsf;
            wsf += p.slab_floats;
            vec4 = vec4 && al16(jb.dout) && (!jb.x || al16(jb.x));
        }
        const dim3 grid(p.tiles_i * p.tiles_j, p.num_slabs, cn);
#define PHC_TGB_LAUNCH(ACTC)                                                                                              \
    PHC_LAUNCH((table_grad_mfma_batched_kernel<ACTC>), grid, dim3(ENC_THREADS), 0, s, b, idx, m, (long long)E,            \
               p.items_per_split, h, p.tiles_j, src, dst, n, k, vec4)
        {
            ProfScope prof(K_TABLE_GRAD, s);
            switch (act) {
                case PHC_ACT_IDENTITY: PHC_TGB_LAUNCH(PHC_ACT_IDENTITY); break;
                case PHC_ACT_RELU: PHC_TGB_LAUNCH(PHC_ACT_RELU); break;
                case PHC_ACT_LRELU: PHC_TGB_LAUNCH(PHC_ACT_LRELU); break;
                case PHC_ACT_ELU: PHC_TGB_LAUNCH(PHC_ACT_ELU); break;
                case PHC_ACT_SWISH: PHC_TGB_LAUNCH(PHC_ACT_SWISH); break;
                default: PHC_TGB_LAUNCH(PHC_ACT_SELU); break;
            }
            rc = check_launch("table_grads_batched");
            if (rc) return rc;
        }
#undef PHC_TGB_LAUNCH
        ProfScope prof(K_TABLE_GRAD_REDUCE, s);
        PHC_LAUNCH(table_grad_reduce_batched_kernel, dim3((m.R * h + 63) / 64, cn), dim3(256), 0, s, b, p.num_slabs, m.R, n, k);
        rc = check_launch("table_grads_batched(reduce)");
        if (rc) return rc;
    }
    return PHC_OK;
}

size_t phc_table_grads_counted_ws_bytes(int64_t n_dst, int F, const int32_t *vocab, int n, int k, int num_jobs) {
    return phc_table_grads_ws_bytes(n_dst, F, vocab, n, k, num_jobs);     // same plan, items = destination nodes
}

int phc_table_grads_counted(const phc_table_grad_job *jobs, int num_jobs, const int32_t *rowptr, const uint32_t *codes,
                            int64_t n_dst, int64_t E, const int32_t *vocab, int F, int n, int k, void *ws, size_t ws_bytes,
                            void *stream) {
    EncMeta m;
    int rc = make_meta("phc_table_grads_counted", F, vocab, &m);
    if (rc) return rc;
    PHC_REQUIRE(num_jobs >= 1 && jobs && n_dst >= 1 && E >= 1 && n >= 1 && k >= 1, PHC_ERR_INVALID_ARG,
                "phc_table_grads_counted: bad arguments");
    PHC_REQUIRE(rowptr && codes, PHC_ERR_INVALID_ARG, "phc_table_grads_counted: null pointer");
    PHC_REQUIRE(F <= 4 && m.R <= 256, PHC_ERR_UNSUPPORTED, "phc_table_grads_counted: packed codes need F <= 4, sum(vocab) <= 256");
    PHC_REQUIRE(ws && ws_bytes >= phc_table_grads_counted_ws_bytes(n_dst, F, vocab, n, k, num_jobs), PHC_ERR_WORKSPACE,
                "phc_table_grads_counted: workspace too small");
    const int h = n * k;
    hipStream_t s = (hipStream_t)stream;
    float *wsf = static_cast<float *>(ws);
    for (int c0 = 0; c0 < num_jobs; c0 += TG_JOBS_PER_LAUNCH) {
        const int cn = num_jobs - c0 < TG_JOBS_PER_LAUNCH ? num_jobs - c0 : TG_JOBS_PER_LAUNCH;
        const TgPlan p = tg_plan(n_dst, m.R, h, cn);
        TgBatch b;
        int vec4 = (k % 4 == 0);
        for (int q = 0; q < cn; ++q) {
            const phc_table_grad_job &jb = jobs[c0 + q];
            PHC_REQUIRE(jb.dout && jb.dtables_flat, PHC_ERR_INVALID_ARG, "phc_table_grads_counted: job %d: null pointer", c0 + q);
            b.j[q].dout = jb.dout; b.j[q].x = nullptr; b.j[q].tables = nullptr; b.j[q].flat = jb.dtables_flat;
            b.j[q].slabs = wsf;
            wsf += p.slab_floats;
            vec4 = vec4 && al16(jb.dout);
        }
        {
            ProfScope prof(K_TABLE_GRAD, s);
            PHC_LAUNCH(table_grad_counted_batched_kernel, dim3(p.tiles_i * p.tiles_j, p.num_slabs, cn), dim3(ENC_THREADS), 0, s, b, rowptr,
                       codes, (long long)E, m, (long long)n_dst, p.items_per_split, h, p.tiles_j, vec4);
            rc = check_launch("table_grads_counted");
            if (rc) return rc;
        }
        ProfScope prof(K_TABLE_GRAD_REDUCE, s);
        PHC_LAUNCH(table_grad_reduce_batched_kernel, dim3((m.R * h + 63) / 64, cn), dim3(256), 0, s, b, p.num_slabs, m.R, n, k);
        rc = check_launch("table_grads_counted(reduce)");
        if (rc) return rc;
    }
    return PHC_OK;
}

// With table-gradient workgroups in the launch, the 128-multiple layers take the 128 x 64 register-resident tiles INSIDE it (two
// wavefronts per SIMD: the latency-bound table gathers run beside the MFMA-bound dH wavefronts); without them the 128 x 128
// tiles in their own launch.
static int tail_direct_form(int num_tjobs, int64_t N_atoms) {
    constexpr int forced = 0;
    if (forced == 2 || forced == 3) return forced;
    return (num_tjobs > 0 || N_atoms > 0) ? 3 : 2;
}

static int tail_plan(const phc_phm_grad_job *jobs, int num_jobs, int num_tjobs, int64_t E, int64_t bond_n_dst, int F_b,
                     const int32_t *vocab_b, int64_t N_atoms, int F_a, const int32_t *vocab_a, int n, int k, EncMeta *mb, EncMeta *ma,
                     TgPlan *pb, TgPlan *pa, size_t *phm_floats) {
    PHC_REQUIRE(num_jobs >= 1 && num_jobs <= GRAD_JOBS_PER_LAUNCH && num_tjobs >= 0 && num_tjobs <= TG_JOBS_PER_LAUNCH && n >= 1 && k >= 1,
                PHC_ERR_UNSUPPORTED, "phc_step_tail: %d PHM jobs (1..%d), %d table jobs (0..%d)", num_jobs, GRAD_JOBS_PER_LAUNCH, num_tjobs,
                TG_JOBS_PER_LAUNCH);
    int rc = check_grad_jobs("phc_step_tail", jobs, num_jobs);
    if (rc) return rc;
    const int h = n * k;
    *pb = TgPlan();
    *pa = TgPlan();
    if (num_tjobs > 0) {
        rc = make_meta("phc_step_tail", F_b, vocab_b, mb);
        if (rc) return rc;
        PHC_REQUIRE(E >= 1 && bond_n_dst >= 0, PHC_ERR_INVALID_ARG, "phc_step_tail: table jobs without edges");
        *pb = tg_plan(bond_n_dst > 0 ? bond_n_dst : E, mb->R, h, num_tjobs);      // counted form: items = destination nodes
        if (bond_n_dst > 0 && mb->R <= 32 && tg_direct_shape_ok(bond_n_dst, h)) {   // either form may run: room for the larger slab set
            const TgPlan d = tg_plan_direct_bond(bond_n_dst, mb->R, h, num_tjobs);
            if (d.slab_floats > pb->slab_floats) pb->slab_floats = d.slab_floats;
        }
    }
    if (N_atoms > 0) {
        rc = make_meta("phc_step_tail", F_a, vocab_a, ma);
        if (rc) return rc;
        *pa = tg_plan(N_atoms, ma->R, h);
        if (tg_direct_shape_ok(N_atoms, h)) {                // either form may run (the register-resident one needs an aligned dout):
            const TgPlan d = tg_plan_direct(N_atoms, ma->R, h);                   // the workspace holds the larger slab set
            if (d.slab_floats > pa->slab_floats) pa->slab_floats = d.slab_floats;
        }
    }
    *phm_floats = grad_chunk_ws_floats(jobs, num_jobs, tail_direct_form(num_tjobs, N_atoms));
    return PHC_OK;
}

size_t phc_step_tail_ws_bytes(const phc_phm_grad_job *jobs, int num_jobs, int num_tjobs, int64_t E, int64_t bond_n_dst, int F_b,
                              const int32_t *vocab_b, int64_t N_atoms, int F_a, const int32_t *vocab_a, int n, int k) {
    EncMeta mb, ma;
    TgPlan pb, pa;
    size_t phm = 0;
    if (tail_plan(jobs, num_jobs, num_tjobs, E, bond_n_dst, F_b, vocab_b, N_atoms, F_a, vocab_a, n, k, &mb, &ma, &pb, &pa, &phm)) return 0;
    return sizeof(float) * (phm + (num_tjobs > 0 ? pb.slab_floats * (size_t)num_tjobs : 0) + (N_atoms > 0 ? pa.slab_floats : 0));
}

static int step_tail_impl(const phc_phm_grad_job *jobs, int num_jobs, const phc_table_grad_job *tjobs, int num_tjobs, const int64_t *bond_i64,
                          const int32_t *vocab_b, const int64_t *edge_src_i64, const int64_t *edge_dst_i64, int64_t E, int F_b, int act,
                          const int32_t *bond_rowptr, const uint32_t *bond_codes, const float *bond_counts, int64_t bond_n_dst,
                          const float *atom_dout, const int64_t *atom_idx_i64, int64_t N_atoms, int F_a, const int32_t *vocab_a,
                          float *atom_dtables_flat, int n, int k, void *ws, size_t ws_bytes, const phc_tail_extras *ex, void *stream);

int phc_step_tail(const phc_phm_grad_job *jobs, int num_jobs, const phc_table_grad_job *tjobs, int num_tjobs, const int64_t *bond_i64,
                  const int32_t *vocab_b, const int64_t *edge_src_i64, const int64_t *edge_dst_i64, int64_t E, int F_b, int act,
                  const int32_t *bond_rowptr, const uint32_t *bond_codes, const float *bond_counts, int64_t bond_n_dst, const float *atom_dout,
                  const int64_t *atom_idx_i64, int64_t N_atoms, int F_a, const int32_t *vocab_a, float *atom_dtables_flat, int n, int k,
                  void *ws, size_t ws_bytes, void *stream) {
    return step_tail_impl(jobs, num_jobs, tjobs, num_tjobs, bond_i64, vocab_b, edge_src_i64, edge_dst_i64, E, F_b, act, bond_rowptr,
                          bond_codes, bond_counts, bond_n_dst, atom_dout, atom_idx_i64, N_atoms, F_a, vocab_a, atom_dtables_flat, n, k, ws,
                          ws_bytes, nullptr, stream);
}

int phc_tail_reg_max_jobs(void) { return TAIL_REG_MAX_JOBS; }
size_t phc_tail_extras_ws_floats(const phc_tail_extras *ex) {
    if (!ex || ex->reg_num <= 0 || !ex->reg_n || !ex->reg_cells) return 0;
    size_t f = 0;
    for (int q = 0; q < ex->reg_num; ++q)
        f += ((size_t)ex->reg_n[q] * ex->reg_cells[q] + 3) / 4 * 4 + (size_t)(ex->reg_cells[q] + 255) / 256;
    return f + 4;
}

int phc_step_tail_ex(const phc_phm_grad_job *jobs, int num_jobs, const phc_table_grad_job *tjobs, int num_tjobs, const int64_t *bond_i64,
                     const int32_t *vocab_b, const int64_t *edge_src_i64, const int64_t *edge_dst_i64, int64_t E, int F_b, int act,
                     const int32_t *bond_rowptr, const uint32_t *bond_codes, const float *bond_counts, int64_t bond_n_dst,
                     const float *atom_dout, const int64_t *atom_idx_i64, int64_t N_atoms, int F_a, const int32_t *vocab_a,
                     float *atom_dtables_flat, int n, int k, void *ws, size_t ws_bytes, const phc_tail_extras *extras, void *stream) {
    return step_tail_impl(jobs, num_jobs, tjobs, num_tjobs, bond_i64, vocab_b, edge_src_i64, edge_dst_i64, E, F_b, act, bond_rowptr,
                          bond_codes, bond_counts, bond_n_dst, atom_dout, atom_idx_i64, N_atoms, F_a, vocab_a, atom_dtables_flat, n, k, ws,
                          ws_bytes, extras, stream);
}

// The launch's 512 workgroup slots (two per CU) among its three families.  The table families' workgroups are latency-bound AND use the
// matrix pipes (one-hot operands); with 256 workgroups each they fill every slot for the first 20 us of the molecule step's tail and the
// dH family starts behind them (scripts/dbg/tail_wg_timeline.py: 252 + 244 table workgroups of 19 us, then 465 dH workgroups of 38 us —
// 64 us for 27 000 workgroup-us of work, 53 us if the slots were full throughout).  Where the tables are a fifth or more of the
// launch's work (a fifth: see the threshold below), every family gets slots in proportion to its work and all of them run from start to end: fewer, longer table
// workgroups (fewer slabs to reduce), the dH family planned for the slots that are left.  Measured (round 6, alternating runs on one
// box): molhiv_b512 0.4986 -> 0.4842 ms and 0.4951 -> 0.4869; at hidden 256 the tables are a sixth of the work and the same rule costs
// 15 - 40 us (their matrix work slows the co-resident dH workgroups for the whole launch instead of for its first 60 us): unchanged
// there.  Work estimates in workgroup-us from the same clocks: table tile rows 0.085 (atoms) / 0.03 per layer (bonds: 0.04 on their own, less beside the dH family), dH tile rows 0.075.
struct TailShares {
    long long target_a, target_b;
    int slots_dh;
};
static TailShares tail_shares(const phc_phm_grad_job *jobs, int num_jobs, int bond_jobs, int64_t bond_n_dst, int64_t N_atoms, int R_a, int h,
                              bool eligible) {
    TailShares sh = {TG_DIRECT_TARGET, TG_DIRECT_TARGET, 512};
#ifdef PHC_TAIL_KNOBS
    // experiment build only: PHC_TAIL_SHARES="atoms,bonds,dH" overrides the rule (scripts/dbg/tail_shares_trace.sh)
    if (const char *e = getenv("PHC_TAIL_SHARES")) {
        long long a = 0, b = 0;
        int d = 0;
        if (sscanf(e, "%lld,%lld,%d", &a, &b, &d) == 3 && eligible) return TailShares{a, b, d};
    }
#endif
    if (!eligible || N_atoms <= 0) return sh;
    const double w_a = 0.085 * (double)N_atoms * ((R_a + 32 * TD_MB - 1) / (32 * TD_MB)) * ((h + 63) / 64);
    const double w_b = 0.03 * (double)bond_n_dst * ((h + 63) / 64) * bond_jobs;
    double w_d = 0.0;
    for (int q = 0; q < num_jobs; ++q) {
        const int in = jobs[q].n * jobs[q].kin, out = jobs[q].n * jobs[q].kout;
        if (in % 4 == 0 && in >= 96 && out >= 32) w_d += 0.075 * (double)jobs[q].N * ((in + 127) / 128) * ((out + 63) / 64);
    }
    const double W = w_a + w_b + w_d;
    if (w_d <= 0.0 || (w_a + w_b) < 0.20 * W) return sh;  // (softmax layers at hidden 128 — atom tables only, 21 % — gain 15 - 20 us; 2 x 200's 18 % nothing)
    auto share = [&](double w) { long long v = (long long)(0.85 * 512.0 * w / W / 8.0 + 0.5) * 8; return v < 32 ? 32LL : (v > 256 ? 256LL : v); };
    sh.target_a = share(w_a);
    sh.target_b = bond_jobs > 0 ? share(w_b) : 0;
    sh.slots_dh = (int)(512 - sh.target_a - sh.target_b);
    if (sh.slots_dh < 192) return TailShares{TG_DIRECT_TARGET, TG_DIRECT_TARGET, 512};      // (tables over three fifths of the work: as before)
    return sh;
}

static int step_tail_impl(const phc_phm_grad_job *jobs, int num_jobs, const phc_table_grad_job *tjobs, int num_tjobs, const int64_t *bond_i64,
                          const int32_t *vocab_b, const int64_t *edge_src_i64, const int64_t *edge_dst_i64, int64_t E, int F_b, int act,
                          const int32_t *bond_rowptr, const uint32_t *bond_codes, const float *bond_counts, int64_t bond_n_dst,
                          const float *atom_dout, const int64_t *atom_idx_i64, int64_t N_atoms, int F_a, const int32_t *vocab_a,
                          float *atom_dtables_flat, int n, int k, void *ws, size_t ws_bytes, const phc_tail_extras *ex, void *stream) {
    EncMeta mb, ma;
    TgPlan pb, pa;
    size_t phm_floats = 0;
    const bool counted = num_tjobs > 0 && bond_n_dst > 0;
    int rc = tail_plan(jobs, num_jobs, num_tjobs, E, bond_n_dst, F_b, vocab_b, N_atoms, F_a, vocab_a, n, k, &mb, &ma, &pb, &pa, &phm_floats);
    if (rc) return rc;
    PHC_REQUIRE(act >= PHC_ACT_IDENTITY && act <= PHC_ACT_SELU, PHC_ERR_INVALID_ARG, "phc_step_tail: unknown activation %d", act);
    if (counted) {
        PHC_REQUIRE(act == PHC_ACT_IDENTITY, PHC_ERR_INVALID_ARG, "phc_step_tail: the counted bond form needs an identity message function");
        PHC_REQUIRE(tjobs && bond_rowptr && bond_codes, PHC_ERR_INVALID_ARG, "phc_step_tail: null pointer (counted bond form)");
        PHC_REQUIRE(F_b <= 4 && mb.R <= 256, PHC_ERR_UNSUPPORTED, "phc_step_tail: packed codes need F <= 4, sum(vocab) <= 256");
    } else {
        PHC_REQUIRE(num_tjobs == 0 || (tjobs && bond_i64 && edge_src_i64 && edge_dst_i64), PHC_ERR_INVALID_ARG, "phc_step_tail: null pointer");
    }
    PHC_REQUIRE(N_atoms == 0 || (atom_dout && atom_idx_i64 && atom_dtables_flat), PHC_ERR_INVALID_ARG, "phc_step_tail: null atom pointer");
    const size_t need = phc_step_tail_ws_bytes(jobs, num_jobs, num_tjobs, E, bond_n_dst, F_b, vocab_b, N_atoms, F_a, vocab_a, n, k);
    PHC_REQUIRE(ws && ws_bytes >= need && (reinterpret_cast<uintptr_t>(ws) & 15) == 0, PHC_ERR_WORKSPACE,
                "phc_step_tail: workspace %zu B < required %zu B (or not 16-byte aligned)", ws_bytes, need);
    hipStream_t s = (hipStream_t)stream;
    const int h = n * k;
    float *wsf = static_cast<float *>(ws);
    // which table families take their register-resident forms (decided before the split plans: the families share the launch's slots)
    bool bond_direct = counted && bond_counts && mb.R <= 32 && tg_direct_shape_ok(bond_n_dst, h);
    for (int q = 0; q < num_tjobs && bond_direct; ++q) bond_direct = tjobs[q].dout && (reinterpret_cast<uintptr_t>(tjobs[q].dout) & 7) == 0;
    const bool atom_direct = N_atoms > 0 && tg_direct_shape_ok(N_atoms, h) && (reinterpret_cast<uintptr_t>(atom_dout) & 7) == 0;
    const TailShares shares = tail_shares(jobs, num_jobs, bond_direct ? num_tjobs : 0, bond_n_dst, atom_direct ? N_atoms : 0, atom_direct ? ma.R : 0, h,
                                          atom_direct && (num_tjobs == 0 || bond_direct));
    GradLaunch gl;
    rc = prepare_grad_launch("phc_step_tail", jobs, num_jobs, wsf, &gl, tail_direct_form(num_tjobs, N_atoms), shares.slots_dh);
    if (rc) return rc;
    wsf += phm_floats;
    TailTables t;
    memset(&t, 0, sizeof(t));
    t.n = n; t.k = k; t.h = h;
    if (num_tjobs > 0) {
        t.mb = mb;
        t.bond_idx = reinterpret_cast<const long long *>(bond_i64);
        t.src = reinterpret_cast<const long long *>(edge_src_i64);
        t.dst = reinterpret_cast<const long long *>(edge_dst_i64);
        t.E = counted ? bond_n_dst : E;
        t.rowptr_b = counted ? bond_rowptr : nullptr;
        t.codes_b = counted ? bond_codes : nullptr;
        t.n_edges_b = E;
        if (bond_direct) {                                   // register-resident form: its own split plan, the same slab stride
            const TgPlan d = tg_plan_direct_bond(bond_n_dst, mb.R, h, num_tjobs, shares.target_b);
            pb.items_per_split = d.items_per_split; pb.tiles_i = d.tiles_i; pb.tiles_j = d.tiles_j; pb.num_slabs = d.num_slabs;
            t.counts_b = bond_counts;
            t.cr_b = mb.R <= 16 ? 16 : 32;
        }
        t.ips_b = pb.items_per_split; t.tiles_j_b = pb.tiles_j; t.tiles_b = pb.tiles_i * pb.tiles_j; t.slabs_b = pb.num_slabs;
        t.jobs_b = num_tjobs;
        int vec4 = (k % 4 == 0);
        for (int q = 0; q < num_tjobs; ++q) {
            PHC_REQUIRE(tjobs[q].dout && (counted || tjobs[q].tables) && tjobs[q].dtables_flat && (act == PHC_ACT_IDENTITY || tjobs[q].x),
                        PHC_ERR_INVALID_ARG, "phc_step_tail: table job %d: null pointer", q);
            t.bond.j[q].dout = tjobs[q].dout; t.bond.j[q].x = tjobs[q].x; t.bond.j[q].tables = tjobs[q].tables;
            t.bond.j[q].flat = tjobs[q].dtables_flat; t.bond.j[q].slabs = wsf;
            wsf += pb.slab_floats;
            vec4 = vec4 && al16(tjobs[q].dout) && (!tjobs[q].x || al16(tjobs[q].x));
        }
        t.vec4_b = vec4;
        if (bond_direct) {
            t.tiles_b = pb.tiles_j;                          // 64-column tiles; the layer groups are the outer index
            t.nb_bond = (unsigned)pb.tiles_i * (unsigned)pb.tiles_j * (unsigned)t.slabs_b;
        } else {
            t.nb_bond = (unsigned)t.tiles_b * (unsigned)t.slabs_b * (unsigned)num_tjobs;
        }
        t.nr_bond = (unsigned)((mb.R * h + 63) / 64) * (unsigned)num_tjobs;
    }
    if (N_atoms > 0) {
        t.ma = ma;
        t.atom_dout = atom_dout;
        t.atom_idx = reinterpret_cast<const long long *>(atom_idx_i64);
        t.atom_slabs = wsf; t.atom_flat = atom_dtables_flat;
        t.atom_direct = atom_direct;
        if (t.atom_direct) pa = tg_plan_direct(N_atoms, ma.R, h, shares.target_a);
        t.N_a = N_atoms; t.ips_a = pa.items_per_split; t.tiles_j_a = pa.tiles_j; t.tiles_a = pa.tiles_i * pa.tiles_j; t.slabs_a = pa.num_slabs;
        t.vec4_a = (k % 4 == 0) && al16(atom_dout);
        t.nb_atom = (unsigned)t.tiles_a * (unsigned)t.slabs_a;
        t.nr_atom = (unsigned)((ma.R * h + 63) / 64);
    }
    t.nr_phm = gl.nb_reduce;
    t.dh_first = 0;                                          // (the long dH workgroups first: round 4 within noise at molecule scale; re-measured late in round 5:
                                                             //  zinc_b1024 1.836 -> 1.866 ms, molpcba_b1024 2.073 -> 2.063, 4 x 180 1.425 -> 1.418, 2 x 200 0.845 -> 0.838 — no rule)
    if (getenv("PHC_DEBUG_PLANS"))
        fprintf(stderr, "[phc tail] atoms: %lld nodes direct=%d wgs=%u | bonds: %d jobs direct=%d wgs=%u | dH wgs=%u (own launch %u)\n",
                (long long)N_atoms, t.atom_direct, t.nb_atom, num_tjobs, t.counts_b ? 1 : 0, t.nb_bond, gl.nb_partial, gl.nb_direct);
    const unsigned nb1 = t.nb_atom + t.nb_bond + gl.nb_partial, nb2 = t.nr_phm + t.nr_bond + t.nr_atom;
    rc = launch_grad_direct("step_tail(direct)", gl, s);     // the 128-multiple layers: register-resident tiles, their own launch
    if (rc) return rc;
    rc = launch_grad_partial128("step_tail(128 x 128 LDS tiles)", gl, s);   // wide layers outside the register-resident form (e.g. a
    if (rc) return rc;                                       // smooth activation folded into the operand): their own launch too
    if (nb1 > 0) {
        ProfScope prof(K_PHM_DH_PARTIAL, s);
        GradBatch b64 = gl.batch;                            // this launch takes the 64 x 64 tile jobs only
        for (int q = 0; q < b64.num; ++q)
            if (b64.j[q].t128 != 0 && b64.j[q].t128 != 3) b64.j[q].blk0_partial = 0xffffffffu;
#define PHC_TAIL_LAUNCH(ACTC) PHC_LAUNCH((tail_partial_kernel<ACTC>), dim3(nb1), dim3(256), 0, s, b64, t)
        switch (act) {
            case PHC_ACT_IDENTITY: PHC_TAIL_LAUNCH(PHC_ACT_IDENTITY); break;
            case PHC_ACT_RELU: PHC_TAIL_LAUNCH(PHC_ACT_RELU); break;
            case PHC_ACT_LRELU: PHC_TAIL_LAUNCH(PHC_ACT_LRELU); break;
            case PHC_ACT_ELU: PHC_TAIL_LAUNCH(PHC_ACT_ELU); break;
            case PHC_ACT_SWISH: PHC_TAIL_LAUNCH(PHC_ACT_SWISH); break;
            default: PHC_TAIL_LAUNCH(PHC_ACT_SELU); break;
        }
#undef PHC_TAIL_LAUNCH
        rc = check_launch("step_tail(partial)");
        if (rc) return rc;
    }

    // ---- extras of the last two launches: the weight regulariser (its cells ride in the reduce launch, its gradient is added
    // where dW is written) and the gate map's parameter gradients (workgroups of the last launch) ----
    TailReg reg;
    memset(&reg, 0, sizeof(reg));
    GateGradHost gate;
    memset(&gate, 0, sizeof(gate));
    bool has_reg = false, has_gate = false;
    unsigned nr_reg = 0;
    if (ex) {
        has_gate = ex->gate_job >= 0;
        if (has_gate) {
            PHC_REQUIRE(ex->gate_job < num_jobs && ex->gate_H && ex->gate_w && ex->gate_A && ex->gate_W, PHC_ERR_INVALID_ARG,
                        "phc_step_tail_ex: gate job %d / null pointer", ex->gate_job);
            PHC_REQUIRE(!ex->gate_db || ex->gate_phm_b, PHC_ERR_INVALID_ARG, "phc_step_tail_ex: gate db without a PHM bias");
            gate.job = ex->gate_job; gate.H = ex->gate_H; gate.gate_w = ex->gate_w; gate.phm_b = ex->gate_phm_b; gate.A = ex->gate_A;
            gate.W = ex->gate_W; gate.n = ex->gate_n; gate.kin = ex->gate_kin; gate.kout = ex->gate_kout; gate.kg = ex->gate_kg;
            gate.dA = ex->gate_dA; gate.dW = ex->gate_dW; gate.db = ex->gate_db; gate.dWg = ex->gate_dWg; gate.dbg = nullptr;
        }
        if (ex->reg_num > 0) {
            PHC_REQUIRE(ex->reg_num <= TAIL_REG_MAX_JOBS && ex->reg_W && ex->reg_n && ex->reg_cells && ex->reg_job && ex->reg_ws &&
                            (ex->reg_p == 1 || ex->reg_p == 2) && (reinterpret_cast<uintptr_t>(ex->reg_ws) & 15) == 0,
                        PHC_ERR_INVALID_ARG, "phc_step_tail_ex: regulariser table (%d layers, p = %d)", ex->reg_num, ex->reg_p);
            has_reg = true;
            reg.num = ex->reg_num; reg.p = ex->reg_p; reg.coef = ex->reg_coef; reg.value = ex->reg_value; reg.loss_inout = ex->reg_loss_inout;
            float *side = ex->reg_ws;
            for (int q = 0; q < ex->reg_num; ++q) {
                const int nq = ex->reg_n[q], cells = ex->reg_cells[q], jq = ex->reg_job[q];
                PHC_REQUIRE(ex->reg_W[q] && nq >= 1 && nq <= TAIL_REG_MAX_N && cells >= 1, PHC_ERR_INVALID_ARG,
                            "phc_step_tail_ex: regularised layer %d: n = %d, cells = %d", q, nq, cells);
                reg.W[q] = ex->reg_W[q]; reg.n[q] = nq; reg.cells[q] = cells; reg.grad[q] = side;
                reg.first_block[q] = (int)nr_reg;
                nr_reg += (unsigned)((cells + 255) / 256);
                // who writes this layer's dW: a PHM job of the batch (same W pointer, not transposed) or the gate map (-2)
                if (jq == -2) {
                    PHC_REQUIRE(has_gate && ex->gate_W == ex->reg_W[q] && ex->gate_n * ex->gate_kin * ex->gate_kout == nq * cells,
                                PHC_ERR_INVALID_ARG, "phc_step_tail_ex: regularised layer %d names the gate map", q);
                    gate.reg_add = side;
                } else {
                    PHC_REQUIRE(jq >= 0 && jq < num_jobs && gl.batch.j[jq].W == ex->reg_W[q] && !gl.batch.j[jq].dw_transposed &&
                                    gl.batch.j[jq].dW && gl.batch.j[jq].n * gl.batch.j[jq].kin * gl.batch.j[jq].kout == nq * cells,
                                PHC_ERR_INVALID_ARG, "phc_step_tail_ex: regularised layer %d does not match job %d", q, jq);
                    gl.batch.j[jq].reg_add = side;
                }
                side += ((size_t)nq * cells + 3) / 4 * 4;
            }
            reg.first_block[ex->reg_num] = (int)nr_reg;
            reg.shares = side;
        }
    }
    {
        ProfScope prof(K_PHM_DH_REDUCE, s);
        PHC_LAUNCH(tail_reduce_kernel, dim3(nb2 + nr_reg), dim3(256), 0, s, gl.batch, t, reg);
        rc = check_launch("step_tail(reduce)");
        if (rc) return rc;
    }
    ProfScope prof(K_PHM_CONTRACT, s);
    return launch_tail_finish("phc_step_tail", gl, has_gate ? &gate : nullptr, has_reg ? &reg : nullptr, s);
}

}  // extern "C"
