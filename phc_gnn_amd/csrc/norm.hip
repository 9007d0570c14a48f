// Per-column batch normalisation of component-major [N, h] activations, fused with the
// activation and skip add that follow it.
//
// Reference: NaivePHMNorm runs one BatchNorm1d(k) per hypercomplex component on a
// reshaped/permuted copy and concatenates (phc/hypercomplex/norm.py:30-35); the model then
// applies ReLU, dropout and the skip add as separate ops (models.py:208-215), PHMMLP applies
// norm -> activation (layers.py:351-353).  Batch-norm is per column, so the n component norms
// are ONE pass over the [N, h] matrix; the n separate weight/bias/running-stat tensors are
// addressed through by-value pointer tables (they stay the reference's Parameters/buffers).
//
//   forward : bn_stats_partial  (shifted sum / sum of squares per column, row ranges)
//             bn_apply          (fixed-order finalise, z = x_hat*w + b, out = act(z) + residual,
//                                running-stat update by workgroup 0)
//   backward: bn_bwd_partial    (sum g, sum g*x_hat per column, g = dy * act'(z))
//             bn_bwd_apply      (dx = w*invstd*(g - mean(g) - x_hat*mean(g*x_hat)), dweight, dbias)
// Rows >= *n_valid (device scalar, optional) are padding: excluded from the statistics,
// written as 0 and given zero gradient.  No atomics; fixed summation order.
#include "phc_common.h"

namespace phc {

constexpr int BN_THREADS = 256;
constexpr int BN_MAX_PARTIALS = 256;

__device__ __forceinline__ float bn_w(const BnParams &p, int col) { return p.has_weight ? p.weight[col / p.k][col % p.k] : 1.f; }
__device__ __forceinline__ float bn_b(const BnParams &p, int col) { return p.has_bias ? p.bias[col / p.k][col % p.k] : 0.f; }

template <int ACT>
__device__ __forceinline__ float bn_act(float z) { return act_fwd<ACT>(z); }

// (BnDrop, philox4x32_10, drop_factors, make_drop: phc_common.h — the H-stationary GEMM's norm-backward transform draws the same masks)

// ---- pass 1 (forward): per-column shifted sums over a row range --------------------------------
// thread -> (vector column c = tid % cols, row lane rl = tid / cols), VEC columns per thread;
// partial[b][0..h) = sum(x - pilot), partial[b][h..2h) = sum((x - pilot)^2), pilot = x[0][col]
template <int VEC>
__device__ __forceinline__ void ldv(const float *p, float (&v)[VEC]) {
    if constexpr (VEC == 4) {
        const float4 t = *reinterpret_cast<const float4 *>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
        v[0] = *p;
    }
}

// Block-local two-pass: partial[b][0..h) = sum_b(x), partial[b][h..2h) = M2_b = sum_b (x - mean_b)^2.
// (the finaliser merges blocks with Chan's formula: exact in the same sense as Welford, no E[x^2]-E[x]^2
// cancellation even when a few hub rows dominate a column)
template <int VEC>
__global__ __launch_bounds__(BN_THREADS) void bn_stats_partial_kernel(const float *__restrict__ x, const int *__restrict__ n_valid,
                                                                    long long N, int h, long long rows_per_block,
                                                                    float *__restrict__ partial) {
    __shared__ float red[VEC * BN_THREADS];
    __shared__ float bmean[VEC * BN_THREADS];
    const long long count = n_valid ? min((long long)*n_valid, N) : N;
    const int tid = threadIdx.x;
    const long long r0 = blockIdx.x * rows_per_block;
    const long long r1 = min(count, r0 + rows_per_block);
    const float nb = (float)max(r1 - r0, 1LL);
    const int hv = h / VEC;
    for (int cbase = 0; cbase < hv; cbase += BN_THREADS) {
        const int cols = min(BN_THREADS, hv - cbase);
        const int lanes = BN_THREADS / cols;                 // row lanes sharing this column block
        const int c = tid % cols, rl = tid / cols;
        const int col = (cbase + c) * VEC;
        float acc[VEC];
        // ---- pass A: column sums of this block's rows
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
        if (rl < lanes) {
#pragma unroll 4
            for (long long r = r0 + rl; r < r1; r += lanes) {
                float xv[VEC];
                ldv<VEC>(x + (size_t)r * h + col, xv);
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[v] += xv[v];
            }
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) red[v * BN_THREADS + tid] = acc[v];
        __syncthreads();
        for (int e = tid; e < cols * VEC; e += BN_THREADS) {
            const int cc = e / VEC, v = e - cc * VEC;
            float a = 0.f;
            for (int l = 0; l < lanes; ++l) a += red[v * BN_THREADS + l * cols + cc];      // fixed order
            partial[(size_t)blockIdx.x * 2 * h + (cbase + cc) * VEC + v] = a;
            bmean[v * BN_THREADS + cc] = a / nb;
        }
        __syncthreads();
        // ---- pass B: squared deviations from the block mean (rows are L1/L2 hot)
        float m[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            m[v] = bmean[v * BN_THREADS + c];
            acc[v] = 0.f;
        }
        if (rl < lanes) {
#pragma unroll 4
            for (long long r = r0 + rl; r < r1; r += lanes) {
                float xv[VEC];
                ldv<VEC>(x + (size_t)r * h + col, xv);
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const float d = xv[v] - m[v];
                    acc[v] = fmaf(d, d, acc[v]);
                }
            }
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) red[v * BN_THREADS + tid] = acc[v];
        __syncthreads();
        for (int e = tid; e < cols * VEC; e += BN_THREADS) {
            const int cc = e / VEC, v = e - cc * VEC;
            float a = 0.f;
            for (int l = 0; l < lanes; ++l) a += red[v * BN_THREADS + l * cols + cc];
            partial[(size_t)blockIdx.x * 2 * h + h + (cbase + cc) * VEC + v] = a;
        }
        __syncthreads();
    }
}

// Merge the per-block (sum, M2) pairs: 8 columns per workgroup, 32 partial lanes each, fixed order.
// Writes stat[0..h) = mean, [h..2h) = invstd, [2h..3h) = scale = invstd*w, [3h..4h) = shift, [4h..5h) = w, [5h..6h) = b
// (the backward reads w / b from there: the by-value pointer tables live in the kernarg segment, and a
// per-thread dynamic index into them costs two dependent long-latency loads); updates the running statistics.
__global__ __launch_bounds__(BN_THREADS) void bn_finalize_kernel(BnParams p, const int *__restrict__ n_valid, long long N, int h,
                                                               int training, float momentum, float eps,
                                                               const float *__restrict__ partial, int num_partials,
                                                               long long rows_per_block, float *__restrict__ stat,
                                                               const float *__restrict__ counts = nullptr) {
    __shared__ float r_sum[32][8], r_m2[32][8];
    const int tid = threadIdx.x;
    const int ci = tid & 7, pl = tid >> 3;
    const int c = blockIdx.x * 8 + ci;
    const bool owner = pl == 0 && c < h;
    // parameter / running-stat values of this column go in flight first (pointer table -> value: two hops)
    float w = 1.f, b = 0.f, rm_old = 0.f, rv_old = 1.f;
    float *rm = nullptr, *rv = nullptr;
    if (owner) {
        const int comp = c / p.k, kk = c - comp * p.k;
        if (p.has_weight) w = p.weight[comp][kk];
        if (p.has_bias) b = p.bias[comp][kk];
        if (p.has_running) {
            rm = p.running_mean[comp] + kk;
            rv = p.running_var[comp] + kk;
            rm_old = *rm;
            rv_old = *rv;
        }
    }
    const long long count = n_valid ? min((long long)*n_valid, N) : N;
    const float cnt = (float)max(count, 1LL);
    float mean = 0.f, invstd = 1.f;
    if (training) {
        // every partial this lane owns is loaded before anything is consumed: one memory latency, not sixteen
        constexpr int PER_LANE = BN_MAX_PARTIALS / 32;
        float ps[PER_LANE], pm[PER_LANE];
#pragma unroll
        for (int t = 0; t < PER_LANE; ++t) {
            const int bb = pl + 32 * t;
            const bool ok = c < h && bb < num_partials;
            const size_t off = (size_t)(ok ? bb : 0) * 2 * h + (c < h ? c : 0);
            const float a = partial[off], m2v = partial[off + h];
            ps[t] = ok ? a : 0.f;
            pm[t] = ok ? m2v : 0.f;
        }
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < PER_LANE; ++t) s += ps[t];
        r_sum[pl][ci] = s;
        __syncthreads();
        float tot = 0.f;
        for (int l = 0; l < 32; ++l) tot += r_sum[l][ci];
        mean = tot / cnt;
        float m2 = 0.f;
#pragma unroll
        for (int t = 0; t < PER_LANE; ++t) {
            const int bb = pl + 32 * t;
            // rows behind partial bb: derived from the row partition, or given (partials emitted by a GEMM epilogue)
            const bool inb = c < h && bb < num_partials;
            const float nbf = counts ? (inb ? counts[bb] : 0.f)
                                     : (float)min(rows_per_block, max(count - (long long)bb * rows_per_block, 0LL));
            if (inb && nbf > 0.f) {
                const float d = ps[t] / nbf - mean;
                m2 += pm[t] + nbf * d * d;
            }
        }
        r_m2[pl][ci] = m2;
        __syncthreads();
        if (owner) {
            float M2 = 0.f;
            for (int l = 0; l < 32; ++l) M2 += r_m2[l][ci];
            const float var = fmaxf(M2 / cnt, 0.f);
            invstd = 1.f / sqrtf(var + eps);
            if (p.has_running) {
                const float unbiased = count > 1 ? M2 / (cnt - 1.f) : var;
                *rm = (1.f - momentum) * rm_old + momentum * mean;
                *rv = (1.f - momentum) * rv_old + momentum * unbiased;
            }
        }
        if (blockIdx.x == 0 && p.has_nbt && tid < p.n_comp) *p.num_batches_tracked[tid] += 1;
    } else if (owner) {
        mean = rm_old;
        invstd = 1.f / sqrtf(rv_old + eps);
    }
    if (owner) {
        stat[c] = mean;
        stat[h + c] = invstd;
        stat[2 * h + c] = invstd * w;
        stat[3 * h + c] = b - mean * invstd * w;
        stat[4 * h + c] = w;
        stat[5 * h + c] = b;
    }
}

// ---- pass 3 (forward): out = act(x * scale + shift) (+ residual) ----------------------------------
template <int ACT>
__global__ __launch_bounds__(BN_THREADS) void bn_apply_kernel(const float *__restrict__ x, const int *__restrict__ n_valid, long long N,
                                                            int h, const float *__restrict__ residual,
                                                            const float *__restrict__ stat, float *__restrict__ out, BnDrop drop = BnDrop{}) {
    extern __shared__ float sm[];                            // scale[h], shift[h]
    float *scale = sm, *shift = sm + h;
    const long long count = n_valid ? min((long long)*n_valid, N) : N;
    const int tid = threadIdx.x;
    const bool dropping = drop.p > 0.f;
    const unsigned long long seed = dropping ? drop.state[0] : 0ull, step = dropping ? drop.state[1] : 0ull;
    for (int c = tid; c < h; c += BN_THREADS) {
        scale[c] = stat[2 * h + c];
        shift[c] = stat[3 * h + c];
    }
    __syncthreads();
    if ((h & 3) == 0) {
        const int h4 = h >> 2;
        const long long total4 = N * (long long)h4;
        for (long long e = (long long)blockIdx.x * BN_THREADS + tid; e < total4; e += (long long)gridDim.x * BN_THREADS) {
            const long long r = e / h4;
            const int c = (int)(e - r * h4) << 2;
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < count) {
                const float4 v = reinterpret_cast<const float4 *>(x)[e];
                o.x = bn_act<ACT>(fmaf(v.x, scale[c], shift[c]));
                o.y = bn_act<ACT>(fmaf(v.y, scale[c + 1], shift[c + 1]));
                o.z = bn_act<ACT>(fmaf(v.z, scale[c + 2], shift[c + 2]));
                o.w = bn_act<ACT>(fmaf(v.w, scale[c + 3], shift[c + 3]));
                if (dropping) {
                    const float4 f = drop_factors(drop, seed, step, r, c, h);
                    o.x *= f.x; o.y *= f.y; o.z *= f.z; o.w *= f.w;
                }
                if (residual) {
                    const float4 rr = reinterpret_cast<const float4 *>(residual)[e];
                    o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
                }
            }
            reinterpret_cast<float4 *>(out)[e] = o;
        }
    } else {
        const long long total = N * (long long)h;
        for (long long e = (long long)blockIdx.x * BN_THREADS + tid; e < total; e += (long long)gridDim.x * BN_THREADS) {
            const long long r = e / h;
            const int c = (int)(e - r * h);
            float o = 0.f;
            if (r < count) {
                o = bn_act<ACT>(fmaf(x[e], scale[c], shift[c]));
                if (residual) o += residual[e];
            }
            out[e] = o;
        }
    }
}

// ---- forward, ONE launch: the statistics merge in the prologue of the normalisation (partials emitted by a GEMM) -------
// The separate merge launch (bn_finalize_kernel: 16 workgroups, two dependent rounds of loads) costs 7.7 us of a replayed step
// (profiles/r01v_ablation_in_graph_costs.txt) for a few hundred KB.  Here a workgroup owns 32 COLUMNS x a row range: its 256
// threads (32 columns x 8 partial lanes) merge the <= 256 (sum, M2) pairs of those 32 columns — all loads of a lane in flight
// together, Chan's formula, fixed order — and then normalise their rows; the first row range also writes save_stat and the
// running statistics.  27 KB of partials per workgroup instead of a launch boundary.
constexpr int BNM_COLS = 32;
constexpr int BNM_LANES = BN_THREADS / BNM_COLS;            // 8
static inline int bn_col_blocks(int h) { return (h + BNM_COLS - 1) / BNM_COLS; }     // the last block may be partial (h % 4 == 0)
// PL = partials per lane: 16 serves <= 128 partials (the molecule-scale GEMM emits 108) with half the loads in flight
template <int ACT, int PL>
__global__ __launch_bounds__(BN_THREADS) void bn_apply_merged_kernel(const float *__restrict__ x, BnParams p, const int *__restrict__ n_valid,
                                                                   long long N, int h, float momentum, float eps,
                                                                   const float *__restrict__ partial, const float *__restrict__ counts,
                                                                   int num_partials, long long rows_per_group,
                                                                   const float *__restrict__ residual, float *__restrict__ stat,
                                                                   float *__restrict__ out, BnDrop drop = BnDrop{}) {
    __shared__ float r_a[BNM_LANES][BNM_COLS], r_b[BNM_LANES][BNM_COLS], s_scale[BNM_COLS], s_shift[BNM_COLS];
    const int tid = threadIdx.x;
    const int ci = tid & (BNM_COLS - 1), pl = tid / BNM_COLS;
    // the last column block may be partial (h % 4 == 0 only): its spare threads work on the last column again and store nothing
    const bool c_ok = blockIdx.x * BNM_COLS + ci < h;
    const int c = c_ok ? blockIdx.x * BNM_COLS + ci : h - 1;
    const long long count = n_valid ? min((long long)*n_valid, N) : N;
    const bool dropping = drop.p > 0.f;
    const unsigned long long seed = dropping ? drop.state[0] : 0ull, step = dropping ? drop.state[1] : 0ull;
    // this thread's partials: b = pl + 8 u
    float ps[PL], pm[PL], pn[PL];
#pragma unroll
    for (int u = 0; u < PL; ++u) {
        const int b = min(pl + BNM_LANES * u, num_partials - 1);
        ps[u] = partial[(size_t)b * 2 * h + c];
        pm[u] = partial[(size_t)b * 2 * h + h + c];
        pn[u] = counts[b];
    }
    // affine parameters / running statistics of the owner threads go in flight too (uniform walk of the pointer tables)
    const int comp = c / p.k, kk = c - comp * p.k;
    float w = 1.f, bq = 0.f, rm_old = 0.f, rv_old = 1.f;
    if (pl == 0) {
        for (int q = 0; q < p.n_comp; ++q)
            if (q == comp) {
                if (p.has_weight) w = p.weight[q][kk];
                if (p.has_bias) bq = p.bias[q][kk];
                if (p.has_running && blockIdx.y == 0) { rm_old = p.running_mean[q][kk]; rv_old = p.running_var[q][kk]; }
            }
    }
    float sl = 0.f;
#pragma unroll
    for (int u = 0; u < PL; ++u) sl += (pl + BNM_LANES * u < num_partials) ? ps[u] : 0.f;
    r_a[pl][ci] = sl;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int l = 0; l < BNM_LANES; ++l) tot += r_a[l][ci];
    const float cnt = (float)max(count, 1LL);
    const float mean = tot / cnt;
    float m2 = 0.f;
#pragma unroll
    for (int u = 0; u < PL; ++u) {
        const bool in = pl + BNM_LANES * u < num_partials;
        if (in && pn[u] > 0.f) {
            const float d = ps[u] / pn[u] - mean;
            m2 += pm[u] + pn[u] * d * d;
        }
    }
    r_b[pl][ci] = m2;
    __syncthreads();
    if (pl == 0) {
        float M2 = 0.f;
#pragma unroll
        for (int l = 0; l < BNM_LANES; ++l) M2 += r_b[l][ci];
        const float var = fmaxf(M2 / cnt, 0.f);
        const float invstd = 1.f / sqrtf(var + eps);
        s_scale[ci] = invstd * w;
        s_shift[ci] = bq - mean * invstd * w;
        if (blockIdx.y == 0 && c_ok) {
            stat[c] = mean;
            stat[h + c] = invstd;
            stat[2 * h + c] = invstd * w;
            stat[3 * h + c] = bq - mean * invstd * w;
            stat[4 * h + c] = w;
            stat[5 * h + c] = bq;
            if (p.has_running) {
                const float unbiased = count > 1 ? M2 / (cnt - 1.f) : var;
                for (int q = 0; q < p.n_comp; ++q)
                    if (q == comp) {
                        p.running_mean[q][kk] = (1.f - momentum) * rm_old + momentum * mean;
                        p.running_var[q][kk] = (1.f - momentum) * rv_old + momentum * unbiased;
                    }
            }
        }
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && p.has_nbt && tid < p.n_comp)
        for (int q = 0; q < p.n_comp; ++q)
            if (q == tid) *p.num_batches_tracked[q] += 1;
    __syncthreads();
    // rows of this group: thread -> (row lane = tid / 8, float4 column = tid % 8)
    const int c4 = tid & 7, rl = tid >> 3;
    const int col = blockIdx.x * BNM_COLS + c4 * 4;
    if (col >= h) return;                                     // partial last column block
    const float4 sc4 = *reinterpret_cast<const float4 *>(s_scale + c4 * 4), sh4 = *reinterpret_cast<const float4 *>(s_shift + c4 * 4);
    const long long r0 = (long long)blockIdx.y * rows_per_group, r1 = min(N, r0 + rows_per_group);
    for (long long r = r0 + rl; r < r1; r += BN_THREADS / 8) {
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        const size_t e = (size_t)r * h + col;
        if (r < count) {
            const float4 v = *reinterpret_cast<const float4 *>(x + e);
            o.x = bn_act<ACT>(fmaf(v.x, sc4.x, sh4.x)); o.y = bn_act<ACT>(fmaf(v.y, sc4.y, sh4.y));
            o.z = bn_act<ACT>(fmaf(v.z, sc4.z, sh4.z)); o.w = bn_act<ACT>(fmaf(v.w, sc4.w, sh4.w));
            if (dropping) {
                const float4 f = drop_factors(drop, seed, step, r, col, h);
                o.x *= f.x; o.y *= f.y; o.z *= f.z; o.w *= f.w;
            }
            if (residual) {
                const float4 rr = *reinterpret_cast<const float4 *>(residual + e);
                o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
            }
        }
        *reinterpret_cast<float4 *>(out + e) = o;
    }
}

// ---- forward, ONE launch, statistics from an ACCUMULATOR (phc_common.h): the producing GEMM added every workgroup's (sum, sum of
// squares) into 64-bit fixed-point words; a workgroup here reads the totals of its 32 columns (4 replicas x 4 words), derives
// mean / invstd in double and normalises its rows.  No merge launch, no per-workgroup partial rows.
template <int ACT>
__global__ __launch_bounds__(BN_THREADS) void bn_apply_acc_kernel(const float *__restrict__ x, BnParams p, const int *__restrict__ n_valid,
                                                                long long N, int h, float momentum, float eps,
                                                                const unsigned long long *__restrict__ acc, long long rows_per_group,
                                                                const float *__restrict__ residual, float *__restrict__ stat,
                                                                float *__restrict__ out, BnDrop drop = BnDrop{}) {
    __shared__ double r_d[BNM_LANES][BNM_COLS];
    __shared__ float s_scale[BNM_COLS], s_shift[BNM_COLS];
    static_assert(BNM_LANES == 2 * STAT_ACC_REP, "one (quantity, replica) pair per partial lane");
    const int tid = threadIdx.x;
    const int ci = tid & (BNM_COLS - 1), pl = tid / BNM_COLS;
    // the last column block may be partial (h % 4 == 0 only): its spare threads work on the last columns again and store nothing
    const bool c_ok = blockIdx.x * BNM_COLS + ci < h;
    const int c = c_ok ? blockIdx.x * BNM_COLS + ci : h - 1;
    const long long count = n_valid ? min((long long)*n_valid, N) : N;
    const bool dropping = drop.p > 0.f;
    const unsigned long long seed = dropping ? drop.state[0] : 0ull, step = dropping ? drop.state[1] : 0ull;
    // the first rounds of this thread's rows go in flight BEFORE the totals prologue (clamped addresses): its L2 miss — the totals
    // were added at the memory side a moment ago — then overlaps with them
    const int c4 = tid & 7, rl = tid >> 3;
    const bool col_ok = blockIdx.x * BNM_COLS + c4 * 4 < h;
    const int col = col_ok ? blockIdx.x * BNM_COLS + c4 * 4 : h - 4;
    const long long r0 = (long long)blockIdx.y * rows_per_group, r1 = min(N, r0 + rows_per_group);
    constexpr int PRE = 4;
    float4 xpre[PRE], rpre[PRE];
#pragma unroll
    for (int u = 0; u < PRE; ++u) {
        const long long r = min(r0 + rl + (long long)u * (BN_THREADS / 8), N - 1);
        xpre[u] = *reinterpret_cast<const float4 *>(x + (size_t)r * h + col);
        rpre[u] = residual ? *reinterpret_cast<const float4 *>(residual + (size_t)r * h + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    r_d[pl][ci] = stat_acc_part(acc, h, pl & (STAT_ACC_REP - 1), pl / STAT_ACC_REP, c);
    const int comp = c / p.k, kk = c - comp * p.k;
    float w = 1.f, bq = 0.f, rm_old = 0.f, rv_old = 1.f;
    if (pl == 0) {
        if (p.has_weight) w = *bn_col_ptr(p.weight, p.flat, comp, kk, c);
        if (p.has_bias) bq = *bn_col_ptr(p.bias, p.flat, comp, kk, c);
        if (p.has_running && blockIdx.y == 0) {
            rm_old = *bn_col_ptr(p.running_mean, p.flat, comp, kk, c);
            rv_old = *bn_col_ptr(p.running_var, p.flat, comp, kk, c);
        }
    }
    __syncthreads();
    if (pl == 0) {
        double S = 0.0, Q = 0.0;
#pragma unroll
        for (int rp = 0; rp < STAT_ACC_REP; ++rp) {
            S += r_d[rp][ci];
            Q += r_d[STAT_ACC_REP + rp][ci];
        }
        const BnFromTotals bt = bn_from_totals(S, Q, count, w, bq, eps);
        s_scale[ci] = bt.scale;
        s_shift[ci] = bt.shift;
        if (blockIdx.y == 0 && c_ok) {
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
    if (blockIdx.x == 0 && blockIdx.y == 0 && p.has_nbt && tid < p.n_comp)
        for (int q = 0; q < p.n_comp; ++q)
            if (q == tid) *p.num_batches_tracked[q] += 1;
    __syncthreads();
    const float4 sc4 = *reinterpret_cast<const float4 *>(s_scale + c4 * 4), sh4 = *reinterpret_cast<const float4 *>(s_shift + c4 * 4);
    auto row = [&](long long r, const float4 v, const float4 rr) {
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < count) {
            o.x = bn_act<ACT>(fmaf(v.x, sc4.x, sh4.x)); o.y = bn_act<ACT>(fmaf(v.y, sc4.y, sh4.y));
            o.z = bn_act<ACT>(fmaf(v.z, sc4.z, sh4.z)); o.w = bn_act<ACT>(fmaf(v.w, sc4.w, sh4.w));
            if (dropping) {
                const float4 f = drop_factors(drop, seed, step, r, col, h);
                o.x *= f.x; o.y *= f.y; o.z *= f.z; o.w *= f.w;
            }
            o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;          // (zeros without a residual)
        }
        *reinterpret_cast<float4 *>(out + (size_t)r * h + col) = o;
    };
    if (!col_ok) return;
#pragma unroll
    for (int u = 0; u < PRE; ++u) {
        const long long r = r0 + rl + (long long)u * (BN_THREADS / 8);
        if (r < r1) row(r, xpre[u], rpre[u]);
    }
    for (long long r = r0 + rl + (long long)PRE * (BN_THREADS / 8); r < r1; r += BN_THREADS / 8) {
        const size_t e = (size_t)r * h + col;
        row(r, *reinterpret_cast<const float4 *>(x + e),
            residual ? *reinterpret_cast<const float4 *>(residual + e) : make_float4(0.f, 0.f, 0.f, 0.f));
    }
}

// ---- pass 1 (backward): sum g and sum g*x_hat per column ------------------------------------------
template <int ACT, int VEC>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_partial_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                                  const float *__restrict__ stat,
                                                                  const int *__restrict__ n_valid, long long N, int h,
                                                                  long long rows_per_block, float *__restrict__ partial, BnDrop drop = BnDrop{},
                                                                  int acc_mode = 0) {
    // acc_mode: `partial` is a statistics accumulator (phc_common.h) — the sums of this row block are ADDED into it
    __shared__ float red[2 * VEC * BN_THREADS];
    const long long count = n_valid ? min((long long)*n_valid, N) : N;
    const bool dropping = VEC == 4 && drop.p > 0.f;
    const unsigned long long seed = dropping ? drop.state[0] : 0ull, step = dropping ? drop.state[1] : 0ull;
    const int tid = threadIdx.x;
    const long long r0 = blockIdx.x * rows_per_block;
    const long long r1 = min(count, r0 + rows_per_block);
    const int hv = h / VEC;
    for (int cbase = 0; cbase < hv; cbase += BN_THREADS) {
        const int cols = min(BN_THREADS, hv - cbase);
        const int lanes = BN_THREADS / cols;
        const int c = tid % cols, rl = tid / cols;
        float sg[VEC], sgx[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) sg[v] = sgx[v] = 0.f;
        if (rl < lanes) {
            const int col = (cbase + c) * VEC;
            float mean[VEC], inv[VEC], w[VEC], b[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                mean[v] = stat[col + v];
                inv[v] = stat[h + col + v];
                w[v] = stat[4 * h + col + v];
                b[v] = stat[5 * h + col + v];
            }
#pragma unroll 4
            for (long long r = r0 + rl; r < r1; r += lanes) {
                float xv[VEC], gv[VEC];
                ldv<VEC>(x + (size_t)r * h + col, xv);
                ldv<VEC>(dy + (size_t)r * h + col, gv);
                if constexpr (VEC == 4) {
                    if (dropping) {
                        const float4 f = drop_factors(drop, seed, step, r, col, h);
                        gv[0] *= f.x; gv[1] *= f.y; gv[2] *= f.z; gv[3] *= f.w;
                    }
                }
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const float xh = (xv[v] - mean[v]) * inv[v];
                    float g = gv[v];
                    if (ACT != PHC_ACT_IDENTITY) g *= act_grad<ACT>(fmaf(xh, w[v], b[v]));
                    sg[v] += g;
                    sgx[v] = fmaf(g, xh, sgx[v]);
                }
            }
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            red[v * BN_THREADS + tid] = sg[v];
            red[(VEC + v) * BN_THREADS + tid] = sgx[v];
        }
        __syncthreads();
        for (int e = tid; e < cols * VEC; e += BN_THREADS) {
            const int cc = e / VEC, v = e - cc * VEC;
            float a = 0.f, bsum = 0.f;
            for (int l = 0; l < lanes; ++l) {
                a += red[v * BN_THREADS + l * cols + cc];
                bsum += red[(VEC + v) * BN_THREADS + l * cols + cc];
            }
            if (acc_mode) {
                unsigned long long *acc64 = reinterpret_cast<unsigned long long *>(partial);
                stat_acc_add(acc64, h, (int)(blockIdx.x % STAT_ACC_REP), 0, (cbase + cc) * VEC + v, (double)a);
                stat_acc_add(acc64, h, (int)(blockIdx.x % STAT_ACC_REP), 1, (cbase + cc) * VEC + v, (double)bsum);
            } else {
                partial[(size_t)blockIdx.x * 2 * h + (cbase + cc) * VEC + v] = a;
                partial[(size_t)blockIdx.x * 2 * h + h + (cbase + cc) * VEC + v] = bsum;
            }
        }
        __syncthreads();
    }
}

// ---- pass 2 (backward): merge partial sums; dwb = [dweight | dbias]; bstat = [mean(g) | mean(g*x_hat)] ------
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_finalize_kernel(const int *__restrict__ n_valid, long long N, int h, int training,
                                                                   const float *__restrict__ partial, int num_partials,
                                                                   float *__restrict__ dwb, float *__restrict__ bstat) {
    __shared__ float r_a[32][8], r_b[32][8];
    const long long count = n_valid ? min((long long)*n_valid, N) : N;
    const int tid = threadIdx.x;
    const int ci = tid & 7, pl = tid >> 3;
    const int c = blockIdx.x * 8 + ci;
    float sg = 0.f, sgx = 0.f;
    {
        constexpr int PER_LANE = BN_MAX_PARTIALS / 32;
        float pa[PER_LANE], pb[PER_LANE];
#pragma unroll
        for (int t = 0; t < PER_LANE; ++t) {
            const int b = pl + 32 * t;
            const bool ok = c < h && b < num_partials;
            const size_t off = (size_t)(ok ? b : 0) * 2 * h + (c < h ? c : 0);
            const float a = partial[off], bb = partial[off + h];
            pa[t] = ok ? a : 0.f;
            pb[t] = ok ? bb : 0.f;
        }
#pragma unroll
        for (int t = 0; t < PER_LANE; ++t) {
            sg += pa[t];
            sgx += pb[t];
        }
    }
    r_a[pl][ci] = sg;
    r_b[pl][ci] = sgx;
    __syncthreads();
    if (pl == 0 && c < h) {
        float a = 0.f, b = 0.f;
        for (int l = 0; l < 32; ++l) {
            a += r_a[l][ci];
            b += r_b[l][ci];
        }
        if (dwb) {
            dwb[c] = b;                                      // dweight = sum g * x_hat
            dwb[h + c] = a;                                  // dbias   = sum g
        }
        const float cnt = (float)max(count, 1LL);
        bstat[c] = training ? a / cnt : 0.f;
        bstat[h + c] = training ? b / cnt : 0.f;
    }
}

// accumulator form without a dx request: only dwb = [dweight | dbias] from the totals
__global__ void bn_bwd_acc_dwb_kernel(int h, const unsigned long long *__restrict__ acc, float *__restrict__ dwb) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < h && dwb) {
        dwb[c] = (float)stat_acc_total(acc, h, 1, c);
        dwb[h + c] = (float)stat_acc_total(acc, h, 0, c);
    }
}

// ---- pass 3 (backward): dx = w*invstd*(g - mean(g) - x_hat*mean(g*x_hat)) ---------------------------
template <int ACT>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_apply_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                                const float *__restrict__ stat,
                                                                const int *__restrict__ n_valid, long long N, int h,
                                                                const float *__restrict__ bstat, float *__restrict__ dx, BnDrop drop = BnDrop{},
                                                                const unsigned long long *__restrict__ acc = nullptr,
                                                                float *__restrict__ dwb = nullptr, int training = 1) {
    // acc != NULL: the column sums (sum g, sum g * x_hat) come as accumulator totals (phc_common.h) instead of bstat; the first
    // workgroup also writes dwb = [dweight | dbias] from them (what bn_bwd_finalize_kernel does on the partial-row path)
    const bool dropping = drop.p > 0.f;
    const unsigned long long seed = dropping ? drop.state[0] : 0ull, step = dropping ? drop.state[1] : 0ull;
    extern __shared__ float sm[];                            // mean, invstd, w, b, mg, mgx : 6 * h
    float *s_mean = sm, *s_inv = sm + h, *s_w = sm + 2 * h, *s_b = sm + 3 * h, *s_mg = sm + 4 * h, *s_mgx = sm + 5 * h;
    const long long count = n_valid ? min((long long)*n_valid, N) : N;
    const int tid = threadIdx.x;
    for (int c = tid; c < h; c += BN_THREADS) {
        s_mean[c] = stat[c];
        s_inv[c] = stat[h + c];
        s_w[c] = stat[4 * h + c];
        s_b[c] = stat[5 * h + c];
        if (acc) {
            const float a = (float)stat_acc_total(acc, h, 0, c), b = (float)stat_acc_total(acc, h, 1, c);
            const float cnt = (float)max(count, 1LL);
            s_mg[c] = training ? a / cnt : 0.f;
            s_mgx[c] = training ? b / cnt : 0.f;
            if (dwb && blockIdx.x == 0) {
                dwb[c] = b;                                  // dweight = sum g * x_hat
                dwb[h + c] = a;                              // dbias   = sum g
            }
        } else {
            s_mg[c] = bstat[c];
            s_mgx[c] = bstat[h + c];
        }
    }
    __syncthreads();
    auto one = [&](float xv, float gv, int c) -> float {
        const float xh = (xv - s_mean[c]) * s_inv[c];
        float g = gv;
        if (ACT != PHC_ACT_IDENTITY) g *= act_grad<ACT>(fmaf(xh, s_w[c], s_b[c]));
        return s_w[c] * s_inv[c] * (g - s_mg[c] - xh * s_mgx[c]);
    };
    if ((h & 3) == 0) {
        const int h4 = h >> 2;
        const long long total4 = N * (long long)h4;
        for (long long e = (long long)blockIdx.x * BN_THREADS + tid; e < total4; e += (long long)gridDim.x * BN_THREADS) {
            const long long r = e / h4;
            const int c = (int)(e - r * h4) << 2;
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < count) {
                const float4 xv = reinterpret_cast<const float4 *>(x)[e];
                float4 gv = reinterpret_cast<const float4 *>(dy)[e];
                if (dropping) {
                    const float4 f = drop_factors(drop, seed, step, r, c, h);
                    gv.x *= f.x; gv.y *= f.y; gv.z *= f.z; gv.w *= f.w;
                }
                o.x = one(xv.x, gv.x, c); o.y = one(xv.y, gv.y, c + 1);
                o.z = one(xv.z, gv.z, c + 2); o.w = one(xv.w, gv.w, c + 3);
            }
            reinterpret_cast<float4 *>(dx)[e] = o;
        }
    } else {
        const long long total = N * (long long)h;
        for (long long e = (long long)blockIdx.x * BN_THREADS + tid; e < total; e += (long long)gridDim.x * BN_THREADS) {
            const long long r = e / h;
            const int c = (int)(e - r * h);
            dx[e] = r < count ? one(x[e], dy[e], c) : 0.f;
        }
    }
}


// ---- pass 3 (backward) with the column sums in an ACCUMULATOR: a workgroup owns 32 COLUMNS x a row range (the layout of
// bn_apply_acc_kernel), reads only those columns' totals (4 KB instead of every column's 16 KB) — and issues the loads of its first
// rows BEFORE that prologue, whose L2 miss (the totals were added at the memory side a moment ago) then costs nothing extra.
constexpr int BNB_PRE = 4;                                   // row rounds whose (dy, x) loads are in flight during the prologue
template <int ACT>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_apply_acc_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                                    const float *__restrict__ stat, const int *__restrict__ n_valid,
                                                                    long long N, int h, const unsigned long long *__restrict__ acc,
                                                                    long long rows_per_group, float *__restrict__ dx,
                                                                    float *__restrict__ dwb, int training, BnDrop drop = BnDrop{}) {
    __shared__ double r_d[BNM_LANES][BNM_COLS];
    __shared__ float s_c[6][BNM_COLS];                       // mean, invstd, w, b, mean g, mean g x_hat
    static_assert(BNM_LANES == 2 * STAT_ACC_REP, "one (quantity, replica) pair per partial lane");
    const int tid = threadIdx.x;
    const int ci = tid & (BNM_COLS - 1), pl = tid / BNM_COLS;
    // the last column block may be partial (h % 4 == 0 only): its spare threads work on the last columns again and store nothing
    const bool c_ok = blockIdx.x * BNM_COLS + ci < h;
    const int c = c_ok ? blockIdx.x * BNM_COLS + ci : h - 1;
    const long long count = n_valid ? min((long long)*n_valid, N) : N;
    const bool dropping = drop.p > 0.f;
    const unsigned long long seed = dropping ? drop.state[0] : 0ull, step = dropping ? drop.state[1] : 0ull;
    const int c4 = tid & 7, rl = tid >> 3;
    const bool col_ok = blockIdx.x * BNM_COLS + c4 * 4 < h;
    const int col = col_ok ? blockIdx.x * BNM_COLS + c4 * 4 : h - 4;
    const long long r0 = (long long)blockIdx.y * rows_per_group, r1 = min(N, r0 + rows_per_group);
    // first rounds of this thread's rows go in flight (clamped addresses; rows past the range are never stored)
    float4 gpre[BNB_PRE], xpre[BNB_PRE];
#pragma unroll
    for (int u = 0; u < BNB_PRE; ++u) {
        const long long r = min(r0 + rl + (long long)u * (BN_THREADS / 8), N - 1);
        gpre[u] = *reinterpret_cast<const float4 *>(dy + (size_t)r * h + col);
        xpre[u] = *reinterpret_cast<const float4 *>(x + (size_t)r * h + col);
    }
    r_d[pl][ci] = stat_acc_part(acc, h, pl & (STAT_ACC_REP - 1), pl / STAT_ACC_REP, c);
    __syncthreads();
    if (pl == 0) {
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int rp = 0; rp < STAT_ACC_REP; ++rp) {
            a += r_d[rp][ci];
            b += r_d[STAT_ACC_REP + rp][ci];
        }
        const float af = (float)a, bf = (float)b;
        const float cnt = (float)max(count, 1LL);
        s_c[0][ci] = stat[c];
        s_c[1][ci] = stat[h + c];
        s_c[2][ci] = stat[4 * h + c];
        s_c[3][ci] = stat[5 * h + c];
        s_c[4][ci] = training ? af / cnt : 0.f;
        s_c[5][ci] = training ? bf / cnt : 0.f;
        if (dwb && blockIdx.y == 0 && c_ok) {
            dwb[c] = bf;                                     // dweight = sum g * x_hat
            dwb[h + c] = af;                                 // dbias   = sum g
        }
    }
    __syncthreads();
    if (!col_ok) return;
    float mean[4], inv[4], w[4], bb[4], mg[4], mgx[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        mean[q] = s_c[0][c4 * 4 + q]; inv[q] = s_c[1][c4 * 4 + q]; w[q] = s_c[2][c4 * 4 + q];
        bb[q] = s_c[3][c4 * 4 + q]; mg[q] = s_c[4][c4 * 4 + q]; mgx[q] = s_c[5][c4 * 4 + q];
    }
    auto one = [&](float xv, float gv, int q) -> float {
        const float xh = (xv - mean[q]) * inv[q];
        float g = gv;
        if (ACT != PHC_ACT_IDENTITY) g *= act_grad<ACT>(fmaf(xh, w[q], bb[q]));
        return w[q] * inv[q] * (g - mg[q] - xh * mgx[q]);
    };
    auto row = [&](long long r, float4 gv, float4 xv) {
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < count) {
            if (dropping) {
                const float4 f = drop_factors(drop, seed, step, r, col, h);
                gv.x *= f.x; gv.y *= f.y; gv.z *= f.z; gv.w *= f.w;
            }
            o.x = one(xv.x, gv.x, 0); o.y = one(xv.y, gv.y, 1); o.z = one(xv.z, gv.z, 2); o.w = one(xv.w, gv.w, 3);
        }
        *reinterpret_cast<float4 *>(dx + (size_t)r * h + col) = o;
    };
#pragma unroll
    for (int u = 0; u < BNB_PRE; ++u) {
        const long long r = r0 + rl + (long long)u * (BN_THREADS / 8);
        if (r < r1) row(r, gpre[u], xpre[u]);
    }
    for (long long r = r0 + rl + (long long)BNB_PRE * (BN_THREADS / 8); r < r1; r += BN_THREADS / 8)
        row(r, *reinterpret_cast<const float4 *>(dy + (size_t)r * h + col), *reinterpret_cast<const float4 *>(x + (size_t)r * h + col));
}

// ---- row-sharded batch-norm (one logical batch split by rows over several ranks) ----------
// Merge this rank's per-block (sum, M2) pairs into ONE pair for all its rows (Chan's formula, fixed order): the 2h floats
// that travel in the all-gather.  Same 8-columns x 32-partial-lanes layout as bn_finalize_kernel.
__global__ __launch_bounds__(BN_THREADS) void bn_merge_moments_kernel(long long N, int h, const float *__restrict__ partial,
                                                                    int num_partials, long long rows_per_block,
                                                                    float *__restrict__ moments) {
    __shared__ float r_sum[32][8], r_m2[32][8];
    const int tid = threadIdx.x;
    const int ci = tid & 7, pl = tid >> 3;
    const int c = blockIdx.x * 8 + ci;
    constexpr int PER_LANE = BN_MAX_PARTIALS / 32;
    float ps[PER_LANE], pm[PER_LANE];
#pragma unroll
    for (int t = 0; t < PER_LANE; ++t) {
        const int bb = pl + 32 * t;
        const bool ok = c < h && bb < num_partials;
        const size_t off = (size_t)(ok ? bb : 0) * 2 * h + (c < h ? c : 0);
        const float a = partial[off], m2v = partial[off + h];
        ps[t] = ok ? a : 0.f;
        pm[t] = ok ? m2v : 0.f;
    }
    float sloc = 0.f;
#pragma unroll
    for (int t = 0; t < PER_LANE; ++t) sloc += ps[t];
    r_sum[pl][ci] = sloc;
    __syncthreads();
    float tot = 0.f;
    for (int l = 0; l < 32; ++l) tot += r_sum[l][ci];
    const float mean = tot / (float)max(N, 1LL);
    float m2 = 0.f;
#pragma unroll
    for (int t = 0; t < PER_LANE; ++t) {
        const int bb = pl + 32 * t;
        const float nbf = (float)min(rows_per_block, max(N - (long long)bb * rows_per_block, 0LL));
        if (c < h && bb < num_partials && nbf > 0.f) {
            const float d = ps[t] / nbf - mean;
            m2 += pm[t] + nbf * d * d;
        }
    }
    r_m2[pl][ci] = m2;
    __syncthreads();
    if (pl == 0 && c < h) {
        float M2 = 0.f;
        for (int l = 0; l < 32; ++l) M2 += r_m2[l][ci];
        moments[c] = tot;
        moments[h + c] = M2;
    }
}

// bstat = [sum g | sum g*x_hat] / N_total from the all-reduced dwb = [sum g*x_hat | sum g]
__global__ void bn_bstat_from_dwb_kernel(const float *__restrict__ dwb, int h, float inv_count, float *__restrict__ bstat) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < h) {
        bstat[c] = dwb[h + c] * inv_count;
        bstat[h + c] = dwb[c] * inv_count;
    }
}

// ---- host ---------------------------------------------------------------------------------------------
struct BnPlan {
    int num_partials;
    long long rows_per_block;
};

static BnPlan bn_plan(long long N) {
    BnPlan p;
    long long parts = (N + 63) / 64;                        // >= 64 rows per workgroup
    if (parts > BN_MAX_PARTIALS) parts = BN_MAX_PARTIALS;
    if (parts < 1) parts = 1;
    p.rows_per_block = (N + parts - 1) / parts;
    if (p.rows_per_block < 1) p.rows_per_block = 1;
    p.num_partials = (int)((N + p.rows_per_block - 1) / p.rows_per_block);
    if (p.num_partials < 1) p.num_partials = 1;
    return p;
}

static inline unsigned apply_grid(long long N, int h) {
    long long blocks = (N * (long long)h / 4 + BN_THREADS * 4 - 1) / (BN_THREADS * 4);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

__global__ void rng_advance_kernel(unsigned long long *state) { state[1] += 1ull; }

#define PHC_BN_ACT_SWITCH(act, MACRO)                 \
    switch (act) {                                    \
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

size_t phc_bn_ws_bytes(int64_t N, int h) {
    if (N < 0 || h < 1) return 0;
    return sizeof(float) * (size_t)h * (2 * BN_MAX_PARTIALS + 2);    // partials + backward [mean(g) | mean(g*x_hat)]
}

int phc_bn_fwd(const float *x, const float *const *weight, const float *const *bias, float *const *running_mean,
               float *const *running_var, int64_t *const *num_batches_tracked, int n_comp, const int32_t *n_valid_dev, int64_t N,
               int h, int training, float momentum, float eps, int act, const float *residual, float *out, float *save_stat,
               void *ws, size_t ws_bytes, float drop_p, int drop_same_k, int drop_site, const uint64_t *rng_state, void *stream) {
    BnParams p;
    BnDrop drop;
    int rc = make_drop("phc_bn_fwd", &drop, training ? drop_p : 0.f, drop_same_k, drop_site, rng_state, h);
    if (rc) return rc;
    rc = fill_params("phc_bn_fwd", &p, weight, bias, running_mean, running_var, num_batches_tracked, n_comp, h);
    if (rc) return rc;
    PHC_REQUIRE(N >= 0, PHC_ERR_INVALID_ARG, "phc_bn_fwd: N < 0");
    PHC_REQUIRE(act >= PHC_ACT_IDENTITY && act <= PHC_ACT_SELU, PHC_ERR_INVALID_ARG, "phc_bn_fwd: unknown activation %d", act);
    PHC_REQUIRE(training || p.has_running, PHC_ERR_INVALID_ARG, "phc_bn_fwd: eval mode needs running statistics");
    if (N == 0) return PHC_OK;
    PHC_REQUIRE(x && out && save_stat, PHC_ERR_INVALID_ARG, "phc_bn_fwd: null pointer");
    PHC_REQUIRE(h <= 8192, PHC_ERR_UNSUPPORTED, "phc_bn_fwd: h=%d too wide", h);
    hipStream_t s = (hipStream_t)stream;
    BnPlan plan = bn_plan(N);
    float *partial = static_cast<float *>(ws);
    const bool vec4 = (h % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0) &&
                      (!residual || (reinterpret_cast<uintptr_t>(residual) & 15) == 0);
    PHC_REQUIRE(vec4 || h % 4 != 0, PHC_ERR_UNSUPPORTED, "phc_bn_fwd: rows must be 16-byte aligned when h %% 4 == 0");
    PHC_REQUIRE(ws && ws_bytes >= phc_bn_ws_bytes(N, h), PHC_ERR_WORKSPACE, "phc_bn_fwd: workspace too small");
    float *stat = save_stat;
    if (training) {
        ProfScope prof(K_BN_STATS, s);
        if (vec4)
            PHC_LAUNCH((bn_stats_partial_kernel<4>), dim3(plan.num_partials), dim3(BN_THREADS), 0, s, x, n_valid_dev,
                               (long long)N, h, plan.rows_per_block, partial);
        else
            PHC_LAUNCH((bn_stats_partial_kernel<1>), dim3(plan.num_partials), dim3(BN_THREADS), 0, s, x, n_valid_dev,
                               (long long)N, h, plan.rows_per_block, partial);
        rc = check_launch("bn_stats_partial");
        if (rc) return rc;
    }
    {
        ProfScope prof(K_BN_FINALIZE, s);
        PHC_LAUNCH(bn_finalize_kernel, dim3((h + 7) / 8), dim3(BN_THREADS), 0, s, p, n_valid_dev, (long long)N, h, training,
                           momentum, eps, partial, plan.num_partials, plan.rows_per_block, stat);
        rc = check_launch("bn_finalize");
        if (rc) return rc;
    }
    const size_t lds = sizeof(float) * 2 * (size_t)h;
    ProfScope prof(K_BN_APPLY, s);
#define PHC_BN_FWD(ACTC)                                                                                                      \
    PHC_LAUNCH((bn_apply_kernel<ACTC>), dim3(apply_grid(N, h)), dim3(BN_THREADS), lds, s, x, n_valid_dev, (long long)N, h,   \
                       residual, stat, out, drop)
    PHC_BN_ACT_SWITCH(act, PHC_BN_FWD)
#undef PHC_BN_FWD
    return check_launch("bn_apply");
}

int phc_bn_finalize_from_partials(const float *stats_partial, const float *stats_counts, int num_partials, const float *const *weight,
                                  const float *const *bias, float *const *running_mean, float *const *running_var,
                                  int64_t *const *num_batches_tracked, int n_comp, const int32_t *n_valid_dev, int64_t N, int h,
                                  float momentum, float eps, float *save_stat, void *stream) {
    BnParams p;
    int rc = fill_params("phc_bn_finalize_from_partials", &p, weight, bias, running_mean, running_var, num_batches_tracked, n_comp, h);
    if (rc) return rc;
    PHC_REQUIRE(N >= 1 && num_partials >= 1 && num_partials <= BN_MAX_PARTIALS && h <= 8192, PHC_ERR_INVALID_ARG,
                "phc_bn_finalize_from_partials: N=%lld, %d partials (max %d), h=%d", (long long)N, num_partials, BN_MAX_PARTIALS, h);
    PHC_REQUIRE(save_stat && stats_partial && stats_counts, PHC_ERR_INVALID_ARG, "phc_bn_finalize_from_partials: null pointer");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_BN_FINALIZE, s);
    PHC_LAUNCH(bn_finalize_kernel, dim3((h + 7) / 8), dim3(BN_THREADS), 0, s, p, n_valid_dev, (long long)N, h, 1, momentum, eps,
               stats_partial, num_partials, 0LL, save_stat, stats_counts);
    return check_launch("bn_finalize(from partials)");
}

int phc_bn_fwd_from_partials(const float *x, const float *stats_partial, const float *stats_counts, int num_partials,
                             const float *const *weight, const float *const *bias, float *const *running_mean,
                             float *const *running_var, int64_t *const *num_batches_tracked, int n_comp, const int32_t *n_valid_dev,
                             int64_t N, int h, float momentum, float eps, int act, const float *residual, float *out,
                             float *save_stat, float drop_p, int drop_same_k, int drop_site, const uint64_t *rng_state, void *stream) {
    BnParams p;
    BnDrop drop;
    int rc = make_drop("phc_bn_fwd_from_partials", &drop, drop_p, drop_same_k, drop_site, rng_state, h);
    if (rc) return rc;
    rc = fill_params("phc_bn_fwd_from_partials", &p, weight, bias, running_mean, running_var, num_batches_tracked, n_comp, h);
    if (rc) return rc;
    const bool acc_mode = stats_counts == nullptr;           // accumulator form: stats_partial holds totals, nothing to merge
    PHC_REQUIRE(N >= 1 && (acc_mode || (num_partials >= 1 && num_partials <= BN_MAX_PARTIALS)), PHC_ERR_INVALID_ARG,
                "phc_bn_fwd_from_partials: N=%lld, %d partials (max %d)", (long long)N, num_partials, BN_MAX_PARTIALS);
    PHC_REQUIRE(act >= PHC_ACT_IDENTITY && act <= PHC_ACT_SELU, PHC_ERR_INVALID_ARG, "phc_bn_fwd_from_partials: unknown activation %d", act);
    PHC_REQUIRE(x && out && save_stat && stats_partial, PHC_ERR_INVALID_ARG, "phc_bn_fwd_from_partials: null pointer");
    PHC_REQUIRE(h <= 8192, PHC_ERR_UNSUPPORTED, "phc_bn_fwd_from_partials: h=%d too wide", h);
    const bool vec4 = (h % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0) &&
                      (!residual || (reinterpret_cast<uintptr_t>(residual) & 15) == 0);
    PHC_REQUIRE(vec4 || h % 4 != 0, PHC_ERR_UNSUPPORTED, "phc_bn_fwd_from_partials: rows must be 16-byte aligned when h %% 4 == 0");
    hipStream_t s = (hipStream_t)stream;
    if (acc_mode) {
        PHC_REQUIRE(vec4 && (reinterpret_cast<uintptr_t>(stats_partial) & 7) == 0, PHC_ERR_UNSUPPORTED,
                    "phc_bn_fwd_from_partials: the accumulator form needs h %% 4 == 0 and aligned rows (h=%d)", h);
        long long groups = (N + 127) / 128;
        const long long cap = (8LL * 256) / bn_col_blocks(h);
        if (groups > cap) groups = cap;
        if (groups < 1) groups = 1;
        const long long rows_per_group = (N + groups - 1) / groups;
        groups = (N + rows_per_group - 1) / rows_per_group;
        const unsigned long long *acc = reinterpret_cast<const unsigned long long *>(stats_partial);
        ProfScope prof(K_BN_APPLY, s);
#define PHC_BN_FWDA(ACTC)                                                                                                      \
    PHC_LAUNCH((bn_apply_acc_kernel<ACTC>), dim3(bn_col_blocks(h), (unsigned)groups), dim3(BN_THREADS), 0, s, x, p, n_valid_dev,     \
               (long long)N, h, momentum, eps, acc, rows_per_group, residual, save_stat, out, drop)
        PHC_BN_ACT_SWITCH(act, PHC_BN_FWDA)
#undef PHC_BN_FWDA
        return check_launch("bn_apply_acc");
    }
    if (vec4) {
        // one launch: the merge of the partials rides in the prologue of the normalisation
        long long groups = (N + 127) / 128;                      // >= 128 rows per workgroup
        const long long cap = (8LL * 256) / bn_col_blocks(h);    // ~8 workgroups per CU at most
        if (groups > cap) groups = cap;
        if (groups < 1) groups = 1;
        const long long rows_per_group = (N + groups - 1) / groups;
        groups = (N + rows_per_group - 1) / rows_per_group;
        ProfScope prof(K_BN_APPLY, s);
#define PHC_BN_FWDM2(ACTC)                                                                                                     \
    if (num_partials <= 16 * BNM_LANES)                                                                                        \
        PHC_LAUNCH((bn_apply_merged_kernel<ACTC, 16>), dim3(bn_col_blocks(h), (unsigned)groups), dim3(BN_THREADS), 0, s, x, p, n_valid_dev, \
                   (long long)N, h, momentum, eps, stats_partial, stats_counts, num_partials, rows_per_group, residual, save_stat, out, \
                   drop);                                                                                                       \
    else                                                                                                                       \
        PHC_LAUNCH((bn_apply_merged_kernel<ACTC, 32>), dim3(bn_col_blocks(h), (unsigned)groups), dim3(BN_THREADS), 0, s, x, p, n_valid_dev, \
                   (long long)N, h, momentum, eps, stats_partial, stats_counts, num_partials, rows_per_group, residual, save_stat, out, \
                   drop)
        PHC_BN_ACT_SWITCH(act, PHC_BN_FWDM2)
#undef PHC_BN_FWDM2
        return check_launch("bn_apply_merged");
    }
    {
        ProfScope prof(K_BN_FINALIZE, s);
        PHC_LAUNCH(bn_finalize_kernel, dim3((h + 7) / 8), dim3(BN_THREADS), 0, s, p, n_valid_dev, (long long)N, h, 1, momentum, eps,
                   stats_partial, num_partials, 0LL, save_stat, stats_counts);
        rc = check_launch("bn_finalize(from partials)");
        if (rc) return rc;
    }
    const size_t lds = sizeof(float) * 2 * (size_t)h;
    ProfScope prof(K_BN_APPLY, s);
#define PHC_BN_FWDP(ACTC)                                                                                                     \
    PHC_LAUNCH((bn_apply_kernel<ACTC>), dim3(apply_grid(N, h)), dim3(BN_THREADS), lds, s, x, n_valid_dev, (long long)N, h, residual,  \
               save_stat, out, drop)
    PHC_BN_ACT_SWITCH(act, PHC_BN_FWDP)
#undef PHC_BN_FWDP
    return check_launch("bn_apply");
}

static int launch_bwd_apply_acc(const float *dy, const float *x, const float *save_stat, const int32_t *n_valid_dev, int64_t N, int h,
                                int act, const unsigned long long *acc, float *dx, float *dwb_flat, int training, const BnDrop &drop,
                                hipStream_t s) {
    long long groups = (N + 127) / 128;
    const long long cap = (8LL * 256) / bn_col_blocks(h);
    if (groups > cap) groups = cap;
    if (groups < 1) groups = 1;
    const long long rows_per_group = (N + groups - 1) / groups;
    groups = (N + rows_per_group - 1) / rows_per_group;
    ProfScope prof(K_BN_BWD_APPLY, s);
#define PHC_BN_BAA(ACTC)                                                                                                     \
    PHC_LAUNCH((bn_bwd_apply_acc_kernel<ACTC>), dim3(bn_col_blocks(h), (unsigned)groups), dim3(BN_THREADS), 0, s, dy, x, save_stat,    \
               n_valid_dev, (long long)N, h, acc, rows_per_group, dx, dwb_flat, training, drop)
    PHC_BN_ACT_SWITCH(act, PHC_BN_BAA)
#undef PHC_BN_BAA
    return check_launch("bn_bwd_apply_acc");
}

int phc_bn_bwd_partial_acc(const float *dy, const float *x, const float *save_stat, const int32_t *n_valid_dev, int64_t N, int h,
                           int act, uint64_t *acc, void *stream) {
    return phc_bn_bwd_partial_acc_drop(dy, x, save_stat, n_valid_dev, N, h, act, acc, 0.f, 0, 0, nullptr, stream);
}

int phc_bn_bwd_partial_acc_drop(const float *dy, const float *x, const float *save_stat, const int32_t *n_valid_dev, int64_t N, int h,
                                int act, uint64_t *acc, float drop_p, int drop_same_k, int drop_site, const uint64_t *rng_state,
                                void *stream) {
    PHC_REQUIRE(N >= 1 && h >= 4 && h % 4 == 0 && h <= 8192, PHC_ERR_INVALID_ARG, "phc_bn_bwd_partial_acc: N=%lld h=%d", (long long)N, h);
    PHC_REQUIRE(act >= PHC_ACT_IDENTITY && act <= PHC_ACT_SELU, PHC_ERR_INVALID_ARG, "phc_bn_bwd_partial_acc: unknown activation %d", act);
    PHC_REQUIRE(dy && x && save_stat && acc, PHC_ERR_INVALID_ARG, "phc_bn_bwd_partial_acc: null pointer");
    PHC_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0 && (reinterpret_cast<uintptr_t>(acc) & 7) == 0,
                PHC_ERR_UNSUPPORTED, "phc_bn_bwd_partial_acc: 16-byte aligned rows, 8-byte aligned accumulator");
    hipStream_t s = (hipStream_t)stream;
    BnPlan plan = bn_plan(N);
    BnDrop drop;
    int rc = make_drop("phc_bn_bwd_partial_acc", &drop, drop_p, drop_same_k, drop_site, rng_state, h);
    if (rc) return rc;
    ProfScope prof(K_BN_BWD_PARTIAL, s);
#define PHC_BN_BPA(ACTC)                                                                                                     \
    PHC_LAUNCH((bn_bwd_partial_kernel<ACTC, 4>), dim3(plan.num_partials), dim3(BN_THREADS), 0, s, dy, x, save_stat, n_valid_dev,     \
               (long long)N, h, plan.rows_per_block, reinterpret_cast<float *>(acc), drop, 1)
    PHC_BN_ACT_SWITCH(act, PHC_BN_BPA)
#undef PHC_BN_BPA
    return check_launch("bn_bwd_partial_acc");
}

int phc_bn_bwd(const float *dy, const float *x, const float *save_stat, const int32_t *n_valid_dev, int64_t N, int h, int training,
               int act, float *dx, float *dwb_flat, void *ws, size_t ws_bytes, float drop_p, int drop_same_k, int drop_site,
               const uint64_t *rng_state, void *stream) {
    BnDrop drop;
    int rc = make_drop("phc_bn_bwd", &drop, training ? drop_p : 0.f, drop_same_k, drop_site, rng_state, h);
    if (rc) return rc;
    PHC_REQUIRE(h >= 1, PHC_ERR_INVALID_ARG, "phc_bn_bwd: h < 1");
    PHC_REQUIRE(N >= 0, PHC_ERR_INVALID_ARG, "phc_bn_bwd: N < 0");
    PHC_REQUIRE(act >= PHC_ACT_IDENTITY && act <= PHC_ACT_SELU, PHC_ERR_INVALID_ARG, "phc_bn_bwd: unknown activation %d", act);
    PHC_REQUIRE(h <= 8192, PHC_ERR_UNSUPPORTED, "phc_bn_bwd: h=%d too wide", h);
    hipStream_t s = (hipStream_t)stream;
    if (N == 0) {
        if (dwb_flat) hipMemsetAsync(dwb_flat, 0, sizeof(float) * 2 * (size_t)h, s);
        return PHC_OK;
    }
    PHC_REQUIRE(dy && x && save_stat, PHC_ERR_INVALID_ARG, "phc_bn_bwd: null pointer");
    const bool acc_mode = ws && ws_bytes == 0;               // accumulator form: ws is a zeroed statistics accumulator
    PHC_REQUIRE(ws && (acc_mode || ws_bytes >= phc_bn_ws_bytes(N, h)), PHC_ERR_WORKSPACE, "phc_bn_bwd: workspace too small");
    PHC_REQUIRE(!acc_mode || (reinterpret_cast<uintptr_t>(ws) & 7) == 0, PHC_ERR_INVALID_ARG, "phc_bn_bwd: the accumulator must be 8-byte aligned");
    BnPlan plan = bn_plan(N);
    float *partial = static_cast<float *>(ws);
    const bool vec4 = (h % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && ((reinterpret_cast<uintptr_t>(dy) & 15) == 0) &&
                      (!dx || (reinterpret_cast<uintptr_t>(dx) & 15) == 0);
    PHC_REQUIRE(vec4 || h % 4 != 0, PHC_ERR_UNSUPPORTED, "phc_bn_bwd: rows must be 16-byte aligned when h %% 4 == 0");
    float *bstat = partial + 2 * (size_t)h * BN_MAX_PARTIALS;
    {
        ProfScope prof(K_BN_BWD_PARTIAL, s);
#define PHC_BN_BP(ACTC)                                                                                                      \
    if (vec4)                                                                                                                \
        PHC_LAUNCH((bn_bwd_partial_kernel<ACTC, 4>), dim3(plan.num_partials), dim3(BN_THREADS), 0, s, dy, x, save_stat,    \
                           n_valid_dev, (long long)N, h, plan.rows_per_block, partial, drop, acc_mode ? 1 : 0);              \
    else                                                                                                                     \
        PHC_LAUNCH((bn_bwd_partial_kernel<ACTC, 1>), dim3(plan.num_partials), dim3(BN_THREADS), 0, s, dy, x, save_stat,    \
                           n_valid_dev, (long long)N, h, plan.rows_per_block, partial, drop, acc_mode ? 1 : 0)
        PHC_BN_ACT_SWITCH(act, PHC_BN_BP)
#undef PHC_BN_BP
        rc = check_launch("bn_bwd_partial");
        if (rc) return rc;
    }
    const unsigned long long *acc = acc_mode ? static_cast<const unsigned long long *>(ws) : nullptr;
    if (!acc_mode || !dx) {                                  // (accumulator form without dx: the finaliser still owes dwb)
        ProfScope prof(K_BN_BWD_FINALIZE, s);
        if (acc_mode)
            PHC_LAUNCH(bn_bwd_acc_dwb_kernel, dim3((h + BN_THREADS - 1) / BN_THREADS), dim3(BN_THREADS), 0, s, h, acc, dwb_flat);
        else
            PHC_LAUNCH(bn_bwd_finalize_kernel, dim3((h + 7) / 8), dim3(BN_THREADS), 0, s, n_valid_dev, (long long)N, h, training,
                       partial, plan.num_partials, dwb_flat, bstat);
        rc = check_launch("bn_bwd_finalize");
        if (rc) return rc;
    }
    if (!dx) return PHC_OK;
    if (acc_mode && vec4) return launch_bwd_apply_acc(dy, x, save_stat, n_valid_dev, N, h, act, acc, dx, dwb_flat, training, drop, s);
    const size_t lds = sizeof(float) * 6 * (size_t)h;
    ProfScope prof(K_BN_BWD_APPLY, s);
#define PHC_BN_BA(ACTC)                                                                                                      \
    PHC_LAUNCH((bn_bwd_apply_kernel<ACTC>), dim3(apply_grid(N, h)), dim3(BN_THREADS), lds, s, dy, x, save_stat,             \
                       n_valid_dev, (long long)N, h, bstat, dx, drop, acc, dwb_flat, training)
    PHC_BN_ACT_SWITCH(act, PHC_BN_BA)
#undef PHC_BN_BA
    return check_launch("bn_bwd_apply");
}

int phc_bn_bwd_from_partials(const float *dy, const float *x, const float *save_stat, const float *sums_partial, int num_partials,
                             const int32_t *n_valid_dev, int64_t N, int h, int act, float *dx, float *dwb_flat, void *ws,
                             size_t ws_bytes, void *stream) {
    const bool acc_mode = num_partials == 0;                 // accumulator form: sums_partial holds totals (no workspace needed)
    PHC_REQUIRE(N >= 1 && h >= 1 && h <= 8192 && num_partials >= 0 && num_partials <= BN_MAX_PARTIALS, PHC_ERR_INVALID_ARG,
                "phc_bn_bwd_from_partials: N=%lld h=%d partials=%d", (long long)N, h, num_partials);
    PHC_REQUIRE(act >= PHC_ACT_IDENTITY && act <= PHC_ACT_SELU, PHC_ERR_INVALID_ARG, "phc_bn_bwd_from_partials: unknown activation %d", act);
    PHC_REQUIRE(dy && x && save_stat && sums_partial, PHC_ERR_INVALID_ARG, "phc_bn_bwd_from_partials: null pointer");
    PHC_REQUIRE(acc_mode || (ws && ws_bytes >= sizeof(float) * 2 * (size_t)h), PHC_ERR_WORKSPACE,
                "phc_bn_bwd_from_partials: workspace too small");
    PHC_REQUIRE(!acc_mode || (reinterpret_cast<uintptr_t>(sums_partial) & 7) == 0, PHC_ERR_INVALID_ARG,
                "phc_bn_bwd_from_partials: the accumulator must be 8-byte aligned");
    const bool vec4 = (h % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && ((reinterpret_cast<uintptr_t>(dy) & 15) == 0) &&
                      (!dx || (reinterpret_cast<uintptr_t>(dx) & 15) == 0);
    PHC_REQUIRE(vec4 || h % 4 != 0, PHC_ERR_UNSUPPORTED, "phc_bn_bwd_from_partials: rows must be 16-byte aligned when h %% 4 == 0");
    hipStream_t s = (hipStream_t)stream;
    float *bstat = static_cast<float *>(ws);
    const unsigned long long *acc = acc_mode ? reinterpret_cast<const unsigned long long *>(sums_partial) : nullptr;
    int rc;
    if (!acc_mode || !dx) {
        ProfScope prof(K_BN_BWD_FINALIZE, s);
        if (acc_mode)
            PHC_LAUNCH(bn_bwd_acc_dwb_kernel, dim3((h + BN_THREADS - 1) / BN_THREADS), dim3(BN_THREADS), 0, s, h, acc, dwb_flat);
        else
            PHC_LAUNCH(bn_bwd_finalize_kernel, dim3((h + 7) / 8), dim3(BN_THREADS), 0, s, n_valid_dev, (long long)N, h, 1, sums_partial,
                       num_partials, dwb_flat, bstat);
        rc = check_launch("bn_bwd_finalize(from partials)");
        if (rc) return rc;
    }
    if (!dx) return PHC_OK;
    if (acc_mode && vec4)
        return launch_bwd_apply_acc(dy, x, save_stat, n_valid_dev, N, h, act, acc, dx, dwb_flat, 1, BnDrop{0.f, 0, h, 0, nullptr}, s);
    const size_t lds = sizeof(float) * 6 * (size_t)h;
    ProfScope prof(K_BN_BWD_APPLY, s);
#define PHC_BN_BAP(ACTC)                                                                                                     \
    PHC_LAUNCH((bn_bwd_apply_kernel<ACTC>), dim3(apply_grid(N, h)), dim3(BN_THREADS), lds, s, dy, x, save_stat, n_valid_dev,  \
               (long long)N, h, bstat, dx, BnDrop{}, acc, dwb_flat, 1)
    PHC_BN_ACT_SWITCH(act, PHC_BN_BAP)
#undef PHC_BN_BAP
    return check_launch("bn_bwd_apply");
}

int phc_rng_advance(uint64_t *rng_state, void *stream) {
    PHC_REQUIRE(rng_state, PHC_ERR_INVALID_ARG, "phc_rng_advance: null state");
    hipStream_t s = (hipStream_t)stream;
    PHC_LAUNCH(rng_advance_kernel, dim3(1), dim3(1), 0, s, reinterpret_cast<unsigned long long *>(rng_state));
    return check_launch("rng_advance");
}

// ---- row-sharded batch-norm: the four phases between which the host places its two collectives -------------------
int phc_bn_moments(const float *x, int64_t N, int h, float *moments, void *ws, size_t ws_bytes, void *stream) {
    PHC_REQUIRE(N >= 0 && h >= 1 && h <= 8192, PHC_ERR_INVALID_ARG, "phc_bn_moments: N=%lld h=%d", (long long)N, h);
    PHC_REQUIRE(moments, PHC_ERR_INVALID_ARG, "phc_bn_moments: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (N == 0) {
        hipMemsetAsync(moments, 0, sizeof(float) * 2 * (size_t)h, s);
        return PHC_OK;
    }
    PHC_REQUIRE(x, PHC_ERR_INVALID_ARG, "phc_bn_moments: null pointer");
    PHC_REQUIRE(ws && ws_bytes >= phc_bn_ws_bytes(N, h), PHC_ERR_WORKSPACE, "phc_bn_moments: workspace too small");
    const bool vec4 = (h % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    PHC_REQUIRE(vec4 || h % 4 != 0, PHC_ERR_UNSUPPORTED, "phc_bn_moments: rows must be 16-byte aligned when h %% 4 == 0");
    BnPlan plan = bn_plan(N);
    float *partial = static_cast<float *>(ws);
    {
        ProfScope prof(K_BN_STATS, s);
        if (vec4)
            PHC_LAUNCH((bn_stats_partial_kernel<4>), dim3(plan.num_partials), dim3(BN_THREADS), 0, s, x, (const int *)nullptr,
                       (long long)N, h, plan.rows_per_block, partial);
        else
            PHC_LAUNCH((bn_stats_partial_kernel<1>), dim3(plan.num_partials), dim3(BN_THREADS), 0, s, x, (const int *)nullptr,
                       (long long)N, h, plan.rows_per_block, partial);
    }
    ProfScope prof(K_BN_FINALIZE, s);
    PHC_LAUNCH(bn_merge_moments_kernel, dim3((h + 7) / 8), dim3(BN_THREADS), 0, s, (long long)N, h, partial, plan.num_partials,
               plan.rows_per_block, moments);
    return check_launch("bn_moments");
}

// A statistics accumulator read back as floats: the bridge from the GEMM epilogues (integer totals) to the row-sharded norm's
// host-side collectives.  count == 0: out = total of quantity 0 | total of quantity 1.  count > 0: the accumulator holds a
// statistics epilogue's (sum y | sum y^2 about zero) over `count` rows and out = sum | M2 about their own mean, what
// phc_bn_moments produces with its two passes over y (the subtraction in double: the totals are exact to 2^-62).
__global__ __launch_bounds__(256) void stat_acc_read_kernel(const unsigned long long *__restrict__ acc, int h, long long count,
                                                            float *__restrict__ out) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= h) return;
    const double S = stat_acc_total(acc, h, 0, col), Q = stat_acc_total(acc, h, 1, col);
    out[col] = (float)S;
    out[h + col] = count > 0 ? (float)fmax(Q - S * S / (double)count, 0.0) : (float)Q;
}

int phc_stat_acc_read(const uint64_t *acc, int h, int64_t count, float *out, void *stream) {
    PHC_REQUIRE(acc && out && h >= 1 && h <= 8192 && count >= 0, PHC_ERR_INVALID_ARG, "phc_stat_acc_read: h=%d count=%lld", h, (long long)count);
    PHC_REQUIRE((reinterpret_cast<uintptr_t>(acc) & 7) == 0, PHC_ERR_UNSUPPORTED, "phc_stat_acc_read: 8-byte aligned accumulator required");
    PHC_LAUNCH(stat_acc_read_kernel, dim3((h + 255) / 256), dim3(256), 0, (hipStream_t)stream,
               reinterpret_cast<const unsigned long long *>(acc), h, (long long)count, out);
    return check_launch("stat_acc_read");
}

int phc_bn_fwd_from_moments(const float *x, int64_t N_local, const float *moments_all, const float *counts_all, int num_ranks,
                            int64_t N_total, const float *const *weight, const float *const *bias, float *const *running_mean,
                            float *const *running_var, int64_t *const *num_batches_tracked, int n_comp, int h, float momentum,
                            float eps, int act, const float *residual, float *out, float *save_stat, void *stream) {
    BnParams p;
    int rc = fill_params("phc_bn_fwd_from_moments", &p, weight, bias, running_mean, running_var, num_batches_tracked, n_comp, h);
    if (rc) return rc;
    PHC_REQUIRE(N_local >= 0 && N_total >= 1 && N_local <= N_total && num_ranks >= 1 && num_ranks <= BN_MAX_PARTIALS && h <= 8192,
                PHC_ERR_INVALID_ARG, "phc_bn_fwd_from_moments: N_local=%lld N_total=%lld ranks=%d h=%d", (long long)N_local,
                (long long)N_total, num_ranks, h);
    PHC_REQUIRE(act >= PHC_ACT_IDENTITY && act <= PHC_ACT_SELU, PHC_ERR_INVALID_ARG, "phc_bn_fwd_from_moments: unknown activation %d", act);
    PHC_REQUIRE(save_stat && moments_all && counts_all && (N_local == 0 || (x && out)), PHC_ERR_INVALID_ARG,
                "phc_bn_fwd_from_moments: null pointer");
    const bool vec4 = (h % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0) &&
                      (!residual || (reinterpret_cast<uintptr_t>(residual) & 15) == 0);
    PHC_REQUIRE(vec4 || h % 4 != 0, PHC_ERR_UNSUPPORTED, "phc_bn_fwd_from_moments: rows must be 16-byte aligned when h %% 4 == 0");
    hipStream_t s = (hipStream_t)stream;
    {
        // every rank merges the same num_ranks pairs in the same order: identical statistics (and running-stat updates) everywhere
        ProfScope prof(K_BN_FINALIZE, s);
        PHC_LAUNCH(bn_finalize_kernel, dim3((h + 7) / 8), dim3(BN_THREADS), 0, s, p, (const int *)nullptr, (long long)N_total, h, 1,
                   momentum, eps, moments_all, num_ranks, 0LL, save_stat, counts_all);
        rc = check_launch("bn_finalize(from moments)");
        if (rc) return rc;
    }
    if (N_local == 0) return PHC_OK;
    const size_t lds = sizeof(float) * 2 * (size_t)h;
    ProfScope prof(K_BN_APPLY, s);
#define PHC_BN_FWDM(ACTC)                                                                                                      \
    PHC_LAUNCH((bn_apply_kernel<ACTC>), dim3(apply_grid(N_local, h)), dim3(BN_THREADS), lds, s, x, (const int *)nullptr,        \
               (long long)N_local, h, residual, save_stat, out)
    PHC_BN_ACT_SWITCH(act, PHC_BN_FWDM)
#undef PHC_BN_FWDM
    return check_launch("bn_apply");
}

int phc_bn_bwd_sums(const float *dy, const float *x, const float *save_stat, int64_t N, int h, int act, float *dwb_local, void *ws,
                    size_t ws_bytes, void *stream) {
    PHC_REQUIRE(N >= 0 && h >= 1 && h <= 8192, PHC_ERR_INVALID_ARG, "phc_bn_bwd_sums: N=%lld h=%d", (long long)N, h);
    PHC_REQUIRE(act >= PHC_ACT_IDENTITY && act <= PHC_ACT_SELU, PHC_ERR_INVALID_ARG, "phc_bn_bwd_sums: unknown activation %d", act);
    PHC_REQUIRE(dwb_local && save_stat, PHC_ERR_INVALID_ARG, "phc_bn_bwd_sums: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (N == 0) {
        hipMemsetAsync(dwb_local, 0, sizeof(float) * 2 * (size_t)h, s);
        return PHC_OK;
    }
    PHC_REQUIRE(dy && x, PHC_ERR_INVALID_ARG, "phc_bn_bwd_sums: null pointer");
    PHC_REQUIRE(ws && ws_bytes >= phc_bn_ws_bytes(N, h), PHC_ERR_WORKSPACE, "phc_bn_bwd_sums: workspace too small");
    const bool vec4 = (h % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && ((reinterpret_cast<uintptr_t>(dy) & 15) == 0);
    PHC_REQUIRE(vec4 || h % 4 != 0, PHC_ERR_UNSUPPORTED, "phc_bn_bwd_sums: rows must be 16-byte aligned when h %% 4 == 0");
    BnPlan plan = bn_plan(N);
    float *partial = static_cast<float *>(ws);
    float *bstat = partial + 2 * (size_t)h * BN_MAX_PARTIALS;
    {
        ProfScope prof(K_BN_BWD_PARTIAL, s);
#define PHC_BN_BPS(ACTC)                                                                                                     \
    if (vec4)                                                                                                                \
        PHC_LAUNCH((bn_bwd_partial_kernel<ACTC, 4>), dim3(plan.num_partials), dim3(BN_THREADS), 0, s, dy, x, save_stat,     \
                   (const int *)nullptr, (long long)N, h, plan.rows_per_block, partial);                                     \
    else                                                                                                                     \
        PHC_LAUNCH((bn_bwd_partial_kernel<ACTC, 1>), dim3(plan.num_partials), dim3(BN_THREADS), 0, s, dy, x, save_stat,     \
                   (const int *)nullptr, (long long)N, h, plan.rows_per_block, partial)
        PHC_BN_ACT_SWITCH(act, PHC_BN_BPS)
#undef PHC_BN_BPS
    }
    ProfScope prof(K_BN_BWD_FINALIZE, s);
    PHC_LAUNCH(bn_bwd_finalize_kernel, dim3((h + 7) / 8), dim3(BN_THREADS), 0, s, (const int *)nullptr, (long long)N, h, 1, partial,
               plan.num_partials, dwb_local, bstat);
    return check_launch("bn_bwd_sums");
}

int phc_bn_bwd_apply(const float *dy, const float *x, const float *save_stat, const float *dwb_global, int64_t N_total, int64_t N_local,
                     int h, int act, float *dx, void *ws, size_t ws_bytes, void *stream) {
    PHC_REQUIRE(N_local >= 0 && N_total >= 1 && h >= 1 && h <= 8192, PHC_ERR_INVALID_ARG, "phc_bn_bwd_apply: N_local=%lld N_total=%lld h=%d",
                (long long)N_local, (long long)N_total, h);
    PHC_REQUIRE(act >= PHC_ACT_IDENTITY && act <= PHC_ACT_SELU, PHC_ERR_INVALID_ARG, "phc_bn_bwd_apply: unknown activation %d", act);
    if (N_local == 0) return PHC_OK;
    PHC_REQUIRE(dy && x && save_stat && dwb_global && dx, PHC_ERR_INVALID_ARG, "phc_bn_bwd_apply: null pointer");
    PHC_REQUIRE(ws && ws_bytes >= sizeof(float) * 2 * (size_t)h, PHC_ERR_WORKSPACE, "phc_bn_bwd_apply: workspace too small");
    const bool vec4 = (h % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && ((reinterpret_cast<uintptr_t>(dy) & 15) == 0) &&
                      ((reinterpret_cast<uintptr_t>(dx) & 15) == 0);
    PHC_REQUIRE(vec4 || h % 4 != 0, PHC_ERR_UNSUPPORTED, "phc_bn_bwd_apply: rows must be 16-byte aligned when h %% 4 == 0");
    hipStream_t s = (hipStream_t)stream;
    float *bstat = static_cast<float *>(ws);
    {
        ProfScope prof(K_BN_BWD_FINALIZE, s);
        PHC_LAUNCH(bn_bstat_from_dwb_kernel, dim3((h + 255) / 256), dim3(256), 0, s, dwb_global, h, 1.f / (float)N_total, bstat);
    }
    const size_t lds = sizeof(float) * 6 * (size_t)h;
    ProfScope prof(K_BN_BWD_APPLY, s);
#define PHC_BN_BAS(ACTC)                                                                                                     \
    PHC_LAUNCH((bn_bwd_apply_kernel<ACTC>), dim3(apply_grid(N_local, h)), dim3(BN_THREADS), lds, s, dy, x, save_stat,       \
               (const int *)nullptr, (long long)N_local, h, bstat, dx)
    PHC_BN_ACT_SWITCH(act, PHC_BN_BAS)
#undef PHC_BN_BAS
    return check_launch("bn_bwd_apply");
}

}  // extern "C"
