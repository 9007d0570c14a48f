// Neighbour aggregation (scatter-sum over edge_index) on gfx950.
//
// Replaces  x.index_select(src) -> msg_encoder(x_j + edge_attr) ->
// torch_scatter.scatter(reduce='add') -> "+ x" self loop
// (phc/hypercomplex/undirectional/messagepassing.py:57-60,72-74 / :133-138,144-146)
// and its autograd.  The reference materialises two [E, h] temporaries and
// scatters with atomics; here one wavefront owns one destination row of the
// destination-sorted edge list, gathers whole source rows with 16-byte-per-lane
// loads, accumulates in registers and writes the row once.  No atomics, one fixed
// summation order (ascending original edge id) => bitwise reproducible.
//
// A wavefront is split into G = 64 / LPR groups of LPR lanes; a group covers one
// feature row (LPR * 4 floats per pass) and the G groups walk different edges of
// the segment concurrently, then fold their partial sums with cross-lane moves.
// HBM-bound: algorithmic bytes per launch are in DESIGN.md §4.
#include "phc_common.h"

namespace phc {

constexpr int AGG_THREADS = 256;                    // 4 wavefronts: 4 rows in flight per workgroup

__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

template <int ACT>
__device__ __forceinline__ float4 f4_act(float4 v) {
    return make_float4(act_fwd<ACT>(v.x), act_fwd<ACT>(v.y), act_fwd<ACT>(v.z), act_fwd<ACT>(v.w));
}

template <int ACT>
__device__ __forceinline__ float4 f4_actgrad_mul(float4 pre, float4 g) {
    return make_float4(act_grad<ACT>(pre.x) * g.x, act_grad<ACT>(pre.y) * g.y, act_grad<ACT>(pre.z) * g.z,
                       act_grad<ACT>(pre.w) * g.w);
}

// Sum the partial float4 held by the G lane groups of a wavefront (group stride LPR lanes).
template <int LPR>
__device__ __forceinline__ float4 fold_groups(float4 v) {
#pragma unroll
    for (int off = LPR; off < WAVE; off <<= 1) {
        v.x += __shfl_xor(v.x, off, WAVE);
        v.y += __shfl_xor(v.y, off, WAVE);
        v.z += __shfl_xor(v.z, off, WAVE);
        v.w += __shfl_xor(v.w, off, WAVE);
    }
    return v;
}

// out[d] = sum_e phi(x[col[e]] + ea[perm[e]]) (+ x[d])      — h % 4 == 0, rows 16-byte aligned
template <int LPR, int ACT, bool HAS_EA>
__global__ __launch_bounds__(AGG_THREADS) void aggregate_fwd_kernel(
    const float *__restrict__ x, const float *__restrict__ ea, const int *__restrict__ rowptr,
    const int *__restrict__ col, const int *__restrict__ perm, long long n_dst, int h, int add_self,
    float *__restrict__ out) {
    constexpr int G = WAVE / LPR;
    const int lane = threadIdx.x & 63;
    const int grp = lane / LPR, li = lane - grp * LPR;
    const long long wave_id = ((long long)blockIdx.x * AGG_THREADS + threadIdx.x) >> 6;
    const long long n_waves = ((long long)gridDim.x * AGG_THREADS) >> 6;
    const int h4 = h >> 2;

    for (long long d = wave_id; d < n_dst; d += n_waves) {
        const int e_begin = rowptr[d], e_end = rowptr[d + 1];
        for (int c4 = li; c4 < h4; c4 += LPR) {
            float4 acc0 = make_float4(0.f, 0.f, 0.f, 0.f), acc1 = acc0;
            int e = e_begin + grp;
            // two edges per group in flight
            for (; e + G < e_end; e += 2 * G) {
                const int s0 = col[e], s1 = col[e + G];
                float4 v0 = reinterpret_cast<const float4 *>(x + (size_t)s0 * h)[c4];
                float4 v1 = reinterpret_cast<const float4 *>(x + (size_t)s1 * h)[c4];
                if (HAS_EA) {
                    const int p0 = perm[e], p1 = perm[e + G];
                    v0 = f4_add(v0, reinterpret_cast<const float4 *>(ea + (size_t)p0 * h)[c4]);
                    v1 = f4_add(v1, reinterpret_cast<const float4 *>(ea + (size_t)p1 * h)[c4]);
                }
                acc0 = f4_add(acc0, f4_act<ACT>(v0));
                acc1 = f4_add(acc1, f4_act<ACT>(v1));
            }
            if (e < e_end) {
                const int s0 = col[e];
                float4 v0 = reinterpret_cast<const float4 *>(x + (size_t)s0 * h)[c4];
                if (HAS_EA) v0 = f4_add(v0, reinterpret_cast<const float4 *>(ea + (size_t)perm[e] * h)[c4]);
                acc0 = f4_add(acc0, f4_act<ACT>(v0));
            }
            float4 acc = fold_groups<LPR>(f4_add(acc0, acc1));
            if (grp == 0) {
                if (add_self & 1) acc = f4_add(acc, reinterpret_cast<const float4 *>(x + (size_t)d * h)[c4]);
                if (add_self & 2) acc = f4_add(acc, reinterpret_cast<const float4 *>(out + (size_t)d * h)[c4]);   // accumulate
                reinterpret_cast<float4 *>(out + (size_t)d * h)[c4] = acc;
            }
        }
    }
}

// Scalar fallback for widths that are not a multiple of 4 floats (or unaligned bases).
template <int ACT>
__global__ __launch_bounds__(AGG_THREADS) void aggregate_fwd_scalar_kernel(
    const float *__restrict__ x, const float *__restrict__ ea, const int *__restrict__ rowptr,
    const int *__restrict__ col, const int *__restrict__ perm, long long n_dst, int h, int add_self,
    float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long long wave_id = ((long long)blockIdx.x * AGG_THREADS + threadIdx.x) >> 6;
    const long long n_waves = ((long long)gridDim.x * AGG_THREADS) >> 6;
    for (long long d = wave_id; d < n_dst; d += n_waves) {
        const int e_begin = rowptr[d], e_end = rowptr[d + 1];
        for (int c = lane; c < h; c += WAVE) {
            float acc = 0.f;
            for (int e = e_begin; e < e_end; ++e) {
                float v = x[(size_t)col[e] * h + c];
                if (ea) v += ea[(size_t)perm[e] * h + c];
                acc += act_fwd<ACT>(v);
            }
            if (add_self & 1) acc += x[(size_t)d * h + c];
            if (add_self & 2) acc += out[(size_t)d * h + c];
            out[(size_t)d * h + c] = acc;
        }
    }
}

// Backward over the source-sorted structure:
//   g_e = phi'(x[s] + ea[e]) * dout[dst_e];  dea[e] = g_e;  dx[s] = sum g_e (+ dout[s])
template <int LPR, int ACT, bool HAS_EA>
__global__ __launch_bounds__(AGG_THREADS) void aggregate_bwd_kernel(
    const float *__restrict__ dout, const float *__restrict__ x, const float *__restrict__ ea,
    const int *__restrict__ rowptr_t, const int *__restrict__ col_t, const int *__restrict__ perm_t, long long n_src,
    long long n_dst, int h, int add_self, float *__restrict__ dx, float *__restrict__ dea) {
    constexpr int G = WAVE / LPR;
    const int lane = threadIdx.x & 63;
    const int grp = lane / LPR, li = lane - grp * LPR;
    const long long wave_id = ((long long)blockIdx.x * AGG_THREADS + threadIdx.x) >> 6;
    const long long n_waves = ((long long)gridDim.x * AGG_THREADS) >> 6;
    const int h4 = h >> 2;

    for (long long s = wave_id; s < n_src; s += n_waves) {
        const int e_begin = rowptr_t[s], e_end = rowptr_t[s + 1];
        for (int c4 = li; c4 < h4; c4 += LPR) {
            float4 xs = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ACT != PHC_ACT_IDENTITY) xs = reinterpret_cast<const float4 *>(x + (size_t)s * h)[c4];
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int e = e_begin + grp; e < e_end; e += G) {
                const int d = col_t[e];
                const int p = perm_t[e];
                float4 g = reinterpret_cast<const float4 *>(dout + (size_t)d * h)[c4];
                if (ACT != PHC_ACT_IDENTITY) {
                    float4 pre = xs;
                    if (HAS_EA) pre = f4_add(pre, reinterpret_cast<const float4 *>(ea + (size_t)p * h)[c4]);
                    g = f4_actgrad_mul<ACT>(pre, g);
                }
                if (dea) reinterpret_cast<float4 *>(dea + (size_t)p * h)[c4] = g;
                acc = f4_add(acc, g);
            }
            acc = fold_groups<LPR>(acc);
            if (grp == 0) {
                if (add_self && s < n_dst) acc = f4_add(acc, reinterpret_cast<const float4 *>(dout + (size_t)s * h)[c4]);
                reinterpret_cast<float4 *>(dx + (size_t)s * h)[c4] = acc;
            }
        }
    }
}

template <int ACT>
__global__ __launch_bounds__(AGG_THREADS) void aggregate_bwd_scalar_kernel(
    const float *__restrict__ dout, const float *__restrict__ x, const float *__restrict__ ea,
    const int *__restrict__ rowptr_t, const int *__restrict__ col_t, const int *__restrict__ perm_t, long long n_src,
    long long n_dst, int h, int add_self, float *__restrict__ dx, float *__restrict__ dea) {
    const int lane = threadIdx.x & 63;
    const long long wave_id = ((long long)blockIdx.x * AGG_THREADS + threadIdx.x) >> 6;
    const long long n_waves = ((long long)gridDim.x * AGG_THREADS) >> 6;
    for (long long s = wave_id; s < n_src; s += n_waves) {
        const int e_begin = rowptr_t[s], e_end = rowptr_t[s + 1];
        for (int c = lane; c < h; c += WAVE) {
            const float xs = (ACT != PHC_ACT_IDENTITY) ? x[(size_t)s * h + c] : 0.f;
            float acc = 0.f;
            for (int e = e_begin; e < e_end; ++e) {
                const int p = perm_t[e];
                float g = dout[(size_t)col_t[e] * h + c];
                if (ACT != PHC_ACT_IDENTITY) g *= act_grad<ACT>(xs + (ea ? ea[(size_t)p * h + c] : 0.f));
                if (dea) dea[(size_t)p * h + c] = g;
                acc += g;
            }
            if (add_self && s < n_dst) acc += dout[(size_t)s * h + c];
            dx[(size_t)s * h + c] = acc;
        }
    }
}

// out[r] = src[idx[r]]
__global__ __launch_bounds__(AGG_THREADS) void gather_rows_kernel(const float *__restrict__ src,
                                                                 const long long *__restrict__ idx, long long n_rows,
                                                                 int h, int vec4, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long long wave_id = ((long long)blockIdx.x * AGG_THREADS + threadIdx.x) >> 6;
    const long long n_waves = ((long long)gridDim.x * AGG_THREADS) >> 6;
    for (long long r = wave_id; r < n_rows; r += n_waves) {
        const long long s = idx[r];
        if (vec4) {
            for (int c4 = lane; c4 < (h >> 2); c4 += WAVE)
                reinterpret_cast<float4 *>(out + (size_t)r * h)[c4] = reinterpret_cast<const float4 *>(src + (size_t)s * h)[c4];
        } else {
            for (int c = lane; c < h; c += WAVE) out[(size_t)r * h + c] = src[(size_t)s * h + c];
        }
    }
}

// ---- dispatch ----------------------------------------------------------------------------------
static inline unsigned agg_grid(long long rows) {
    long long blocks = (rows + 3) / 4;                  // one wavefront per row, 4 per workgroup
    const long long cap = 256LL * 8 * 4;                // fill every CU 8 workgroups deep, stride the rest
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

static inline int pick_lpr(int h4) { return h4 <= 8 ? 8 : h4 <= 16 ? 16 : h4 <= 32 ? 32 : 64; }

#define PHC_DISPATCH_ACT(ACTVAR, ...)                                              \
    switch (ACTVAR) {                                                              \
        case PHC_ACT_IDENTITY: { constexpr int ACT = PHC_ACT_IDENTITY; __VA_ARGS__; } break; \
        case PHC_ACT_RELU: { constexpr int ACT = PHC_ACT_RELU; __VA_ARGS__; } break;       \
        case PHC_ACT_LRELU: { constexpr int ACT = PHC_ACT_LRELU; __VA_ARGS__; } break;     \
        case PHC_ACT_ELU: { constexpr int ACT = PHC_ACT_ELU; __VA_ARGS__; } break;         \
        case PHC_ACT_SWISH: { constexpr int ACT = PHC_ACT_SWISH; __VA_ARGS__; } break;     \
        case PHC_ACT_SELU: { constexpr int ACT = PHC_ACT_SELU; __VA_ARGS__; } break;       \
        default: break;                                                            \
    }

#define PHC_DISPATCH_LPR(LPRVAR, ...)                                  \
    switch (LPRVAR) {                                                  \
        case 8: { constexpr int LPR = 8; __VA_ARGS__; } break;         \
        case 16: { constexpr int LPR = 16; __VA_ARGS__; } break;       \
        case 32: { constexpr int LPR = 32; __VA_ARGS__; } break;       \
        default: { constexpr int LPR = 64; __VA_ARGS__; } break;       \
    }

static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace phc

using namespace phc;

extern "C" {

int phc_aggregate_fwd(const float *x, const float *ea, const int32_t *rowptr, const int32_t *col, const int32_t *perm,
                      int64_t n_dst, int h, int act, int add_self, float *out, void *stream) {
    PHC_REQUIRE(n_dst >= 0 && h >= 1, PHC_ERR_INVALID_ARG, "phc_aggregate_fwd: n_dst=%lld h=%d", (long long)n_dst, h);
    PHC_REQUIRE(act >= PHC_ACT_IDENTITY && act <= PHC_ACT_SELU, PHC_ERR_INVALID_ARG, "phc_aggregate_fwd: unknown activation %d", act);
    if (n_dst == 0) return PHC_OK;
    PHC_REQUIRE(x && rowptr && col && out && (!ea || perm), PHC_ERR_INVALID_ARG, "phc_aggregate_fwd: null pointer");
    PHC_REQUIRE(out != x, PHC_ERR_INVALID_ARG, "phc_aggregate_fwd: out must not alias x");
    hipStream_t s = (hipStream_t)stream;
    const unsigned grid = agg_grid(n_dst);
    const bool vec = (h % 4 == 0) && aligned16(x) && aligned16(out) && (!ea || aligned16(ea));
    ProfScope prof(K_AGG_FWD, s);
    if (vec) {
        const int lpr = pick_lpr(h >> 2);
        PHC_DISPATCH_ACT(act, PHC_DISPATCH_LPR(lpr, {
            if (ea)
                PHC_LAUNCH((aggregate_fwd_kernel<LPR, ACT, true>), dim3(grid), dim3(AGG_THREADS), 0, s, x, ea, rowptr,
                                   col, perm, (long long)n_dst, h, add_self, out);
            else
                PHC_LAUNCH((aggregate_fwd_kernel<LPR, ACT, false>), dim3(grid), dim3(AGG_THREADS), 0, s, x, ea, rowptr,
                                   col, perm, (long long)n_dst, h, add_self, out);
        }));
    } else {
        PHC_DISPATCH_ACT(act, PHC_LAUNCH((aggregate_fwd_scalar_kernel<ACT>), dim3(grid), dim3(AGG_THREADS), 0, s, x, ea,
                                                 rowptr, col, perm, (long long)n_dst, h, add_self, out));
    }
    return check_launch("aggregate_fwd");
}

int phc_aggregate_bwd(const float *dout, const float *x, const float *ea, const int32_t *rowptr_t, const int32_t *col_t,
                      const int32_t *perm_t, int64_t n_src, int64_t n_dst, int h, int act, int add_self, float *dx, float *dea,
                      void *stream) {
    PHC_REQUIRE(n_src >= 0 && n_dst >= 0 && h >= 1, PHC_ERR_INVALID_ARG, "phc_aggregate_bwd: n_src=%lld n_dst=%lld h=%d",
                (long long)n_src, (long long)n_dst, h);
    PHC_REQUIRE(act >= PHC_ACT_IDENTITY && act <= PHC_ACT_SELU, PHC_ERR_INVALID_ARG, "phc_aggregate_bwd: unknown activation %d", act);
    if (n_src == 0) return PHC_OK;
    PHC_REQUIRE(dout && rowptr_t && col_t && perm_t && dx, PHC_ERR_INVALID_ARG, "phc_aggregate_bwd: null pointer");
    PHC_REQUIRE(act == PHC_ACT_IDENTITY || x, PHC_ERR_INVALID_ARG, "phc_aggregate_bwd: x is required when act != identity");
    hipStream_t s = (hipStream_t)stream;
    const unsigned grid = agg_grid(n_src);
    const bool vec = (h % 4 == 0) && aligned16(dout) && aligned16(dx) && (!x || aligned16(x)) && (!ea || aligned16(ea)) &&
                     (!dea || aligned16(dea));
    ProfScope prof(K_AGG_BWD, s);
    if (vec) {
        const int lpr = pick_lpr(h >> 2);
        PHC_DISPATCH_ACT(act, PHC_DISPATCH_LPR(lpr, {
            if (ea)
                PHC_LAUNCH((aggregate_bwd_kernel<LPR, ACT, true>), dim3(grid), dim3(AGG_THREADS), 0, s, dout, x, ea,
                                   rowptr_t, col_t, perm_t, (long long)n_src, (long long)n_dst, h, add_self, dx, dea);
            else
                PHC_LAUNCH((aggregate_bwd_kernel<LPR, ACT, false>), dim3(grid), dim3(AGG_THREADS), 0, s, dout, x, ea,
                                   rowptr_t, col_t, perm_t, (long long)n_src, (long long)n_dst, h, add_self, dx, dea);
        }));
    } else {
        PHC_DISPATCH_ACT(act, PHC_LAUNCH((aggregate_bwd_scalar_kernel<ACT>), dim3(grid), dim3(AGG_THREADS), 0, s, dout, x,
                                                 ea, rowptr_t, col_t, perm_t, (long long)n_src, (long long)n_dst, h, add_self, dx,
                                                 dea));
    }
    return check_launch("aggregate_bwd");
}

int phc_gather_rows_i64(const float *src, const int64_t *idx_i64, int64_t n_rows, int h, float *out, void *stream) {
    PHC_REQUIRE(n_rows >= 0 && h >= 1, PHC_ERR_INVALID_ARG, "phc_gather_rows_i64: n_rows=%lld h=%d", (long long)n_rows, h);
    if (n_rows == 0) return PHC_OK;
    PHC_REQUIRE(src && idx_i64 && out, PHC_ERR_INVALID_ARG, "phc_gather_rows_i64: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int vec4 = (h % 4 == 0) && aligned16(src) && aligned16(out);
    ProfScope prof(K_GATHER_ROWS, s);
    PHC_LAUNCH(gather_rows_kernel, dim3(agg_grid(n_rows)), dim3(AGG_THREADS), 0, s, src,
                       reinterpret_cast<const long long *>(idx_i64), (long long)n_rows, h, vec4, out);
    return check_launch("gather_rows");
}

}  // extern "C"
