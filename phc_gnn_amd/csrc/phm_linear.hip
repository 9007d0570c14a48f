// PHM linear transform on gfx950:  y = x . H(A, W) + b,  H = sum_i A_i (x) W_i
//
// Replaces the reference's einsum-Kronecker + sum + matmul + bias-add chain
// (phc/hypercomplex/kronecker.py:35-48, layers.py:198-219) and its autograd.
//
// Forward / dX kernel (phm_gemm_kernel):
//   * the n*n*n contribution tensor A is staged in LDS once per workgroup;
//   * the [K x 128] panel of H (or H^T for dX) a workgroup needs is ASSEMBLED IN
//     LDS from W (n*kin*kout floats, L2 resident) — H never exists in HBM;
//   * x tiles [64 x K-chunk] are staged through LDS with 16-byte loads and fed
//     to v_mfma_f32_32x32x2_f32 (exact fp32; 4 waves as 2x2, each 32 rows x 64 cols).
//   When K <= 128 the H panel is built once and the workgroup streams M tiles
//   under it.
// Backward w.r.t. parameters:
//   dH = x^T . dy as fixed-order split-K slabs (phm_dH_partial_kernel), a slab
//   reduction, then the two small contractions dW = <A, dH>, dA = <W, dH>.
//   No atomics anywhere: results are bitwise reproducible.
#include <stdlib.h>

#include "phc_common.h"
#include "phm_grad_device.h"
#include "gemm_hs_device.h"
#include "gemm_hs_ag_device.h"
#include <vector>

namespace phc {


constexpr int GEMM_BK = 128;       // K chunk held in LDS
constexpr int XS_STRIDE = GEMM_BK + 4;   // floats; +4 keeps ds_read_b128 conflict-free (16-lane groups hit 16 distinct 4-bank slots)
constexpr int MAX_PHM_DIM = 16;
constexpr int HB = 8;              // W-load batch of the panel builder (HB * n loads in flight per thread)

// Two tile configurations of the same kernel (chosen by the row count on the host):
//   latency    : 64 x 64 tile, 256 threads, 2 workgroups per CU  — molecule batches (a few hundred tiles in all)
//   throughput : 128 x 128 tile, 512 threads (2 waves per SIMD), x streamed once per 128 outputs — large graphs
template <int BM, int BN>
struct GemmCfg {
    static constexpr int CT = BN / 64;                       // 32-column MFMA tiles per wave (waves are 32 rows x CT*32 cols)
    static constexpr int WAVES_M = BM / 32, WAVES_N = 2;
    static constexpr int THREADS = 64 * WAVES_M * WAVES_N;
    static constexpr int HS_STRIDE = BN + 1;                 // odd: conflict-free for both fill orders and the b32 operand reads
    static constexpr int X_REGS = BM * GEMM_BK / 4 / THREADS;
    static constexpr size_t lds_floats(int n) { return (size_t)BM * XS_STRIDE + (size_t)GEMM_BK * HS_STRIDE + (size_t)n * n * n; }
};

__host__ __device__ inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// Element (k, j) of the GEMM's right operand (used by the exported H builder).
//   TRANS = 0: operand is H    : k = a*kin + kap (input feature), j = c*kout + pi (output feature)
//   TRANS = 1: operand is H^T  : k = c*kout + pi,                 j = a*kin + kap
template <int TRANS>
__device__ __forceinline__ float h_element(const float *__restrict__ As, const float *__restrict__ W, int n, int kin,
                                           int kout, int k, int j) {
    int a, kap, c, pi;
    if (TRANS == 0) {
        a = k / kin; kap = k - a * kin;
        c = j / kout; pi = j - c * kout;
    } else {
        c = k / kout; pi = k - c * kout;
        a = j / kin; kap = j - a * kin;
    }
    const float *w = W + (size_t)kap * kout + pi;
    const float *arow = As + a * n + c;
    const int wstride = kin * kout;
    const int astride = n * n;
    float h = 0.f;
#pragma unroll 4
    for (int i = 0; i < n; ++i) h = fmaf(arow[i * astride], w[(size_t)i * wstride], h);
    return h;
}

// Assemble rows [k0, k0+kc) x columns [j0, j0+BN) of the operand in LDS (hs[kk * HS_STRIDE + jj]).
// W-stationary: each work item loads the n values W[0..n)[kap][pi] ONCE (contiguous in pi across
// lanes) and emits every operand element that uses them (one per a for H, one per c for H^T), mixing
// with the contribution tensor held in LDS.  Items are processed HB at a time so that HB*n global
// loads are in flight per thread.  NPHM > 0 fixes n at compile time.
template <int TRANS, int NPHM, int BN, int THREADS>
__device__ __forceinline__ void build_h_chunk(const float *__restrict__ As, const float *__restrict__ W, float *__restrict__ hs,
                                              int n_rt, int kin, int kout, int k0, int kc, int kc8, int j0, int Nout, int tid) {
    constexpr int HS_STRIDE = BN + 1;
    const int n = NPHM ? NPHM : n_rt;
    constexpr int NN = NPHM ? NPHM : MAX_PHM_DIM;
    const int wstride = kin * kout;
    const int astride = n * n;
    // rows kc..kc8 are MFMA padding
    for (int e = tid; e < (kc8 - kc) * BN; e += THREADS) hs[(kc + e / BN) * HS_STRIDE + (e % BN)] = 0.f;
    if (TRANS == 0) {
        constexpr int STEP = THREADS / BN;
        const int jj = tid & (BN - 1);
        const int j = j0 + jj;
        const bool jvalid = j < Nout;
        const int jc = min(j, Nout - 1);                     // clamped: loads are unconditional (no wait at a branch join)
        const int c = jc / kout;
        const int pi = jc - c * kout;
        const float *wbase = W + pi;
        const float *abase = As + c;
        for (int kap0 = tid / BN; kap0 < kin; kap0 += HB * STEP) {
            float w[HB][NN];
#pragma unroll
            for (int b = 0; b < HB; ++b) {
                const int kap = min(kap0 + b * STEP, kin - 1);
#pragma unroll
                for (int i = 0; i < NN; ++i) w[b][i] = wbase[(size_t)min(i, n - 1) * wstride + kap * kout];
            }
#pragma unroll
            for (int b = 0; b < HB; ++b) {
                const int kap = kap0 + b * STEP;
                if (kap >= kin) break;
                int a = k0 > kap ? (k0 - kap + kin - 1) / kin : 0;
                for (; a < n; ++a) {
                    const int kk = a * kin + kap - k0;
                    if (kk >= kc) break;
                    float hsum = 0.f;
#pragma unroll
                    for (int i = 0; i < NN; ++i)
                        if (i < n) hsum = fmaf(abase[i * astride + a * n], w[b][i], hsum);
                    hs[kk * HS_STRIDE + jj] = jvalid ? hsum : 0.f;
                }
            }
        }
    } else {
        const int items = BN * kout;
        for (int it0 = tid; it0 < items; it0 += HB * THREADS) {
            float w[HB][NN];
            int jjv[HB], piv[HB], av[HB];
            bool okv[HB];
#pragma unroll
            for (int b = 0; b < HB; ++b) {
                const int it = min(it0 + b * THREADS, items - 1);
                const int jj = it / kout;
                const int pi = it - jj * kout;
                const int j = j0 + jj;
                const int jc = min(j, Nout - 1);             // clamped: unconditional loads
                const int a = jc / kin;
                const int kap = jc - a * kin;
                jjv[b] = jj; piv[b] = pi; av[b] = a; okv[b] = j < Nout;
                const float *wbase = W + (size_t)kap * kout + pi;
#pragma unroll
                for (int i = 0; i < NN; ++i) w[b][i] = wbase[(size_t)min(i, n - 1) * wstride];
            }
#pragma unroll
            for (int b = 0; b < HB; ++b) {
                if (it0 + b * THREADS >= items) break;
                const int pi = piv[b], a = av[b];
                int c = k0 > pi ? (k0 - pi + kout - 1) / kout : 0;
                for (; c < n; ++c) {
                    const int kk = c * kout + pi - k0;
                    if (kk >= kc) break;
                    float hsum = 0.f;
#pragma unroll
                    for (int i = 0; i < NN; ++i)
                        if (i < n) hsum = fmaf(As[i * astride + a * n + c], w[b][i], hsum);
                    hs[kk * HS_STRIDE + jjv[b]] = okv[b] ? hsum : 0.f;
                }
            }
        }
    }
}

// out[M, Nout] = X[M, K] . Op(K, Nout) (+ bias[Nout]) (+ residual[M, Nout])
// grid.x = number of BN-wide output panels, grid.y = workgroups striding over M tiles.
// Per (M tile, K chunk): x chunk -> registers (issued first / prefetched under the previous MFMA loop),
// operand panel assembled in LDS (once per workgroup when K fits one chunk), registers -> LDS, MFMA.
template <int TRANS, int NPHM, int BM, int BN>
__global__ __launch_bounds__((GemmCfg<BM, BN>::THREADS)) void phm_gemm_kernel(
    const float *__restrict__ X, const float *__restrict__ A, const float *__restrict__ W,
    const float *__restrict__ bias, const float *__restrict__ residual, float *__restrict__ out,
    long long M, int K, int Nout, int n, int kin, int kout, int vec4) {
    using Cfg = GemmCfg<BM, BN>;
    constexpr int THREADS = Cfg::THREADS, HS_STRIDE = Cfg::HS_STRIDE, X_REGS = Cfg::X_REGS, CT = Cfg::CT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *xs = smem;                                   // [BM][XS_STRIDE]
    float *hs = xs + BM * XS_STRIDE;                    // [GEMM_BK][HS_STRIDE]
    float *As = hs + GEMM_BK * HS_STRIDE;               // [n*n*n]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;            // WAVES_M x 2 waves, each 32 rows x (CT*32) columns
    const int r = lane & 31, hh = lane >> 5;
    const int j0 = blockIdx.x * BN;

    for (int i = tid; i < n * n * n; i += THREADS) As[i] = A[i];

    const int num_chunks = (K + GEMM_BK - 1) / GEMM_BK;
    const long long num_mtiles = (M + BM - 1) / BM;
    const long long my_tiles = blockIdx.y < num_mtiles ? (num_mtiles - blockIdx.y + gridDim.y - 1) / gridDim.y : 0;
    const long long total_iters = my_tiles * num_chunks;
    bool h_ready = false;                               // panel resident across M tiles when K fits one chunk

    float4 xr[X_REGS];
    auto load_x = [&](long long it) {                   // global -> registers for iteration `it`
        const long long mt = blockIdx.y + (it / num_chunks) * gridDim.y;
        const int ch = (int)(it % num_chunks);
        const long long m0 = mt * BM;
        const int k0 = ch * GEMM_BK;
        const int kc = min(GEMM_BK, K - k0);
        const int c4n = round_up(kc, 8) >> 2;
#pragma unroll
        for (int t = 0; t < X_REGS; ++t) {
            const int e = tid + t * THREADS;
            const int row = e / c4n, c4 = e - row * c4n;
            // clamped, unconditional: nothing consumes the registers until the LDS write one iteration later
            const long long rr = min(m0 + row, M - 1);
            const int cc = min(c4 * 4, kc - 4);
            xr[t] = *reinterpret_cast<const float4 *>(X + (size_t)rr * K + k0 + cc);
        }
    };

    if (vec4 && total_iters > 0) load_x(0);
    f32x16 acc[CT];
    for (long long it = 0; it < total_iters; ++it) {
        const long long mt = blockIdx.y + (it / num_chunks) * gridDim.y;
        const int ch = (int)(it % num_chunks);
        const long long m0 = mt * BM;
        const int k0 = ch * GEMM_BK;
        const int kc = min(GEMM_BK, K - k0);            // valid k in this chunk
        const int kc8 = round_up(kc, 8);                // MFMA consumes k in groups of 8
        if (ch == 0) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[ct][i] = 0.f;
        }
        __syncthreads();                                // previous MFMA loop has finished reading xs / hs (and As is filled)

        if (!h_ready) {
#ifndef PHC_EXP_NO_HBUILD
            build_h_chunk<TRANS, NPHM, BN, THREADS>(As, W, hs, n, kin, kout, k0, kc, kc8, j0, Nout, tid);
#endif
            if (num_chunks == 1) h_ready = true;
        }
        if (vec4) {
            const int c4n = kc8 >> 2;
#pragma unroll
            for (int t = 0; t < X_REGS; ++t) {
                const int e = tid + t * THREADS;
                const int row = e / c4n, c4 = e - row * c4n;
                const bool ok = m0 + row < M && c4 * 4 < kc;                     // tail rows / k padding read as zero
                const float4 v = ok ? xr[t] : make_float4(0.f, 0.f, 0.f, 0.f);
                if (row < BM) *reinterpret_cast<float4 *>(xs + row * XS_STRIDE + c4 * 4) = v;
            }
        } else {
            for (int e = tid; e < BM * kc8; e += THREADS) {
                const int row = e / kc8;
                const int kk = e - row * kc8;
                float v = 0.f;
                if (m0 + row < M && kk < kc) v = X[(size_t)(m0 + row) * K + k0 + kk];
                xs[row * XS_STRIDE + kk] = v;
            }
        }
        __syncthreads();
        if (vec4 && it + 1 < total_iters) load_x(it + 1);      // in flight under the MFMA loop

        // lane (r, hh) supplies A[row r][k] and B[k][col r] for k = kg*8 + 4*hh + q, q = 0..3
        const float *xrow = xs + (wm * 32 + r) * XS_STRIDE + 4 * hh;
        const float *hcol = hs + (4 * hh) * HS_STRIDE + wn * (CT * 32) + r;
#ifdef PHC_EXP_NO_MFMA
        if (tid == 100000)
#endif
#pragma unroll 2
        for (int kg = 0; kg < kc8; kg += 8) {
            const float4 av = *reinterpret_cast<const float4 *>(xrow + kg);
            const float a4[4] = {av.x, av.y, av.z, av.w};
            const float *hb = hcol + kg * HS_STRIDE;
            float bq[4][CT];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) bq[q][ct] = hb[q * HS_STRIDE + ct * 32];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[q], bq[q][ct], acc[ct], 0, 0, 0);
        }

#ifdef PHC_EXP_NO_EPI
        if (ch == num_chunks - 1 && acc[0][0] == 123.456f) {
#else
        if (ch == num_chunks - 1) {                     // epilogue: bias, residual, store
#endif
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int col = j0 + wn * (CT * 32) + ct * 32 + r;
                const int colc = min(col, Nout - 1);
                const float bv = bias ? bias[colc] : 0.f;
                float rv[16];
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {   // residual rows: clamped, unconditional, all 16 loads in flight
                    const long long row = min(m0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hh, M - 1);
                    rv[reg] = residual ? residual[(size_t)row * Nout + colc] : 0.f;
                }
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const long long row = m0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
                    if (row < M && col < Nout) out[(size_t)row * Nout + col] = acc[ct][reg] + bv + rv[reg];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Streaming kernel for K <= 256 (every PHM layer of hidden width <= 256): wave-decoupled.
//   * the whole [K x BN] operand panel is assembled in LDS once per workgroup (one barrier);
//   * after that every WAVEFRONT is on its own: it owns 32-row slabs of x, stages them through a
//     private LDS buffer (no workgroup barrier), runs the MFMAs against the shared panel, transposes
//     its 32x32 result tiles through the same buffer and stores 16 bytes per lane.
//   Waves of one SIMD drift apart, so one wave's loads/stores overlap another's MFMAs.
// ------------------------------------------------------------------------------------------
#ifdef PHC_KTIMING
// experiment build only: wave 0 of the first and the last workgroup stamp the phases of the launch (100 MHz wall clock)
__device__ unsigned long long g_st_ts[64];
#define ST_STAMP(i)                                                                                              \
    do {                                                                                                         \
        if (threadIdx.x == 0 && ((blockIdx.x == 0 && blockIdx.y == 0) || (blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1))) \
            g_st_ts[(blockIdx.y == 0 ? 0 : 32) + (i)] = wall_clock64();                                          \
    } while (0)
#else
#define ST_STAMP(i)
#endif

constexpr int ST_KH = 64;                       // k columns staged per step
constexpr int ST_STRIDE = ST_KH + 4;            // floats; conflict-free ds_read_b128 of the A operand
constexpr int ST_FLOATS = 32 * ST_STRIDE;       // private staging floats per wave (8.5 KiB); also holds a 32x36 output tile

// STATS: the workgroup also emits, per output column, the (sum, M2) of the rows it produced below *n_valid —
// the batch-norm statistics of y come out of the GEMM's registers instead of a second pass over y
// (stats_partial[blockIdx.y][0..Nout) = sum, [Nout..2Nout) = M2 about the workgroup's own mean; stats_counts[blockIdx.y]).
// (XfAcc — an XF operand whose batch statistics are still in an accumulator — is declared in gemm_hs_device.h)
template <int TRANS, int NPHM, int BN, int WAVES, int STATS = 0>
__global__ __launch_bounds__(64 * WAVES) void phm_stream_kernel(
    const float *__restrict__ X, const float *__restrict__ A, const float *__restrict__ W,
    const float *__restrict__ bias, const float *__restrict__ residual, float *__restrict__ out,
    long long M, int K, int Nout, int n, int kin, int kout, int out_vec4,
    const int *__restrict__ n_valid = nullptr, float *__restrict__ stats_partial = nullptr,
    float *__restrict__ stats_counts = nullptr, const float *__restrict__ xf_stat = nullptr, int xf_act = 0,
    const float *__restrict__ bw_x = nullptr, const float *__restrict__ bw_stat = nullptr, int bw_act = 0, int stats_acc = 0,
    XfAcc xa = XfAcc{}) {
    // stats_acc != 0: stats_partial is a statistics accumulator (phc_common.h: 64-bit fixed-point words added with integer
    // atomics, no merge launch behind this one) instead of one [2 * Nout] partial row per workgroup row
    // STATS bit 0: emit batch-norm partials of the output; bit 1 (XF): the INPUT is a batch-norm that has not been
    // applied yet — x := act(x * scale + shift) per column (xf_stat = the norm's [mean|invstd|scale|shift|w|b], width K),
    // rows >= *n_valid read as 0 — evaluated while the slab is staged, so the normalised tensor never exists in HBM
    // STATS bit 2 (BW): this launch is dz = dy . H^T of a layer whose input was act(batch_norm(bw_x)); besides dz it emits
    // the batch-norm backward's column sums of g = dz * act'(.) and g * x_hat over its rows (stats_partial[blockIdx.y]
    // [0..Nout) / [Nout..2 Nout)), so that the norm's backward needs no separate pass over dz and bw_x for them
    constexpr bool ST = (STATS & 1) != 0, XF = (STATS & 2) != 0, BW = (STATS & 4) != 0;
    constexpr int THREADS = 64 * WAVES, HS_STRIDE = BN + 1, CT = BN / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *hs = smem;                                          // [round_up(K, 8)][HS_STRIDE]
    float *stage_all = hs + round_up(K, 8) * HS_STRIDE;        // [WAVES][ST_FLOATS]
    float *As = stage_all + WAVES * ST_FLOATS;                 // [n*n*n]
    float *xf_lds = As + round_up(n * n * n, 4);               // XF from an accumulator: scale[K] | shift[K] (16-byte aligned)

    // XCD-aware block order: workgroups go to the 8 XCDs round-robin by linear id, and the gridDim.x column panels of one
    // row-slab group read the same x rows.  Inside every run of 8 * gridDim.x consecutive ids the (panel, slab group) pairs
    // are permuted so that the panels of one slab group sit 8 ids apart — on ONE XCD, whose L2 then serves the re-reads.
    unsigned pbx = blockIdx.x, pby = blockIdx.y;
    if (PHC_XCD_MAP && gridDim.x > 1) {
        const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x, run = 8u * gridDim.x;
        const unsigned chunk = lin / run, rem = lin - chunk * run;
        if (chunk * 8u + 8u <= gridDim.y) {
            pbx = rem >> 3;
            pby = chunk * 8u + (rem & 7u);
        }
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int j0 = pbx * BN;
    float *stage = stage_all + wave * ST_FLOATS;
    ST_STAMP(0);

    const long long num_slabs = (M + 31) / 32;
    long long slab = (long long)pby * WAVES + wave;
    const long long slab_step = (long long)gridDim.y * WAVES;
    const int halves = (K + ST_KH - 1) / ST_KH;
    const int lrow = lane >> 4, lc4 = lane & 15;               // staging: lane -> (row lrow + 4t, float4 column lc4)

    // x staging registers (plain loads: hipcc drains them before the MFMA loop, so the overlap comes from the
    // other waves of the SIMD; an inline-asm variant that hid them from hipcc made it copy the still-pending
    // registers — cdna guide §5.7 item 1 — and was dropped)
    f32x4 xr[8];
    f32x4 xsc = {1.f, 1.f, 1.f, 1.f}, xsh = {0.f, 0.f, 0.f, 0.f};      // XF: scale / shift of the 4 columns held in xr
    int xf_k = 0;                                                      // first of those 4 columns
    auto load_half = [&](long long sl, int half) {
        const int k0 = half * ST_KH;
        const int cc = min(lc4 * 4, K - k0 - 4);               // clamped: unconditional loads
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const long long row = min(sl * 32 + lrow + 4 * t, M - 1);
            xr[t] = *reinterpret_cast<const f32x4 *>(X + (size_t)row * K + k0 + cc);
        }
        if constexpr (XF) {
            if (!xa.acc) {
                xsc = *reinterpret_cast<const f32x4 *>(xf_stat + 2 * (size_t)K + k0 + cc);
                xsh = *reinterpret_cast<const f32x4 *>(xf_stat + 3 * (size_t)K + k0 + cc);
            }
        }
        xf_k = k0 + cc;                                        // (accumulator form: scale / shift come from LDS at staging time)
    };
    long long xf_count = M;
    if constexpr (XF) {
        if (n_valid) xf_count = min((long long)*n_valid, M);
    }
    // BW: per-column norm constants of the 4 columns this lane stores (col = j0 + ct*32 + (lane & 7)*4) and its sums
    float4 bw_mean[CT], bw_inv[CT], bw_w[CT], bw_b[CT];
    float bw_sg[CT][4], bw_sgx[CT][4];
    long long bw_count = M;
    if constexpr (BW) {
        if (n_valid) bw_count = min((long long)*n_valid, M);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int col = min(j0 + ct * 32 + (lane & 7) * 4, max(Nout - 4, 0));
            bw_mean[ct] = *reinterpret_cast<const float4 *>(bw_stat + col);
            bw_inv[ct] = *reinterpret_cast<const float4 *>(bw_stat + (size_t)Nout + col);
            bw_w[ct] = *reinterpret_cast<const float4 *>(bw_stat + 4 * (size_t)Nout + col);
            bw_b[ct] = *reinterpret_cast<const float4 *>(bw_stat + 5 * (size_t)Nout + col);
#pragma unroll
            for (int q = 0; q < 4; ++q) bw_sg[ct][q] = bw_sgx[ct][q] = 0.f;
        }
    }
    if (slab < num_slabs) load_half(slab, 0);
    // XF from an accumulator, part 1: the totals and the affine parameters of column `tid` go in flight right behind the x slab
    // (K <= 256 <= THREADS: one column per thread); they are consumed after the operand panel has been issued too
    double xa_S = 0.0, xa_Q = 0.0;
    float xa_w = 1.f, xa_b = 0.f;
    if constexpr (XF) {
        if (xa.acc && tid < K) {
            xa_S = stat_acc_total(xa.acc, K, 0, tid);
            xa_Q = stat_acc_total(xa.acc, K, 1, tid);
            const int comp = tid / xa.p.k, kk = tid - comp * xa.p.k;
            if (xa.p.has_weight) xa_w = *bn_col_ptr(xa.p.weight, xa.p.flat, comp, kk, tid);
            if (xa.p.has_bias) xa_b = *bn_col_ptr(xa.p.bias, xa.p.flat, comp, kk, tid);
        }
    }

    // bias columns this lane will store (fixed across slabs): col = j0 + ct*32 + (lane & 7)*4
    float4 bias4[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int col = min(j0 + ct * 32 + (lane & 7) * 4, max(Nout - 4, 0));
        bias4[ct] = (bias && out_vec4) ? *reinterpret_cast<const float4 *>(bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // STATS: running (count, sum, M2) of this wavefront's rows for the column this lane owns in the MFMA layout
    float st_bias[CT], st_sum[CT], st_m2[CT];
    float st_n = 0.f;
    long long st_count = M;
    if constexpr (ST) {
        if (n_valid) st_count = min((long long)*n_valid, M);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int col = j0 + ct * 32 + r;
            st_bias[ct] = (bias && col < Nout) ? bias[col] : 0.f;
            st_sum[ct] = st_m2[ct] = 0.f;
        }
    }

#ifndef PHC_EXP_NO_HBUILD
    if constexpr (NPHM < 0) {
        // prebuilt operand (phc_phm_build_H_batched): W points at a row-major [K, Nout] matrix; the panel is one
        // round of coalesced loads instead of the A -> W -> assemble chain (3.6 us of a 13 us launch at molecule scale)
        const int K8 = round_up(K, 8);
        if ((Nout & 3) == 0) {
            // four clamped (never branched) 16-byte loads per thread in flight, then the LDS writes: a conditional load
            // is a branch with a vmcnt(0) at its join, i.e. one L2 round trip per iteration
            constexpr int C4 = BN / 4;
            auto panel_chunk = [&](int e0) {
                f32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = e0 + tid + u * THREADS;
                    const int k = e / C4, c4 = e - k * C4;
                    v[u] = *reinterpret_cast<const f32x4 *>(W + (size_t)min(k, K - 1) * Nout + min(j0 + c4 * 4, Nout - 4));
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = e0 + tid + u * THREADS;
                    const int k = e / C4, c4 = e - k * C4;
                    if (e < K8 * C4) {
                        const bool ok = k < K && j0 + c4 * 4 < Nout;
                        float *d = hs + k * HS_STRIDE + c4 * 4;
                        d[0] = ok ? v[u][0] : 0.f; d[1] = ok ? v[u][1] : 0.f; d[2] = ok ? v[u][2] : 0.f; d[3] = ok ? v[u][3] : 0.f;
                    }
                }
            };
            // the first chunk (all of a K <= 128, 32-column panel) is straight-line code: at a loop header hipcc drains
            // every load in flight (vmcnt(0)), which put the x slab's round trip in front of the panel's instead of beside it
            panel_chunk(0);
            for (int e0 = 4 * THREADS; e0 < K8 * C4; e0 += 4 * THREADS) panel_chunk(e0);
        } else {
            for (int e = tid; e < K8 * BN; e += THREADS) {
                const int k = e / BN, jj = e - k * BN;
                hs[k * HS_STRIDE + jj] = (k < K && j0 + jj < Nout) ? W[(size_t)k * Nout + j0 + jj] : 0.f;
            }
        }
    } else {
        for (int i = tid; i < n * n * n; i += THREADS) As[i] = A[i];
        __syncthreads();
        build_h_chunk<TRANS, (NPHM < 0 ? 0 : NPHM), BN, THREADS>(As, W, hs, n, kin, kout, 0, K, round_up(K, 8), j0, Nout, tid);
    }
#endif
    if constexpr (XF) {
        if (xa.acc) {
            // part 2: scale / shift of the K operand columns into LDS (read after the barrier below); workgroup (0, 0) also leaves
            // save_stat and the running statistics behind
            const bool owner = pbx == 0 && pby == 0;
            const int c = tid;
            if (c < K) {
                const float w = xa_w, bq = xa_b;
                const BnFromTotals bt = bn_from_totals(xa_S, xa_Q, xf_count, w, bq, xa.eps);
                xf_lds[c] = bt.scale;
                xf_lds[K + c] = bt.shift;
                if (owner) {
                    float *st = xa.stat_out;
                    st[c] = bt.mean; st[K + c] = bt.invstd; st[2 * K + c] = bt.scale; st[3 * K + c] = bt.shift;
                    st[4 * K + c] = w; st[5 * K + c] = bq;
                    if (xa.p.has_running) {
                        const int comp = c / xa.p.k, kk = c - comp * xa.p.k;
                        float *rm = bn_col_ptr(xa.p.running_mean, xa.p.flat, comp, kk, c), *rv = bn_col_ptr(xa.p.running_var, xa.p.flat, comp, kk, c);
                        *rm = bn_running_update(*rm, xa.momentum, bt.mean);
                        *rv = bn_running_update(*rv, xa.momentum, bt.unbiased);
                    }
                }
            }
            if (owner && xa.p.has_nbt && tid < xa.p.n_comp)
                for (int q = 0; q < xa.p.n_comp; ++q)
                    if (q == tid) *xa.p.num_batches_tracked[q] += 1;
        }
    }
    __syncthreads();
    ST_STAMP(1);

    for (; slab < num_slabs; slab += slab_step) {
        f32x16 acc[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[ct][i] = 0.f;
        // BW: the norm-input rows the epilogue needs go in flight now (clamped, unconditional), under the MFMAs
        f32x4 bwx[BW ? CT : 1][4];
        if constexpr (BW) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int idx = lane + 64 * t;
                    const long long grow = min(slab * 32 + (idx >> 3), M - 1);
                    const int col = min(j0 + ct * 32 + (idx & 7) * 4, max(Nout - 4, 0));
                    bwx[ct][t] = *reinterpret_cast<const f32x4 *>(bw_x + (size_t)grow * Nout + col);
                }
        }

        for (int half = 0; half < halves; ++half) {
            const int k0 = half * ST_KH;
            const int kc = min(ST_KH, K - k0);
            const int kc8 = round_up(kc, 8);
            // registers -> private LDS (k padding and the clamped column read as zero)
            const bool colok = lc4 * 4 < kc;
            if constexpr (XF) {
                if (xa.acc) {                                  // written in the prologue, in front of the panel's barrier
                    xsc = *reinterpret_cast<const f32x4 *>(xf_lds + xf_k);
                    xsh = *reinterpret_cast<const f32x4 *>(xf_lds + K + xf_k);
                }
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                f32x4 v = xr[t];
                if constexpr (XF) {
                    const bool live = slab * 32 + lrow + 4 * t < xf_count;
                    if (xf_act == PHC_ACT_RELU) {            // the common case stays branch-free: fma, max, select
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = live ? fmaxf(fmaf(v[q], xsc[q], xsh[q]), 0.f) : 0.f;
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = live ? act_fwd_rt(xf_act, fmaf(v[q], xsc[q], xsh[q])) : 0.f;
                    }
                }
                *reinterpret_cast<f32x4 *>(stage + (lrow + 4 * t) * ST_STRIDE + lc4 * 4) = colok ? v : z;
            }
            // next half (or the next slab's first half) goes in flight under the MFMAs
#ifndef PHC_EXP_NO_XLOAD
            if (half + 1 < halves) load_half(slab, half + 1);
            else if (slab + slab_step < num_slabs) load_half(slab + slab_step, 0);
#endif

            ST_STAMP(2 + 2 * half);
            const float *xrow = stage + r * ST_STRIDE + 4 * hh;
            const float *hcol = hs + (k0 + 4 * hh) * HS_STRIDE + r;
#ifdef PHC_EXP_NO_MFMA
            if (tid == 100000)
#endif
#pragma unroll 2
            for (int kg = 0; kg < kc8; kg += 8) {
                const float4 av = *reinterpret_cast<const float4 *>(xrow + kg);
                const float a4[4] = {av.x, av.y, av.z, av.w};
                const float *hb = hcol + kg * HS_STRIDE;
                float bq[4][CT];
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) bq[q][ct] = hb[q * HS_STRIDE + ct * 32];
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[q], bq[q][ct], acc[ct], 0, 0, 0);
            }
        }

        ST_STAMP(6);
        const long long row_base = slab * 32;
        if constexpr (ST) {
            // statistics of what is STORED: a residual (PHMConv's self loop behind the transform) joins the accumulators in front
            // of them, in the MFMA layout (register reg = row (reg & 3) + 8 (reg >> 2) + 4 hh, lane = column r: 128-byte runs per row)
            if (residual) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const int col = j0 + ct * 32 + r;
                    float rv[16];
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const long long grow = min(row_base + (reg & 3) + 8 * (reg >> 2) + 4 * hh, M - 1);
                        rv[reg] = residual[(size_t)grow * Nout + min(col, Nout - 1)];
                    }
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) acc[ct][reg] += rv[reg];
                }
            }
            // rows of this slab that count: [row_base, min(row_base + 32, st_count))
            const int nv = (int)max(0LL, min(32LL, st_count - row_base));
            if (nv > 0) {                                                    // wave-uniform
                const float cnt = (float)nv;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    float sv = 0.f;
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int rw = (reg & 3) + 8 * (reg >> 2) + 4 * hh;
                        sv += rw < nv ? acc[ct][reg] + st_bias[ct] : 0.f;
                    }
                    sv += __shfl_xor(sv, 32);
                    const float mean_s = sv / cnt;
                    float m2 = 0.f;
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int rw = (reg & 3) + 8 * (reg >> 2) + 4 * hh;
                        const float d = acc[ct][reg] + st_bias[ct] - mean_s;
                        m2 = rw < nv ? fmaf(d, d, m2) : m2;
                    }
                    m2 += __shfl_xor(m2, 32);
                    if (st_n == 0.f) {
                        st_sum[ct] = sv;
                        st_m2[ct] = m2;
                    } else {                                                 // Chan merge of (st_n, st_sum, st_m2) and the slab
                        const float delta = mean_s - st_sum[ct] / st_n;
                        st_m2[ct] += m2 + delta * delta * (st_n * cnt / (st_n + cnt));
                        st_sum[ct] += sv;
                    }
                }
                st_n += cnt;
            }
        }
        // epilogue: transpose each 32x32 tile through the private buffer, 16-byte stores
#ifdef PHC_EXP_NO_EPI
        if (acc[0][0] == 123.456f)
#endif
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int col0 = j0 + ct * 32;
            if (out_vec4) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg)
                    stage[((reg & 3) + 8 * (reg >> 2) + 4 * hh) * 36 + r] = acc[ct][reg];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int idx = lane + 64 * t;
                    const int row = idx >> 3, c4 = idx & 7;
                    const int col = col0 + c4 * 4;
                    const long long grow = row_base + row;
                    float4 v = *reinterpret_cast<const float4 *>(stage + row * 36 + c4 * 4);
                    if (grow < M && col < Nout) {
                        v.x += bias4[ct].x; v.y += bias4[ct].y; v.z += bias4[ct].z; v.w += bias4[ct].w;
                        if (!ST && residual) {
                            const float4 r4 = *reinterpret_cast<const float4 *>(residual + (size_t)grow * Nout + col);
                            v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
                        }
                        if constexpr (BW) {
                            if (grow < bw_count) {
                                const f32x4 x4 = bwx[ct][t];
                                const float xs_[4] = {x4[0], x4[1], x4[2], x4[3]}, vs_[4] = {v.x, v.y, v.z, v.w};
                                const float mm[4] = {bw_mean[ct].x, bw_mean[ct].y, bw_mean[ct].z, bw_mean[ct].w};
                                const float ii[4] = {bw_inv[ct].x, bw_inv[ct].y, bw_inv[ct].z, bw_inv[ct].w};
                                const float ww[4] = {bw_w[ct].x, bw_w[ct].y, bw_w[ct].z, bw_w[ct].w};
                                const float bb[4] = {bw_b[ct].x, bw_b[ct].y, bw_b[ct].z, bw_b[ct].w};
                                // (the ReLU test outside the element loop: inside, the run-time activation switch became a
                                // chain of scalar branches per element)
                                if (bw_act == PHC_ACT_RELU) {
#pragma unroll
                                    for (int q = 0; q < 4; ++q) {
                                        const float xh = (xs_[q] - mm[q]) * ii[q];
                                        const float gq = fmaf(xh, ww[q], bb[q]) > 0.f ? vs_[q] : 0.f;
                                        bw_sg[ct][q] += gq;
                                        bw_sgx[ct][q] = fmaf(gq, xh, bw_sgx[ct][q]);
                                    }
                                } else {
#pragma unroll
                                    for (int q = 0; q < 4; ++q) {
                                        const float xh = (xs_[q] - mm[q]) * ii[q];
                                        const float gq = vs_[q] * act_grad_rt(bw_act, fmaf(xh, ww[q], bb[q]));
                                        bw_sg[ct][q] += gq;
                                        bw_sgx[ct][q] = fmaf(gq, xh, bw_sgx[ct][q]);
                                    }
                                }
                            }
                        }
                        *reinterpret_cast<float4 *>(out + (size_t)grow * Nout + col) = v;
                    }
                }
            } else {
                const int col = col0 + r;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const long long grow = row_base + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
                    if (grow < M && col < Nout) {
                        float v = acc[ct][reg];
                        if (bias) v += bias[col];
                        if (!ST && residual) v += residual[(size_t)grow * Nout + col];
                        out[(size_t)grow * Nout + col] = v;
                    }
                }
            }
        }
    }
    ST_STAMP(7);
    if constexpr (BW) {
        // lanes sharing (lane & 7) hold the same 4 columns for different rows: fold them, then the WAVES wavefronts in order
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float a = bw_sg[ct][q], b2 = bw_sgx[ct][q];
#pragma unroll
                for (int off = 8; off < 64; off <<= 1) {
                    a += __shfl_xor(a, off);
                    b2 += __shfl_xor(b2, off);
                }
                bw_sg[ct][q] = a;
                bw_sgx[ct][q] = b2;
            }
        __syncthreads();                                                     // every wave is done with its staging buffer
        float *sw = stage_all + wave * ST_FLOATS;                            // [ct*32 + c] = sum g, [BN + ct*32 + c] = sum g*x_hat
        if (lane < 8) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    sw[ct * 32 + lane * 4 + q] = bw_sg[ct][q];
                    sw[BN + ct * 32 + lane * 4 + q] = bw_sgx[ct][q];
                }
        }
        __syncthreads();
        for (int cc = tid; cc < BN; cc += 64 * WAVES) {
            float a = 0.f, b2 = 0.f;
            for (int w = 0; w < WAVES; ++w) {
                a += stage_all[w * ST_FLOATS + cc];
                b2 += stage_all[w * ST_FLOATS + BN + cc];
            }
            const int col = j0 + cc;
            if (col < Nout) {
                if (stats_acc) {
                    unsigned long long *acc64 = reinterpret_cast<unsigned long long *>(stats_partial);
                    stat_acc_add(acc64, Nout, (int)(pby % STAT_ACC_REP), 0, col, (double)a);
                    stat_acc_add(acc64, Nout, (int)(pby % STAT_ACC_REP), 1, col, (double)b2);
                } else {
                    stats_partial[(size_t)pby * 2 * Nout + col] = a;
                    stats_partial[(size_t)pby * 2 * Nout + Nout + col] = b2;
                }
            }
        }
    }
    if constexpr (ST) {
        // merge the WAVES wavefronts of this workgroup in wave order (fixed) and publish one partial per column
        __syncthreads();                                                     // every wave is done with its staging buffer
        float *sw = stage_all + wave * ST_FLOATS;                            // [0] = n, [1 + ct*32 + r] = sum, [1 + CT*32 + ...] = M2
        if (hh == 0) {
            if (r == 0) sw[0] = st_n;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                sw[1 + ct * 32 + r] = st_sum[ct];
                sw[1 + CT * 32 + ct * 32 + r] = st_m2[ct];
            }
        }
        __syncthreads();
        for (int cc = tid; cc < BN; cc += 64 * WAVES) {
            float nt = 0.f, sum = 0.f, m2 = 0.f;
            for (int w = 0; w < WAVES; ++w) {
                const float *q = stage_all + w * ST_FLOATS;
                const float nw = q[0];
                if (nw > 0.f) {
                    const float sw_ = q[1 + cc], mw = q[1 + CT * 32 + cc];
                    if (nt == 0.f) {
                        sum = sw_;
                        m2 = mw;
                    } else {
                        const float delta = sw_ / nw - sum / nt;
                        m2 += mw + delta * delta * (nt * nw / (nt + nw));
                        sum += sw_;
                    }
                    nt += nw;
                }
            }
            const int col = j0 + cc;
            if (col < Nout) {
                if (stats_acc) {
                    // additive form of the workgroup's (sum, M2 about its own mean): sum and sum of squares, in double
                    unsigned long long *acc64 = reinterpret_cast<unsigned long long *>(stats_partial);
                    const double sd = (double)sum, qd = (double)m2 + (nt > 0.f ? sd * sd / (double)nt : 0.0);
                    stat_acc_add(acc64, Nout, (int)(pby % STAT_ACC_REP), 0, col, sd);
                    stat_acc_add(acc64, Nout, (int)(pby % STAT_ACC_REP), 1, col, qd);
                } else {
                    stats_partial[(size_t)pby * 2 * Nout + col] = sum;
                    stats_partial[(size_t)pby * 2 * Nout + Nout + col] = m2;
                }
            }
            if (!stats_acc && pbx == 0 && cc == 0) stats_counts[pby] = nt;
        }
    }
    ST_STAMP(8);
}

// H[k, j] for the whole matrix (exported helper; the hot kernels do not use it).
__global__ void phm_build_H_kernel(const float *__restrict__ A, const float *__restrict__ W, float *__restrict__ H, int n,
                                   int kin, int kout) {
    extern __shared__ float As[];
    for (int i = threadIdx.x; i < n * n * n; i += blockDim.x) As[i] = A[i];
    __syncthreads();
    const int in = n * kin, out = n * kout;
    const long long total = (long long)in * out;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(e / out), j = (int)(e - (long long)k * out);
        H[e] = h_element<0>(As, W, n, kin, kout, k, j);
    }
}

// H and H^T of MANY layers in one launch (phc_phm_build_H_batched): the operand matrices of a whole
// training step are assembled once, right after the weights changed, and every GEMM of the step
// loads its panel from them.
constexpr int H_JOBS_PER_LAUNCH = 32;
struct HJobDev {
    const float *A, *W;
    float *H, *HT;
    float *Hk4, *HTk4;           // k4-interleaved copies for the H-stationary GEMM (gemm_hs_device.h), or NULL
    int n, kin, kout;
    int dense_t;                 // n == 1 and W is an nn.Linear weight [kout, kin]: H = W^T (H^T is W itself); 2: zero fill; 3: gate map
    unsigned blk0;
    // dense_t == 3 (the read-out's gate as ONE linear map, pooling.py:57-66): H := M = H(A, W) . gate_w^T [in, kg], HT := M^T,
    // gate_c[j] = sum_col phm_b[col] gate_w[j, col] + gate_b[j]
    const float *gate_w, *gate_b, *phm_b;
    float *gate_c;
    int kg;
};
struct HBatch {
    int num;
    HJobDev j[H_JOBS_PER_LAUNCH];
};

// Gate-map job, one workgroup per row r of M (r == in: the bias row): the row of H(A, W) goes through LDS, then every wavefront
// takes gate columns j = wave, wave + 4, ... and sums over the row in a fixed order (lane-strided partials, xor tree).
__device__ __forceinline__ void gate_compose_block(const HJobDev &jb, int r) {
    __shared__ float hrow[1024];
    const int n = jb.n, kin = jb.kin, kout = jb.kout, in = n * kin, out = n * kout, kg = jb.kg;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int a = r / kin, kap = r - a * kin;
    for (int col = tid; col < out; col += 256) {
        float v;
        if (r == in) {
            v = jb.phm_b ? jb.phm_b[col] : 0.f;
        } else {
            const int c = col / kout, pi = col - c * kout;
            v = 0.f;
            for (int i = 0; i < n; ++i) v = fmaf(jb.A[(i * n + a) * n + c], jb.W[((size_t)i * kin + kap) * kout + pi], v);
        }
        hrow[col] = v;
    }
    __syncthreads();
    // eight gate columns of this wavefront per round, their gate_w words all in flight before the first is used (one at a time, a
    // round trip per column stood in the launch's critical path)
    constexpr int JR = 8;
    for (int j0 = wave; j0 < kg; j0 += 4 * JR) {
        float p[JR];
#pragma unroll
        for (int q = 0; q < JR; ++q) p[q] = 0.f;
        for (int col = lane; col < out; col += 64) {
            float gw[JR];
#pragma unroll
            for (int q = 0; q < JR; ++q) gw[q] = jb.gate_w[(size_t)min(j0 + 4 * q, kg - 1) * out + col];
            const float hv = hrow[col];
#pragma unroll
            for (int q = 0; q < JR; ++q) p[q] = fmaf(hv, gw[q], p[q]);
        }
#pragma unroll
        for (int q = 0; q < JR; ++q) {
            float v = p[q];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
            const int j = j0 + 4 * q;
            if (lane == 0 && j < kg) {
                if (r == in) {
                    jb.gate_c[j] = v + (jb.gate_b ? jb.gate_b[j] : 0.f);
                } else {
                    jb.H[(size_t)r * kg + j] = v;
                    if (jb.HT) jb.HT[(size_t)j * in + r] = v;
                }
            }
        }
    }
}

// The atom encoder of the step (encoder.py:31-34: out[i, c k + kap] = sum_f T_{c,f}[idx[i, f]][kap]) as PASSENGER workgroups of the
// operand-build launch: both depend on nothing but the parameters and the batch, and both are latency-bound — side by side they
// take what the longer one takes (one launch and ~8 us per step less than back to back).  One thread per (row, float4 column
// group) and round; the F indices of a row are read by every thread of the row (one line, broadcast), an out-of-range index reads
// row 0 and raises PHC_IDX_ERR_ATOM exactly as phc_encode_fwd does.
constexpr int HB_ENC_MAX_F = 10;
struct EncFront {
    const long long *idx;            // [N, F]
    const float *const *tables;      // device array of n * F table pointers ([c][f] -> [V_f, k])
    float *out;                      // [N, n k]
    long long N;
    int F, n, k;
    int vocab[HB_ENC_MAX_F];
    unsigned blk0, blocks;           // its workgroups sit behind the operand jobs'
    int *err;
};
__device__ __forceinline__ void encode_front_block(const EncFront &e, unsigned bx) {
    const int h4 = (e.n * e.k) >> 2;
    const long long total = e.N * h4;
    for (long long t = (long long)bx * 256 + threadIdx.x; t < total; t += (long long)e.blocks * 256) {
        const long long i = t / h4;
        const int c4 = (int)(t - i * h4), col = c4 << 2;
        const int c = col / e.k, kap = col - c * e.k;
        long long iv[HB_ENC_MAX_F];
#pragma unroll
        for (int f = 0; f < HB_ENC_MAX_F; ++f) {
            const int ff = min(f, e.F - 1);
            iv[f] = checked_index(e.idx[i * e.F + ff], e.vocab[ff], e.err, PHC_IDX_ERR_ATOM);
        }
        float4 rows[HB_ENC_MAX_F];
#pragma unroll
        for (int f = 0; f < HB_ENC_MAX_F; ++f)
            rows[f] = *reinterpret_cast<const float4 *>(e.tables[c * e.F + min(f, e.F - 1)] + iv[f] * e.k + kap);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int f = 0; f < HB_ENC_MAX_F; ++f)
            if (f < e.F) { acc.x += rows[f].x; acc.y += rows[f].y; acc.z += rows[f].z; acc.w += rows[f].w; }
        reinterpret_cast<float4 *>(e.out + (size_t)i * e.n * e.k)[c4] = acc;
    }
}

// A PHM job in 32 x 32 tiles (round 4; in, out, kout multiples of 4 and 16-byte aligned outputs): a thread computes four consecutive
// columns of one row (the n terms' W values as ONE float4 each, the A values shared by the four columns), the tile meets in LDS and
// all four forms leave as 16-byte stores in runs of 128-512 bytes.  One thread per element — the form this replaces and the fallback —
// wrote H^T, the k4 form and its transpose as scattered 4-byte stores: 46 us for the twelve 256 x 256 operands of the ZINC model.
// Same sums in the same order (ascending i).
__device__ __forceinline__ void phm_build_tile(const HJobDev &jb, int tile) {
    __shared__ float t_[32][33];
    const int n = jb.n, kin = jb.kin, kout = jb.kout, in = n * kin, out = n * kout;
    const int tj = (out + 31) / 32;
    const int k0 = (tile / tj) * 32, j0 = (tile % tj) * 32;
    const int tid = threadIdx.x;
    {
        const int kk = tid >> 3, jq = (tid & 7) * 4;
        const int k = min(k0 + kk, in - 1), jc = min(j0 + jq, out - 4);
        const int a = k / kin, kap = k - a * kin;
        const int c = jc / kout, pi = jc - c * kout;           // (kout % 4 == 0: the four columns share the component)
        float4 hs = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i0 = 0; i0 < n; i0 += 4) {                     // four terms per round trip (clamped, selected; ascending i)
            float av[4];
            float4 wv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = min(i0 + q, n - 1);
                av[q] = jb.A[(i * n + a) * n + c];
                wv[q] = *reinterpret_cast<const float4 *>(jb.W + ((size_t)i * kin + kap) * kout + pi);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (i0 + q < n) {
                    hs.x = fmaf(av[q], wv[q].x, hs.x); hs.y = fmaf(av[q], wv[q].y, hs.y);
                    hs.z = fmaf(av[q], wv[q].z, hs.z); hs.w = fmaf(av[q], wv[q].w, hs.w);
                }
        }
        t_[kk][jq] = hs.x; t_[kk][jq + 1] = hs.y; t_[kk][jq + 2] = hs.z; t_[kk][jq + 3] = hs.w;
    }
    __syncthreads();
    const int r = tid >> 3, q4 = (tid & 7) * 4;               // a row of the tile and a quad of its columns (or the transpose)
    // H: row k0 + r, columns j0 + q4 .. + 3
    if (k0 + r < in && j0 + q4 < out)
        *reinterpret_cast<float4 *>(jb.H + (size_t)(k0 + r) * out + j0 + q4) = make_float4(t_[r][q4], t_[r][q4 + 1], t_[r][q4 + 2], t_[r][q4 + 3]);
    // H^T: row j0 + r, columns k0 + q4 .. + 3
    if (jb.HT && j0 + r < out && k0 + q4 < in)
        *reinterpret_cast<float4 *>(jb.HT + (size_t)(j0 + r) * in + k0 + q4) = make_float4(t_[q4][r], t_[q4 + 1][r], t_[q4 + 2][r], t_[q4 + 3][r]);
    // k4 form: float4 slot (k / 4) * out + jc = H[4 (k / 4) .. + 3][jc]: k-quad tid / 32, column tid % 32
    {
        const int kq_ = tid >> 5, jj = tid & 31;
        if (jb.Hk4 && k0 + 4 * kq_ < in && j0 + jj < out)
            *reinterpret_cast<float4 *>(jb.Hk4 + ((size_t)((k0 >> 2) + kq_) * out + j0 + jj) * 4) =
                make_float4(t_[4 * kq_][jj], t_[4 * kq_ + 1][jj], t_[4 * kq_ + 2][jj], t_[4 * kq_ + 3][jj]);
        // transpose's k4 form: float4 slot (jc / 4) * in + k = H[k][4 (jc / 4) .. + 3]: column quad tid / 32, row tid % 32
        if (jb.HTk4 && j0 + 4 * kq_ < out && k0 + jj < in)
            *reinterpret_cast<float4 *>(jb.HTk4 + ((size_t)((j0 >> 2) + kq_) * in + k0 + jj) * 4) =
                make_float4(t_[jj][4 * kq_], t_[jj][4 * kq_ + 1], t_[jj][4 * kq_ + 2], t_[jj][4 * kq_ + 3]);
    }
}

__global__ __launch_bounds__(256) void phm_build_H_batched_kernel(HBatch b, EncFront enc) {
    if (enc.blocks && blockIdx.x >= enc.blk0) {              // (block-uniform)
        encode_front_block(enc, blockIdx.x - enc.blk0);
        return;
    }
    int job = 0;
    for (int q = 1; q < b.num; ++q)
        if (blockIdx.x >= b.j[q].blk0) job = q;
    const HJobDev &jb = b.j[job];
    const int n = jb.n, kin = jb.kin, kout = jb.kout;
    const int in = n * kin, out = n * kout;
    if (jb.dense_t == 3) {                                    // (block-uniform)
        gate_compose_block(jb, (int)(blockIdx.x - jb.blk0));
        return;
    }
    if (jb.dense_t == 4) {                                    // a PHM job in 32 x 32 tiles
        phm_build_tile(jb, (int)(blockIdx.x - jb.blk0));
        return;
    }
    const int e = (blockIdx.x - jb.blk0) * 256 + threadIdx.x;
    if (e >= in * out) return;
    const int k = e / out, jc = e - k * out;
    const int a = k / kin, kap = k - a * kin;
    const int c = jc / kout, pi = jc - c * kout;
    if (jb.dense_t == 2) {                                   // zero-fill job
        jb.H[e] = 0.f;
        return;
    }
    if (jb.dense_t) {
        jb.H[e] = jb.W[(size_t)jc * kin + k];
        return;
    }
    float hsum = 0.f;
    for (int i0 = 0; i0 < n; i0 += 8) {                      // the n terms eight per round trip (clamped, selected; ascending i)
        float av[8], wv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int i = min(i0 + q, n - 1);
            av[q] = jb.A[(i * n + a) * n + c];
            wv[q] = jb.W[((size_t)i * kin + kap) * kout + pi];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) hsum = i0 + q < n ? fmaf(av[q], wv[q], hsum) : hsum;
    }
    jb.H[e] = hsum;
    if (jb.HT) jb.HT[(size_t)jc * in + k] = hsum;
    // k4 form of a [Kd, Nd] operand: float4 slot (kd / 4) * Nd + jd holds rows 4 (kd / 4) .. +3 of column jd
    if (jb.Hk4) jb.Hk4[((size_t)(k >> 2) * out + jc) * 4 + (k & 3)] = hsum;
    if (jb.HTk4) jb.HTk4[((size_t)(jc >> 2) * in + k) * 4 + (jc & 3)] = hsum;
}

__global__ __launch_bounds__(256) void phm_dH_partial_kernel(const float *__restrict__ X, const float *__restrict__ DY,
                                                            float *__restrict__ slabs, long long N, int in, int out,
                                                            int tiles_j, long long rows_per_split, int vec4) {
    __shared__ __attribute__((aligned(16))) float smem[2 * DH_RB * DH_STRIDE];
    dH_partial_body(X, DY, slabs, N, in, out, tiles_j, rows_per_split, vec4, blockIdx.x, blockIdx.y, smem);
}

__global__ __launch_bounds__(256) void phm_dH_reduce_kernel(const float *__restrict__ slabs, float *__restrict__ dH, size_t slab_elems,
                                                           int splits) {
    dH_reduce_body(slabs, dH, slab_elems, splits, blockIdx.x);
}

// blocks [0, n^3): dA[i,a,c] = sum_{kap,pi} W[i,kap,pi] * dH[a*kin+kap, c*kout+pi]
// blocks [n^3, ..): dW[i,kap,pi] = sum_{a,c} A[i,a,c] * dH[a*kin+kap, c*kout+pi]; the last block also copies db.
__device__ __forceinline__ void contract_body(const float *__restrict__ A, const float *__restrict__ W,
                                              const float *__restrict__ dH, float *__restrict__ dW, float *__restrict__ dA,
                                              float *__restrict__ db, int n, int kin, int kout, int dA_blocks, int bx,
                                              int dw_transposed = 0, const float *__restrict__ reg_add = nullptr) {
    const int out = n * kout;
    const int tid = threadIdx.x;
    if (bx < dA_blocks) {
        if (!dA) return;
        const int iac = bx;
        const int i = iac / (n * n), a = (iac / n) % n, c = iac % n;
        float s = 0.f;
        // four (W, dH) pairs per thread in one round trip (clamped, selected; ascending order as the plain loop)
        const int kk = kin * kout;
        for (int e0 = 0; e0 < kk; e0 += 4 * 256) {
            float wv[4], hv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = min(e0 + tid + 256 * q, kk - 1);
                const int kap = e / kout, pi = e - kap * kout;
                wv[q] = W[(size_t)i * kk + e];
                hv[q] = dH[(size_t)(a * kin + kap) * out + c * kout + pi];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) s = e0 + tid + 256 * q < kk ? fmaf(wv[q], hv[q], s) : s;
        }
        __shared__ float red[256];
        red[tid] = s;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {           // fixed tree: reproducible
            if (tid < st) red[tid] += red[tid + st];
            __syncthreads();
        }
        if (tid == 0) dA[iac] = red[0];
        return;
    }
    const int blk = bx - dA_blocks;
    const int e = blk * 256 + tid;                       // index into dW [n, kin, kout]
    if (dW && e < n * kin * kout) {
        const int i = e / (kin * kout);
        const int rem = e - i * kin * kout;
        const int kap = rem / kout, pi = rem - kap * kout;
        float s = 0.f;
        // the n*n (a, c) terms sixteen at a time in one round trip (clamped, selected; same ascending (a, c) order): a
        // runtime-bounded double loop costs one memory latency per term
        const int nn = n * n;
        for (int t0 = 0; t0 < nn; t0 += 16) {
            float av[16], hv[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int t = min(t0 + q, nn - 1);
                const int a = t / n, c = t - a * n;
                av[q] = A[(i * n + a) * n + c];
                hv[q] = dH[(size_t)(a * kin + kap) * out + c * kout + pi];
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) s = t0 + q < nn ? fmaf(av[q], hv[q], s) : s;
        }
        if (reg_add) s += reg_add[e];                    // the weight regulariser's gradient (tail_reg_body, the launch in front)
        // dw_transposed: the parameter is an nn.Linear weight [kout, kin] (n = 1)
        dW[dw_transposed ? ((size_t)i * kout + pi) * kin + kap : (size_t)e] = s;
    }
    if (db && blk == 0)
        for (int j = tid; j < out; j += 256) db[j] = dH[(size_t)n * kin * out + j];
}

__global__ __launch_bounds__(256) void phm_contract_kernel(const float *__restrict__ A, const float *__restrict__ W,
                                                          const float *__restrict__ dH, float *__restrict__ dW,
                                                          float *__restrict__ dA, float *__restrict__ db, int n, int kin,
                                                          int kout, int dA_blocks) {
    contract_body(A, W, dH, dW, dA, db, n, kin, kout, dA_blocks, blockIdx.x);
}

__global__ __launch_bounds__(256) void phm_dH_partial128_kernel(const float *__restrict__ X, const float *__restrict__ DY,
                                                               float *__restrict__ slabs, long long N, int in, int out,
                                                               int tiles_j, long long rows_per_split, int vec4) {
    __shared__ __attribute__((aligned(16))) float smem[2 * DH_RB2 * DH_T2];
    dH_partial_body128(X, DY, slabs, N, in, out, tiles_j, rows_per_split, vec4, blockIdx.x, blockIdx.y, smem);
}

// one launch per tile shape (a job of the other shape has blk0_partial = UINT_MAX and is never selected): the two
// bodies have very different register needs, and a merged kernel would run the 64 x 64 jobs at the 128 x 128 body's occupancy
template <int T128>
__global__ __launch_bounds__(256) void phm_grads_partial_kernel(GradBatch b) {
    int k = 0;
    for (int q = 0; q < b.num; ++q)
        if (b.j[q].blk0_partial != 0xffffffffu && blockIdx.x >= b.j[q].blk0_partial) k = q;
    const GradJobDev &j = b.j[k];
    const unsigned local = blockIdx.x - j.blk0_partial;
    int tile;
    long long split;
    xcd_tile_split(blockIdx.x, local, (unsigned)j.tiles, (unsigned)j.splits, tile, split);
    __shared__ __attribute__((aligned(16))) float smem[2 * DH_RB * DH_STRIDE];      // 32 KB, either tile shape
    static_assert(2 * DH_RB * DH_STRIDE == 2 * DH_RB2 * DH_T2, "both tile shapes stage 32 KB");
    const long long xf_count = j.xf_stat ? (j.xf_n_valid ? min((long long)*j.xf_n_valid, j.N) : j.N) : 0;
    if constexpr (T128)
        dH_partial_body128(j.x, j.dy, j.slabs, j.N, j.in, j.out, j.tiles_j, j.rows_per_split, j.vec4, tile, split, smem, j.xf_stat,
                           j.xf_act, xf_count);
    else
        dH_partial_body(j.x, j.dy, j.slabs, j.N, j.in, j.out, j.tiles_j, j.rows_per_split, j.vec4, tile, split, smem, j.xf_stat,
                        j.xf_act, xf_count);
}

// register-resident jobs (tile form 2): one workgroup per (tile, split), four wavefronts on four row quarters
__global__ __launch_bounds__(256) void phm_grads_direct_kernel(GradBatch b) {
    extern __shared__ __attribute__((aligned(16))) float direct_smem[];
    int k = 0;
    for (int q = 0; q < b.num; ++q)
        if (b.j[q].blk0_partial != 0xffffffffu && blockIdx.x >= b.j[q].blk0_partial) k = q;
    const GradJobDev &j = b.j[k];
    const unsigned local = blockIdx.x - j.blk0_partial;
    int tile;
    long long split;
    xcd_tile_split(blockIdx.x, local, (unsigned)j.tiles, (unsigned)j.splits, tile, split);
    const long long xf_count = j.xf_stat ? (j.xf_n_valid ? min((long long)*j.xf_n_valid, j.N) : j.N) : 0;
    dH_direct_body<4>(j.x, j.dy, j.slabs, j.N, j.in, j.out, j.tiles_j, j.rows_per_split, tile, split, direct_smem, j.xf_stat, j.xf_act,
                   xf_count);
}

__global__ __launch_bounds__(256) void phm_grads_reduce_kernel(GradBatch b) {
    const int k = grad_job_of_block<1>(b, blockIdx.x);
    const GradJobDev &j = b.j[k];
    dH_reduce_body(j.slabs, j.dH, (size_t)j.in * j.out + j.out, j.splits, blockIdx.x - j.blk0_reduce);
}

__global__ __launch_bounds__(256) void phm_grads_contract_kernel(GradBatch b) {
    const int k = grad_job_of_block<2>(b, blockIdx.x);
    const GradJobDev &j = b.j[k];
    contract_body(j.A, j.W, j.dH, j.dW, j.dA, j.db, j.n, j.kin, j.kout, j.dA_blocks, (int)(blockIdx.x - j.blk0_contract),
                  j.dw_transposed, j.reg_add);
}

// ---- the gate map's parameter gradients (pooling.py:57-66 as ONE linear map M = H(A, W) . gate_w^T) ----
// From dM^T = dg^T . x [kg, in] and dbg = sum dg [kg] (a dense parameter-gradient job's outputs):
//   dH[r, col]  = sum_j dM^T[j, r] gate_w[j, col]                 (never stored: re-evaluated where it is needed, kg MACs)
//   dA[i, a, c] = sum_{kap, pi} W[i, kap, pi] dH[a kin + kap, c kout + pi]          blocks [0, n^3)
//   dW[i, kap, pi] = sum_{a, c} A[i, a, c] dH[a kin + kap, c kout + pi]             blocks [n^3, n^3 + ceil(n kin kout / 256))
//   dWg[j, col] = sum_r dM^T[j, r] H[r, col] + dbg[j] b[col]                        then ceil(kg out / 256) blocks
//   db[col]     = sum_j dbg[j] gate_w[j, col]                                       last block
// every sum in a fixed order (thread-strided partials, LDS tree): bit-reproducible.
struct GateGradArgs {
    const float *dMT, *dbg, *H, *gate_w, *phm_b, *A, *W;
    float *dA, *dW, *db, *dWg;
    int n, k, kg;                 // kin == kout == k
    int blk_rows, blk_dWg, blk_db;
    // dM^T[j, r] = dMT[j * sj + r * sr]: (h, 1) for the nn.Linear layout [kg, in] of the stand-alone entry point, (1, kg) when the
    // launch reads the dense job's dH = x^T . dg [in, kg] where the reduce launch left it (dbg = its bias row)
    int sj, sr;
    const float *reg_add;         // added to dW as it is written (the weight regulariser's gradient), or NULL
    float *dbg_out;               // a copy of dbg for the gate bias's gradient, or NULL
};

// Operands staged in LDS (dynamic): tile blocks kg * k * 2 + k * (k + 1) floats, row blocks kg * (n k) + kg * n + n^3 + n * (n k) floats.
// Every phase issues its global loads together and sums in a fixed order; none of the sums depends on the grid's shape.
__device__ __forceinline__ void gate_param_grads_body(const GateGradArgs &g, float *gsm, float (*red)[256], int bx) {
    const int n = g.n, k = g.k, kg = g.kg, h = n * k;
    const int tid = threadIdx.x;
    if (bx < g.blk_rows) {
        // ---- tile (a, c): dH[a k .. , c k ..] = dM[a-rows, :] . gate_w[:, c-cols] in LDS, then dA[i, a, c] for every i ----
        const int a = bx / n, c = bx - a * n;
        float *sM = gsm, *sG = gsm + kg * k, *tile = gsm + 2 * kg * k;          // [kg][k], [kg][k], [k][k + 1]
        for (int e = tid; e < kg * k; e += 256) {
            const int j = e / k, t = e - j * k;
            sM[e] = g.dMT[(size_t)j * g.sj + (size_t)(a * k + t) * g.sr];
            sG[e] = g.gate_w[(size_t)j * h + c * k + t];
        }
        __syncthreads();
        for (int e = tid; e < k * k; e += 256) {
            const int kap = e / k, pi = e - kap * k;
            float v = 0.f;
#pragma unroll 8
            for (int j = 0; j < kg; ++j) v = fmaf(sM[j * k + kap], sG[j * k + pi], v);
            tile[kap * (k + 1) + pi] = v;
        }
        __syncthreads();
        if (!g.dA) return;
        for (int i0 = 0; i0 < n; i0 += 4) {                                       // four i per round: one reduction tree for the four
            float p[4] = {0.f, 0.f, 0.f, 0.f};
            for (int e = tid; e < k * k; e += 256) {
                const int kap = e / k, pi = e - kap * k;
                float wv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) wv[q] = g.W[((size_t)min(i0 + q, n - 1) * k + kap) * k + pi];
                const float tv = tile[kap * (k + 1) + pi];
#pragma unroll
                for (int q = 0; q < 4; ++q) p[q] = fmaf(wv[q], tv, p[q]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) red[q][tid] = p[q];
            __syncthreads();
            for (int st = 128; st >= 1; st >>= 1) {
                if (tid < st) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) red[q][tid] += red[q][tid + st];
                }
                __syncthreads();
            }
            if (tid < 4 && i0 + tid < n) g.dA[((i0 + tid) * n + a) * n + c] = red[tid][0];
            __syncthreads();
        }
        return;
    }
    if (bx < g.blk_dWg) {
        // ---- row kap of every tile: dW[i, kap, pi] = sum_{a, c} A[i, a, c] dH[a k + kap, c k + pi], gate_w whole in LDS ----
        if (!g.dW) return;
        const int kap = bx - g.blk_rows;
        float *sG = gsm, *sMk = gsm + kg * h, *sA = sMk + kg * n;                 // [kg][h], [kg][n], [n][n][n]
        for (int e = tid; e < kg * h; e += 256) sG[e] = g.gate_w[e];
        for (int e = tid; e < kg * n; e += 256) {
            const int j = e / n, a = e - j * n;
            sMk[e] = g.dMT[(size_t)j * g.sj + (size_t)(a * k + kap) * g.sr];
        }
        for (int e = tid; e < n * n * n; e += 256) sA[e] = g.A[e];
        float *sD = sA + n * n * n;                                               // [n][h]: dH[a k + kap, col] of this row
        __syncthreads();
        // stage 1: one column per thread, its n values in one pass over gate_w's column (each word read once, n FMAs on it; the
        // LDS reads of eight j go out together — one at a time the chain waited a round trip per term)
        for (int col = tid; col < h; col += 256) {
            for (int a0 = 0; a0 < n; a0 += 4) {
                float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
                for (int j = 0; j < kg; ++j) {
                    const float gw = sG[j * h + col];
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = fmaf(sMk[j * n + min(a0 + q, n - 1)], gw, v[q]);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (a0 + q < n) sD[(a0 + q) * h + col] = v[q];
            }
        }
        __syncthreads();
        // stage 2: dW[i, kap, pi] = sum_{a, c} A[i, a, c] dH[a k + kap, c k + pi]
        for (int o = tid; o < h; o += 256) {                                      // o = (i, pi)
            const int i = o / k, pi = o - i * k;
            float acc = 0.f;
            for (int a = 0; a < n; ++a)
#pragma unroll 4
                for (int c = 0; c < n; ++c) acc = fmaf(sA[(i * n + a) * n + c], sD[a * h + c * k + pi], acc);
            if (g.reg_add) acc += g.reg_add[((size_t)i * k + kap) * k + pi];
            g.dW[((size_t)i * k + kap) * k + pi] = acc;
        }
        return;
    }
    if (bx < g.blk_db) {
        // ---- dWg[j, col] = sum_r dM^T[j, r] H[r, col] + dbg[j] b[col]: one j and 32 columns per workgroup, r dealt over 8 thread rows ----
        if (!g.dWg) return;
        const int cb = h / 32 + (h % 32 ? 1 : 0);
        const int w = bx - g.blk_dWg, j = w / cb, col = (w - j * cb) * 32 + (tid & 31), part = tid >> 5;
        float p = 0.f;
        if (col < h) {
#pragma unroll 8
            for (int r = part; r < h; r += 8) p = fmaf(g.dMT[(size_t)j * g.sj + (size_t)r * g.sr], g.H[(size_t)r * h + col], p);
        }
        red[0][tid] = p;
        __syncthreads();
        if (part == 0 && col < h) {
            float v = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) v += red[0][q * 32 + (tid & 31)];
            if (g.phm_b) v = fmaf(g.dbg[j], g.phm_b[col], v);
            g.dWg[(size_t)j * h + col] = v;
        }
        return;
    }
    if (g.db)                                                 // db[col] = sum_j dbg[j] gate_w[j, col]
        for (int col = tid; col < h; col += 256) {
            float p = 0.f;
            for (int j = 0; j < kg; ++j) p = fmaf(g.dbg[j], g.gate_w[(size_t)j * h + col], p);
            g.db[col] = p;
        }
    if (g.dbg_out)
        for (int j = tid; j < kg; j += 256) g.dbg_out[j] = g.dbg[j];
}

__global__ __launch_bounds__(256) void gate_param_grads_kernel(GateGradArgs g) {
    extern __shared__ float gsm[];
    __shared__ float red[4][256];
    gate_param_grads_body(g, gsm, red, (int)blockIdx.x);
}

// The step tail's LAST launch: the contractions dH -> dW / dA / db of every job, the gate map's parameter gradients (reading the
// dense job's dH in place) and the weight regulariser's value as three families of workgroups of one launch — they all depend on
// the reduce launch only (three launches in a row before: 7 + 9.5 + 7 us).
__global__ __launch_bounds__(256) void tail_finish_kernel(GradBatch b, GateGradArgs g, TailReg r, unsigned nb_contract, unsigned nb_gate) {
    extern __shared__ float gsm[];
    __shared__ float red[4][256];
    const unsigned bx = blockIdx.x;
    if (bx < nb_contract) {
        const int k = grad_job_of_block<2>(b, bx);
        const GradJobDev &j = b.j[k];
        contract_body(j.A, j.W, j.dH, j.dW, j.dA, j.db, j.n, j.kin, j.kout, j.dA_blocks, (int)(bx - j.blk0_contract), j.dw_transposed,
                      j.reg_add);
        return;
    }
    if (bx < nb_contract + nb_gate) {
        gate_param_grads_body(g, gsm, red, (int)(bx - nb_contract));
        return;
    }
    tail_reg_finish(r);
}

// ---- host side ---------------------------------------------------------------------------
static int g_num_cu = 0;
static int num_cu() {
    if (g_num_cu == 0) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) g_num_cu = p.multiProcessorCount;
        if (g_num_cu <= 0) g_num_cu = 256;
    }
    return g_num_cu;
}

static int check_phm_dims(const char *who, int64_t N, int n, int kin, int kout) {
    PHC_REQUIRE(N >= 0, PHC_ERR_INVALID_ARG, "%s: N = %lld < 0", who, (long long)N);
    PHC_REQUIRE(n >= 1 && kin >= 1 && kout >= 1, PHC_ERR_INVALID_ARG, "%s: n, kin, kout must be >= 1 (got %d, %d, %d)", who,
                n, kin, kout);
    PHC_REQUIRE(n <= MAX_PHM_DIM, PHC_ERR_UNSUPPORTED, "%s: phm_dim %d > %d", who, n, MAX_PHM_DIM);
    PHC_REQUIRE((int64_t)n * kin < (1 << 20) && (int64_t)n * kout < (1 << 20), PHC_ERR_UNSUPPORTED,
                "%s: feature width too large", who);
    return PHC_OK;
}

template <int TRANS, int NPHM, int BM, int BN>
static int launch_gemm_cfg(const float *X, const float *A, const float *W, const float *bias, const float *residual,
                           float *out, int64_t M, int K, int Nout, int n, int kin, int kout, int wg_per_cu, hipStream_t s) {
    using Cfg = GemmCfg<BM, BN>;
    const int panels = (Nout + BN - 1) / BN;
    const int64_t mtiles = (M + BM - 1) / BM;
    int64_t gy = ((int64_t)wg_per_cu * num_cu() + panels - 1) / panels;   // fill the chip; further M tiles are strided in-kernel
    if (gy > mtiles) gy = mtiles;
    if (gy < 1) gy = 1;
    const int vec4 = (K % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
    const size_t lds = sizeof(float) * Cfg::lds_floats(n);
    PHC_REQUIRE(ensure_dynamic_lds(&phm_gemm_kernel<TRANS, NPHM, BM, BN>, 150 * 1024), PHC_ERR_LAUNCH,
                "phm_gemm_kernel: the device refused %d bytes of dynamic LDS", 150 * 1024);
    PHC_LAUNCH((phm_gemm_kernel<TRANS, NPHM, BM, BN>), dim3(panels, (unsigned)gy), dim3(Cfg::THREADS), lds, s, X, A, W,
                       bias, residual, out, (long long)M, K, Nout, n, kin, kout, vec4);
    return PHC_OK;
}


constexpr int GEMM_STATS_MAX_PARTIALS = 256;

template <int TRANS, int NPHM, int BN, int WAVES, int STATS = 0>
static int launch_stream_cfg(const float *X, const float *A, const float *W, const float *bias, const float *residual,
                             float *out, int64_t M, int K, int Nout, int n, int kin, int kout, int wg_per_cu, hipStream_t s,
                             const int *n_valid = nullptr, float *stats_partial = nullptr, float *stats_counts = nullptr,
                             int *num_partials = nullptr, const float *xf_stat = nullptr, int xf_act = 0,
                             const float *bw_x = nullptr, const float *bw_stat = nullptr, int bw_act = 0, int stats_acc = 0,
                             const XfAcc &xa = XfAcc{}) {
    const int panels = (Nout + BN - 1) / BN;
    const int64_t slabs = (M + 31) / 32;
    int64_t gy = ((int64_t)wg_per_cu * num_cu() + panels - 1) / panels;
    const int64_t max_gy = (slabs + WAVES - 1) / WAVES;
    if (gy > max_gy) gy = max_gy;
    if ((STATS & 5) && gy > GEMM_STATS_MAX_PARTIALS) gy = GEMM_STATS_MAX_PARTIALS;      // one statistics partial per workgroup row
    {
        // experiment: fewer, longer workgroup rows for the statistics-emitting launches (fewer partials to merge downstream)
        constexpr int cap = 0;
        if ((STATS & 1) && cap > 0 && gy > cap) gy = cap;
    }
    if (gy < 1) gy = 1;
    if (num_partials) *num_partials = (int)gy;
    const size_t lds = sizeof(float) * ((size_t)round_up(K, 8) * (BN + 1) + (size_t)WAVES * ST_FLOATS + (size_t)round_up(n * n * n, 4) +
                                        ((STATS & 2) ? 2 * (size_t)K : 0));
    const int out_vec4 = (Nout % 4 == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0) &&
                         (!bias || (reinterpret_cast<uintptr_t>(bias) & 15) == 0) &&
                         (!residual || (reinterpret_cast<uintptr_t>(residual) & 15) == 0);
    PHC_REQUIRE(ensure_dynamic_lds(&phm_stream_kernel<TRANS, NPHM, BN, WAVES, STATS>, 150 * 1024), PHC_ERR_LAUNCH,
                "phm_stream_kernel: the device refused %d bytes of dynamic LDS", 150 * 1024);
    PHC_LAUNCH((phm_stream_kernel<TRANS, NPHM, BN, WAVES, STATS>), dim3(panels, (unsigned)gy), dim3(64 * WAVES), lds, s, X, A, W,
                       bias, residual, out, (long long)M, K, Nout, n, kin, kout, out_vec4, n_valid, stats_partial, stats_counts,
                       xf_stat, xf_act, bw_x, bw_stat, bw_act, stats_acc, xa);
    return PHC_OK;
}


// ---- H-stationary streaming GEMM (gemm_hs_device.h): launcher ----------------------------------------------------------
// Shapes it serves: K in {128, 256}, Nout in {128, 256} (every hidden-width PHM layer of the BASELINE configs), 16-byte aligned
// dense rows, operand available in k4 form; statistics only in the accumulator form.  Everything else stays on phm_stream_kernel.
constexpr int HS_RING = 4;
// run-time switch (phc_gemm_hs_set_enabled): the load-time self check of the Python layer turns the kernel off when its results
// disagree with the LDS-panel kernel's (a compiler upgrade that breaks the inline-asm contract of gemm_hs_device.h)
static int g_hs_enabled = -1;                                 // -1: not decided yet (environment: PHC_GEMM_NO_HS)
static inline bool hs_enabled() {
    int v = __atomic_load_n(&g_hs_enabled, __ATOMIC_RELAXED);
    if (v < 0) {
        v = getenv("PHC_GEMM_NO_HS") ? 0 : 1;
        __atomic_store_n(&g_hs_enabled, v, __ATOMIC_RELAXED);
    }
    return v != 0;
}
// Where it is the faster kernel (A/B inside the replayed training steps, one MI355X): hidden 256 at 23-27 k rows -19..-20 % per
// launch (zinc_b1024 2.46 -> 2.31 ms, molpcba_b1024 3.00 -> 2.87 ms per step), 1 M rows x 256: 1205 -> 992 us (132 TFLOP/s);
// at K = Nout = 128 with ONE workgroup per CU the 32-column staged kernel was 0.2-0.4 us per launch ahead at 12.9 k rows (805
// sixteen-row steps on 256 CUs: 4 rounds for 3.1 rounds of work); with two workgroups per CU (launch_hs_cfg) the H-stationary
// kernel wins there too: molhiv_b512 0.694 / 0.692 / 0.697 -> 0.680 / 0.688 / 0.676 ms, molhiv_b32 (1 152 rows) 0.457 -> 0.425 ms,
// the softmax step 0.801 -> 0.796 ms (alternating runs on one box).  (The threshold knob went with round 6: 0 rows.)
// ... and the reference's PUBLISHED widths (benchmarks/README.md:45-91, run_script_zinc_phm4.sh: hidden 200 = 4 x 50 = 5 x 40 and
// 180 = 4 x 45): square 180- / 200-wide layers run on the 192 / 208 instantiations — one row per DMA instruction, the last
// k-group and the last column tiles partly empty (operand read as zero there).  PHC_GEMM_HS_PUB=0 keeps them on the LDS-panel kernel.
static inline bool hs_pub_width(int K, int Nout) {
    static const bool off = getenv("PHC_GEMM_HS_PUB") && atoi(getenv("PHC_GEMM_HS_PUB")) == 0;
    return !off && K == Nout && (K == 180 || K == 200);
}
static inline bool hs_widths_ok(int K, int Nout) {
    return ((K == 128 || K == 256) && (Nout == 128 || Nout == 256)) || hs_pub_width(K, Nout);
}
static bool hs_shape_ok(const float *x, const float *Mk4, int64_t N, int K, int Nout) {
    constexpr long long min128 = 0;
    return hs_enabled() && Mk4 && N >= 1 && hs_widths_ok(K, Nout) && (K != 128 || Nout != 128 || N >= min128) &&
           ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(Mk4)) & 15) == 0;
}

static unsigned long long *g_hs_stamps = nullptr;             // diagnostics (phc_debug_hs_stamps): per-wavefront phase clocks

template <int K, int TPW, int WAVES, int FLAGS, int RB = 1>
static int launch_hs_cfg(GemmHsArgs a, const XfAcc &xa, hipStream_t s) {
    // 16-row steps (RB = 1).  32-row steps (RB = 2) were measured inside the replayed zinc_b1024 step: plain 28.3 -> 33.6 us,
    // with statistics 41.9 -> 46.9 us per launch (5.8 -> 2.9 steps per workgroup: the last round is half empty); they are for
    // launches with hundreds of steps per workgroup (config 5: launch_hs)
    constexpr int R = HS_RING;
    a.stamps = g_hs_stamps;
    a.panels = (a.Nout + 16 * TPW * WAVES - 1) / (16 * TPW * WAVES);
    const long long steps = (a.M + 16 * RB - 1) / (16 * RB);
    // workgroups per CU: one for the 8-wavefront configurations (two wavefronts per SIMD; a second workgroup made the zinc_b1024
    // step 2.10 -> 2.23 ms).  The 128 x 128 configuration (4 wavefronts, 64 operand registers, 33 KB of ring) leaves a SIMD with
    // one wavefront: two workgroups per CU are worth 3-4 % at scale (10 M rows: 3.07 -> 2.95 ms, scripts/dbg/gemm_hs_vs_panel.py)
    // and 2 % of the molhiv_b512 step, whose folded-norm dx GEMMs then run 1.7 instead of 3.4 steps per workgroup
    // (0.706 / 0.711 / 0.714 -> 0.695 / 0.692 / 0.698 ms, alternating runs on one box with PHC_HS_GROUPS_PER_CU = 1 / 2)
    constexpr int per_cu_env = 0;
    const int per_cu = per_cu_env > 0 ? per_cu_env : (WAVES == 4 && K == 128 ? 2 : 1);
    long long groups = (long long)per_cu * num_cu() / a.panels;
    if (groups > steps) groups = steps;
    if (groups < 1) groups = 1;
    a.groups = (int)groups;
    const size_t lds = hs_lds_bytes<K, RB, R, FLAGS>();
    PHC_REQUIRE(ensure_dynamic_lds(&gemm_hs16_kernel<K, TPW, WAVES, RB, R, FLAGS>, 150 * 1024), PHC_ERR_LAUNCH,
                "gemm_hs16_kernel: the device refused %d bytes of dynamic LDS", 150 * 1024);
    PHC_LAUNCH((gemm_hs16_kernel<K, TPW, WAVES, RB, R, FLAGS>), dim3((unsigned)(a.panels * a.groups)), dim3(64 * WAVES), lds, s, a, xa);
    return PHC_OK;
}

template <int FLAGS>
static int launch_hs(const GemmHsArgs &a, int K, const XfAcc &xa, hipStream_t s) {
    // Nout = 256: 8 wavefronts x 2 tiles; Nout = 128: 4 wavefronts x 2 tiles (two independent accumulators per wavefront keep
    // the 16x16x4 MFMA's 40-cycle dependent latency off the issue path)
    // (Nout = 256 as two 128-column workgroups of 4 wavefronts, two per CU: zinc_b1024 2.11 -> 3.98 ms — the doubled ring of the
    // folded-norm variants leaves room for one workgroup per CU only)
    // 180 x 180: 12 of 16 column tiles (8 wavefronts x 2), 12 k-groups; 200 x 200: 13 of 16 tiles, 13 k-groups
    // (4 wavefronts x 3 tiles for 180: the folded-norm variants then need > 256 registers and hipcc parks arrays in scratch)
    if (K == 180 && a.Nout == 180) return launch_hs_cfg<192, 2, 8, FLAGS>(a, xa, s);
    if (K == 200 && a.Nout == 200) return launch_hs_cfg<208, 2, 8, FLAGS>(a, xa, s);
    if (K == 256 && a.Nout == 256) return launch_hs_cfg<256, 2, 8, FLAGS>(a, xa, s);
    if (K == 256 && a.Nout == 128) return launch_hs_cfg<256, 2, 4, FLAGS>(a, xa, s);
    if (K == 128 && a.Nout == 256) return launch_hs_cfg<128, 2, 8, FLAGS>(a, xa, s);
    // (128 x 128 with 32-row steps at config-5 scale was measured worth nothing — 10 M rows, two workgroups per CU: 2868 -> 2857 us —
    // and its instantiations were removed in round 5; the kernel's RB parameter remains)
    return launch_hs_cfg<128, 2, 4, FLAGS>(a, xa, s);
}

// the square configurations only (the dropout variants of the folded norm backward: fewer instantiations of a heavy kernel)
template <int FLAGS>
static int launch_hs_square(const GemmHsArgs &a, int K, hipStream_t s) {
    if (K == 180) return launch_hs_cfg<192, 2, 8, FLAGS>(a, XfAcc{}, s);
    if (K == 200) return launch_hs_cfg<208, 2, 8, FLAGS>(a, XfAcc{}, s);
    if (K == 256) return launch_hs_cfg<256, 2, 8, FLAGS>(a, XfAcc{}, s);
    return launch_hs_cfg<128, 2, 4, FLAGS>(a, XfAcc{}, s);
}

static GemmHsArgs hs_args(const float *x, const float *Mk4, const float *b, const float *residual, float *y, int64_t N, int K, int Nout) {
    GemmHsArgs a = {};
    a.x = x; a.hk4 = Mk4; a.bias = b; a.residual = residual; a.y = y; a.M = N; a.K = K; a.Nout = Nout;
    return a;
}

constexpr int GEMM_STREAM_MAX_K = 256;
// plain and statistics launches on a PREBUILT operand also take deeper contractions (the 768-wide head of the molpcba
// configuration, run_script_pcba_phm4.sh: --d_units 768,256): the [K x 32] panel of a 32-column workgroup still fits the LDS
// (768 x 33 floats = 99 KB + 4 x 8.5 KB staging).  Operand transforms and backward sums keep the 256 limit (one column per thread).
constexpr int GEMM_PREBUILT_MAX_K = 768;
// rows at which the 128-column / 8-wave streaming configuration takes over from the 64-column / 4-wave one
constexpr int64_t GEMM_BIG_M = 64 * 1024;
// K > 128 (hidden 256): rows from which the 64-column / 8-wave configuration (one panel per CU shared by 8 wavefronts, x
// read 4x) takes over from the 32-column / 4-wave one (two workgroups per CU, x read 8x).
// ... and the contraction depth from which it does: the published 180-wide model (run_script_zinc_phm4.sh) runs 1.756 -> 1.622 ms per
// step on the 32-column configuration, the 200-wide one 1.940 -> 1.972 ms (alternating runs, PHC_GEMM_WIDE_K_M = 16384 / 1000000)
constexpr int GEMM_WIDE_K_MIN = 192;
// ... the 128-column / 8-wave configuration: its [K x 129] panel + staging must fit the 150 KB of dynamic LDS (K <= 160; a 180-deep
// contraction at >= 64 k rows used to be refused by the runtime: "invalid argument")
static inline bool gemm_big_cfg(int64_t N, int K, int Nout) { return N >= 64 * 1024 && Nout > 64 && K <= 160; }
static inline int64_t gemm_wide_k_m() {
    // measured (bench.py, one MI355X): zinc_b1024 (23.6 k rows, PHM-8 hidden 256) 2.75 -> 2.58 ms, molpcba_b1024 (27.2 k rows)
    // 3.42 -> 3.19 ms per training step with the 8-wave configuration instead of 2 x (32 columns, 4 waves) per CU
    constexpr int64_t v = 16 * 1024;
    return v;
}

template <int TRANS>
static int launch_gemm(const float *X, const float *A, const float *W, const float *bias, const float *residual,
                       float *out, int64_t M, int K, int Nout, int n, int kin, int kout, hipStream_t s) {
    if (M == 0) return PHC_OK;
    ProfScope prof(TRANS ? K_PHM_GEMM_DX : K_PHM_GEMM_FWD, s);
    const bool vec4 = (K % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
    const bool stream = vec4 && K <= GEMM_STREAM_MAX_K;
    const bool wide_k = K > 128;                       // the [K x 128] panel no longer fits next to 8 staging buffers
    const bool big = M >= GEMM_BIG_M && Nout > 64;
    // (32-column panels for molecule batches: measured no gain for y = x.H, slower for dx — removed in round 5)
#define PHC_GEMM_LAUNCH(NP)                                                                                              \
    if (stream && big && wide_k)                                                                                         \
        launch_stream_cfg<TRANS, NP, 64, 8>(X, A, W, bias, residual, out, M, K, Nout, n, kin, kout, 1, s);                \
    else if (stream && big)                                                                                              \
        launch_stream_cfg<TRANS, NP, 128, 8>(X, A, W, bias, residual, out, M, K, Nout, n, kin, kout, 1, s);               \
    else if (stream)                                                                                                     \
        launch_stream_cfg<TRANS, NP, 64, 4>(X, A, W, bias, residual, out, M, K, Nout, n, kin, kout, 2, s);                \
    else if (big)                                                                                                        \
        launch_gemm_cfg<TRANS, NP, 128, 128>(X, A, W, bias, residual, out, M, K, Nout, n, kin, kout, 1, s);               \
    else                                                                                                                 \
        launch_gemm_cfg<TRANS, NP, 64, 64>(X, A, W, bias, residual, out, M, K, Nout, n, kin, kout, 2, s)
    switch (n) {
        case 1: PHC_GEMM_LAUNCH(1); break;                  // dense maps (the hypercomplex -> real heads)
        case 2: PHC_GEMM_LAUNCH(2); break;
        case 4: PHC_GEMM_LAUNCH(4); break;
        case 8: PHC_GEMM_LAUNCH(8); break;
        default: PHC_GEMM_LAUNCH(0); break;
    }
#undef PHC_GEMM_LAUNCH
    return check_launch(TRANS ? "phm_gemm_dx" : "phm_gemm_fwd");
}

// 128 x 128 tiles when that still leaves >= 4 tiles per matrix, or when there are so many rows that the extra row
// splits a single big tile needs cost nothing next to the operand stream (>= 64k rows): otherwise the slab
// traffic of those splits outweighs the halved operand reads (measured at 128 x 128, N = 13.8k)
static inline bool dh_use_128(int64_t N, int in, int out) {
    constexpr bool no128 = false;
    if (no128 || in < DH_T2 || out < DH_T2 || in % 4 || out % 4) return false;
    return (in >= 2 * DH_T2 && out >= 2 * DH_T2) || N >= 64 * 1024;
}

struct DhPlan {
    int t128;
    int tiles_i, tiles_j, splits;
    int64_t rows_per_split;
    size_t slab_elems;
};

static DhPlan dh_plan(int64_t N, int in, int out) {
    DhPlan p;
    p.t128 = dh_use_128(N, in, out) ? 1 : 0;
    const int T = p.t128 ? DH_T2 : DH_T;
    p.tiles_i = (in + T - 1) / T;
    p.tiles_j = (out + T - 1) / T;
    const int tiles = p.tiles_i * p.tiles_j;
    constexpr int target_blocks = 512;
    int64_t splits = (target_blocks + tiles - 1) / tiles;       // ~2 workgroups per CU (fixed: plan must not depend on the device)
    const int64_t max_splits = (N + DH_RB - 1) / DH_RB;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    int64_t rps = (N + splits - 1) / splits;
    rps = (rps + DH_RB - 1) / DH_RB * DH_RB;
    if (rps < DH_RB) rps = DH_RB;
    splits = N > 0 ? (N + rps - 1) / rps : 1;
    p.splits = (int)splits;
    p.rows_per_split = rps;
    p.slab_elems = (size_t)in * out + out;
    return p;
}

// register-resident form of ONE layer (phc_phm_linear_bwd): splits = workgroups / tiles with every workgroup the same number
// of rows (a multiple of 64), at most 256 workgroups (one round of one workgroup per CU on a 256-CU part; shape-only: fixed order)
static bool dh_direct_shape_ok(int64_t N, int in, int out) {
    constexpr bool off = false;
    if ((int64_t)(in > out ? in : out) * 4 * (N / 256 + 128) >= ((int64_t)1 << 31)) return false;       // 32-bit offsets per wavefront
    return !off && in % 4 == 0 && out % 4 == 0 && in >= 96 && out >= 32 && N >= 8;             // partial edge tiles: dH_direct_body
}
static DhPlan dh_plan_direct(int64_t N, int in, int out) {
    DhPlan p;
    p.t128 = 2;
    p.tiles_i = (in + DH_D - 1) / DH_D;
    p.tiles_j = (out + DH_D - 1) / DH_D;
    const int64_t tiles = (int64_t)p.tiles_i * p.tiles_j;
    constexpr int64_t dtarget = 256;
    int64_t rps = (tiles * N / dtarget + 63) / 64 * 64;
    if (rps < 64) rps = 64;
    while (tiles * ((N + rps - 1) / rps) > dtarget && rps < N) rps += 64;
    p.rows_per_split = rps;
    p.splits = (int)((N + rps - 1) / rps);
    p.slab_elems = (size_t)in * out + out;
    return p;
}

struct GradPlan {
    int tiles_i, tiles_j, splits, t128, pair;
    int64_t rows_per_split;
    size_t slab_elems;
};

// register-resident tile form (dH_direct_body): float4-addressable operands; widths that are not multiples of the tile end in a
// partial tile (form 2: 128 x 128 tiles; form 3: 128 x 64, or 128 x 32 when out is an odd multiple of 32 — dh3_np)
// (form 3 reads dy two columns per lane, or one when the width is odd: any width — the 200 -> 50 gate map of the published 2 x 200
// model, which as 848 LDS-tile workgroups held the tail's slots for 20 us of its 130 (scripts/dbg/tail_wg_timeline.py), the 180 -> 45
// one of the 4 x 180 model; form 2 reads four per lane)
static inline bool dh_use_direct(const phc_phm_grad_job &jb, int form = 2) {
    constexpr bool off = false;
    const int in = jb.n * jb.kin, out = jb.n * jb.kout;
    if ((int64_t)(in > out ? in : out) * 4 * (jb.N / 256 + 128) >= ((int64_t)1 << 31)) return false;    // 32-bit offsets per wavefront
    return !off && in % 4 == 0 && (form == 3 || out % 4 == 0) && in >= 96 && out >= 32 && jb.N >= 8 && (reinterpret_cast<uintptr_t>(jb.x) & 15) == 0 &&
           (reinterpret_cast<uintptr_t>(jb.dy) & 15) == 0 &&
           (!jb.xf_stat || ((reinterpret_cast<uintptr_t>(jb.xf_stat) & 15) == 0 &&
                            (jb.xf_act == PHC_ACT_IDENTITY || jb.xf_act == PHC_ACT_RELU)));
}

// Split counts of a whole batch: about 1024 workgroups in total, shared among the jobs in proportion to
// their row-block counts (depends on the shapes only, never on the device: fixed summation order).
// direct_form: 2 = 128 x 128 tiles per wavefront in their own launch (512 registers, one wavefront per SIMD); 3 = 128 x 64 tiles
// (256 registers, two per SIMD) for a launch SHARED with other workgroup families (phc_step_tail: the table-gradient workgroups
// are latency-bound gathers and run in the slots beside the MFMA-bound dH wavefronts)
// slots3: the workgroup slots the form-3 family may take (512 = two per CU; the step tail leaves some to its table families)
static void grad_plans(const phc_phm_grad_job *jobs, int num, GradPlan *plans, int direct_form = 2, int64_t slots3 = 512) {
    int64_t total = 0, direct_rows = 0;
    for (int k = 0; k < num; ++k) {
        const int in = jobs[k].n * jobs[k].kin, out = jobs[k].n * jobs[k].kout;
        plans[k].t128 = dh_use_direct(jobs[k], direct_form) ? direct_form : (dh_use_128(jobs[k].N, in, out) ? 1 : 0);
        plans[k].pair = 1;
        const int T = plans[k].t128 ? DH_T2 : DH_T;
        plans[k].tiles_i = (in + T - 1) / T;
        plans[k].tiles_j = plans[k].t128 == 3 ? (out + 32 * dh3_np(out) - 1) / (32 * dh3_np(out)) : (out + T - 1) / T;
        plans[k].slab_elems = (size_t)in * out + out;
        if (plans[k].t128 >= 2)
            direct_rows += (int64_t)plans[k].tiles_i * plans[k].tiles_j * jobs[k].N;
        else
            total += (int64_t)plans[k].tiles_i * plans[k].tiles_j * ((jobs[k].N + DH_RB - 1) / DH_RB);
    }
    if (direct_rows > 0) {
        // register-resident jobs: one workgroup per CU of a 256-CU part in ONE round (a workgroup owns its CU: 512 registers per
        // lane), every workgroup the same number of rows: the smallest multiple of 8 whose workgroup count fits the target
        constexpr int64_t dtarget2 = 256;
        const int64_t dtarget3 = slots3;
        const int64_t dtarget = direct_form == 3 ? dtarget3 : dtarget2;       // form 3: two workgroups per CU
        // multiples of 64 rows: each of the four wavefronts then runs whole 16-row iterations of the unmasked loop
        // form 3 (the step tail), even shares: a job's shorter last split is `pair` tiles' worth of ONE workgroup when that many of them
        // fit in rows_per_split rows (tail kernel, GradJobDev.pair) — the workgroup count is then (nearly) continuous in rows_per_split
        // and the family fills its slots instead of stopping at the last whole split count that fits (PHC_DH_PAIR=0: whole counts)
        static const bool pairing = !(getenv("PHC_DH_PAIR") && atoi(getenv("PHC_DH_PAIR")) == 0);
        auto pair_of = [&](int k, int64_t rps) -> int {
            if (direct_form != 3 || !pairing) return 1;
            const int64_t rem = jobs[k].N < rps ? jobs[k].N : jobs[k].N % rps;      // (a job of fewer rows than one split: its only split)
            if (rem == 0) return 1;
            const int64_t m = rps / ((rem + 63) / 64 * 64);
            return (int)(m < 1 ? 1 : (m > 8 ? 8 : m));
        };
        auto wgs_of = [&](int k, int64_t rps) -> int64_t {
            const int64_t tiles = (int64_t)plans[k].tiles_i * plans[k].tiles_j, splits = (jobs[k].N + rps - 1) / rps;
            const int m = pair_of(k, rps);
            return m > 1 ? tiles * (splits - 1) + (tiles + m - 1) / m : tiles * splits;
        };
        int64_t rps = (direct_rows / dtarget + 63) / 64 * 64;
        if (rps < 64) rps = 64;
        for (;; rps += 64) {
            int64_t wgs = 0;
            for (int k = 0; k < num; ++k)
                if (plans[k].t128 >= 2) wgs += wgs_of(k, rps);
            if (wgs <= dtarget || rps >= direct_rows) break;
        }
        for (int k = 0; k < num; ++k)
            if (plans[k].t128 >= 2) {
                plans[k].rows_per_split = rps;
                plans[k].splits = (int)((jobs[k].N + rps - 1) / rps);
                plans[k].pair = pair_of(k, rps);
            }
    }
    constexpr int64_t target = 1024;
    static const bool dbg = getenv("PHC_DEBUG_PLANS") != nullptr;
    if (dbg)
        for (int k = 0; k < num; ++k)
            fprintf(stderr, "[phc plan] job %d: N=%lld in=%d out=%d form=%d xf=%d splits=%d rows_per_split=%lld pair=%d\n", k, (long long)jobs[k].N,
                    jobs[k].n * jobs[k].kin, jobs[k].n * jobs[k].kout, plans[k].t128, jobs[k].xf_stat ? jobs[k].xf_act : -1,
                    plans[k].t128 >= 2 ? plans[k].splits : -1, plans[k].t128 >= 2 ? (long long)plans[k].rows_per_split : -1LL, plans[k].pair);
    for (int k = 0; k < num; ++k) {
        if (plans[k].t128 >= 2) continue;
        const int64_t max_splits = (jobs[k].N + DH_RB - 1) / DH_RB;
        int64_t splits = total > target ? (max_splits * target + total - 1) / total : max_splits;
        if (splits < 1) splits = 1;
        if (splits > max_splits && max_splits > 0) splits = max_splits;
        int64_t rows = (jobs[k].N + splits - 1) / splits;
        rows = (rows + DH_RB - 1) / DH_RB * DH_RB;
        if (rows < DH_RB) rows = DH_RB;
        plans[k].rows_per_split = rows;
        plans[k].splits = (int)((jobs[k].N + rows - 1) / rows);
        if (plans[k].splits < 1) plans[k].splits = 1;
    }
}

int check_grad_jobs(const char *who, const phc_phm_grad_job *jobs, int num_jobs) {
    PHC_REQUIRE(num_jobs >= 0 && (num_jobs == 0 || jobs), PHC_ERR_INVALID_ARG, "%s: bad job table", who);
    for (int k = 0; k < num_jobs; ++k) {
        int rc = check_phm_dims(who, jobs[k].N, jobs[k].n, jobs[k].kin, jobs[k].kout);
        if (rc) return rc;
        PHC_REQUIRE(jobs[k].N > 0, PHC_ERR_INVALID_ARG, "%s: job %d has no rows (its gradients are zero: do not submit it)", who, k);
    }
    return PHC_OK;
}

size_t grad_chunk_ws_floats(const phc_phm_grad_job *jobs, int num_jobs, int direct_form) {
    GradPlan plans[GRAD_JOBS_PER_LAUNCH];
    grad_plans(jobs, num_jobs, plans, direct_form);
    size_t off = 0;
    for (int k = 0; k < num_jobs; ++k) off += (plans[k].slab_elems * ((size_t)plans[k].splits + 1) + 3) / 4 * 4;   // 16-byte segments
    return off;
}

int prepare_grad_launch(const char *who, const phc_phm_grad_job *jobs, int cn, float *ws_base, GradLaunch *out, int direct_form, int slots3) {
    PHC_REQUIRE(cn >= 1 && cn <= GRAD_JOBS_PER_LAUNCH, PHC_ERR_INVALID_ARG, "%s: %d jobs in one launch (1..%d)", who, cn,
                GRAD_JOBS_PER_LAUNCH);
    PHC_REQUIRE(slots3 >= 64 && slots3 <= 512, PHC_ERR_INVALID_ARG, "%s: %d workgroup slots for the parameter-gradient family", who, slots3);
    GradPlan plans[GRAD_JOBS_PER_LAUNCH];
    grad_plans(jobs, cn, plans, direct_form, slots3);    // (fewer slots: longer splits, fewer slabs — grad_chunk_ws_floats' 512-slot size holds)
    GradBatch &b = out->batch;
    b.num = cn;
    unsigned nb_partial = 0, nb_partial128 = 0, nb_direct = 0, nb_reduce = 0, nb_contract = 0;
    size_t off = 0;
    for (int k = 0; k < cn; ++k) {
        const phc_phm_grad_job &jb = jobs[k];
        PHC_REQUIRE(jb.x && jb.dy && jb.A && jb.W && (jb.dW || jb.dA || jb.db), PHC_ERR_INVALID_ARG, "%s: job %d: null pointer", who, k);
        PHC_REQUIRE(!jb.xf_stat || (jb.xf_act >= PHC_ACT_IDENTITY && jb.xf_act <= PHC_ACT_SELU), PHC_ERR_INVALID_ARG,
                    "%s: job %d: unknown activation %d", who, k, jb.xf_act);
        GradJobDev &d = b.j[k];
        d.x = jb.x; d.dy = jb.dy; d.A = jb.A; d.W = jb.W;
        d.dW = jb.dW; d.dA = jb.dA; d.db = jb.db;
        d.N = jb.N; d.n = jb.n; d.kin = jb.kin; d.kout = jb.kout;
        d.in = jb.n * jb.kin; d.out = jb.n * jb.kout;
        d.tiles_j = plans[k].tiles_j;
        d.tiles = plans[k].tiles_i * plans[k].tiles_j;
        d.splits = plans[k].splits;
        d.t128 = plans[k].t128;
        d.dw_transposed = jb.flags & 1;
        d.reg_add = nullptr;
        d.xf_stat = jb.xf_stat;
        d.xf_n_valid = jb.xf_n_valid_dev;
        d.xf_act = jb.xf_act;
        d.rows_per_split = plans[k].rows_per_split;
        d.vec4 = (d.in % 4 == 0) && (d.out % 4 == 0) && ((reinterpret_cast<uintptr_t>(jb.x) & 15) == 0) &&
                 ((reinterpret_cast<uintptr_t>(jb.dy) & 15) == 0);
        d.slabs = ws_base + off;
        d.dH = d.slabs + plans[k].slab_elems * (size_t)plans[k].splits;
        off += (plans[k].slab_elems * ((size_t)plans[k].splits + 1) + 3) / 4 * 4;
        d.dA_blocks = jb.n * jb.n * jb.n;
        unsigned &nb_mine = d.t128 == 2 ? nb_direct : (d.t128 == 1 ? nb_partial128 : nb_partial);     // forms 0 and 3 share a launch
        d.blk0_partial = nb_mine;
        d.blk0_reduce = nb_reduce;
        d.blk0_contract = nb_contract;
        d.pair = d.t128 == 3 ? plans[k].pair : 1;
        nb_mine += d.pair > 1 ? (unsigned)d.tiles * (unsigned)(d.splits - 1) + (unsigned)((d.tiles + d.pair - 1) / d.pair)
                              : (unsigned)d.tiles * (unsigned)d.splits;
        nb_reduce += (unsigned)((plans[k].slab_elems + 63) / 64);
        nb_contract += (unsigned)(d.dA_blocks + (jb.n * jb.kin * jb.kout + 255) / 256);
    }
    out->nb_partial = nb_partial;
    out->nb_partial128 = nb_partial128;
    out->nb_direct = nb_direct;
    out->nb_reduce = nb_reduce;
    out->nb_contract = nb_contract;
    return PHC_OK;
}

int launch_grad_direct(const char *who, const GradLaunch &gl, hipStream_t s, bool profile) {
    if (gl.nb_direct == 0) return PHC_OK;
    GradBatch bs = gl.batch;                                 // jobs of the other tile forms are masked out
    for (int k = 0; k < bs.num; ++k)
        if (bs.j[k].t128 != 2) bs.j[k].blk0_partial = 0xffffffffu;
    PHC_REQUIRE(ensure_dynamic_lds(&phm_grads_direct_kernel, DH_DIRECT_LDS_BYTES), PHC_ERR_LAUNCH,
                "%s: the device refused %d bytes of dynamic LDS", who, (int)DH_DIRECT_LDS_BYTES);
    if (profile) {
        ProfScope prof(K_PHM_DH_PARTIAL, s);
        PHC_LAUNCH(phm_grads_direct_kernel, dim3(gl.nb_direct), dim3(256), DH_DIRECT_LDS_BYTES, s, bs);
    } else {
        PHC_LAUNCH(phm_grads_direct_kernel, dim3(gl.nb_direct), dim3(256), DH_DIRECT_LDS_BYTES, s, bs);
    }
    return check_launch(who);
}

int launch_grad_partial128(const char *who, const GradLaunch &gl, hipStream_t s) {
    if (gl.nb_partial128 == 0) return PHC_OK;
    GradBatch bs = gl.batch;                                 // jobs of the other tile forms are masked out
    for (int k = 0; k < bs.num; ++k)
        if (bs.j[k].t128 != 1) bs.j[k].blk0_partial = 0xffffffffu;
    ProfScope prof(K_PHM_DH_PARTIAL, s);
    PHC_LAUNCH(phm_grads_partial_kernel<1>, dim3(gl.nb_partial128), dim3(256), 0, s, bs);
    return check_launch(who);
}

int launch_grad_contract(const char *who, const GradLaunch &gl, hipStream_t s) {
    PHC_LAUNCH(phm_grads_contract_kernel, dim3(gl.nb_contract), dim3(256), 0, s, gl.batch);
    return check_launch(who);
}

int launch_tail_finish(const char *who, const GradLaunch &gl, const GateGradHost *gate, const TailReg *reg, hipStream_t s) {
    if (!gate && !reg) return launch_grad_contract(who, gl, s);
    GateGradArgs g = {};
    unsigned nb_gate = 0;
    size_t lds = 0;
    if (gate) {
        PHC_REQUIRE(gate->job >= 0 && gate->job < gl.batch.num && gate->kin == gate->kout && gate->kg >= 1, PHC_ERR_INVALID_ARG,
                    "%s: gate job %d / shape", who, gate->job);
        const GradJobDev &dj = gl.batch.j[gate->job];
        const int k = gate->kin, h = gate->n * k;
        PHC_REQUIRE(dj.n == 1 && dj.in == h && dj.out == gate->kg && dj.dH, PHC_ERR_INVALID_ARG,
                    "%s: the gate's dense job is [%d x %d], expected [%d x %d]", who, dj.in, dj.out, h, gate->kg);
        const int lds_tile = 2 * gate->kg * k + k * (k + 1), lds_row = gate->kg * h + gate->kg * gate->n + gate->n * gate->n * gate->n + gate->n * h;
        lds = sizeof(float) * (size_t)(lds_tile > lds_row ? lds_tile : lds_row);
        PHC_REQUIRE(lds <= 128 * 1024, PHC_ERR_UNSUPPORTED, "%s: gate_w [%d x %d] does not fit the LDS staging", who, gate->kg, h);
        g.dMT = dj.dH; g.dbg = dj.dH + (size_t)h * gate->kg;                     // dH = x^T . dg [h, kg], then its bias row = sum dg
        g.sj = 1; g.sr = gate->kg;
        g.H = gate->H; g.gate_w = gate->gate_w; g.phm_b = gate->phm_b; g.A = gate->A; g.W = gate->W;
        g.dA = gate->dA; g.dW = gate->dW; g.db = gate->db; g.dWg = gate->dWg; g.dbg_out = gate->dbg; g.reg_add = gate->reg_add;
        g.n = gate->n; g.k = k; g.kg = gate->kg;
        g.blk_rows = g.n * g.n;
        g.blk_dWg = g.blk_rows + k;
        g.blk_db = g.blk_dWg + g.kg * ((h + 31) / 32);
        nb_gate = (unsigned)(g.blk_db + 1);
    }
    TailReg r = {};
    if (reg) r = *reg;
    PHC_REQUIRE(ensure_dynamic_lds(&tail_finish_kernel, 128 * 1024), PHC_ERR_LAUNCH, "%s: the device refused %d bytes of dynamic LDS", who,
                128 * 1024);
    PHC_LAUNCH(tail_finish_kernel, dim3(gl.nb_contract + nb_gate + (reg ? 1u : 0u)), dim3(256), lds, s, gl.batch, g, r, gl.nb_contract, nb_gate);
    return check_launch(who);
}

// the 128 x 128 partial launch of a prepared batch (if it has such jobs), then reduce and contract; the 64 x 64 partial
// launch is the caller's (phc_phm_param_grads launches it alone, phc_step_tail side by side with the table gradients)
int launch_grad_rest(const char *who, const GradLaunch &gl, bool partial64_too, hipStream_t s) {
    int rc = launch_grad_direct(who, gl, s);
    if (rc) return rc;
    for (int shape = partial64_too ? 0 : 1; shape < 2; ++shape) {
        const unsigned nb = shape ? gl.nb_partial128 : gl.nb_partial;
        if (nb == 0) continue;
        GradBatch bs = gl.batch;                             // jobs of the other tile shape are masked out
        for (int k = 0; k < bs.num; ++k)
            if (bs.j[k].t128 != shape) bs.j[k].blk0_partial = 0xffffffffu;
        ProfScope prof(K_PHM_DH_PARTIAL, s);
        if (shape)
            PHC_LAUNCH(phm_grads_partial_kernel<1>, dim3(nb), dim3(256), 0, s, bs);
        else
            PHC_LAUNCH(phm_grads_partial_kernel<0>, dim3(nb), dim3(256), 0, s, bs);
        rc = check_launch(who);
        if (rc) return rc;
    }
    {
        ProfScope prof(K_PHM_DH_REDUCE, s);
        PHC_LAUNCH(phm_grads_reduce_kernel, dim3(gl.nb_reduce), dim3(256), 0, s, gl.batch);
        rc = check_launch(who);
        if (rc) return rc;
    }
    ProfScope prof(K_PHM_CONTRACT, s);
    PHC_LAUNCH(phm_grads_contract_kernel, dim3(gl.nb_contract), dim3(256), 0, s, gl.batch);
    return check_launch(who);
}

}  // namespace phc

using namespace phc;

extern "C" {

#ifdef PHC_KTIMING
int phc_debug_dh_timestamps(unsigned long long *out64) {
    return hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_dh_ts), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -1;
}
int phc_debug_stream_timestamps(unsigned long long *out64) {
    return hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_st_ts), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -1;
}
#endif

int phc_phm_build_H(const float *A, const float *W, int n, int kin, int kout, float *H, void *stream) {
    int rc = check_phm_dims("phc_phm_build_H", 0, n, kin, kout);
    if (rc) return rc;
    PHC_REQUIRE(A && W && H, PHC_ERR_INVALID_ARG, "phc_phm_build_H: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const long long total = (long long)n * kin * n * kout;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    ProfScope prof(K_PHM_BUILD_H, s);
    PHC_LAUNCH(phm_build_H_kernel, dim3(blocks), dim3(256), sizeof(float) * n * n * n, s, A, W, H, n, kin, kout);
    return check_launch("phm_build_H");
}

int phc_phm_linear_fwd(const float *x, const float *A, const float *W, const float *b, const float *residual, int64_t N,
                       int n, int kin, int kout, float *y, void *stream) {
    int rc = check_phm_dims("phc_phm_linear_fwd", N, n, kin, kout);
    if (rc) return rc;
    PHC_REQUIRE(A && W && (N == 0 || (x && y)), PHC_ERR_INVALID_ARG, "phc_phm_linear_fwd: null pointer");
    PHC_REQUIRE(x != y || N == 0, PHC_ERR_INVALID_ARG, "phc_phm_linear_fwd: y must not alias x");
    return launch_gemm<0>(x, A, W, b, residual, y, N, n * kin, n * kout, n, kin, kout, (hipStream_t)stream);
}

size_t phc_phm_linear_bwd_ws_bytes(int64_t N, int n, int kin, int kout) {
    if (N < 0 || n < 1 || kin < 1 || kout < 1) return 0;
    DhPlan p = dh_plan(N, n * kin, n * kout);
    size_t need = p.slab_elems * ((size_t)p.splits + 1);
    if (dh_direct_shape_ok(N, n * kin, n * kout)) {          // either form may run (the register-resident one needs aligned operands)
        DhPlan d = dh_plan_direct(N, n * kin, n * kout);
        const size_t nd = d.slab_elems * ((size_t)d.splits + 1);
        if (nd > need) need = nd;
    }
    return sizeof(float) * need;
}

int phc_phm_linear_bwd(const float *x, const float *A, const float *W, const float *dy, int64_t N, int n, int kin, int kout,
                       float *dx, float *dW, float *dA, float *db, void *ws, size_t ws_bytes, void *stream) {
    int rc = check_phm_dims("phc_phm_linear_bwd", N, n, kin, kout);
    if (rc) return rc;
    PHC_REQUIRE(A && W && (N == 0 || (x && dy)), PHC_ERR_INVALID_ARG, "phc_phm_linear_bwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int in = n * kin, out = n * kout;
    const bool want_params = dW || dA || db;
    if (want_params) {
        const size_t need = phc_phm_linear_bwd_ws_bytes(N, n, kin, kout);
        PHC_REQUIRE(ws && ws_bytes >= need, PHC_ERR_WORKSPACE, "phc_phm_linear_bwd: workspace %zu B < required %zu B", ws_bytes,
                    need);
        PHC_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 15) == 0, PHC_ERR_WORKSPACE, "phc_phm_linear_bwd: workspace not 16-byte aligned");
    }
    hipStream_t sp = s;
    if (want_params) {
        const bool direct = dh_direct_shape_ok(N, in, out) && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
                            (reinterpret_cast<uintptr_t>(dy) & 15) == 0;
        DhPlan p = direct ? dh_plan_direct(N, in, out) : dh_plan(N, in, out);
        float *slabs = static_cast<float *>(ws);
        float *dH = slabs + p.slab_elems * (size_t)p.splits;
        if (N == 0) {
            hipMemsetAsync(dH, 0, sizeof(float) * p.slab_elems, sp);
        } else {
            const int vec4 = (in % 4 == 0) && (out % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) &&
                             ((reinterpret_cast<uintptr_t>(dy) & 15) == 0);
            {
                ProfScope prof(K_PHM_DH_PARTIAL, sp);
                if (p.t128 == 2) {
                    GradLaunch gl = {};
                    gl.batch.num = 1;
                    GradJobDev &d = gl.batch.j[0];
                    d.x = x; d.dy = dy; d.slabs = slabs; d.N = N; d.in = in; d.out = out;
                    d.tiles_j = p.tiles_j; d.tiles = p.tiles_i * p.tiles_j; d.splits = p.splits; d.rows_per_split = p.rows_per_split;
                    d.t128 = 2;
                    gl.nb_direct = (unsigned)d.tiles * (unsigned)d.splits;
                    rc = launch_grad_direct("phm_dH_partial", gl, sp, false);
                    if (rc) return rc;
                } else if (p.t128)
                    PHC_LAUNCH(phm_dH_partial128_kernel, dim3(p.tiles_i * p.tiles_j, p.splits), dim3(256), 0, sp, x, dy, slabs,
                               (long long)N, in, out, p.tiles_j, (long long)p.rows_per_split, vec4);
                else
                    PHC_LAUNCH(phm_dH_partial_kernel, dim3(p.tiles_i * p.tiles_j, p.splits), dim3(256), 0, sp, x, dy, slabs,
                               (long long)N, in, out, p.tiles_j, (long long)p.rows_per_split, vec4);
                rc = check_launch("phm_dH_partial");
                if (rc) return rc;
            }
            {
                ProfScope prof(K_PHM_DH_REDUCE, sp);
                PHC_LAUNCH(phm_dH_reduce_kernel, dim3((unsigned)((p.slab_elems + 63) / 64)), dim3(256), 0, sp, slabs,
                                   dH, p.slab_elems, p.splits);
                rc = check_launch("phm_dH_reduce");
                if (rc) return rc;
            }
        }
        const int dA_blocks = n * n * n;
        const int dW_blocks = (n * kin * kout + 255) / 256;
        ProfScope prof(K_PHM_CONTRACT, sp);
        PHC_LAUNCH(phm_contract_kernel, dim3(dA_blocks + dW_blocks), dim3(256), 0, sp, A, W, dH, dW, dA, db, n, kin, kout,
                           dA_blocks);
        rc = check_launch("phm_contract");
        if (rc) return rc;
    }
    if (dx) {
        rc = launch_gemm<1>(dy, A, W, nullptr, nullptr, dx, N, out, in, n, kin, kout, s);
        if (rc) return rc;
    }
    return PHC_OK;
}

// ---- prebuilt operand matrices --------------------------------------------------------------
static int build_H_batched_impl(const phc_phm_H_job *jobs, int num_jobs, const phc_encode_front *enc, void *stream);

int phc_phm_build_H_batched(const phc_phm_H_job *jobs, int num_jobs, void *stream) {
    return build_H_batched_impl(jobs, num_jobs, nullptr, stream);
}
int phc_encode_front_supported(int64_t N, int F, int n, int k) {
    return (N >= 1 && F >= 1 && F <= HB_ENC_MAX_F && n >= 1 && n <= PHC_MAX_COMPONENTS && k >= 4 && k % 4 == 0) ? 1 : 0;
}
int phc_phm_build_H_batched_ex(const phc_phm_H_job *jobs, int num_jobs, const phc_encode_front *enc, void *stream) {
    return build_H_batched_impl(jobs, num_jobs, enc, stream);
}

static int build_H_batched_impl(const phc_phm_H_job *jobs, int num_jobs, const phc_encode_front *enc, void *stream) {
    PHC_REQUIRE(num_jobs >= 0 && (num_jobs == 0 || jobs), PHC_ERR_INVALID_ARG, "phc_phm_build_H_batched: bad job table");
    hipStream_t s = (hipStream_t)stream;
    EncFront ef = {};
    if (enc) {
        PHC_REQUIRE(num_jobs >= 1 && phc_encode_front_supported(enc->N, enc->F, enc->n, enc->k) && enc->idx_i64 && enc->tables && enc->vocab &&
                        enc->out && (reinterpret_cast<uintptr_t>(enc->out) & 15) == 0,
                    PHC_ERR_INVALID_ARG, "phc_phm_build_H_batched_ex: encoder passenger N=%lld F=%d n=%d k=%d (F <= %d, k %% 4 == 0, aligned out)",
                    (long long)enc->N, enc->F, enc->n, enc->k, HB_ENC_MAX_F);
        ef.idx = reinterpret_cast<const long long *>(enc->idx_i64);
        ef.tables = enc->tables; ef.out = enc->out; ef.N = enc->N; ef.F = enc->F; ef.n = enc->n; ef.k = enc->k;
        for (int f = 0; f < enc->F; ++f) {
            PHC_REQUIRE(enc->vocab[f] >= 1, PHC_ERR_INVALID_ARG, "phc_phm_build_H_batched_ex: vocab[%d] = %d", f, enc->vocab[f]);
            ef.vocab[f] = enc->vocab[f];
        }
        ef.err = err_word();
    }
    for (int c0 = 0; c0 < num_jobs; c0 += H_JOBS_PER_LAUNCH) {
        const int cn = num_jobs - c0 < H_JOBS_PER_LAUNCH ? num_jobs - c0 : H_JOBS_PER_LAUNCH;
        HBatch b;
        b.num = cn;
        unsigned blocks = 0;
        for (int k = 0; k < cn; ++k) {
            const phc_phm_H_job &jb = jobs[c0 + k];
            int rc = check_phm_dims("phc_phm_build_H_batched", 1, jb.n, jb.kin, jb.kout);
            if (rc) return rc;
            if (jb.flags == 2) {                                 // zero fill: H[0 .. kin * kout) = 0 (n = 1; the step's statistics arena)
                PHC_REQUIRE(jb.H && jb.n == 1 && !jb.HT && !jb.Hk4 && !jb.HTk4, PHC_ERR_INVALID_ARG,
                            "phc_phm_build_H_batched: job %d: a zero-fill job takes H, n = 1 and nothing else", c0 + k);
                b.j[k] = HJobDev{};
                b.j[k].H = jb.H; b.j[k].n = 1; b.j[k].kin = jb.kin; b.j[k].kout = jb.kout; b.j[k].dense_t = 2;
                b.j[k].blk0 = blocks;
                blocks += (unsigned)(((size_t)jb.kin * jb.kout + 255) / 256);
                continue;
            }
            if (jb.flags == 3) {                                 // gate map: M = H(A, W) . gate_w^T, M^T, composed bias
                const int in_ = jb.n * jb.kin, out_ = jb.n * jb.kout;
                PHC_REQUIRE(jb.A && jb.W && jb.H && jb.gate_w && jb.gate_c && jb.gate_k >= 1 && out_ <= 1024 && !jb.Hk4 && !jb.HTk4,
                            PHC_ERR_INVALID_ARG, "phc_phm_build_H_batched: job %d: a gate job takes A, W, H (= M), gate_w, gate_c, "
                            "gate_k >= 1 and an inner width <= 1024 (got %d)", c0 + k, out_);
                b.j[k] = HJobDev{};
                b.j[k].A = jb.A; b.j[k].W = jb.W; b.j[k].H = jb.H; b.j[k].HT = jb.HT;
                b.j[k].n = jb.n; b.j[k].kin = jb.kin; b.j[k].kout = jb.kout; b.j[k].dense_t = 3;
                b.j[k].gate_w = jb.gate_w; b.j[k].gate_b = jb.gate_b; b.j[k].phm_b = jb.phm_b; b.j[k].gate_c = jb.gate_c; b.j[k].kg = jb.gate_k;
                b.j[k].blk0 = blocks;
                blocks += (unsigned)(in_ + 1);
                continue;
            }
            const int dense_t = jb.flags & 1;
            PHC_REQUIRE((jb.A || dense_t) && jb.W && jb.H, PHC_ERR_INVALID_ARG, "phc_phm_build_H_batched: job %d: null pointer", c0 + k);
            PHC_REQUIRE(!dense_t || jb.n == 1, PHC_ERR_INVALID_ARG, "phc_phm_build_H_batched: job %d: a dense weight needs n = 1", c0 + k);
            b.j[k] = HJobDev{};
            b.j[k].A = jb.A; b.j[k].W = jb.W; b.j[k].H = jb.H; b.j[k].HT = jb.HT;
            const int in_ = jb.n * jb.kin, out_ = jb.n * jb.kout;
            PHC_REQUIRE((!jb.Hk4 || in_ % 4 == 0) && (!jb.HTk4 || out_ % 4 == 0) && (!dense_t || (!jb.Hk4 && !jb.HTk4)), PHC_ERR_INVALID_ARG,
                        "phc_phm_build_H_batched: job %d: the k4 form needs a contraction width that is a multiple of 4 (PHM jobs only)", c0 + k);
            b.j[k].Hk4 = jb.Hk4; b.j[k].HTk4 = jb.HTk4;
            b.j[k].n = jb.n; b.j[k].kin = jb.kin; b.j[k].kout = jb.kout; b.j[k].dense_t = dense_t;
            b.j[k].blk0 = blocks;
            constexpr bool no_tiles = false;
            const bool tiled = !no_tiles && !dense_t && in_ % 4 == 0 && out_ % 4 == 0 && jb.kout % 4 == 0 && in_ >= 4 && out_ >= 4 &&
                               ((reinterpret_cast<uintptr_t>(jb.H) | reinterpret_cast<uintptr_t>(jb.HT) | reinterpret_cast<uintptr_t>(jb.Hk4) |
                                 reinterpret_cast<uintptr_t>(jb.HTk4) | reinterpret_cast<uintptr_t>(jb.W)) & 15) == 0;
            if (tiled) {
                b.j[k].dense_t = 4;
                blocks += (unsigned)(((in_ + 31) / 32) * ((out_ + 31) / 32));
            } else {
                blocks += (unsigned)(((size_t)jb.n * jb.kin * jb.n * jb.kout + 255) / 256);
            }
        }
        EncFront e = {};
        if (enc && c0 + cn >= num_jobs) {                        // the passenger rides in the last chunk's launch
            e = ef;
            const long long units = ef.N * ((ef.n * ef.k) >> 2);
            long long eb = (units + 255) / 256;
            if (eb > 8192) eb = 8192;
            e.blk0 = blocks;
            e.blocks = (unsigned)eb;
            blocks += (unsigned)eb;
        }
        ProfScope prof(K_PHM_BUILD_H, s);
        PHC_LAUNCH(phm_build_H_batched_kernel, dim3(blocks), dim3(256), 0, s, b, e);
        int rc = check_launch("phm_build_H_batched");
        if (rc) return rc;
    }
    return PHC_OK;
}

int phc_gate_param_grads(const float *dMT, const float *dbg, const float *H, const float *gate_w, const float *phm_b, const float *A,
                         const float *W, int n, int kin, int kout, int kg, float *dA, float *dW, float *db, float *dWg, void *stream) {
    int rc = check_phm_dims("phc_gate_param_grads", 1, n, kin, kout);
    if (rc) return rc;
    PHC_REQUIRE(dMT && dbg && H && gate_w && A && W && kg >= 1, PHC_ERR_INVALID_ARG, "phc_gate_param_grads: null pointer / kg = %d", kg);
    PHC_REQUIRE(!db || phm_b, PHC_ERR_INVALID_ARG, "phc_gate_param_grads: db without a PHM bias");
    PHC_REQUIRE(kin == kout, PHC_ERR_UNSUPPORTED, "phc_gate_param_grads: a square PHM layer (kin == kout), got %d x %d", kin, kout);
    const int k = kin, h = n * k;
    const int lds_tile = 2 * kg * k + k * (k + 1), lds_row = kg * h + kg * n + n * n * n + n * h;
    const size_t lds = sizeof(float) * (size_t)(lds_tile > lds_row ? lds_tile : lds_row);
    PHC_REQUIRE(lds <= 128 * 1024, PHC_ERR_UNSUPPORTED, "phc_gate_param_grads: gate_w [%d x %d] does not fit the LDS staging", kg, h);
    GateGradArgs g;
    g.dMT = dMT; g.dbg = dbg; g.H = H; g.gate_w = gate_w; g.phm_b = phm_b; g.A = A; g.W = W;
    g.dA = dA; g.dW = dW; g.db = db; g.dWg = dWg;
    g.n = n; g.k = k; g.kg = kg;
    g.sj = h; g.sr = 1; g.reg_add = nullptr; g.dbg_out = nullptr;
    g.blk_rows = n * n;
    g.blk_dWg = g.blk_rows + k;
    g.blk_db = g.blk_dWg + kg * ((h + 31) / 32);
    PHC_REQUIRE(ensure_dynamic_lds(&gate_param_grads_kernel, 128 * 1024), PHC_ERR_LAUNCH,
                "phc_gate_param_grads: the device refused %d bytes of dynamic LDS", 128 * 1024);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_PHM_CONTRACT, s);
    PHC_LAUNCH(gate_param_grads_kernel, dim3((unsigned)(g.blk_db + 1)), dim3(256), lds, s, g);
    return check_launch("gate_param_grads");
}

int phc_gemm_prebuilt(const float *x, const float *M, const float *Mk4, const float *b, const float *residual, int64_t N, int K, int Nout,
                      float *y, int is_dx, void *stream) {
    PHC_REQUIRE(N >= 0 && K >= 1 && Nout >= 1, PHC_ERR_INVALID_ARG, "phc_gemm_prebuilt: bad shape");
    if (N == 0) return PHC_OK;
    PHC_REQUIRE(x && M && y, PHC_ERR_INVALID_ARG, "phc_gemm_prebuilt: null pointer");
    if (hs_shape_ok(x, Mk4, N, K, Nout) && ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(residual)) & 15) == 0) {
        hipStream_t hs = (hipStream_t)stream;
        ProfScope prof(is_dx ? K_PHM_GEMM_DX : K_PHM_GEMM_FWD, hs);
        launch_hs<0>(hs_args(x, Mk4, b, residual, y, N, K, Nout), K, XfAcc{}, hs);
        return check_launch("phc_gemm_prebuilt(hs)");
    }
    PHC_REQUIRE(K <= GEMM_PREBUILT_MAX_K && K % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(M) & 15) == 0,
                PHC_ERR_UNSUPPORTED, "phc_gemm_prebuilt: needs K <= 768, K %% 4 == 0 and 16-byte aligned x, M (K=%d)", K);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(is_dx ? K_PHM_GEMM_DX : K_PHM_GEMM_FWD, s);
    if (K > GEMM_STREAM_MAX_K) {                             // deep contraction: 32-column panels, one workgroup per CU
        launch_stream_cfg<0, -1, 32, 4>(x, nullptr, M, b, residual, y, N, K, Nout, 1, K, Nout, 1, s);
        return check_launch("phc_gemm_prebuilt(deep)");
    }
    // Few rows (molecule batches: about one 32-row slab per wavefront): 32-column panels give twice the wavefronts,
    // each half as long — measured 0.8 us per launch faster than 64-column panels at N = 13.8k; above that the extra
    // re-reads of x cost more than the occupancy gains.
    constexpr int small_bn = 32;
    if (N >= gemm_wide_k_m() && Nout > 64 && K > GEMM_WIDE_K_MIN)
        launch_stream_cfg<0, -1, 64, 8>(x, nullptr, M, b, residual, y, N, K, Nout, 1, K, Nout, 1, s);
    else if (gemm_big_cfg(N, K, Nout))
        launch_stream_cfg<0, -1, 128, 8>(x, nullptr, M, b, residual, y, N, K, Nout, 1, K, Nout, 1, s);
    else if (small_bn == 32 && Nout > 32 && N <= 32 * 1024)
        launch_stream_cfg<0, -1, 32, 4>(x, nullptr, M, b, residual, y, N, K, Nout, 1, K, Nout, 3, s);
    else
        launch_stream_cfg<0, -1, 64, 4>(x, nullptr, M, b, residual, y, N, K, Nout, 1, K, Nout, 2, s);
    return check_launch("phc_gemm_prebuilt");
}

int phc_gemm_stats_max_partials(void) { return GEMM_STATS_MAX_PARTIALS; }
size_t phc_stat_acc_words(int h) { return h > 0 ? (size_t)STAT_ACC_REP * 4 * (size_t)h : 0; }
int phc_stat_acc_replicas(void) { return STAT_ACC_REP; }

int phc_gemm_prebuilt_stats(const float *x, const float *M, const float *Mk4, const float *b, int64_t N, int K, int Nout, float *y,
                            const int32_t *n_valid_dev, float *stats_partial, float *stats_counts, int *num_partials, void *stream) {
    return phc_gemm_prebuilt_stats_res(x, M, Mk4, b, nullptr, N, K, Nout, y, n_valid_dev, stats_partial, stats_counts, num_partials, stream);
}

int phc_gemm_prebuilt_stats_res(const float *x, const float *M, const float *Mk4, const float *b, const float *residual, int64_t N, int K,
                                int Nout, float *y, const int32_t *n_valid_dev, float *stats_partial, float *stats_counts,
                                int *num_partials, void *stream) {
    PHC_REQUIRE(N >= 1 && K >= 1 && Nout >= 1, PHC_ERR_INVALID_ARG, "phc_gemm_prebuilt_stats: bad shape");
    PHC_REQUIRE(!residual || ((reinterpret_cast<uintptr_t>(residual) & 15) == 0 && Nout % 4 == 0), PHC_ERR_UNSUPPORTED,
                "phc_gemm_prebuilt_stats_res: 16-byte aligned residual rows required");
    const int acc = stats_counts ? 0 : 1;                    // accumulator form: stats_partial is a statistics accumulator
    PHC_REQUIRE(x && M && y && stats_partial && (acc || num_partials), PHC_ERR_INVALID_ARG, "phc_gemm_prebuilt_stats: null pointer");
    PHC_REQUIRE(!acc || (reinterpret_cast<uintptr_t>(stats_partial) & 7) == 0, PHC_ERR_INVALID_ARG,
                "phc_gemm_prebuilt_stats: the accumulator must be 8-byte aligned");
    PHC_REQUIRE(K <= GEMM_PREBUILT_MAX_K && K % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(M) & 15) == 0,
                PHC_ERR_UNSUPPORTED, "phc_gemm_prebuilt_stats: needs K <= 768, K %% 4 == 0 and 16-byte aligned x, M (K=%d)", K);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_PHM_GEMM_FWD, s);
    if (acc && hs_shape_ok(x, Mk4, N, K, Nout) && ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(b)) & 15) == 0) {
        GemmHsArgs a = hs_args(x, Mk4, b, residual, y, N, K, Nout);
        a.n_valid = n_valid_dev;
        a.stat_acc = reinterpret_cast<unsigned long long *>(stats_partial);
        launch_hs<HS_ST>(a, K, XfAcc{}, s);
        return check_launch("phc_gemm_prebuilt_stats(hs)");
    }
    constexpr int small_bn = 32;
    if (K > GEMM_STREAM_MAX_K)                               // deep contraction: 32-column panels, one workgroup per CU
        launch_stream_cfg<0, -1, 32, 4, 1>(x, nullptr, M, b, residual, y, N, K, Nout, 1, K, Nout, 1, s, n_valid_dev, stats_partial,
                                           stats_counts, num_partials, nullptr, 0, nullptr, nullptr, 0, acc);
    else if (N >= gemm_wide_k_m() && Nout > 64 && K > GEMM_WIDE_K_MIN)
        launch_stream_cfg<0, -1, 64, 8, 1>(x, nullptr, M, b, residual, y, N, K, Nout, 1, K, Nout, 1, s, n_valid_dev, stats_partial,
                                           stats_counts, num_partials, nullptr, 0, nullptr, nullptr, 0, acc);
    else if (gemm_big_cfg(N, K, Nout))
        launch_stream_cfg<0, -1, 128, 8, 1>(x, nullptr, M, b, residual, y, N, K, Nout, 1, K, Nout, 1, s, n_valid_dev, stats_partial,
                                            stats_counts, num_partials, nullptr, 0, nullptr, nullptr, 0, acc);
    else if (small_bn == 32 && Nout > 32 && N <= 32 * 1024)
        launch_stream_cfg<0, -1, 32, 4, 1>(x, nullptr, M, b, residual, y, N, K, Nout, 1, K, Nout, 3, s, n_valid_dev, stats_partial,
                                           stats_counts, num_partials, nullptr, 0, nullptr, nullptr, 0, acc);
    else
        launch_stream_cfg<0, -1, 64, 4, 1>(x, nullptr, M, b, residual, y, N, K, Nout, 1, K, Nout, 2, s, n_valid_dev, stats_partial,
                                           stats_counts, num_partials, nullptr, 0, nullptr, nullptr, 0, acc);
    return check_launch("phc_gemm_prebuilt_stats");
}

static int gemm_prebuilt_xf_impl(const float *x, const float *xf_stat, int xf_act, const int32_t *n_valid_dev, const float *M,
                                 const float *Mk4, const float *b, int64_t N, int K, int Nout, float *y, float *stats_partial,
                                 float *stats_counts, int *num_partials, void *stream, const XfAcc &xa);

int phc_gemm_prebuilt_xf(const float *x, const float *xf_stat, int xf_act, const int32_t *n_valid_dev, const float *M, const float *Mk4,
                         const float *b, int64_t N, int K, int Nout, float *y, float *stats_partial, float *stats_counts,
                         int *num_partials, void *stream) {
    return gemm_prebuilt_xf_impl(x, xf_stat, xf_act, n_valid_dev, M, Mk4, b, N, K, Nout, y, stats_partial, stats_counts, num_partials,
                                 stream, XfAcc{});
}

int phc_gemm_prebuilt_xf_acc(const float *x, const uint64_t *xf_acc, const float *const *weight, const float *const *bias,
                             float *const *running_mean, float *const *running_var, int64_t *const *num_batches_tracked, int n_comp,
                             float momentum, float eps, float *save_stat, int xf_act, const int32_t *n_valid_dev, const float *M,
                             const float *Mk4, const float *b, int64_t N, int K, int Nout, float *y, float *stats_partial,
                             float *stats_counts, int *num_partials, void *stream) {
    PHC_REQUIRE(xf_acc && save_stat && (reinterpret_cast<uintptr_t>(xf_acc) & 7) == 0, PHC_ERR_INVALID_ARG,
                "phc_gemm_prebuilt_xf_acc: accumulator / save_stat pointer");
    XfAcc xa;
    xa.acc = reinterpret_cast<const unsigned long long *>(xf_acc);
    int rc = fill_params("phc_gemm_prebuilt_xf_acc", &xa.p, weight, bias, running_mean, running_var, num_batches_tracked, n_comp, K);
    if (rc) return rc;
    xa.momentum = momentum;
    xa.eps = eps;
    xa.stat_out = save_stat;
    return gemm_prebuilt_xf_impl(x, save_stat, xf_act, n_valid_dev, M, Mk4, b, N, K, Nout, y, stats_partial, stats_counts, num_partials,
                                 stream, xa);
}

static int gemm_prebuilt_xf_impl(const float *x, const float *xf_stat, int xf_act, const int32_t *n_valid_dev, const float *M,
                                 const float *Mk4, const float *b, int64_t N, int K, int Nout, float *y, float *stats_partial,
                                 float *stats_counts, int *num_partials, void *stream, const XfAcc &xa) {
    PHC_REQUIRE(N >= 1 && K >= 1 && Nout >= 1, PHC_ERR_INVALID_ARG, "phc_gemm_prebuilt_xf: bad shape");
    PHC_REQUIRE(x && xf_stat && M && y, PHC_ERR_INVALID_ARG, "phc_gemm_prebuilt_xf: null pointer");
    PHC_REQUIRE(xf_act >= PHC_ACT_IDENTITY && xf_act <= PHC_ACT_SELU, PHC_ERR_INVALID_ARG, "phc_gemm_prebuilt_xf: unknown activation %d",
                xf_act);
    const int acc = (stats_partial && !stats_counts) ? 1 : 0;        // accumulator form of the statistics output
    PHC_REQUIRE(!stats_partial || acc || num_partials, PHC_ERR_INVALID_ARG, "phc_gemm_prebuilt_xf: statistics outputs");
    PHC_REQUIRE(!acc || (reinterpret_cast<uintptr_t>(stats_partial) & 7) == 0, PHC_ERR_INVALID_ARG,
                "phc_gemm_prebuilt_xf: the accumulator must be 8-byte aligned");
    PHC_REQUIRE(K <= GEMM_STREAM_MAX_K && K % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(M) & 15) == 0 && (reinterpret_cast<uintptr_t>(xf_stat) & 15) == 0,
                PHC_ERR_UNSUPPORTED, "phc_gemm_prebuilt_xf: needs K <= 256, K %% 4 == 0 and 16-byte aligned x, M, stat (K=%d)", K);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_PHM_GEMM_FWD, s);
    if ((!stats_partial || acc) && hs_shape_ok(x, Mk4, N, K, Nout) &&
        ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(xf_stat)) & 15) == 0) {
        GemmHsArgs a = hs_args(x, Mk4, b, nullptr, y, N, K, Nout);
        a.n_valid = n_valid_dev;
        a.xf_stat = xf_stat;
        a.xf_act = xf_act;
        a.stat_acc = reinterpret_cast<unsigned long long *>(stats_partial);
        if (stats_partial)
            launch_hs<HS_XF | HS_ST>(a, K, xa, s);
        else
            launch_hs<HS_XF>(a, K, xa, s);
        return check_launch("phc_gemm_prebuilt_xf(hs)");
    }
    constexpr int small_bn = 32;
    const bool big = gemm_big_cfg(N, K, Nout);
    const bool small = small_bn == 32 && Nout > 32 && N <= 32 * 1024;
#define PHC_XF_LAUNCH(SV)                                                                                                      \
    if (N >= gemm_wide_k_m() && Nout > 64 && K > GEMM_WIDE_K_MIN)                                                                          \
        launch_stream_cfg<0, -1, 64, 8, SV>(x, nullptr, M, b, nullptr, y, N, K, Nout, 1, K, Nout, 1, s, n_valid_dev, stats_partial, \
                                            stats_counts, num_partials, xf_stat, xf_act, nullptr, nullptr, 0, acc, xa);              \
    else if (big)                                                                                                               \
        launch_stream_cfg<0, -1, 128, 8, SV>(x, nullptr, M, b, nullptr, y, N, K, Nout, 1, K, Nout, 1, s, n_valid_dev, stats_partial, \
                                             stats_counts, num_partials, xf_stat, xf_act, nullptr, nullptr, 0, acc, xa);             \
    else if (small)                                                                                                             \
        launch_stream_cfg<0, -1, 32, 4, SV>(x, nullptr, M, b, nullptr, y, N, K, Nout, 1, K, Nout, 3, s, n_valid_dev, stats_partial, \
                                            stats_counts, num_partials, xf_stat, xf_act, nullptr, nullptr, 0, acc, xa);              \
    else                                                                                                                        \
        launch_stream_cfg<0, -1, 64, 4, SV>(x, nullptr, M, b, nullptr, y, N, K, Nout, 1, K, Nout, 2, s, n_valid_dev, stats_partial, \
                                            stats_counts, num_partials, xf_stat, xf_act, nullptr, nullptr, 0, acc, xa)
    if (stats_partial) {
        PHC_XF_LAUNCH(3);
    } else {
        PHC_XF_LAUNCH(2);
    }
#undef PHC_XF_LAUNCH
    return check_launch("phc_gemm_prebuilt_xf");
}

int phc_gemm_prebuilt_bwsums(const float *dy, const float *M, const float *Mk4, int64_t N, int K, int Nout, float *dz, const float *bn_x,
                             const float *bn_stat, int act, const int32_t *n_valid_dev, float *sums_partial, int *num_partials,
                             void *stream) {
    PHC_REQUIRE(N >= 1 && K >= 1 && Nout >= 1, PHC_ERR_INVALID_ARG, "phc_gemm_prebuilt_bwsums: bad shape");
    const int acc = num_partials ? 0 : 1;                    // accumulator form: sums_partial is a statistics accumulator
    PHC_REQUIRE(dy && M && dz && bn_x && bn_stat && sums_partial, PHC_ERR_INVALID_ARG, "phc_gemm_prebuilt_bwsums: null pointer");
    PHC_REQUIRE(!acc || (reinterpret_cast<uintptr_t>(sums_partial) & 7) == 0, PHC_ERR_INVALID_ARG,
                "phc_gemm_prebuilt_bwsums: the accumulator must be 8-byte aligned");
    PHC_REQUIRE(act >= PHC_ACT_IDENTITY && act <= PHC_ACT_SELU, PHC_ERR_INVALID_ARG, "phc_gemm_prebuilt_bwsums: unknown activation %d", act);
    PHC_REQUIRE(K <= GEMM_STREAM_MAX_K && K % 4 == 0 && Nout % 4 == 0 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(M) & 15) == 0 && (reinterpret_cast<uintptr_t>(dz) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(bn_x) & 15) == 0 && (reinterpret_cast<uintptr_t>(bn_stat) & 15) == 0,
                PHC_ERR_UNSUPPORTED, "phc_gemm_prebuilt_bwsums: needs K <= 256, K %% 4 == 0, Nout %% 4 == 0, 16-byte aligned operands");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_PHM_GEMM_DX, s);
    if (acc && hs_shape_ok(dy, Mk4, N, K, Nout)) {
        GemmHsArgs a = hs_args(dy, Mk4, nullptr, nullptr, dz, N, K, Nout);
        a.n_valid = n_valid_dev;
        a.stat_acc = reinterpret_cast<unsigned long long *>(sums_partial);
        a.bw_x = bn_x;
        a.bw_stat = bn_stat;
        a.bw_act = act;
        launch_hs<HS_BW>(a, K, XfAcc{}, s);
        return check_launch("phc_gemm_prebuilt_bwsums(hs)");
    }
    constexpr int small_bn = 32;
    if (N >= gemm_wide_k_m() && Nout > 64 && K > GEMM_WIDE_K_MIN)
        launch_stream_cfg<0, -1, 64, 8, 4>(dy, nullptr, M, nullptr, nullptr, dz, N, K, Nout, 1, K, Nout, 1, s, n_valid_dev, sums_partial,
                                           nullptr, num_partials, nullptr, 0, bn_x, bn_stat, act, acc);
    else if (gemm_big_cfg(N, K, Nout))
        launch_stream_cfg<0, -1, 128, 8, 4>(dy, nullptr, M, nullptr, nullptr, dz, N, K, Nout, 1, K, Nout, 1, s, n_valid_dev, sums_partial,
                                            nullptr, num_partials, nullptr, 0, bn_x, bn_stat, act, acc);
    else if (small_bn == 32 && Nout > 32 && N <= 32 * 1024)
        launch_stream_cfg<0, -1, 32, 4, 4>(dy, nullptr, M, nullptr, nullptr, dz, N, K, Nout, 1, K, Nout, 3, s, n_valid_dev, sums_partial,
                                           nullptr, num_partials, nullptr, 0, bn_x, bn_stat, act, acc);
    else
        launch_stream_cfg<0, -1, 64, 4, 4>(dy, nullptr, M, nullptr, nullptr, dz, N, K, Nout, 1, K, Nout, 2, s, n_valid_dev, sums_partial,
                                           nullptr, num_partials, nullptr, 0, bn_x, bn_stat, act, acc);
    return check_launch("phc_gemm_prebuilt_bwsums");
}

// The norm backward's element-wise pass folded into the dx GEMM behind it (HS_BA), optionally with the sums epilogue of the
// norm in front of that GEMM (HS_BW): see include/phc_hip.h.
static inline bool agw_enabled() {
    static const bool off = getenv("PHC_GEMM_AGGREGATE_WIDE") && atoi(getenv("PHC_GEMM_AGGREGATE_WIDE")) == 0;
    return !off;
}
int phc_gemm_aggregate_supported(int64_t N, int h, int Nout, int F, int vocab_rows) {
    static const bool off = getenv("PHC_GEMM_AGGREGATE") && atoi(getenv("PHC_GEMM_AGGREGATE")) == 0;
    const bool shape = (h == 128 && Nout == 128) || (agw_enabled() && gemm_agw_shape_ok(h, Nout, vocab_rows) && hs_widths_ok(h, Nout));
    return (!off && hs_enabled() && N >= 1 && N <= (1 << 20) && shape && F >= 1 && F <= HS_AG_MAX_F && vocab_rows >= 1 &&
            vocab_rows <= HS_AG_MAX_ROWS) ? 1 : 0;
}

int phc_gemm_aggregate_stats(const phc_gemm_ag_desc *d, void *stream) {
    PHC_REQUIRE(d, PHC_ERR_INVALID_ARG, "phc_gemm_aggregate_stats: null descriptor");
    PHC_REQUIRE(d->n >= 1 && d->k >= 1 && d->vocab && d->F >= 1 && d->F <= HS_AG_MAX_F, PHC_ERR_INVALID_ARG,
                "phc_gemm_aggregate_stats: bad shape (n %d, k %d, F %d)", d->n, d->k, d->F);
    const int h = d->n * d->k;
    HsAgArgs ag = {};
    ag.F = d->F;
    ag.off[0] = 0;
    for (int f = 0; f < d->F; ++f) {
        PHC_REQUIRE(d->vocab[f] >= 1, PHC_ERR_INVALID_ARG, "phc_gemm_aggregate_stats: vocab[%d] = %d", f, d->vocab[f]);
        ag.off[f + 1] = ag.off[f] + d->vocab[f];
    }
    PHC_REQUIRE(phc_gemm_aggregate_supported(d->N, h, d->Nout, d->F, ag.off[d->F]) && d->k % 4 == 0 && d->n * d->F <= PHC_MAX_COMPONENTS * 4,
                PHC_ERR_UNSUPPORTED, "phc_gemm_aggregate_stats: %lld rows, h = %d, Nout = %d, F = %d, sum(vocab) = %d outside the fused form's range",
                (long long)d->N, h, d->Nout, d->F, ag.off[d->F]);
    PHC_REQUIRE(d->x && d->codes && d->tables && d->rowptr && d->col && d->s_out && d->Mk4 && d->y && d->stat_acc, PHC_ERR_INVALID_ARG,
                "phc_gemm_aggregate_stats: null pointer");
    const bool normed = d->norm_acc != nullptr;
    PHC_REQUIRE(!normed || (d->save_stat && d->x_out), PHC_ERR_INVALID_ARG, "phc_gemm_aggregate_stats: a pending norm needs save_stat and x_out");
    PHC_REQUIRE(!normed || (d->norm_act >= PHC_ACT_IDENTITY && d->norm_act <= PHC_ACT_SELU), PHC_ERR_INVALID_ARG,
                "phc_gemm_aggregate_stats: unknown activation %d", d->norm_act);
    PHC_REQUIRE(((reinterpret_cast<uintptr_t>(d->x) | reinterpret_cast<uintptr_t>(d->s_out) | reinterpret_cast<uintptr_t>(d->Mk4) |
                  reinterpret_cast<uintptr_t>(d->y) | reinterpret_cast<uintptr_t>(d->b) |
                  (normed ? (reinterpret_cast<uintptr_t>(d->x_out) | reinterpret_cast<uintptr_t>(d->skip)) : 0)) & 15) == 0 &&
                    ((reinterpret_cast<uintptr_t>(d->stat_acc) | reinterpret_cast<uintptr_t>(d->norm_acc)) & 7) == 0,
                PHC_ERR_UNSUPPORTED, "phc_gemm_aggregate_stats: 16-byte aligned rows / 8-byte aligned accumulators");
    if (normed) {
        int rc = fill_params("phc_gemm_aggregate_stats", &ag.p, d->weight, d->bias, d->running_mean, d->running_var, d->num_batches_tracked,
                             d->n_comp, h);
        if (rc) return rc;
        ag.acc = reinterpret_cast<const unsigned long long *>(d->norm_acc);
        ag.skip = d->skip; ag.x_out = d->x_out; ag.stat = d->save_stat;
        ag.momentum = d->momentum; ag.eps = d->eps; ag.norm_act = d->norm_act;
    }
    ag.y = d->x; ag.codes = d->codes; ag.tables = d->tables; ag.rowptr = d->rowptr; ag.col = d->col;
    ag.s_out = d->s_out; ag.n = d->n; ag.k = d->k; ag.add_self = d->add_self ? 1 : 0;
    GemmHsArgs a = hs_args(nullptr, d->Mk4, d->b, nullptr, d->y, d->N, h, d->Nout);
    a.n_valid = d->n_valid_dev;
    a.stat_acc = reinterpret_cast<unsigned long long *>(d->stat_acc);
    a.panels = 1;
    a.stamps = g_hs_stamps;
    hipStream_t s = (hipStream_t)stream;
    const long long steps = (d->N + 15) / 16;
    if (h != 128) {
        // the wide layers (gemm_hs_agw_device.h): one workgroup per CU, as launch_hs_cfg<K, 2, 8> counts them
        constexpr int wide_per_cu_env = 0;
        long long groups = (long long)(wide_per_cu_env > 0 ? wide_per_cu_env : 1) * num_cu();
        if (groups > steps) groups = steps;
        a.groups = (int)groups;
        ProfScope prof(K_PHM_GEMM_AG, s);
        int rc = launch_gemm_agw(a, ag, normed, s);
        if (rc) return rc;
        return check_launch("phc_gemm_aggregate_stats");
    }
    // workgroups: as many as the statistics-epilogue GEMM on the same rows takes (two per CU) — a workgroup's steps, and with them
    // the partial sums it adds to the accumulator, are then the same sets of rows: the totals are bit-identical to that launch's
    constexpr int per_cu_env = 0;
    long long groups = (long long)(per_cu_env > 0 ? per_cu_env : 2) * num_cu();
    if (groups > steps) groups = steps;
    a.groups = (int)groups;
    const size_t tbl_bytes = (size_t)(128 + ag.off[d->F]) * 128 * sizeof(float);     // the operand (k4 form) + the bond tables in LDS
    ProfScope prof(K_PHM_GEMM_AG, s);
#define PHC_AG_LAUNCH1(FC, NC, RC, LC)                                                                                             \
    do {                                                                                                                           \
        PHC_REQUIRE(ensure_dynamic_lds(&gemm_hs_ag_kernel<FC, NC, RC, LC>, (128 + HS_AG_MAX_ROWS) * 128 * sizeof(float)), PHC_ERR_LAUNCH, \
                    "phc_gemm_aggregate_stats: the device refused the dynamic LDS");                                              \
        PHC_LAUNCH((gemm_hs_ag_kernel<FC, NC, RC, LC>), dim3((unsigned)groups), dim3(HS_AG_THREADS), tbl_bytes, s, a, ag);        \
    } while (0)
    // (the norm parameters' layout is a template parameter: a run-time branch around the loads costs a full drain at its join)
#define PHC_AG_LAUNCH(FC)                                   \
    if (normed && ag.norm_act == PHC_ACT_RELU && ag.p.flat) \
        PHC_AG_LAUNCH1(FC, true, true, true);               \
    else if (normed && ag.norm_act == PHC_ACT_RELU)         \
        PHC_AG_LAUNCH1(FC, true, true, false);              \
    else if (normed && ag.p.flat)                           \
        PHC_AG_LAUNCH1(FC, true, false, true);              \
    else if (normed)                                        \
        PHC_AG_LAUNCH1(FC, true, false, false);             \
    else                                                    \
        PHC_AG_LAUNCH1(FC, false, true, true)
    switch (d->F) {
        case 1: PHC_AG_LAUNCH(1); break;
        case 2: PHC_AG_LAUNCH(2); break;
        case 3: PHC_AG_LAUNCH(3); break;
        default: PHC_AG_LAUNCH(4); break;
    }
#undef PHC_AG_LAUNCH
#undef PHC_AG_LAUNCH1
    return check_launch("phc_gemm_aggregate_stats");
}

int phc_gemm_hs_enabled(void) { return hs_enabled() ? 1 : 0; }
int phc_gemm_hs_set_enabled(int on) {
    const int prev = hs_enabled() ? 1 : 0;
    __atomic_store_n(&g_hs_enabled, on ? 1 : 0, __ATOMIC_RELAXED);
    return prev;
}

int phc_debug_hs_stamps(unsigned long long *dev_words) {
    g_hs_stamps = dev_words;
    return PHC_OK;
}

int phc_gemm_bwd_apply_supported(int K, int Nout) {
    return (hs_enabled() && hs_widths_ok(K, Nout)) ? 1 : 0;
}

int phc_gemm_bwd_apply(const phc_gemm_ba_desc *d, void *stream) {
    PHC_REQUIRE(d, PHC_ERR_INVALID_ARG, "phc_gemm_bwd_apply: null descriptor");
    PHC_REQUIRE(d->N >= 1 && phc_gemm_bwd_apply_supported(d->K, d->Nout), PHC_ERR_UNSUPPORTED,
                "phc_gemm_bwd_apply: N=%lld K=%d Nout=%d outside the fused form's range (K, Nout in {128, 256}, 180 x 180, 200 x 200)", (long long)d->N, d->K, d->Nout);
    PHC_REQUIRE(d->dout && d->bn_x && d->bn_stat && d->sums_acc && d->dz && d->Mk4 && d->y, PHC_ERR_INVALID_ARG, "phc_gemm_bwd_apply: null pointer");
    PHC_REQUIRE(d->act >= PHC_ACT_IDENTITY && d->act <= PHC_ACT_SELU, PHC_ERR_INVALID_ARG, "phc_gemm_bwd_apply: unknown activation %d", d->act);
    PHC_REQUIRE(d->count == 0 || d->count >= d->N, PHC_ERR_INVALID_ARG, "phc_gemm_bwd_apply: count=%lld < N=%lld", (long long)d->count, (long long)d->N);
    const bool bw = d->bw_sums_acc != nullptr;
    PHC_REQUIRE(!bw || (d->bw_x && d->bw_stat && d->bw_act >= PHC_ACT_IDENTITY && d->bw_act <= PHC_ACT_SELU), PHC_ERR_INVALID_ARG,
                "phc_gemm_bwd_apply: second-stage sums need bw_x, bw_stat and an activation");
    PHC_REQUIRE(((reinterpret_cast<uintptr_t>(d->dout) | reinterpret_cast<uintptr_t>(d->bn_x) | reinterpret_cast<uintptr_t>(d->bn_stat) |
                  reinterpret_cast<uintptr_t>(d->dz) | reinterpret_cast<uintptr_t>(d->Mk4) | reinterpret_cast<uintptr_t>(d->y) |
                  reinterpret_cast<uintptr_t>(d->bw_x) | reinterpret_cast<uintptr_t>(d->bw_stat)) & 15) == 0 &&
                    ((reinterpret_cast<uintptr_t>(d->sums_acc) | reinterpret_cast<uintptr_t>(d->bw_sums_acc)) & 7) == 0,
                PHC_ERR_UNSUPPORTED, "phc_gemm_bwd_apply: 16-byte aligned rows / 8-byte aligned accumulators required");
    hipStream_t s = (hipStream_t)stream;
    GemmHsArgs a = hs_args(d->dout, d->Mk4, nullptr, nullptr, d->y, d->N, d->K, d->Nout);
    a.n_valid = d->n_valid_dev;
    a.ba_x = d->bn_x;
    a.ba_stat = d->bn_stat;
    a.ba_acc = reinterpret_cast<const unsigned long long *>(d->sums_acc);
    a.ba_act = d->act;
    a.ba_dz = d->dz;
    a.ba_dwb = d->dwb;
    a.ba_count = d->count;
    // ABI 6: dropout behind the norm's activation — the mask multiplies the upstream gradient inside the transform (HS_DR)
    int rc = make_drop("phc_gemm_bwd_apply", &a.ba_drop, d->drop_p, d->drop_same_k, d->drop_site, d->rng_state, d->K);
    if (rc) return rc;
    const bool dr = a.ba_drop.p > 0.f;
    PHC_REQUIRE(!dr || (d->K == d->Nout && (d->K == 128 || d->K == 256 || d->K == 180 || d->K == 200)), PHC_ERR_UNSUPPORTED,
                "phc_gemm_bwd_apply: the dropout form serves square layers of width 128 / 256 / 180 / 200 (K = %d, Nout = %d)", d->K, d->Nout);
    ProfScope prof(K_PHM_GEMM_DX, s);
    if (bw) {
        a.stat_acc = reinterpret_cast<unsigned long long *>(d->bw_sums_acc);
        a.bw_x = d->bw_x;
        a.bw_stat = d->bw_stat;
        a.bw_act = d->bw_act;
        if (dr)
            launch_hs_square<HS_BA | HS_BW | HS_DR>(a, d->K, s);
        else
            launch_hs<HS_BA | HS_BW>(a, d->K, XfAcc{}, s);
    } else if (dr) {
        launch_hs_square<HS_BA | HS_DR>(a, d->K, s);
    } else {
        launch_hs<HS_BA>(a, d->K, XfAcc{}, s);
    }
    return check_launch("phc_gemm_bwd_apply");
}

// ---- batched parameter gradients ------------------------------------------------------------
size_t phc_phm_param_grads_ws_bytes(const phc_phm_grad_job *jobs, int num_jobs) {
    if (check_grad_jobs("phc_phm_param_grads_ws_bytes", jobs, num_jobs)) return 0;
    size_t total = 0;
    for (int c0 = 0; c0 < num_jobs; c0 += GRAD_JOBS_PER_LAUNCH)
        total += grad_chunk_ws_floats(jobs + c0, num_jobs - c0 < GRAD_JOBS_PER_LAUNCH ? num_jobs - c0 : GRAD_JOBS_PER_LAUNCH);
    return sizeof(float) * total;
}

int phc_phm_param_grads(const phc_phm_grad_job *jobs, int num_jobs, void *ws, size_t ws_bytes, void *stream) {
    int rc = check_grad_jobs("phc_phm_param_grads", jobs, num_jobs);
    if (rc) return rc;
    if (num_jobs == 0) return PHC_OK;
    const size_t need = phc_phm_param_grads_ws_bytes(jobs, num_jobs);
    PHC_REQUIRE(ws && ws_bytes >= need, PHC_ERR_WORKSPACE, "phc_phm_param_grads: workspace %zu B < required %zu B", ws_bytes, need);
    PHC_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 15) == 0, PHC_ERR_WORKSPACE, "phc_phm_param_grads: workspace not 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    float *base = static_cast<float *>(ws);
    for (int c0 = 0; c0 < num_jobs; c0 += GRAD_JOBS_PER_LAUNCH) {
        const int cn = num_jobs - c0 < GRAD_JOBS_PER_LAUNCH ? num_jobs - c0 : GRAD_JOBS_PER_LAUNCH;
        GradLaunch gl;
        rc = prepare_grad_launch("phc_phm_param_grads", jobs + c0, cn, base, &gl);
        if (rc) return rc;
        base += grad_chunk_ws_floats(jobs + c0, cn);
        rc = launch_grad_rest("phc_phm_param_grads", gl, true, s);
        if (rc) return rc;
    }
    return PHC_OK;
}

}  // extern "C"
