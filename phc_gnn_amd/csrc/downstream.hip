// The downstream prediction head as ONE launch each way (second half).
//
// Reference: PHMDownstreamNet.forward (phc/hypercomplex/downstream.py:90-120) — PHMLinear -> norm -> act -> PHMLinear ->
// norm -> act -> PHMLinear (-> n*target) — followed by RealTransformer 'linear' (layers.py:399-414): B rows only (one per
// graph), i.e. ~22 MFLOP that used to be 6 + 8 latency-bound launches (GEMM, norm merge, GEMM, norm merge, GEMM, small
// linear; and the same backwards).  Here a workgroup owns 32 rows for the whole chain:
//   * all operand matrices (prebuilt H / H^T, phc_phm_build_H_batched) are staged into LDS once, next to the 32-row slab;
//   * the three GEMMs run on the matrix cores (v_mfma_f32_32x32x2_f32) straight from LDS, activations never leave the CU;
//   * batch-norm needs statistics over ALL rows: every workgroup publishes its (sum, M2) pair per column with write-through
//     (sc1) stores, arrives at a device-scope counter and, once all have arrived, merges the G pairs in workgroup order
//     (Chan's formula; the same order on every workgroup => identical statistics, bitwise reproducible).  Two such
//     exchanges per direction.  The G = ceil(B / 32) <= 64 workgroups must be co-resident: the host checks the device's real
//     capacity (occupancy for this launch's LDS size x the CU count of the — possibly partitioned — device; a CU mask in the
//     environment disqualifies) in phc_downstream_supported / every launch and answers PHC_ERR_UNSUPPORTED otherwise, and the
//     poll loops are bounded (DN_SPIN_MAX -> PHC_ERR_RENDEZVOUS in the device error word) so that even an unforeseen
//     scheduling stall ends the launch instead of hanging the GPU.  The hand-off is MI355X_MICROARCH.md's data-tagged granule: every value travels as ONE
//     naturally aligned 8-byte {value, tag} written by one sc1 (write-through) store and read by sc1 loads that are
//     repeated until every tag equals this launch's; no flag, no counter, no fence, no L2 write-back — one memory round
//     trip after the last producer's store has landed (a counter + payload hand-off measured 4.2 + 3.5 us per exchange
//     here, profiles/r02_downstream_phases.txt).
//   * the tag is (generation + 1): a word that workgroup 0 bumps at the end of every launch (all workgroups read it at
//     their start, and workgroup 0 cannot get to the bump before every other one has published, i.e. has read it).
// Padding rows of a fixed-shape batch (rows >= *n_valid) stay out of the statistics, read as 0 after each norm and get zero
// gradient, as in norm.hip.
#include <stdlib.h>

#include "phc_common.h"

namespace phc {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int DN_THREADS = 256;
constexpr int DN_ROWS = 32;
constexpr int DN_MAXW = 128;                         // widest layer handled (d0, d1, d2)
constexpr int DN_MAXD3 = 32;                         // n * target
constexpr int DN_MAXT = 8;
constexpr int DN_EXTRA = DN_MAXT * DN_MAXD3 + DN_MAXT;          // dWr | dbr partials ride with the first backward rendezvous
constexpr int DN_PSTRIDE = 2 * DN_MAXW + DN_EXTRA;   // floats per workgroup per rendezvous
constexpr int DN_MAX_WG = 64;                        // B <= 2048

struct DnNorm {
    const float *weight[PHC_MAX_COMPONENTS];
    const float *bias[PHC_MAX_COMPONENTS];
    float *rmean[PHC_MAX_COMPONENTS];
    float *rvar[PHC_MAX_COMPONENTS];
    long long *nbt[PHC_MAX_COMPONENTS];
    int n_comp, k, affine;
    float momentum, eps;
};

struct DnArgs {
    int B, d0, d1, d2, d3, T, act, G;
    const int *n_valid;
    // forward operands: H* row-major [K][N]; backward operands: H*T row-major [N][K] (= H^T)
    const float *x, *H1, *b1, *H2, *b2, *H3, *b3, *H1T, *H2T, *H3T, *Wr, *br;
    DnNorm bn1, bn2;
    float *z1, *z2, *z3, *logits, *stat1, *stat2;             // forward outputs (z*, stat* are what the backward reads)
    const float *dlogits;
    float *dx, *dz1, *dz2, *dz3, *dwb1, *dwb2, *dWr, *dbr;    // backward outputs
    unsigned long long *partial;                              // [2][G][DN_PSTRIDE] {value, tag} granules
    unsigned *sync;                                           // generation word
    int *err;                                                 // device error word (phc_set_error_word) or NULL
};

// A rendezvous that never completes (a producer workgroup that is not resident: CU mask, another stream owning the CUs, a
// partitioned device the capacity query did not see) must not hang the GPU: every poll loop gives up after DN_SPIN_MAX
// rounds, raises PHC_ERR_RENDEZVOUS in the device error word and lets the launch run to its end (its outputs are then
// garbage; the host raises at its next check point).  2^21 rounds of >= 1 us each: seconds, never reached by a live exchange.
constexpr int DN_SPIN_MAX = 1 << 21;

#ifdef PHC_DN_TIMING
// experiment build only: workgroup 0 / G-1 stamp the phases of the forward launch (100 MHz wall clock)
__device__ unsigned long long g_dn_ts[64];
#define DN_STAMP(i)                                                                              \
    do {                                                                                         \
        if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1))                \
            g_dn_ts[(blockIdx.x == 0 ? 0 : 32) + (i)] = wall_clock64();                          \
    } while (0)
#else
#define DN_STAMP(i)
#endif

// {value, tag} granule: one 8-byte write-through store / one 8-byte L1-bypassing load
__device__ __forceinline__ void put_granule(unsigned long long *p, float v, unsigned tag) {
    __hip_atomic_store(p, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long get_granule(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr int DN_MERGE = 32;                                 // granule pairs in flight per thread while collecting

// Collect column `col` of partial vectors A (offset 0) and Bv (offset DN_MAXW) of workgroups g0 .. g0+DN_MERGE-1 (clamped to
// G-1): all loads of a round in flight together, repeated until every tag matches.
__device__ __forceinline__ void collect_pairs(const unsigned long long *part, int G, int g0, int col, unsigned tag, float (&va)[DN_MERGE],
                                              float (&vb)[DN_MERGE], int *err) {
    bool ok;
    int spins = 0;
    do {
        ok = true;
#pragma unroll
        for (int u = 0; u < DN_MERGE; ++u) {
            const int gg = min(g0 + u, G - 1);
            const unsigned long long qa = get_granule(part + (size_t)gg * DN_PSTRIDE + col);
            const unsigned long long qb = get_granule(part + (size_t)gg * DN_PSTRIDE + DN_MAXW + col);
            va[u] = __uint_as_float((unsigned)qa);
            vb[u] = __uint_as_float((unsigned)qb);
            ok = ok && (unsigned)(qa >> 32) == tag && (unsigned)(qb >> 32) == tag;
        }
        if (!ok) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > DN_SPIN_MAX) {
                if (err) atomicOr(err, PHC_ERR_RENDEZVOUS);
                // a launch whose exchange timed out must not look like a result: the statistics become NaN, and with them the
                // logits / gradients of every row (the loss turns NaN at once); the running statistics are left alone (fwd_norm)
#pragma unroll
                for (int u = 0; u < DN_MERGE; ++u) va[u] = vb[u] = __int_as_float(0x7fc00000);
                break;
            }
        }
    } while (!ok);
}

// One 32 x 32 output tile (column tile ct) over k in [k0, k1) (multiples of 8): A in LDS [32][lda], Hs in LDS [K][ldh].
__device__ __forceinline__ f32x16 tile_mfma(const float *A, int lda, const float *Hs, int ldh, int ct, int k0, int k1, int lane) {
    const int r = lane & 31, hh = lane >> 5;
    // two independent accumulator chains (a dependent 32x32x2 MFMA waits for its predecessor's 16 passes; alternating chains
    // keeps the pipe full), operands of the next k-group read from LDS while the current MFMAs run
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.f;
    const float *arow = A + r * lda + 4 * hh;
    const float *hcol = Hs + (size_t)(4 * hh) * ldh + ct * 32 + r;
    float4 av = *reinterpret_cast<const float4 *>(arow + k0);
    const float *hb = hcol + (size_t)k0 * ldh;
    float b0 = hb[0], b1 = hb[ldh], b2 = hb[2 * ldh], b3 = hb[3 * ldh];
    for (int kg = k0; kg < k1; kg += 8) {
        const int kn = min(kg + 8, k1 - 8);                     // clamped prefetch of the next group (re-reads the last one)
        const float4 an = *reinterpret_cast<const float4 *>(arow + kn);
        const float *hn = hcol + (size_t)kn * ldh;
        const float n0 = hn[0], n1 = hn[ldh], n2 = hn[2 * ldh], n3 = hn[3 * ldh];
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b1, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b2, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b3, acc1, 0, 0, 0);
        av = an; b0 = n0; b1 = n1; b2 = n2; b3 = n3;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) acc0[i] += acc1[i];
    return acc0;
}

// out[32][N] (LDS, leading dimension ldo) = A[32][K] . Hs[K][N] (+ bias).  N in {32, 64, 128}: tiles = N / 32 column tiles,
// the 4 wavefronts cover tiles x (4 / tiles) k-splits, partial tiles meet in `red` ([4][32][33]).  Ends with a barrier.
__device__ __forceinline__ void wg_gemm(const float *A, int lda, const float *Hs, int ldh, int K, int N, const float *bias, float *red,
                                        float *out, int ldo) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles = N >> 5, ks_n = 4 / tiles;
    const int tile = wave % tiles, ks = wave / tiles;
    const int kspan = K / ks_n;
    const f32x16 acc = tile_mfma(A, lda, Hs, ldh, tile, ks * kspan, (ks + 1) * kspan, lane);
    const int r = lane & 31, hh = lane >> 5;
    float *mine = red + wave * (32 * 33);
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) mine[((reg & 3) + 8 * (reg >> 2) + 4 * hh) * 33 + r] = acc[reg];
    __syncthreads();
    const int lgN = 31 - __clz(N);
    for (int e = tid; e < DN_ROWS * N; e += DN_THREADS) {
        const int row = e >> lgN, col = e & (N - 1);
        const int t = col >> 5, c = col & 31;
        float v = bias ? bias[col] : 0.f;
        for (int s = 0; s < ks_n; ++s) v += red[(s * tiles + t) * (32 * 33) + row * 33 + c];
        out[row * ldo + col] = v;
    }
    __syncthreads();
}

// ---- staging: every global load of a launch phase is issued before the first one is consumed --------------------------
// (hipcc drains all loads in flight at a loop header, so "load, store to LDS" loops pay one memory round trip per
// iteration: 16 of them for a 128 x 128 operand matrix.  The loads below are unconditional (clamped index) and fully
// unrolled; the LDS writes follow in a second unrolled pass.)
template <int U>
__device__ __forceinline__ void ld_batch4(const float *__restrict__ src, int total4, float4 (&v)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int e = min((int)threadIdx.x + u * DN_THREADS, max(total4 - 1, 0));
        v[u] = reinterpret_cast<const float4 *>(src)[e];
    }
}
template <int U>
__device__ __forceinline__ void ld_batch1(const float *__restrict__ src, int total, float (&v)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = src ? src[min((int)threadIdx.x + u * DN_THREADS, max(total - 1, 0))] : 0.f;
}
// flat row-major [K][N] (N = 4 * n4) -> LDS with leading dimension ld
template <int U>
__device__ __forceinline__ void st_matrix4(const float4 (&v)[U], int total4, int n4, float *dst, int ld) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int e = (int)threadIdx.x + u * DN_THREADS;
        if (e < total4) {
            const int k = e / n4, c4 = e - k * n4;
            *reinterpret_cast<float4 *>(dst + k * ld + c4 * 4) = v[u];           // ld % 4 == 0
        }
    }
}
// 32 rows of width w = 4 * w4 starting at row rb (rows >= B become 0) -> LDS [32][ld] (ld % 4 == 0)
template <int U>
__device__ __forceinline__ void st_rows4(const float4 (&v)[U], int w4, int rows_here, float *dst, int ld) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int e = (int)threadIdx.x + u * DN_THREADS;
        if (e < DN_ROWS * w4) {
            const int row = e / w4, c4 = e - row * w4;
            *reinterpret_cast<float4 *>(dst + row * ld + c4 * 4) = row < rows_here ? v[u] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}
template <int U>
__device__ __forceinline__ void st_flat1(const float (&v)[U], int total, float *dst) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int e = (int)threadIdx.x + u * DN_THREADS;
        if (e < total) dst[e] = v[u];
    }
}

__device__ __forceinline__ void store_rows(const float *src, int ld, int B, int rb, int w, float *__restrict__ dst) {
    const int w4 = w >> 2;                                   // w % 4 == 0 (widths are multiples of 32)
    for (int e = threadIdx.x; e < DN_ROWS * w4; e += DN_THREADS) {
        const int row = e / w4, c4 = e - row * w4;
        if (rb + row < B)
            *reinterpret_cast<float4 *>(dst + (size_t)(rb + row) * w + c4 * 4) = *reinterpret_cast<const float4 *>(src + row * ld + c4 * 4);
    }
}

__host__ __device__ __forceinline__ int r4(int v) { return (v + 3) & ~3; }          // LDS segments start 16-byte aligned

__device__ __forceinline__ int rows_counted(int nvalid, int g) { return max(0, min(DN_ROWS, nvalid - g * DN_ROWS)); }

// act(x * scale + shift) over the 32 x N slab Z -> Aout (rows >= n_valid: 0); N is a power of two >= 32: float4, shifts
__device__ __forceinline__ void apply_norm_act(const float *Z, float *Aout, int ld, int N, int lgN, const float *sc, int KM, int act, int rb,
                                               int nvalid) {
    const int n4 = N >> 2, lg4 = lgN - 2;
    for (int e = threadIdx.x; e < DN_ROWS * n4; e += DN_THREADS) {
        const int row = e >> lg4, c = (e & (n4 - 1)) << 2;
        const float4 z = *reinterpret_cast<const float4 *>(Z + row * ld + c);
        const float4 s4 = *reinterpret_cast<const float4 *>(sc + c), h4 = *reinterpret_cast<const float4 *>(sc + KM + c);
        float4 o;
        if (act == PHC_ACT_RELU) {
            o.x = fmaxf(fmaf(z.x, s4.x, h4.x), 0.f); o.y = fmaxf(fmaf(z.y, s4.y, h4.y), 0.f);
            o.z = fmaxf(fmaf(z.z, s4.z, h4.z), 0.f); o.w = fmaxf(fmaf(z.w, s4.w, h4.w), 0.f);
        } else {
            o.x = act_fwd_rt(act, fmaf(z.x, s4.x, h4.x)); o.y = act_fwd_rt(act, fmaf(z.y, s4.y, h4.y));
            o.z = act_fwd_rt(act, fmaf(z.z, s4.z, h4.z)); o.w = act_fwd_rt(act, fmaf(z.w, s4.w, h4.w));
        }
        if (rb + row >= nvalid) o = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4 *>(Aout + row * ld + c) = o;
    }
    __syncthreads();
}

// Forward norm over all rows of z (this workgroup's slab is in Z, [32][ld]): publish this workgroup's (sum, M2) per column,
// collect everybody's, merge in workgroup order, write scale / shift (sc[0..N) / sc[KM..KM+N)) to LDS; workgroup 0 also writes
// save_stat and the running statistics.  w / b: the norm's affine parameters of this thread's column, rm_old / rv_old its
// running statistics (loaded at kernel start).
__device__ __forceinline__ void fwd_norm(const DnArgs &a, const DnNorm &bn, const float *Z, int ld, int N, int nvalid,
                                         unsigned long long *part, unsigned tag, float *stat, float *sc, int KM, float w, float b,
                                         float rm_old, float rv_old, int stamp) {
    const int tid = threadIdx.x, g = blockIdx.x;
    if (tid < N) {
        const int nv = rows_counted(nvalid, g);
        float s = 0.f;
        for (int r = 0; r < nv; ++r) s += Z[r * ld + tid];
        const float mean_l = nv > 0 ? s / (float)nv : 0.f;
        float m2 = 0.f;
        for (int r = 0; r < nv; ++r) {
            const float d = Z[r * ld + tid] - mean_l;
            m2 = fmaf(d, d, m2);
        }
        put_granule(part + (size_t)g * DN_PSTRIDE + tid, s, tag);
        put_granule(part + (size_t)g * DN_PSTRIDE + DN_MAXW + tid, m2, tag);
        DN_STAMP(stamp);
        float n = 0.f, mean = 0.f, M2 = 0.f;
        for (int g0 = 0; g0 < a.G; g0 += DN_MERGE) {
            float ps[DN_MERGE], pm[DN_MERGE];
            collect_pairs(part, a.G, g0, tid, tag, ps, pm, a.err);
#pragma unroll
            for (int u = 0; u < DN_MERGE; ++u) {
                const float ng = g0 + u < a.G ? (float)rows_counted(nvalid, g0 + u) : 0.f;
                if (ng > 0.f) {
                    const float delta = ps[u] / ng - mean, tot = n + ng;
                    mean += delta * (ng / tot);
                    M2 += pm[u] + delta * delta * (n * ng / tot);
                    n = tot;
                }
            }
        }
        DN_STAMP(stamp + 1);
        const float cnt = fmaxf(n, 1.f);
        const float var = fmaxf(M2 / cnt, 0.f);
        const float invstd = 1.f / sqrtf(var + bn.eps);
        sc[tid] = invstd * w;
        sc[KM + tid] = b - mean * invstd * w;
        if (g == 0 && mean == mean && M2 == M2) {                // (NaN: the exchange timed out — nothing persistent is touched)
            stat[tid] = mean;
            stat[N + tid] = invstd;
            stat[2 * N + tid] = invstd * w;
            stat[3 * N + tid] = b - mean * invstd * w;
            stat[4 * N + tid] = w;
            stat[5 * N + tid] = b;
            const int comp = tid / bn.k, kk = tid - comp * bn.k;
            const float unbiased = n > 1.f ? M2 / (n - 1.f) : var;
            for (int q = 0; q < bn.n_comp; ++q) {                // uniform index into the kernarg pointer tables
                if (q == comp) {
                    bn.rmean[q][kk] = (1.f - bn.momentum) * rm_old + bn.momentum * mean;
                    bn.rvar[q][kk] = (1.f - bn.momentum) * rv_old + bn.momentum * unbiased;
                }
                if (q == tid && bn.nbt[q]) *bn.nbt[q] += 1;
            }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void norm_column(const DnNorm &bn, int col, int N, float &w, float &b, float &rm, float &rv) {
    // the pointer tables sit in the kernarg segment: index them with a UNIFORM loop counter (a per-lane index would send
    // the whole struct through scratch memory) and select this lane's component afterwards; all loads are unconditional
    const int c = min(col, N - 1);
    const int comp = c / bn.k, kk = c - comp * bn.k;
    w = 1.f; b = 0.f; rm = 0.f; rv = 1.f;
#pragma unroll 4
    for (int q = 0; q < bn.n_comp; ++q) {
        const float wq = bn.affine ? bn.weight[q][kk] : 1.f, bq = bn.affine ? bn.bias[q][kk] : 0.f;
        const float mq = bn.rmean[q][kk], vq = bn.rvar[q][kk];
        const bool mine = q == comp;
        w = mine ? wq : w; b = mine ? bq : b; rm = mine ? mq : rm; rv = mine ? vq : rv;
    }
}

__global__ __launch_bounds__(DN_THREADS) void downstream_fwd_kernel(DnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, g = blockIdx.x, rb = g * DN_ROWS;
    const int d0 = a.d0, d1 = a.d1, d2 = a.d2, d3 = a.d3, T = a.T;
    const int KM = max(d0, max(d1, d2)), lda = KM + 4;
    float *A0 = sm, *A1 = A0 + DN_ROWS * lda;
    float *Hs1 = A1 + DN_ROWS * lda, *Hs2 = Hs1 + d0 * (d1 + 4), *Hs3 = Hs2 + d1 * (d2 + 4);
    float *red = Hs3 + r4(d2 * d3), *sc = red + 4 * 32 * 33, *wr = sc + 2 * KM, *z3s = wr + r4(T * d3 + T),
          *bs = z3s + r4(DN_ROWS * d3);                        // bs: b1 | b2 | b3
    const unsigned tag = __hip_atomic_load(a.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    const int nvalid = a.n_valid ? min(*a.n_valid, a.B) : a.B;
    const int lg1 = 31 - __clz(d1), lg2 = 31 - __clz(d2);
    DN_STAMP(0);

    // ---- everything this launch reads from memory goes in flight now ----
    float4 xr[4], h1[16], h2[16];
    float h3[4], wrv[2], b1v, b2v, b3v;
    const int rows_here = max(0, min(DN_ROWS, a.B - rb));
    ld_batch4<4>(a.x + (size_t)min(rb, a.B - 1) * d0, rows_here * (d0 >> 2), xr);
    ld_batch4<16>(a.H1, d0 * (d1 >> 2), h1);
    ld_batch4<16>(a.H2, d1 * (d2 >> 2), h2);
    ld_batch1<4>(a.H3, d2 * d3, h3);
    ld_batch1<2>(a.Wr, T * d3, wrv);
    b1v = a.b1 ? a.b1[min(tid, d1 - 1)] : 0.f;
    b2v = a.b2 ? a.b2[min(tid, d2 - 1)] : 0.f;
    b3v = a.b3 ? a.b3[min(tid, d3 - 1)] : 0.f;
    const float brv = a.br ? a.br[min(tid, T - 1)] : 0.f;
    float w1, bb1, rm1, rv1, w2, bb2, rm2, rv2;
    norm_column(a.bn1, tid, d1, w1, bb1, rm1, rv1);
    norm_column(a.bn2, tid, d2, w2, bb2, rm2, rv2);
    st_rows4<4>(xr, d0 >> 2, rows_here, A0, lda);
    st_matrix4<16>(h1, d0 * (d1 >> 2), d1 >> 2, Hs1, d1 + 4);
    st_matrix4<16>(h2, d1 * (d2 >> 2), d2 >> 2, Hs2, d2 + 4);
    st_flat1<4>(h3, d2 * d3, Hs3);
    st_flat1<2>(wrv, T * d3, wr);
    if (tid < T) wr[T * d3 + tid] = brv;
    if (tid < d1) bs[tid] = b1v;
    if (tid < d2) bs[DN_MAXW + tid] = b2v;
    if (tid < d3) bs[2 * DN_MAXW + tid] = b3v;
    __syncthreads();
    DN_STAMP(1);

    // layer 1: z1 = x . H1 + b1 -> norm -> act
    wg_gemm(A0, lda, Hs1, d1 + 4, d0, d1, bs, red, A1, lda);
    DN_STAMP(2);
    store_rows(A1, lda, a.B, rb, d1, a.z1);
    fwd_norm(a, a.bn1, A1, lda, d1, nvalid, a.partial, tag, a.stat1, sc, KM, w1, bb1, rm1, rv1, 3);
    DN_STAMP(5);
    apply_norm_act(A1, A0, lda, d1, lg1, sc, KM, a.act, rb, nvalid);
    // layer 2
    DN_STAMP(6);
    wg_gemm(A0, lda, Hs2, d2 + 4, d1, d2, bs + DN_MAXW, red, A1, lda);
    DN_STAMP(7);
    store_rows(A1, lda, a.B, rb, d2, a.z2);
    fwd_norm(a, a.bn2, A1, lda, d2, nvalid, a.partial + (size_t)a.G * DN_PSTRIDE, tag, a.stat2, sc, KM, w2, bb2, rm2, rv2, 8);
    DN_STAMP(10);
    apply_norm_act(A1, A0, lda, d2, lg2, sc, KM, a.act, rb, nvalid);
    // layer 3 (d2 -> d3 = n * target, a handful of columns) and the real head (d3 -> T): plain FMAs
    for (int e = tid; e < DN_ROWS * d3; e += DN_THREADS) {
        const int row = e / d3, j = e - row * d3;
        float v = bs[2 * DN_MAXW + j];
        for (int k = 0; k < d2; ++k) v = fmaf(A0[row * lda + k], Hs3[k * d3 + j], v);
        z3s[row * d3 + j] = v;
        if (rb + row < a.B) a.z3[(size_t)(rb + row) * d3 + j] = v;
    }
    __syncthreads();
    for (int e = tid; e < DN_ROWS * T; e += DN_THREADS) {
        const int row = e / T, t = e - row * T;
        float v = wr[T * d3 + t];
        for (int j = 0; j < d3; ++j) v = fmaf(z3s[row * d3 + j], wr[t * d3 + j], v);
        if (rb + row < a.B) a.logits[(size_t)(rb + row) * T + t] = v;
    }
    DN_STAMP(11);
    if (g == 0 && tid == 0) __hip_atomic_fetch_add(a.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Backward norm: Gm holds g = d(act out) * act'(.) (rows >= n_valid: 0), XH holds x_hat; both [32][ld] in LDS.  Publishes the
// column sums, collects everybody's, then Gm := dz = w * invstd * (g - mean(g) - x_hat * mean(g * x_hat)) in place (st: the
// norm's [mean | invstd | w | b] in LDS, stride DN_MAXW); workgroup 0 writes dwb = [sum g*x_hat | sum g] and (first exchange
// only) merges the extra partials (dWr | dbr).
__device__ __forceinline__ void bwd_norm(const DnArgs &a, float *Gm, const float *XH, int ld, int N, int lgN, int nvalid, int rb,
                                         unsigned long long *part, unsigned tag, const float *st, float *dwb, float *sc, int n_extra,
                                         float *extra_out) {
    const int tid = threadIdx.x, g = blockIdx.x;
    if (tid < N) {
        float sg = 0.f, sgx = 0.f;
        for (int r = 0; r < DN_ROWS; ++r) {
            const float gv = Gm[r * ld + tid];
            sg += gv;
            sgx = fmaf(gv, XH[r * ld + tid], sgx);
        }
        put_granule(part + (size_t)g * DN_PSTRIDE + tid, sg, tag);
        put_granule(part + (size_t)g * DN_PSTRIDE + DN_MAXW + tid, sgx, tag);
        float SG = 0.f, SGX = 0.f;
        for (int g0 = 0; g0 < a.G; g0 += DN_MERGE) {
            float pa[DN_MERGE], pb[DN_MERGE];
            collect_pairs(part, a.G, g0, tid, tag, pa, pb, a.err);
#pragma unroll
            for (int u = 0; u < DN_MERGE; ++u)
                if (g0 + u < a.G) {
                    SG += pa[u];
                    SGX += pb[u];
                }
        }
        const float cnt = (float)max(nvalid, 1);
        sc[tid] = SG / cnt;
        sc[DN_MAXW + tid] = SGX / cnt;
        if (g == 0 && dwb) {
            dwb[tid] = SGX;
            dwb[N + tid] = SG;
        }
    }
    if (g == 0 && extra_out && tid >= DN_THREADS - 64) {        // the last wavefront merges the extras (dWr | dbr), workgroup order
        for (int j = tid - (DN_THREADS - 64); j < n_extra; j += 64) {
            float s = 0.f;
            for (int gg = 0; gg < a.G; ++gg) {
                unsigned long long q;
                int spins = 0;
                do {
                    q = get_granule(part + (size_t)gg * DN_PSTRIDE + 2 * DN_MAXW + j);
                    if (++spins > DN_SPIN_MAX) {
                        if (a.err) atomicOr(a.err, PHC_ERR_RENDEZVOUS);
                        break;
                    }
                } while ((unsigned)(q >> 32) != tag);
                s += (unsigned)(q >> 32) == tag ? __uint_as_float((unsigned)q) : __int_as_float(0x7fc00000);
            }
            extra_out[j] = s;
        }
    }
    __syncthreads();
    const int n4 = N >> 2, lg4 = lgN - 2;
    for (int e = tid; e < DN_ROWS * n4; e += DN_THREADS) {
        const int row = e >> lg4, c = (e & (n4 - 1)) << 2;
        float4 gq = *reinterpret_cast<const float4 *>(Gm + row * ld + c);
        const float4 xh = *reinterpret_cast<const float4 *>(XH + row * ld + c);
        const float4 w4 = *reinterpret_cast<const float4 *>(st + 2 * DN_MAXW + c), i4 = *reinterpret_cast<const float4 *>(st + DN_MAXW + c);
        const float4 mg = *reinterpret_cast<const float4 *>(sc + c), mx = *reinterpret_cast<const float4 *>(sc + DN_MAXW + c);
        const bool live = rb + row < nvalid;
        gq.x = live ? w4.x * i4.x * (gq.x - mg.x - xh.x * mx.x) : 0.f;
        gq.y = live ? w4.y * i4.y * (gq.y - mg.y - xh.y * mx.y) : 0.f;
        gq.z = live ? w4.z * i4.z * (gq.z - mg.z - xh.z * mx.z) : 0.f;
        gq.w = live ? w4.w * i4.w * (gq.w - mg.w - xh.w * mx.w) : 0.f;
        *reinterpret_cast<float4 *>(Gm + row * ld + c) = gq;
    }
    __syncthreads();
}

__global__ __launch_bounds__(DN_THREADS) void downstream_bwd_kernel(DnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, g = blockIdx.x, rb = g * DN_ROWS;
    const int d0 = a.d0, d1 = a.d1, d2 = a.d2, d3 = a.d3, T = a.T;
    const int KM = max(d0, max(d1, d2)), lda = KM + 4;
    float *A0 = sm, *A1 = A0 + DN_ROWS * lda;
    float *Hs1T = A1 + DN_ROWS * lda, *Hs2T = Hs1T + d1 * (d0 + 4), *Hs3T = Hs2T + d2 * (d1 + 4);
    float *red = Hs3T + r4(d3 * d2), *sc = red + 4 * 32 * 33, *wr = sc + 2 * DN_MAXW, *dls = wr + r4(T * d3), *z3s = dls + r4(DN_ROWS * T),
          *dz3s = z3s + r4(DN_ROWS * d3), *st1 = dz3s + r4(DN_ROWS * d3), *st2 = st1 + 4 * DN_MAXW;     // st*: mean | invstd | w | b
    const unsigned tag = __hip_atomic_load(a.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    const int nvalid = a.n_valid ? min(*a.n_valid, a.B) : a.B;
    const int lg1 = 31 - __clz(d1), lg2 = 31 - __clz(d2);

    // ---- everything this launch reads from memory goes in flight now (z1 stays in registers until its turn) ----
    float4 z2r[4], z1r[4], h1[16], h2[16];
    float h3[4], wrv[2], dlv[1], z3v[4], s1v[4], s2v[4];
    const int rows_here = max(0, min(DN_ROWS, a.B - rb));
    const size_t r0 = (size_t)min(rb, a.B - 1);
    ld_batch4<4>(a.z2 + r0 * d2, rows_here * (d2 >> 2), z2r);
    ld_batch4<4>(a.z1 + r0 * d1, rows_here * (d1 >> 2), z1r);
    ld_batch4<16>(a.H1T, d1 * (d0 >> 2), h1);
    ld_batch4<16>(a.H2T, d2 * (d1 >> 2), h2);
    ld_batch1<4>(a.H3T, d3 * d2, h3);
    ld_batch1<2>(a.Wr, T * d3, wrv);
    ld_batch1<1>(a.dlogits + r0 * T, rows_here * T, dlv);
    ld_batch1<4>(a.z3 + r0 * d3, rows_here * d3, z3v);
#pragma unroll
    for (int q = 0; q < 4; ++q) {                            // mean, invstd, w, b of both norms (save_stat slots 0, 1, 4, 5)
        const int slot = q < 2 ? q : q + 2;
        s1v[q] = a.stat1[slot * d1 + min(tid, d1 - 1)];
        s2v[q] = a.stat2[slot * d2 + min(tid, d2 - 1)];
    }
    st_rows4<4>(z2r, d2 >> 2, rows_here, A1, lda);
    st_matrix4<16>(h1, d1 * (d0 >> 2), d0 >> 2, Hs1T, d0 + 4);
    st_matrix4<16>(h2, d2 * (d1 >> 2), d1 >> 2, Hs2T, d1 + 4);
    st_flat1<4>(h3, d3 * d2, Hs3T);
    st_flat1<2>(wrv, T * d3, wr);
    if (tid < DN_ROWS * T) dls[tid] = tid < rows_here * T ? dlv[0] : 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int e = tid + u * DN_THREADS;
        if (e < DN_ROWS * d3) z3s[e] = e < rows_here * d3 ? z3v[u] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (tid < d1) st1[q * DN_MAXW + tid] = s1v[q];
        if (tid < d2) st2[q * DN_MAXW + tid] = s2v[q];
    }
    __syncthreads();

    // head and layer 3 backwards (a handful of columns): dz3 = dlogits . Wr, da2 = dz3 . H3^T; partial dWr | dbr of these rows
    for (int e = tid; e < DN_ROWS * d3; e += DN_THREADS) {
        const int row = e / d3, j = e - row * d3;
        float v = 0.f;
        for (int t = 0; t < T; ++t) v = fmaf(dls[row * T + t], wr[t * d3 + j], v);
        dz3s[row * d3 + j] = v;
        if (rb + row < a.B) a.dz3[(size_t)(rb + row) * d3 + j] = v;
    }
    unsigned long long *part1 = a.partial, *part2 = a.partial + (size_t)a.G * DN_PSTRIDE;
    for (int e = tid; e < T * d3 + T; e += DN_THREADS) {
        float s = 0.f;
        if (e < T * d3) {
            const int t = e / d3, j = e - t * d3;
            for (int r = 0; r < DN_ROWS; ++r) s = fmaf(dls[r * T + t], z3s[r * d3 + j], s);
        } else {
            const int t = e - T * d3;
            for (int r = 0; r < DN_ROWS; ++r) s += dls[r * T + t];
        }
        put_granule(part1 + (size_t)g * DN_PSTRIDE + 2 * DN_MAXW + e, s, tag);
    }
    __syncthreads();
    // g2 = da2 * act'(z_hat2 * w + b) -> A0, x_hat2 -> A1 (over z2)
    // (the ReLU test sits outside the element loops here and below: inside, the run-time activation switch became a chain of
    // scalar branches per element)
    const bool relu = a.act == PHC_ACT_RELU;
    auto g2_rows = [&](auto is_relu) {
        for (int e = tid; e < DN_ROWS * d2; e += DN_THREADS) {
            const int row = e >> lg2, c = e & (d2 - 1);
            float da = 0.f;
            for (int j = 0; j < d3; ++j) da = fmaf(dz3s[row * d3 + j], Hs3T[j * d2 + c], da);
            const float xh = (A1[row * lda + c] - st2[c]) * st2[DN_MAXW + c];
            const float pre = fmaf(xh, st2[2 * DN_MAXW + c], st2[3 * DN_MAXW + c]);
            float g;
            if constexpr (decltype(is_relu)::value)
                g = pre > 0.f ? da : 0.f;
            else
                g = da * act_grad_rt(a.act, pre);
            A0[row * lda + c] = rb + row < nvalid ? g : 0.f;
            A1[row * lda + c] = xh;
        }
    };
    if (relu)
        g2_rows(std::true_type{});
    else
        g2_rows(std::false_type{});
    __syncthreads();
    bwd_norm(a, A0, A1, lda, d2, lg2, nvalid, rb, part1, tag, st2, a.dwb2, sc, T * d3 + T, a.dWr);
    store_rows(A0, lda, a.B, rb, d2, a.dz2);
    // da1 = dz2 . H2^T -> A1 (the GEMM's output slab); then g1 -> A0, x_hat1 -> A1 (z1 comes out of the registers)
    wg_gemm(A0, lda, Hs2T, d1 + 4, d2, d1, nullptr, red, A1, lda);
    auto g1_rows = [&](auto is_relu) {
        const int w4 = d1 >> 2;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + u * DN_THREADS;
            if (e < DN_ROWS * w4) {
                const int row = e >> (lg1 - 2), c0 = (e & (w4 - 1)) << 2;
                const float zq[4] = {z1r[u].x, z1r[u].y, z1r[u].z, z1r[u].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = c0 + q;
                    const float xh = row < rows_here ? (zq[q] - st1[c]) * st1[DN_MAXW + c] : 0.f;
                    const float pre = fmaf(xh, st1[2 * DN_MAXW + c], st1[3 * DN_MAXW + c]);
                    const float da = A1[row * lda + c];
                    float g;
                    if constexpr (decltype(is_relu)::value)
                        g = pre > 0.f ? da : 0.f;
                    else
                        g = da * act_grad_rt(a.act, pre);
                    A0[row * lda + c] = rb + row < nvalid ? g : 0.f;
                    A1[row * lda + c] = xh;
                }
            }
        }
    };
    if (relu)
        g1_rows(std::true_type{});
    else
        g1_rows(std::false_type{});
    __syncthreads();
    bwd_norm(a, A0, A1, lda, d1, lg1, nvalid, rb, part2, tag, st1, a.dwb1, sc, 0, nullptr);
    store_rows(A0, lda, a.B, rb, d1, a.dz1);
    // dx = dz1 . H1^T
    if (a.dx) {
        wg_gemm(A0, lda, Hs1T, d0 + 4, d1, d0, nullptr, red, A1, lda);
        store_rows(A1, lda, a.B, rb, d0, a.dx);
    }
    if (g == 0 && tid == 0) __hip_atomic_fetch_add(a.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

static size_t dn_lds_floats(int d0, int d1, int d2, int d3, int T, bool bwd) {
    const int KM = d0 > d1 ? (d0 > d2 ? d0 : d2) : (d1 > d2 ? d1 : d2);
    size_t f = 2 * (size_t)DN_ROWS * (KM + 4) + 4 * 32 * 33;
    if (!bwd)
        f += (size_t)d0 * (d1 + 4) + (size_t)d1 * (d2 + 4) + r4(d2 * d3) + 2 * KM + r4(T * d3 + T) + r4(DN_ROWS * d3) + 3 * DN_MAXW;
    else
        f += (size_t)d1 * (d0 + 4) + (size_t)d2 * (d1 + 4) + r4(d3 * d2) + 2 * DN_MAXW + r4(T * d3) + r4(DN_ROWS * T) + 2 * r4(DN_ROWS * d3) +
             8 * DN_MAXW;
    return f + 16;
}

static bool dn_shape_ok(int B, int d0, int d1, int d2, int d3, int T) {
    auto wide = [](int d) { return d == 32 || d == 64 || d == 128; };
    // every k-split must stay a multiple of 8: K / (4 / tiles) with tiles = N / 32
    auto split_ok = [](int K, int N) { return (K % (8 * (4 / (N >> 5)))) == 0; };
    if (!(B >= 1 && B <= DN_MAX_WG * DN_ROWS && wide(d0) && wide(d1) && wide(d2) && d3 >= 1 && d3 <= DN_MAXD3 && T >= 1 && T <= DN_MAXT)) return false;
    if (!(split_ok(d0, d1) && split_ok(d1, d2) && split_ok(d2, d1) && split_ok(d1, d0))) return false;
    if (d2 * d3 > 4 * DN_THREADS || T * d3 > 2 * DN_THREADS) return false;          // register staging of H3 / Wr (ld_batch1<4>, <2>)
    return sizeof(float) * dn_lds_floats(d0, d1, d2, d3, T, false) <= 150 * 1024 && sizeof(float) * dn_lds_floats(d0, d1, d2, d3, T, true) <= 150 * 1024;
}

// How many workgroups of a fused-head launch the CURRENT device can hold at the same time (the rendezvous needs all G of
// them resident): occupancy per CU for the launch's real LDS size x the CU count the runtime reports for this (possibly
// partitioned: CPX = 32 CUs) device.  0 when a CU mask is in the environment (the runtime reports the unmasked count) or
// when any query fails — the caller then takes the per-layer path.  Cached per (device, direction, LDS size).
static int dn_resident_capacity(bool bwd, size_t lds_bytes) {
    static const bool masked = getenv("HSA_CU_MASK") || getenv("ROC_GLOBAL_CU_MASK") || getenv("HSA_CU_MASK_SKIP_INIT");
    if (masked) return 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    struct Slot { size_t lds; int cap; };
    static Slot cache[64][2] = {};
    Slot &sl = cache[dev][bwd ? 1 : 0];
    if (sl.lds == lds_bytes && sl.cap > 0) return sl.cap;
    const void *fn = bwd ? reinterpret_cast<const void *>(&downstream_bwd_kernel) : reinterpret_cast<const void *>(&downstream_fwd_kernel);
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    int per_cu = 0;
    hipError_t e = bwd ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, downstream_bwd_kernel, DN_THREADS, lds_bytes)
                       : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, downstream_fwd_kernel, DN_THREADS, lds_bytes);
    hipDeviceProp_t prop;
    if (e != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess || per_cu < 1 || prop.multiProcessorCount < 1) {
        (void)hipGetLastError();
        return 0;
    }
    sl.lds = lds_bytes;
    sl.cap = per_cu * prop.multiProcessorCount;
    return sl.cap;
}

// all G workgroups of BOTH launches must be able to be resident at once on this device
static bool dn_device_ok(int B, int d0, int d1, int d2, int d3, int T) {
    const int G = (B + DN_ROWS - 1) / DN_ROWS;
    return dn_resident_capacity(false, sizeof(float) * dn_lds_floats(d0, d1, d2, d3, T, false)) >= G &&
           dn_resident_capacity(true, sizeof(float) * dn_lds_floats(d0, d1, d2, d3, T, true)) >= G;
}

static int fill_norm(const char *who, DnNorm *o, const phc_downstream_norm &n, int width) {
    PHC_REQUIRE(n.n_comp >= 1 && n.n_comp <= PHC_MAX_COMPONENTS && width % n.n_comp == 0, PHC_ERR_INVALID_ARG, "%s: norm with %d components on width %d",
                who, n.n_comp, width);
    PHC_REQUIRE(n.running_mean && n.running_var, PHC_ERR_INVALID_ARG, "%s: running statistics are required", who);
    o->n_comp = n.n_comp;
    o->k = width / n.n_comp;
    o->affine = n.weight != nullptr && n.bias != nullptr;
    o->momentum = n.momentum;
    o->eps = n.eps;
    for (int c = 0; c < n.n_comp; ++c) {
        o->weight[c] = o->affine ? n.weight[c] : nullptr;
        o->bias[c] = o->affine ? n.bias[c] : nullptr;
        o->rmean[c] = n.running_mean[c];
        o->rvar[c] = n.running_var[c];
        o->nbt[c] = n.num_batches_tracked ? reinterpret_cast<long long *>(n.num_batches_tracked[c]) : nullptr;
        PHC_REQUIRE(o->rmean[c] && o->rvar[c] && (!o->affine || (o->weight[c] && o->bias[c])), PHC_ERR_INVALID_ARG, "%s: null norm pointer", who);
    }
    return PHC_OK;
}

static int dn_common(const char *who, const phc_downstream_desc *d, DnArgs *a) {
    PHC_REQUIRE(d, PHC_ERR_INVALID_ARG, "%s: null descriptor", who);
    PHC_REQUIRE(dn_shape_ok((int)d->B, d->d0, d->d1, d->d2, d->d3, d->T), PHC_ERR_UNSUPPORTED,
                "%s: B=%lld widths %d-%d-%d-%d-%d outside the fused head's range (use the per-layer path)", who, (long long)d->B, d->d0, d->d1,
                d->d2, d->d3, d->T);
    PHC_REQUIRE(d->act >= PHC_ACT_IDENTITY && d->act <= PHC_ACT_SELU, PHC_ERR_INVALID_ARG, "%s: unknown activation %d", who, d->act);
    PHC_REQUIRE(dn_device_ok((int)d->B, d->d0, d->d1, d->d2, d->d3, d->T), PHC_ERR_UNSUPPORTED,
                "%s: the %lld workgroups of the fused head cannot all be resident on this device (partitioned device or CU mask): use "
                "the per-layer path", who, (long long)((d->B + DN_ROWS - 1) / DN_ROWS));
    PHC_REQUIRE(d->partial && d->sync, PHC_ERR_WORKSPACE, "%s: partial / sync buffers missing", who);
    a->B = (int)d->B; a->d0 = d->d0; a->d1 = d->d1; a->d2 = d->d2; a->d3 = d->d3; a->T = d->T; a->act = d->act;
    a->G = (a->B + DN_ROWS - 1) / DN_ROWS;
    a->n_valid = d->n_valid_dev;
    a->partial = reinterpret_cast<unsigned long long *>(d->partial);
    a->sync = reinterpret_cast<unsigned *>(d->sync);
    a->err = err_word();
    a->Wr = d->Wr; a->br = d->br;
    a->z1 = d->z1; a->z2 = d->z2; a->z3 = d->z3; a->stat1 = d->stat1; a->stat2 = d->stat2;
    PHC_REQUIRE(a->Wr && a->z1 && a->z2 && a->z3 && a->stat1 && a->stat2, PHC_ERR_INVALID_ARG, "%s: null pointer", who);
    return PHC_OK;
}

}  // namespace phc

using namespace phc;

extern "C" {

#ifdef PHC_DN_TIMING
int phc_debug_dn_timestamps(unsigned long long *out64) {
    return hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_dn_ts), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -1;
}
#endif

int phc_downstream_supported(int64_t B, int d0, int d1, int d2, int d3, int T) {
    const int Bc = (int)(B > (1LL << 30) ? 1 << 30 : B);
    return (dn_shape_ok(Bc, d0, d1, d2, d3, T) && dn_device_ok(Bc, d0, d1, d2, d3, T)) ? 1 : 0;
}

size_t phc_downstream_partial_floats(int64_t B) {
    if (B < 1) return 0;
    return 2 * 2 * (size_t)((B + DN_ROWS - 1) / DN_ROWS) * DN_PSTRIDE;      // 8-byte {value, tag} granules
}

int phc_downstream_fwd(const phc_downstream_desc *d, void *stream) {
    DnArgs a = {};
    int rc = dn_common("phc_downstream_fwd", d, &a);
    if (rc) return rc;
    rc = fill_norm("phc_downstream_fwd", &a.bn1, d->norm1, d->d1);
    if (rc) return rc;
    rc = fill_norm("phc_downstream_fwd", &a.bn2, d->norm2, d->d2);
    if (rc) return rc;
    a.x = d->x; a.H1 = d->H1; a.b1 = d->b1; a.H2 = d->H2; a.b2 = d->b2; a.H3 = d->H3; a.b3 = d->b3; a.logits = d->logits;
    PHC_REQUIRE(a.x && a.H1 && a.H2 && a.H3 && a.logits, PHC_ERR_INVALID_ARG, "phc_downstream_fwd: null pointer");
    PHC_REQUIRE(((reinterpret_cast<uintptr_t>(a.x) | reinterpret_cast<uintptr_t>(a.H1) | reinterpret_cast<uintptr_t>(a.H2) |
                  reinterpret_cast<uintptr_t>(a.z1) | reinterpret_cast<uintptr_t>(a.z2)) & 15) == 0,
                PHC_ERR_UNSUPPORTED, "phc_downstream_fwd: 16-byte aligned rows required");
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = sizeof(float) * dn_lds_floats(a.d0, a.d1, a.d2, a.d3, a.T, false);
    // (hipFuncAttributeMaxDynamicSharedMemorySize was set, per device and checked, by dn_device_ok's capacity query)
    ProfScope prof(K_DOWNSTREAM_FWD, s);
    PHC_LAUNCH(downstream_fwd_kernel, dim3(a.G), dim3(DN_THREADS), lds, s, a);
    return check_launch("downstream_fwd");
}

int phc_downstream_bwd(const phc_downstream_desc *d, void *stream) {
    DnArgs a = {};
    int rc = dn_common("phc_downstream_bwd", d, &a);
    if (rc) return rc;
    a.H1T = d->H1T; a.H2T = d->H2T; a.H3T = d->H3T; a.dlogits = d->dlogits;
    a.dx = d->dx; a.dz1 = d->dz1; a.dz2 = d->dz2; a.dz3 = d->dz3; a.dwb1 = d->dwb1; a.dwb2 = d->dwb2; a.dWr = d->dWr; a.dbr = d->dWr ? d->dWr + d->T * d->d3 : nullptr;
    PHC_REQUIRE(a.H1T && a.H2T && a.H3T && a.dlogits && a.dz1 && a.dz2 && a.dz3 && a.dWr, PHC_ERR_INVALID_ARG, "phc_downstream_bwd: null pointer");
    PHC_REQUIRE(((reinterpret_cast<uintptr_t>(a.H1T) | reinterpret_cast<uintptr_t>(a.H2T) | reinterpret_cast<uintptr_t>(a.z2) |
                  reinterpret_cast<uintptr_t>(a.dz1) | reinterpret_cast<uintptr_t>(a.dz2) | reinterpret_cast<uintptr_t>(a.dx)) & 15) == 0,
                PHC_ERR_UNSUPPORTED, "phc_downstream_bwd: 16-byte aligned rows required");
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = sizeof(float) * dn_lds_floats(a.d0, a.d1, a.d2, a.d3, a.T, true);
    ProfScope prof(K_DOWNSTREAM_BWD, s);
    PHC_LAUNCH(downstream_bwd_kernel, dim3(a.G), dim3(DN_THREADS), lds, s, a);
    return check_launch("downstream_bwd");
}

}  // extern "C"
