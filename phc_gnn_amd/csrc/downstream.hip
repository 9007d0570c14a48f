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
using f32x4_t = __attribute__((ext_vector_type(4))) float;

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
    const float *H1k4, *H2k4, *H1Tk4, *H2Tk4;                 // v2: k4 forms of H1, H2 (forward) and of their transposes (backward)
    DnNorm bn1, bn2;
    float *z1, *z2, *z3, *logits, *stat1, *stat2;             // forward outputs (z*, stat* are what the backward reads)
    const float *dlogits;
    float *dx, *dz1, *dz2, *dz3, *dwb1, *dwb2, *dWr, *dbr;    // backward outputs
    // v2 only: phm_dropout behind the two activations (downstream.py:104-107; layers.py:31-55) — keep-and-scale factors of
    // drop_factors_v(p, same, k, site, seed, step, row, column, width); a1 / a2: the dropped activations [B, d1] / [B, d2] the
    // forward leaves for the parameter-gradient jobs of layers 2 and 3 (without dropout those re-derive them from z and save_stat)
    float drop_p1, drop_p2;
    int drop_same1, drop_same2, drop_k1, drop_k2, drop_site1, drop_site2;
    const unsigned long long *rng;                            // device {seed, step}; NULL: no dropout
    float *a1, *a2;
    unsigned long long *partial;                              // [2][G][DN_PSTRIDE] {value, tag} granules
    unsigned *sync;                                           // generation word
    int *err;                                                 // device error word (phc_set_error_word) or NULL
    // v2 forward only (round 6): the NaN-masked mean loss of the logits inside the launch — every workgroup also publishes its
    // logits as {value, tag} granules, workgroup 0 collects them at its end, writes loss | sum | count and d loss / d logits for
    // an upstream gradient of 1 (what bce_masked_fwd_unit_kernel did in a launch of its own: 16 us inside a replayed step)
    const float *loss_y;                                      // [B, T] labels, NaN = unlabeled; NULL: off
    unsigned long long *loss_gran;                            // [B T] granules
    float *loss_out, *loss_dz;                                // [3], [B, T]
    int loss_kind;                                            // 0 BCE with logits, 1 L1    // forward + backward in ONE launch (downstream2_fb_kernel): the backward part's own exchange buffers and generation word
    unsigned long long *partial_b;
    unsigned *sync_b;
};

// A rendezvous that never completes (a producer workgroup that is not resident: CU mask, another stream owning the CUs, a
// partitioned device the capacity query did not see) must not hang the GPU: every poll loop gives up after DN_SPIN_MAX
// rounds, raises PHC_ERR_RENDEZVOUS in the device error word and lets the launch run to its end (its outputs are then
// garbage; the host raises at its next check point).  2^21 rounds of >= 1 us each: seconds, never reached by a live exchange.
constexpr int DN_SPIN_MAX = 1 << 21;
constexpr int DN_LOSS_MAX = 4096;                            // logits whose loss the forward launch can take (DnArgs.loss_y)

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

// =====================================================================================================================
// v2 of the fused head (round 4): the same chain and the same exchange protocol with
//   * 16 rows per workgroup (32 above 1024 rows) and 512 threads: twice the workgroups, each GEMM half as long — the launch is a
//     chain of latency-bound phases, so the phases are what must shrink;
//   * the three big GEMMs on v_mfma_f32_16x16x4_f32 with the operand matrix taken straight from memory in its k4 form
//     (phc_phm_H_job.Hk4 / HTk4: one 16-byte load = four consecutive k of one column) into REGISTERS — all of a wavefront's
//     loads in flight at once, issued at kernel entry (first GEMM) or behind a publish (the next GEMM's operand arrives while
//     the statistics travel) — no 64 KB LDS staging pass in front of the first MFMA (6 us of the 32 us v1 launch);
//   * widths that are any multiple of 4 up to 256 (zero-padded k-groups / column tiles): the 256-wide ZINC / molpcba-style
//     heads and the reference's 180 -> 180 -> 80 head run fused too.
// MFMA roles as in gemm_hs_device.h: A := operand slice (A[i][k] = H[k][col0 + i]), B := activations (B[k][j] = a[row j][k]),
// lane (c = lane % 16, kq = lane / 16) ends with y[row c][col0 + 4 kq .. + 3].  Activations live in LDS as [rows][ld] with
// ld = width rounded up to 16, + 4 (one 16-byte pad: the 16 lanes of a kq group read 16 different bank groups).
// =====================================================================================================================
constexpr int D2_THREADS = 512, D2_WAVES = 8;
constexpr int D2_MAXW = 256;                           // widest layer (d0, d1, d2)
constexpr int D2_EXTRA = DN_MAXT * DN_MAXD3 + DN_MAXT;
constexpr int D2_PSTRIDE = 2 * D2_MAXW + D2_EXTRA;     // granules per workgroup per exchange
constexpr int D2_MAX_WG = 128;
constexpr int D2_MERGE = 16;                           // granule pairs in flight per thread while collecting
constexpr int D2_FB_STASH = 15 * D2_MAXW;              // floats behind the carve-up: the norms' statistics across the one-launch forward + backward

__host__ __device__ __forceinline__ int r16(int v) { return (v + 15) & ~15; }

// one wavefront's share of a GEMM  out[R, N] = A[R, K] . H[K, N]:  tiles of 16 columns; with fewer than 8 tiles the
// contraction is split over the wavefronts that share a tile (partials meet in LDS)
struct D2Plan {
    int tiles, tp2, ks_n, tile0, tpw, kg0, kgn;
};
__device__ __forceinline__ D2Plan d2_plan(int K, int N, int wave) {
    D2Plan p;
    const int kg = (K + 15) >> 4;
    p.tiles = (N + 15) >> 4;
    if (p.tiles > 4) {
        p.tp2 = 8; p.ks_n = 1; p.tile0 = wave; p.tpw = (p.tiles + 7) >> 3; p.kg0 = 0; p.kgn = kg;
    } else {
        p.tp2 = p.tiles <= 1 ? 1 : (p.tiles <= 2 ? 2 : 4);
        p.ks_n = 8 / p.tp2;
        p.tile0 = wave % p.tp2;
        const int ks = wave / p.tp2;
        p.tpw = 1;
        p.kg0 = ks * kg / p.ks_n;
        p.kgn = (ks + 1) * kg / p.ks_n - p.kg0;
    }
    // (every field is the same for all lanes of a wavefront: tell the compiler, so that the branches on them are scalar)
    p.tile0 = __builtin_amdgcn_readfirstlane(p.tile0); p.kg0 = __builtin_amdgcn_readfirstlane(p.kg0);
    p.kgn = __builtin_amdgcn_readfirstlane(p.kgn); p.tpw = __builtin_amdgcn_readfirstlane(p.tpw);
    p.tiles = __builtin_amdgcn_readfirstlane(p.tiles); p.ks_n = __builtin_amdgcn_readfirstlane(p.ks_n);
    p.tp2 = __builtin_amdgcn_readfirstlane(p.tp2);
    return p;
}

// A barrier that orders LDS traffic only: __syncthreads() also waits for every global load / store in flight (s_waitcnt
// vmcnt(0)) — here that would put the next GEMM's operand loads (issued early on purpose) and the slab's write-back in
// front of every phase boundary (measured: 4.2 us for a 0.3 us element-wise phase).
__device__ __forceinline__ void d2_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

struct D2Regs {
    f32x4_t v[2][16];                                   // [tile of this wavefront][k-group of this wavefront]
};

// every operand load of the wavefront in flight together.  Whole blocks of four k-groups (and the second tile) are skipped by
// WAVE-UNIFORM branches — d2_gemm_t skips the same blocks, so their registers are never read; a CU's address unit needs 16
// cycles per 16-byte wave-load: with all 32 slots loaded (24 of them dummies at 128 x 128) the eight wavefronts of a workgroup
// spent 1.9 us just issuing.  Inside a block the loads are unconditional (clamped address, selected value): a per-lane
// conditional load is a branch with a full wait at its join.
__device__ __forceinline__ void d2_load_operand(const float *__restrict__ Hk4, int K, int N, const D2Plan &p, int lane, D2Regs &h) {
    const int c = lane & 15, kq = lane >> 4;
    const f32x4_t *src = reinterpret_cast<const f32x4_t *>(Hk4);
    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            if (t < p.tpw && 4 * g4 < p.kgn) {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int gi = 4 * g4 + gq;
                    const int tile = p.tile0 + 8 * t, col = 16 * tile + c, srow = 4 * (p.kg0 + gi) + kq;
                    const bool ok = gi < p.kgn && tile < p.tiles && col < N && srow < (K >> 2);
                    const f32x4_t v = src[ok ? (size_t)srow * N + col : 0];
                    h.v[t][gi] = ok ? v : zero;
                }
            }
        }
}

// out[r][col] = sum_k A[r][k] H[k][col] (+ bias[col]); A in LDS [16 RB][lda] (columns K .. r16(K) zero), out in LDS [16 RB][ldo]
// (columns N .. r16(N) come out as 0), red: LDS scratch of 8 * 16 RB * 20 floats.  Ends with a barrier.
template <int RB, int TPW>
__device__ __forceinline__ void d2_gemm_t(const float *A, int lda, const D2Regs &h, const D2Plan &p, int N, const float *__restrict__ bias,
                                          float *red, float *out, int ldo) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, kq = lane >> 4;
    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
    f32x4_t acc[RB][TPW][2];                              // [row block][tile][chain]: two independent accumulator chains per tile
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int t = 0; t < TPW; ++t) acc[rb][t][0] = acc[rb][t][1] = zero;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
        if (4 * g4 < p.kgn) {                            // wave-uniform: whole groups of four k-groups are skipped
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int gi = 4 * g4 + gq;
                const int g = p.kg0 + min(gi, p.kgn - 1);     // (beyond kgn the operand registers are zero)
                f32x4_t fr[RB];
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) fr[rb] = *reinterpret_cast<const f32x4_t *>(A + (rb * 16 + c) * lda + 16 * g + 4 * kq);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                        for (int t = 0; t < TPW; ++t)
                            acc[rb][t][q & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(h.v[t][gi][q], fr[rb][q], acc[rb][t][q & 1], 0, 0, 0);
            }
        }
    }
    if (p.ks_n == 1) {
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int tile = p.tile0 + 8 * t;
            if (tile < p.tiles) {
                const int col = 16 * tile + 4 * kq;
                f32x4_t b4 = zero;
                if (bias) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) b4[q] = col + q < N ? bias[col + q] : 0.f;
                }
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
                    *reinterpret_cast<f32x4_t *>(out + (rb * 16 + c) * ldo + col) = acc[rb][t][0] + acc[rb][t][1] + b4;
            }
        }
        d2_lds_barrier();
        return;
    }
    // split contraction: partial tiles -> red[wave][16 RB][20], then one output element per thread and round
    {
        float *mine = red + (size_t)wave * (16 * RB * 20);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) *reinterpret_cast<f32x4_t *>(mine + (rb * 16 + c) * 20 + 4 * kq) = acc[rb][0][0] + acc[rb][0][1];
    }
    d2_lds_barrier();
    const int Np = 16 * p.tiles;
    for (int e = tid; e < 16 * RB * Np; e += D2_THREADS) {
        const int row = e / Np, col = e - row * Np;
        const int tile = col >> 4, cc = col & 15;
        float v = (bias && col < N) ? bias[col] : 0.f;
        for (int ks = 0; ks < p.ks_n; ++ks) v += red[(size_t)(ks * p.tp2 + tile) * (16 * RB * 20) + row * 20 + cc];
        out[row * ldo + col] = v;
    }
    d2_lds_barrier();
}

template <int RB>
__device__ __forceinline__ void d2_gemm(const float *A, int lda, const D2Regs &h, const D2Plan &p, int N, const float *__restrict__ bias,
                                        float *red, float *out, int ldo) {
    if (p.tpw > 1)                                       // (uniform over the workgroup: every wavefront's plan has the same tpw)
        d2_gemm_t<RB, 2>(A, lda, h, p, N, bias, red, out, ldo);
    else
        d2_gemm_t<RB, 1>(A, lda, h, p, N, bias, red, out, ldo);
}

template <int RB>
__device__ __forceinline__ int d2_rows_counted(int nvalid, int g) { return max(0, min(16 * RB, nvalid - g * 16 * RB)); }

// A column's two values travel as ONE 16-byte pair of granules {a, tag, b, tag} (words 2 col, 2 col + 1 of the workgroup's
// record): written by two 8-byte write-through stores, read by one 16-byte sc1 load — each half carries its own tag, so a pair
// torn between the halves is simply "not there yet".  (Half the load instructions of the collect, and 16-byte sc1 loads run at
// 1.4-1.8x the 8-byte rate: the collect's time is its bytes per CU, MI355X_MICROARCH.md 'allgather'.)
using u32x4_t = __attribute__((ext_vector_type(4))) unsigned;
__device__ __forceinline__ void d2_put_pair(unsigned long long *rec, int col, float va, float vb, unsigned tag) {
    put_granule(rec + 2 * col, va, tag);
    put_granule(rec + 2 * col + 1, vb, tag);
}
// collect column `col` of workgroups g0 .. g0 + D2_MERGE - 1 (clamped to G - 1)
__device__ __forceinline__ void d2_collect(const unsigned long long *part, int G, int g0, int col, unsigned tag, float (&va)[D2_MERGE],
                                           float (&vb)[D2_MERGE], int *err) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)part, 0, 0x7ffffff0, 0x00020000);
    bool ok;
    int spins = 0;
    do {
        ok = true;
#pragma unroll
        for (int u = 0; u < D2_MERGE; ++u) {
            const int gg = min(g0 + u, G - 1);
            const u32x4_t q = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(((size_t)gg * D2_PSTRIDE + 2 * col) * 8), 0, 16);   // aux 16 = sc1
            va[u] = __uint_as_float(q[0]);
            vb[u] = __uint_as_float(q[2]);
            ok = ok && q[1] == tag && q[3] == tag;
        }
        if (!ok) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > DN_SPIN_MAX) {
                if (err) atomicOr(err, PHC_ERR_RENDEZVOUS);
#pragma unroll
                for (int u = 0; u < D2_MERGE; ++u) va[u] = vb[u] = __int_as_float(0x7fc00000);      // poisoned, see collect_pairs
                break;
            }
        }
    } while (!ok);
}

// rows [rb, rb + 16 RB) of a [B, w] matrix <-> LDS [16 RB][ld]; rows >= B read as 0, columns w .. r16(w) are written 0
template <int RB>
__device__ __forceinline__ void d2_rows_to_lds(const float *__restrict__ src, int B, int rb, int w, float *dst, int ld) {
    // (all loads of the slab in flight together, the LDS writes in a second pass: a "load, store" loop pays one memory round
    // trip per iteration — hipcc drains the loads at the loop header)
    constexpr int U = 2 * RB;                            // 16 RB rows x up to 64 float4 over 512 threads
    const int w4 = r16(w) >> 2;
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int e = threadIdx.x + u * D2_THREADS;
        const int row = e / w4, c4 = e - row * w4;
        const bool ok = e < 16 * RB * w4 && rb + row < B && 4 * c4 < w;
        v[u] = *reinterpret_cast<const float4 *>(src + (ok ? (size_t)(rb + row) * w + 4 * c4 : 0));
        if (!ok) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int e = threadIdx.x + u * D2_THREADS;
        const int row = e / w4, c4 = e - row * w4;
        if (e < 16 * RB * w4) *reinterpret_cast<float4 *>(dst + row * ld + 4 * c4) = v[u];
    }
}
template <int RB>
__device__ __forceinline__ void d2_store_rows(const float *src, int ld, int B, int rb, int w, float *__restrict__ dst) {
    const int w4 = w >> 2;
    for (int e = threadIdx.x; e < 16 * RB * w4; e += D2_THREADS) {
        const int row = e / w4, c4 = e - row * w4;
        if (rb + row < B)
            *reinterpret_cast<float4 *>(dst + (size_t)(rb + row) * w + 4 * c4) = *reinterpret_cast<const float4 *>(src + row * ld + 4 * c4);
    }
}

// How the collect of one exchange is dealt over the 512 threads: column = tid % nc, part = tid / nc; part p merges the
// workgroups [p gp, (p + 1) gp) in ascending order, the column's part-0 thread then merges the parts in ascending order (a
// fixed order: bit-reproducible).  One round = 16 granule pairs in flight per thread; at 34 workgroups and 128 columns every
// thread has 9: ONE round trip instead of three (the rounds were the exchange's cost: ~3 us each).
struct D2Deal {
    int nc, parts, gp;
};
__device__ __forceinline__ D2Deal d2_deal(int N, int G) {
    D2Deal d;
    d.nc = (N + 31) & ~31;
    d.parts = min(min(D2_THREADS / d.nc, 4), G);              // (at most four: the column's last merge walks the parts one by one)
    d.gp = (G + d.parts - 1) / d.parts;
    return d;
}

// What workgroup 0 leaves behind of a forward norm (save_stat, running statistics): kept in registers and written at the END
// of the launch — written in place, the stores sat in front of the next barrier of the one workgroup every other one waits for.
struct D2Keep {
    float mean, invstd, w, b, unbiased, rm_old, rv_old;
    int valid;
};
__device__ __forceinline__ void d2_write_norm_state(const DnNorm &bn, const D2Keep &k, float *stat, int N) {
    const int tid = threadIdx.x;
    if (tid < N && k.valid) {
        stat[tid] = k.mean;
        stat[N + tid] = k.invstd;
        stat[2 * N + tid] = k.invstd * k.w;
        stat[3 * N + tid] = k.b - k.mean * k.invstd * k.w;
        stat[4 * N + tid] = k.w;
        stat[5 * N + tid] = k.b;
        const int comp = tid / bn.k, kk = tid - comp * bn.k;
        for (int q = 0; q < bn.n_comp; ++q) {                    // uniform index into the kernarg pointer tables
            if (q == comp) {
                bn.rmean[q][kk] = (1.f - bn.momentum) * k.rm_old + bn.momentum * k.mean;
                bn.rvar[q][kk] = (1.f - bn.momentum) * k.rv_old + bn.momentum * k.unbiased;
            }
            if (q == tid && bn.nbt[q]) *bn.nbt[q] += 1;
        }
    }
}

// forward norm over all rows (as fwd_norm): Z = this workgroup's slab [16 RB][ld]; scale / shift -> sc[0 .. N) / sc[D2_MAXW ..);
// scratch: 3 x 512 floats of LDS (the parts' partial merges)
template <int RB>
__device__ __forceinline__ void d2_fwd_norm(const DnArgs &a, const DnNorm &bn, const float *Z, int ld, int N, int nvalid,
                                            unsigned long long *part, unsigned tag, float *sc, float *scratch, float w, float b,
                                            float rm_old, float rv_old, D2Keep &keep, int stamp) {
    const int tid = threadIdx.x, g = blockIdx.x;
    if (tid < N) {
        const int nv = d2_rows_counted<RB>(nvalid, g);
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16 * RB; ++r) s += r < nv ? Z[r * ld + tid] : 0.f;
        const float mean_l = nv > 0 ? s / (float)nv : 0.f;
        float m2 = 0.f;
#pragma unroll
        for (int r = 0; r < 16 * RB; ++r) {
            const float d = Z[r * ld + tid] - mean_l;
            m2 = r < nv ? fmaf(d, d, m2) : m2;
        }
        d2_put_pair(part + (size_t)g * D2_PSTRIDE, tid, s, m2, tag);
    }
    DN_STAMP(stamp);
    const D2Deal dl = d2_deal(N, a.G);
    const int col = tid % dl.nc, pt = tid / dl.nc;
    const bool active = pt < dl.parts && col < N;
    float n = 0.f, mean = 0.f, M2 = 0.f;
    if (active) {
        const int gbeg = pt * dl.gp, gend = min(gbeg + dl.gp, a.G);
        for (int g0 = gbeg; g0 < gend; g0 += D2_MERGE) {
            float ps[D2_MERGE], pm[D2_MERGE];
            d2_collect(part, gend, g0, col, tag, ps, pm, a.err);
#pragma unroll
            for (int u = 0; u < D2_MERGE; ++u) {
                const float ng = g0 + u < gend ? (float)d2_rows_counted<RB>(nvalid, g0 + u) : 0.f;
                if (ng > 0.f) {
                    const float delta = ps[u] / ng - mean, tot = n + ng;
                    mean += delta * (ng / tot);
                    M2 += pm[u] + delta * delta * (n * ng / tot);
                    n = tot;
                }
            }
        }
        scratch[pt * dl.nc + col] = n;
        scratch[512 + pt * dl.nc + col] = mean;
        scratch[1024 + pt * dl.nc + col] = M2;
    }
    d2_lds_barrier();
    DN_STAMP(stamp + 1);
    if (tid < N) {
        for (int q = 1; q < dl.parts; ++q) {                     // part 0 is this thread's own (n, mean, M2)
            const float nq = scratch[q * dl.nc + tid], mq = scratch[512 + q * dl.nc + tid], Mq = scratch[1024 + q * dl.nc + tid];
            if (nq > 0.f) {
                const float delta = mq - mean, tot = n + nq;
                mean += delta * (nq / tot);
                M2 += Mq + delta * delta * (n * nq / tot);
                n = tot;
            }
        }
        const float cnt = fmaxf(n, 1.f);
        const float var = fmaxf(M2 / cnt, 0.f);
        const float invstd = 1.f / sqrtf(var + bn.eps);
        sc[tid] = invstd * w;
        sc[D2_MAXW + tid] = b - mean * invstd * w;
        keep.mean = mean; keep.invstd = invstd; keep.w = w; keep.b = b; keep.rm_old = rm_old; keep.rv_old = rv_old;
        keep.unbiased = n > 1.f ? M2 / (n - 1.f) : var;
        keep.valid = (mean == mean && M2 == M2) ? 1 : 0;         // (NaN: the exchange timed out — nothing persistent is touched)
    }
    d2_lds_barrier();
}

// Aout = act(Z * scale + shift) on the slab (rows >= n_valid: 0; columns N .. r16(N): 0)
template <int RB>
__device__ __forceinline__ void d2_apply_norm_act(const float *Z, int ldz, float *Aout, int lda, int N, const float *sc, int act, int rb,
                                                  int nvalid, float drop_p = 0.f, int drop_same = 0, int drop_k = 4, int drop_site = 0,
                                                  unsigned long long seed = 0, unsigned long long step = 0) {
    const int n4 = r16(N) >> 2;
    const bool relu = act == PHC_ACT_RELU;
    for (int e = threadIdx.x; e < 16 * RB * n4; e += D2_THREADS) {
        const int row = e / n4, c = (e - row * n4) << 2;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < N && rb + row < nvalid) {
            const float4 z = *reinterpret_cast<const float4 *>(Z + row * ldz + c);
            const float4 s4 = *reinterpret_cast<const float4 *>(sc + c), h4 = *reinterpret_cast<const float4 *>(sc + D2_MAXW + c);
            if (relu) {
                o.x = fmaxf(fmaf(z.x, s4.x, h4.x), 0.f); o.y = fmaxf(fmaf(z.y, s4.y, h4.y), 0.f);
                o.z = fmaxf(fmaf(z.z, s4.z, h4.z), 0.f); o.w = fmaxf(fmaf(z.w, s4.w, h4.w), 0.f);
            } else {
                o.x = act_fwd_rt(act, fmaf(z.x, s4.x, h4.x)); o.y = act_fwd_rt(act, fmaf(z.y, s4.y, h4.y));
                o.z = act_fwd_rt(act, fmaf(z.z, s4.z, h4.z)); o.w = act_fwd_rt(act, fmaf(z.w, s4.w, h4.w));
            }
            if (drop_p > 0.f) {                                  // (uniform)
                const float4 f = drop_factors_v(drop_p, drop_same, drop_k, drop_site, seed, step, rb + row, c, N);
                o.x *= f.x; o.y *= f.y; o.z *= f.z; o.w *= f.w;
            }
        }
        *reinterpret_cast<float4 *>(Aout + row * lda + c) = o;
    }
    d2_lds_barrier();
}

// LDS carve-up shared by host and device (floats)
struct D2Lds {
    int ldx, ld1, ld2, off_a, off_b, off_red, off_sc, off_small, total;
};
__host__ __device__ inline D2Lds d2_lds(int RB, int d0, int d1, int d2, int d3, int T) {
    D2Lds l;
    l.ldx = r16(d0) + 4; l.ld1 = r16(d1) + 4; l.ld2 = r16(d2) + 4;
    const int ldm = (l.ldx > l.ld1 ? (l.ldx > l.ld2 ? l.ldx : l.ld2) : (l.ld1 > l.ld2 ? l.ld1 : l.ld2));
    const int R = 16 * RB;
    l.off_a = 0;                                         // buffer A  [R][ldm]
    l.off_b = l.off_a + R * ldm;                         // buffer B  [R][ldm]
    l.off_red = l.off_b + R * ldm;                       // split-contraction partials [8][R][20]
    l.off_sc = l.off_red + 8 * R * 20;                   // scale | shift (forward), mean g | mean g x_hat (backward): 2 x D2_MAXW
    l.off_small = l.off_sc + 2 * D2_MAXW;                // H3 / H3T, Wr, br, biases, z3 slab, statistics rows, ...
    // forward: H3 [d2 d3] + Wr [T d3 + T] + z3s [R d3] + biases 3 x D2_MAXW
    // backward: H3T [d3 d2] + Wr [T d3] + dls [R T] + z3s, dz3s [R d3] x 2 + st1, st2 [4 D2_MAXW] x 2 + third slab [R][ldm]
    const int fwd = r4(d2 * d3) + r4(T * d3 + T) + r4(R * d3) + 3 * D2_MAXW;
    const int bwd = r4(d3 * d2) + r4(T * d3) + r4(R * T) + 2 * r4(R * d3) + 8 * D2_MAXW + R * ldm;
    l.total = l.off_small + (fwd > bwd ? fwd : bwd) + 16;
    return l;
}

// The loss at the end of workgroup 0 (DnArgs.loss_y): entry e by thread e % 512; per-thread sums in entry order, the 64 lanes
// of a wavefront by a fixed butterfly, the 8 wavefronts in order — one fixed summation tree whatever the grid.
__device__ __forceinline__ float d2_bce_entry(float z, float y) { return fmaxf(z, 0.f) - z * y + log1pf(expf(-fabsf(z))); }
template <int KIND>
__device__ __forceinline__ void d2_loss_tail(const DnArgs &a, unsigned tag, float *scratch) {
    const int tid = threadIdx.x, total = a.B * a.T;
    constexpr int PER = 8;                                    // total <= 4096 = 8 x 512 (host-checked)
    float zv[PER], yv[PER];
    bool dead = false;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int e = tid + D2_THREADS * q;
        zv[q] = 0.f;
        yv[q] = __int_as_float(0x7fc00000);
        if (e < total) {
            unsigned long long w;
            int spins = 0;
            do {
                w = get_granule(a.loss_gran + e);
                if ((unsigned)(w >> 32) == tag) break;
                __builtin_amdgcn_s_sleep(1);
            } while (++spins <= DN_SPIN_MAX);
            dead = dead || (unsigned)(w >> 32) != tag;
            zv[q] = __uint_as_float((unsigned)w);
            yv[q] = a.loss_y[e];
        }
    }
    float s = 0.f, c = 0.f;
#pragma unroll
    for (int q = 0; q < PER; ++q)
        if (yv[q] == yv[q]) {                                 // labeled (not NaN)
            s += KIND == 0 ? d2_bce_entry(zv[q], yv[q]) : fabsf(zv[q] - yv[q]);
            c += 1.f;
        }
    if (dead) {
        if (a.err) atomicOr(a.err, PHC_ERR_RENDEZVOUS);
        s = __int_as_float(0x7fc00000);
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        s += __shfl_xor(s, off, 64);
        c += __shfl_xor(c, off, 64);
    }
    d2_lds_barrier();                                         // (scratch: whoever used it last is done)
    if ((tid & 63) == 0) { scratch[tid >> 6] = s; scratch[8 + (tid >> 6)] = c; }
    d2_lds_barrier();
    s = c = 0.f;
#pragma unroll
    for (int w = 0; w < D2_WAVES; ++w) { s += scratch[w]; c += scratch[8 + w]; }
    if (tid == 0) {
        a.loss_out[0] = s / c;                                // no labeled entry: NaN, like the mean of an empty selection
        a.loss_out[1] = s;
        a.loss_out[2] = c;
    }
    const float scale = 1.f / c;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int e = tid + D2_THREADS * q;
        if (e < total) {
            float gq = 0.f;
            if (yv[q] == yv[q]) {
                if (KIND == 0) {
                    gq = (1.f / (1.f + expf(-zv[q])) - yv[q]) * scale;
                } else {
                    const float d = zv[q] - yv[q];
                    gq = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * scale;      // torch's sign(0) = 0
                }
            }
            a.loss_dz[e] = gq;
        }
    }
}

// The forward of a 16 RB-row workgroup up to its logits (everything but workgroup 0's leftovers: d2_fwd_tail).  k1 / k2: the norms'
// statistics of column tid (identical on every workgroup: fixed merge order); my_logit: logits entry tid of this workgroup's rows
// (R T <= 256 entries: one per thread) — what the one-launch forward + backward (downstream2_fb_kernel) carries over.
// COUNT (the one-launch forward + backward): this thread's share of the labeled entries of loss_y (entry e by thread e % 512) comes
// back in `labeled` — the labels travel with the launch's first loads and are counted behind the staging barrier, where they have
// landed anyway (counted in front of the body they held the operand loads back for a whole first-touch round trip; kept as eight
// raw values across the body they cost eight registers of a kernel that has none to spare).
template <int RB, bool COUNT = false>
__device__ __forceinline__ void d2_fwd_body(const DnArgs &a, float *sm, unsigned tag, D2Keep &k1, D2Keep &k2, float &my_logit,
                                            float *labeled = nullptr) {
    constexpr int R = 16 * RB;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = blockIdx.x, rb = g * R;
    const int d0 = a.d0, d1 = a.d1, d2 = a.d2, d3 = a.d3, T = a.T;
    const D2Lds L = d2_lds(RB, d0, d1, d2, d3, T);
    float *bufA = sm + L.off_a, *bufB = sm + L.off_b, *red = sm + L.off_red, *sc = sm + L.off_sc;
    float *Hs3 = sm + L.off_small, *wr = Hs3 + r4(d2 * d3), *z3s = wr + r4(T * d3 + T), *bs = z3s + r4(R * d3);     // bs: b1 | b2 | b3
    const int nvalid = a.n_valid ? min(*a.n_valid, a.B) : a.B;
    const unsigned long long rng_seed = a.rng ? a.rng[0] : 0ull, rng_step = a.rng ? a.rng[1] : 0ull;    // (uniform: scalar loads)

    // ---- everything this launch reads from memory goes in flight now: the first GEMM's operand slice into registers, the
    // x slab and the small operands into LDS ----
    DN_STAMP(0);
    const D2Plan p1 = d2_plan(d0, d1, wave), p2 = d2_plan(d1, d2, wave);
    D2Regs h;
    d2_load_operand(a.H1k4, d0, d1, p1, lane, h);
    float yv[COUNT ? 8 : 1];                                 // B T <= 4096 = 8 x 512 (host-checked)
    if constexpr (COUNT) {
#pragma unroll
        for (int q = 0; q < 8; ++q) yv[q] = a.loss_y[min(tid + D2_THREADS * q, a.B * T - 1)];
    }
    // First touches: a workgroup's first access to a buffer costs a page walk on top of the round trip (the first exchange of a launch
    // took 6.4 - 7.7 us against 2.3 - 3.5 for the later ones, scripts/dbg/dn_fb_time.py).  The exchange records this workgroup will
    // sweep are touched NOW, under the staging loads; the values are dropped behind the staging barrier.
    unsigned long long warm[COUNT ? 4 : 2];
    warm[0] = get_granule(a.partial + (size_t)g * D2_PSTRIDE);
    warm[1] = get_granule(a.partial + ((size_t)a.G + g) * D2_PSTRIDE);
    if constexpr (COUNT) {
        warm[2] = get_granule(a.partial_b + (size_t)g * D2_PSTRIDE);
        warm[3] = get_granule(a.partial_b + ((size_t)a.G + g) * D2_PSTRIDE);
    }
    DN_STAMP(12);
    // the norms' per-component parameter pointers -> LDS (uniform kernarg reads), each column thread fetches its own four values
    // behind the staging barrier: they arrive under the first GEMM (as run-time loops over the components with the loads inside,
    // hipcc drained the loads once per component: 3.3 us in front of the staging barrier)
    unsigned long long *ptab = reinterpret_cast<unsigned long long *>(sc);       // [2 norms][4 arrays][16 components]; sc is free until the first norm
    if (tid == 0) {
        for (int q = 0; q < a.bn1.n_comp; ++q) {
            ptab[0 * 16 + q] = (unsigned long long)a.bn1.weight[q]; ptab[1 * 16 + q] = (unsigned long long)a.bn1.bias[q];
            ptab[2 * 16 + q] = (unsigned long long)a.bn1.rmean[q]; ptab[3 * 16 + q] = (unsigned long long)a.bn1.rvar[q];
        }
        for (int q = 0; q < a.bn2.n_comp; ++q) {
            ptab[4 * 16 + q] = (unsigned long long)a.bn2.weight[q]; ptab[5 * 16 + q] = (unsigned long long)a.bn2.bias[q];
            ptab[6 * 16 + q] = (unsigned long long)a.bn2.rmean[q]; ptab[7 * 16 + q] = (unsigned long long)a.bn2.rvar[q];
        }
    }
    DN_STAMP(13);
    d2_rows_to_lds<RB>(a.x, a.B, rb, d0, bufA, L.ldx);
    DN_STAMP(14);
    for (int e = tid; e < d2 * d3; e += D2_THREADS) Hs3[e] = a.H3[e];
    for (int e = tid; e < T * d3; e += D2_THREADS) wr[e] = a.Wr[e];
    if (tid < T) wr[T * d3 + tid] = a.br ? a.br[tid] : 0.f;
    if (tid < d1) bs[tid] = a.b1 ? a.b1[tid] : 0.f;
    if (tid < d2) bs[D2_MAXW + tid] = a.b2 ? a.b2[tid] : 0.f;
    if (tid < d3) bs[2 * D2_MAXW + tid] = a.b3 ? a.b3[tid] : 0.f;
    __syncthreads();
    DN_STAMP(1);
#pragma unroll
    for (int q = 0; q < (COUNT ? 4 : 2); ++q) asm volatile("" ::"v"(warm[q]));
    if constexpr (COUNT) {
        float c = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) c += (tid + D2_THREADS * q < a.B * T && yv[q] == yv[q]) ? 1.f : 0.f;
        *labeled = c;
    }
    float w1 = 1.f, bb1 = 0.f, rm1 = 0.f, rv1 = 1.f, w2 = 1.f, bb2 = 0.f, rm2 = 0.f, rv2 = 1.f;
    {
        const int c1 = min(tid, d1 - 1), c2 = min(tid, d2 - 1);
        const int q1 = c1 / a.bn1.k, k1 = c1 - q1 * a.bn1.k, q2 = c2 / a.bn2.k, k2 = c2 - q2 * a.bn2.k;
        const float *pw1 = reinterpret_cast<const float *>(ptab[0 * 16 + q1]), *pb1 = reinterpret_cast<const float *>(ptab[1 * 16 + q1]);
        const float *pm1 = reinterpret_cast<const float *>(ptab[2 * 16 + q1]), *pv1 = reinterpret_cast<const float *>(ptab[3 * 16 + q1]);
        const float *pw2 = reinterpret_cast<const float *>(ptab[4 * 16 + q2]), *pb2 = reinterpret_cast<const float *>(ptab[5 * 16 + q2]);
        const float *pm2 = reinterpret_cast<const float *>(ptab[6 * 16 + q2]), *pv2 = reinterpret_cast<const float *>(ptab[7 * 16 + q2]);
        // (unconditional loads: without affine parameters the weight / bias slots are read from the running mean's address)
        const float lw1 = (a.bn1.affine ? pw1 : pm1)[k1], lb1 = (a.bn1.affine ? pb1 : pm1)[k1], lw2 = (a.bn2.affine ? pw2 : pm2)[k2],
                    lb2 = (a.bn2.affine ? pb2 : pm2)[k2];
        rm1 = pm1[k1]; rv1 = pv1[k1]; rm2 = pm2[k2]; rv2 = pv2[k2];
        if (a.bn1.affine) { w1 = lw1; bb1 = lb1; }
        if (a.bn2.affine) { w2 = lw2; bb2 = lb2; }
    }
    d2_lds_barrier();                                            // (the pointer table sits in `sc`: every read is done before the first norm writes it)

    // layer 1: z1 = x . H1 + b1 -> norm -> act
    d2_gemm<RB>(bufA, L.ldx, h, p1, d1, bs, red, bufB, L.ld1);
    DN_STAMP(2);
    d2_store_rows<RB>(bufB, L.ld1, a.B, rb, d1, a.z1);
    d2_fwd_norm<RB>(a, a.bn1, bufB, L.ld1, d1, nvalid, a.partial, tag, sc, red, w1, bb1, rm1, rv1, k1, 3);
    // the second GEMM's operand: issued BEHIND the exchange (its lines would compete with the granule sweep for the CU's few
    // outstanding misses: the collect took 6.5 us with the operand in flight beside it, 2.6 us without), it arrives under the
    // element-wise pass
    d2_load_operand(a.H2k4, d1, d2, p2, lane, h);
    DN_STAMP(5);
    d2_apply_norm_act<RB>(bufB, L.ld1, bufA, L.ld1, d1, sc, a.act, rb, nvalid, a.drop_p1, a.drop_same1, a.drop_k1, a.drop_site1, rng_seed, rng_step);
    if (a.a1) d2_store_rows<RB>(bufA, L.ld1, a.B, rb, d1, a.a1);
    // layer 2
    DN_STAMP(6);
    d2_gemm<RB>(bufA, L.ld1, h, p2, d2, bs + D2_MAXW, red, bufB, L.ld2);
    DN_STAMP(7);
    d2_store_rows<RB>(bufB, L.ld2, a.B, rb, d2, a.z2);
    d2_fwd_norm<RB>(a, a.bn2, bufB, L.ld2, d2, nvalid, a.partial + (size_t)a.G * D2_PSTRIDE, tag, sc, red, w2, bb2, rm2, rv2, k2, 8);
    DN_STAMP(10);
    d2_apply_norm_act<RB>(bufB, L.ld2, bufA, L.ld2, d2, sc, a.act, rb, nvalid, a.drop_p2, a.drop_same2, a.drop_k2, a.drop_site2, rng_seed, rng_step);
    if (a.a2) d2_store_rows<RB>(bufA, L.ld2, a.B, rb, d2, a.a2);
    // layer 3 (d2 -> d3 = n * target, a handful of columns) and the real head (d3 -> T): plain FMAs
    // (eight lanes per output element, each a strided eighth of the contraction, fixed-order butterfly: one thread per element
    // walked d2 dependent LDS reads — 1.6 us for 64 outputs)
    for (int e0 = 0; e0 < R * d3; e0 += D2_THREADS / 8) {
        const int e = e0 + (tid >> 3), part = tid & 7;
        const int row = min(e, R * d3 - 1) / d3, j = min(e, R * d3 - 1) - row * d3;
        float v = 0.f;
        for (int k = part; k < d2; k += 8) v = fmaf(bufA[row * L.ld2 + k], Hs3[k * d3 + j], v);
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 4, 64);
        v += bs[2 * D2_MAXW + j];
        if (part == 0 && e < R * d3) {
            z3s[row * d3 + j] = v;
            if (rb + row < a.B) a.z3[(size_t)(rb + row) * d3 + j] = v;
        }
    }
    d2_lds_barrier();
    for (int e = tid; e < R * T; e += D2_THREADS) {
        const int row = e / T, t = e - row * T;
        float v = wr[T * d3 + t];
        for (int j = 0; j < d3; ++j) v = fmaf(z3s[row * d3 + j], wr[t * d3 + j], v);
        if (e == tid) my_logit = v;
        if (rb + row < a.B) {
            a.logits[(size_t)(rb + row) * T + t] = v;
            if (a.loss_y) put_granule(a.loss_gran + (size_t)(rb + row) * T + t, v, tag);
        }
    }
    DN_STAMP(11);
}

// workgroup 0's leftovers of the forward: what the norms leave behind, the loss, the generation bump — off every other workgroup's path
template <int RB>
__device__ __forceinline__ void d2_fwd_tail(const DnArgs &a, float *sm, unsigned tag, const D2Keep &k1, const D2Keep &k2) {
    const D2Lds L = d2_lds(RB, a.d0, a.d1, a.d2, a.d3, a.T);
    float *red = sm + L.off_red;
    d2_write_norm_state(a.bn1, k1, a.stat1, a.d1);
    d2_write_norm_state(a.bn2, k2, a.stat2, a.d2);
    if (a.loss_y) {
        if (a.loss_kind == 0)
            d2_loss_tail<0>(a, tag, red);
        else
            d2_loss_tail<1>(a, tag, red);
    }
    // (the bump comes behind the collect: every workgroup has published its logits, i.e. has read the generation word)
    if (threadIdx.x == 0) __hip_atomic_fetch_add(a.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int RB>
__global__ __launch_bounds__(D2_THREADS) void downstream2_fwd_kernel(DnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const unsigned tag = __hip_atomic_load(a.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    D2Keep k1 = {}, k2 = {};
    float my_logit = 0.f;
    d2_fwd_body<RB>(a, sm, tag, k1, k2, my_logit);
    if (blockIdx.x == 0) d2_fwd_tail<RB>(a, sm, tag, k1, k2);
}


// backward norm (as bwd_norm): Gm = g, XH = x_hat, both [16 RB][ld] in LDS; Gm := dz in place.  keep_sg / keep_sgx: the column
// totals (workgroup 0 writes dweight | dbias from them at the end of the launch); scratch: 2 x 512 floats of LDS
template <int RB>
__device__ __forceinline__ void d2_bwd_norm(const DnArgs &a, float *Gm, const float *XH, int ld, int N, int nvalid, int rb,
                                            unsigned long long *part, unsigned tag, const float *st, float *sc, float *scratch,
                                            float &keep_sg, float &keep_sgx, int stamp) {
    const int tid = threadIdx.x, g = blockIdx.x;
    if (tid < N) {
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int r = 0; r < 16 * RB; ++r) {
            const float gv = Gm[r * ld + tid];
            sg += gv;
            sgx = fmaf(gv, XH[r * ld + tid], sgx);
        }
        d2_put_pair(part + (size_t)g * D2_PSTRIDE, tid, sg, sgx, tag);
    }
    DN_STAMP(stamp);
    const D2Deal dl = d2_deal(N, a.G);
    const int col = tid % dl.nc, pt = tid / dl.nc;
    float SG = 0.f, SGX = 0.f;
    if (pt < dl.parts && col < N) {
        const int gbeg = pt * dl.gp, gend = min(gbeg + dl.gp, a.G);
        for (int g0 = gbeg; g0 < gend; g0 += D2_MERGE) {
            float pa[D2_MERGE], pb[D2_MERGE];
            d2_collect(part, gend, g0, col, tag, pa, pb, a.err);
#pragma unroll
            for (int u = 0; u < D2_MERGE; ++u)
                if (g0 + u < gend) {
                    SG += pa[u];
                    SGX += pb[u];
                }
        }
        scratch[pt * dl.nc + col] = SG;
        scratch[512 + pt * dl.nc + col] = SGX;
    }
    d2_lds_barrier();
    DN_STAMP(stamp + 1);
    if (tid < N) {
        for (int q = 1; q < dl.parts; ++q) {
            SG += scratch[q * dl.nc + tid];
            SGX += scratch[512 + q * dl.nc + tid];
        }
        const float cnt = (float)max(nvalid, 1);
        sc[tid] = SG / cnt;
        sc[D2_MAXW + tid] = SGX / cnt;
        keep_sg = SG;
        keep_sgx = SGX;
    }
    d2_lds_barrier();
    const int n4 = r16(N) >> 2;
    for (int e = tid; e < 16 * RB * n4; e += D2_THREADS) {
        const int row = e / n4, c = (e - row * n4) << 2;
        float4 gq = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < N && rb + row < nvalid) {
            gq = *reinterpret_cast<const float4 *>(Gm + row * ld + c);
            const float4 xh = *reinterpret_cast<const float4 *>(XH + row * ld + c);
            const float4 w4 = *reinterpret_cast<const float4 *>(st + 2 * D2_MAXW + c), i4 = *reinterpret_cast<const float4 *>(st + D2_MAXW + c);
            const float4 mg = *reinterpret_cast<const float4 *>(sc + c), mx = *reinterpret_cast<const float4 *>(sc + D2_MAXW + c);
            gq.x = w4.x * i4.x * (gq.x - mg.x - xh.x * mx.x);
            gq.y = w4.y * i4.y * (gq.y - mg.y - xh.y * mx.y);
            gq.z = w4.z * i4.z * (gq.z - mg.z - xh.z * mx.z);
            gq.w = w4.w * i4.w * (gq.w - mg.w - xh.w * mx.w);
        }
        *reinterpret_cast<float4 *>(Gm + row * ld + c) = gq;
    }
    d2_lds_barrier();
}

// The extras of the first backward exchange (dWr | dbr partials of every workgroup) summed by workgroup 0 at the END of the
// launch: extra j by wavefront j % 8, the G <= 128 workgroups over its lanes (two per lane), eight extras in flight per round,
// fixed-order butterfly.  (Merged in place by one wavefront, one granule at a time, this was 34 dependent round trips in front
// of the barrier that every other workgroup's next exchange waits behind.)
__device__ __forceinline__ void d2_merge_extras(const unsigned long long *part, int G, unsigned tag, int n_extra, float *out, int *err) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j0 = 8 * wave; j0 < n_extra; j0 += 64) {           // this wavefront's extras j0 .. j0 + 7 (of every group of 64)
        float v[8][2];
        bool ok;
        int spins = 0;
        do {
            ok = true;
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int j = min(j0 + u, n_extra - 1), gg = lane + 64 * hh;
                    const unsigned long long q = get_granule(part + (size_t)min(gg, G - 1) * D2_PSTRIDE + 2 * D2_MAXW + j);
                    v[u][hh] = gg < G ? __uint_as_float((unsigned)q) : 0.f;
                    ok = ok && (gg >= G || (unsigned)(q >> 32) == tag);
                }
            if (!ok && ++spins > DN_SPIN_MAX) {
                if (err) atomicOr(err, PHC_ERR_RENDEZVOUS);
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u][0] = v[u][1] = __int_as_float(0x7fc00000);
                break;
            }
        } while (!ok);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float sum = v[u][0] + v[u][1];
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) sum += __shfl_xor(sum, off, 64);
            if (lane == 0 && j0 + u < n_extra) out[j0 + u] = sum;
        }
    }
}

// g = da * act'(x_hat w + b), x_hat = (z - mean) invstd for the slab: DA (LDS) and Z (LDS) -> Gout, XHout (may alias DA / Z)
template <int RB>
__device__ __forceinline__ void d2_gate_rows(const float *DA, const float *Z, int ld, int N, const float *st, int act, int rb, int nvalid,
                                             float *Gout, float *XHout, float drop_p = 0.f, int drop_same = 0, int drop_k = 4,
                                             int drop_site = 0, unsigned long long seed = 0, unsigned long long step = 0) {
    const bool relu = act == PHC_ACT_RELU;
    const int Np = r16(N);
    for (int e = threadIdx.x; e < 16 * RB * Np; e += D2_THREADS) {
        const int row = e / Np, c = e - row * Np;
        float gv = 0.f, xh = 0.f;
        if (c < N) {
            xh = (Z[row * ld + c] - st[c]) * st[D2_MAXW + c];
            const float pre = fmaf(xh, st[2 * D2_MAXW + c], st[3 * D2_MAXW + c]);
            const float da = DA[row * ld + c];
            gv = relu ? (pre > 0.f ? da : 0.f) : da * act_grad_rt(act, pre);
            if (rb + row >= nvalid) gv = 0.f;
            if (drop_p > 0.f) {                                  // the forward's mask on the upstream gradient (uniform branch)
                const float4 f = drop_factors_v(drop_p, drop_same, drop_k, drop_site, seed, step, rb + row, c & ~3, N);
                const int q = c & 3;
                gv *= q == 0 ? f.x : (q == 1 ? f.y : (q == 2 ? f.z : f.w));
            }
        }
        Gout[row * ld + c] = gv;
        XHout[row * ld + c] = xh;
    }
    d2_lds_barrier();
}

// The backward of a 16 RB-row workgroup (everything but workgroup 0's leftovers: d2_bwd_tail).  FUSED: the part behind the forward
// in ONE launch (downstream2_fb_kernel) — d loss / d logits of the workgroup's rows is the caller's (entry tid: my_dlogit), the
// norms' mean | invstd | w | b were put in place by the caller (d2_bwd_stats: workgroup 0 has not written save_stat yet), z1 / z2 / z3
// are re-read from memory (this workgroup's own stores, complete behind the caller's barrier).
template <int RB, bool FUSED>
__device__ __forceinline__ void d2_bwd_body(const DnArgs &a, unsigned long long *partial, float *sm, unsigned tag, float my_dlogit,
                                            float &sg1, float &sgx1, float &sg2, float &sgx2) {
    constexpr int R = 16 * RB;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = blockIdx.x, rb = g * R;
    const int d0 = a.d0, d1 = a.d1, d2 = a.d2, d3 = a.d3, T = a.T;
    const D2Lds L = d2_lds(RB, d0, d1, d2, d3, T);
    const int ldm = max(L.ldx, max(L.ld1, L.ld2));
    float *bufA = sm + L.off_a, *bufB = sm + L.off_b, *red = sm + L.off_red, *sc = sm + L.off_sc;
    float *Hs3T = sm + L.off_small, *wr = Hs3T + r4(d3 * d2), *dls = wr + r4(T * d3), *z3s = dls + r4(R * T), *dz3s = z3s + r4(R * d3),
          *st1 = dz3s + r4(R * d3), *st2 = st1 + 4 * D2_MAXW, *bufC = st2 + 4 * D2_MAXW;        // st*: mean | invstd | w | b
    const int nvalid = a.n_valid ? min(*a.n_valid, a.B) : a.B;
    const unsigned long long rng_seed = a.rng ? a.rng[0] : 0ull, rng_step = a.rng ? a.rng[1] : 0ull;    // (uniform: scalar loads)
    (void)ldm;

    // ---- everything in flight: the first GEMM's operand (H2^T) into registers, the slabs of z2 / z1 and the small operands into LDS ----
    DN_STAMP(16);
    const D2Plan pa = d2_plan(d2, d1, wave), pb = d2_plan(d1, d0, wave);
    D2Regs h;
    d2_load_operand(a.H2Tk4, d2, d1, pa, lane, h);
    d2_rows_to_lds<RB>(a.z2, a.B, rb, d2, bufB, L.ld2);          // bufB: z2 slab -> x_hat2
    d2_rows_to_lds<RB>(a.z1, a.B, rb, d1, bufC, L.ld1);          // bufC: z1 slab (kept until its turn)
    for (int e = tid; e < d3 * d2; e += D2_THREADS) Hs3T[e] = a.H3T[e];
    for (int e = tid; e < T * d3; e += D2_THREADS) wr[e] = a.Wr[e];
    if constexpr (FUSED) {
        if (tid < R * T) dls[tid] = rb * T + tid < a.B * T ? my_dlogit : 0.f;        // (R T <= 256: one entry per thread)
    } else {
        for (int e = tid; e < R * T; e += D2_THREADS) dls[e] = rb * T + e < a.B * T ? a.dlogits[(size_t)rb * T + e] : 0.f;
    }
    for (int e = tid; e < R * d3; e += D2_THREADS) z3s[e] = rb * d3 + e < a.B * d3 ? a.z3[(size_t)rb * d3 + e] : 0.f;
    if constexpr (!FUSED) {                                  // (FUSED: the caller wrote st1 / st2 out of the forward's registers: d2_bwd_stats)
#pragma unroll
        for (int q = 0; q < 4; ++q) {                        // mean, invstd, w, b of both norms (save_stat slots 0, 1, 4, 5)
            const int slot = q < 2 ? q : q + 2;
            if (tid < d1) st1[q * D2_MAXW + tid] = a.stat1[slot * d1 + tid];
            if (tid < d2) st2[q * D2_MAXW + tid] = a.stat2[slot * d2 + tid];
        }
    }
    __syncthreads();
    DN_STAMP(17);

    // head and layer 3 backwards (a handful of columns): dz3 = dlogits . Wr, da2 = dz3 . H3^T; partial dWr | dbr of these rows
    for (int e = tid; e < R * d3; e += D2_THREADS) {
        const int row = e / d3, j = e - row * d3;
        float v = 0.f;
        for (int t = 0; t < T; ++t) v = fmaf(dls[row * T + t], wr[t * d3 + j], v);
        dz3s[row * d3 + j] = v;
        if (rb + row < a.B) a.dz3[(size_t)(rb + row) * d3 + j] = v;
    }
    unsigned long long *part1 = partial, *part2 = partial + (size_t)a.G * D2_PSTRIDE;      // (the caller's: a.partial, or the one-launch form's a.partial_b)
    for (int e = tid; e < T * d3 + T; e += D2_THREADS) {
        float s = 0.f;
        if (e < T * d3) {
            const int t = e / d3, j = e - t * d3;
            for (int r = 0; r < R; ++r) s = fmaf(dls[r * T + t], z3s[r * d3 + j], s);
        } else {
            const int t = e - T * d3;
            for (int r = 0; r < R; ++r) s += dls[r * T + t];
        }
        put_granule(part1 + (size_t)g * D2_PSTRIDE + 2 * D2_MAXW + e, s, tag);
    }
    d2_lds_barrier();
    // da2 = dz3 . H3^T -> bufA; then g2 -> bufA, x_hat2 -> bufB (over z2)
    {
        const int Np = r16(d2);
        for (int e = tid; e < R * Np; e += D2_THREADS) {
            const int row = e / Np, c = e - row * Np;
            float da = 0.f;
            if (c < d2)
                for (int j = 0; j < d3; ++j) da = fmaf(dz3s[row * d3 + j], Hs3T[j * d2 + c], da);
            bufA[row * L.ld2 + c] = da;
        }
    }
    d2_lds_barrier();
    DN_STAMP(18);
    d2_gate_rows<RB>(bufA, bufB, L.ld2, d2, st2, a.act, rb, nvalid, bufA, bufB, a.drop_p2, a.drop_same2, a.drop_k2, a.drop_site2, rng_seed, rng_step);
    DN_STAMP(19);
    d2_bwd_norm<RB>(a, bufA, bufB, L.ld2, d2, nvalid, rb, part1, tag, st2, sc, red, sg2, sgx2, 20);
    DN_STAMP(22);
    d2_store_rows<RB>(bufA, L.ld2, a.B, rb, d2, a.dz2);
    // da1 = dz2 . H2^T -> bufB; the last GEMM's operand (H1^T) goes in flight behind it
    d2_gemm<RB>(bufA, L.ld2, h, pa, d1, nullptr, red, bufB, L.ld1);
    DN_STAMP(23);
    if (a.dx) d2_load_operand(a.H1Tk4, d1, d0, pb, lane, h);
    d2_gate_rows<RB>(bufB, bufC, L.ld1, d1, st1, a.act, rb, nvalid, bufA, bufB, a.drop_p1, a.drop_same1, a.drop_k1, a.drop_site1, rng_seed,
                     rng_step);                                                          // g1 -> bufA, x_hat1 -> bufB
    DN_STAMP(24);
    d2_bwd_norm<RB>(a, bufA, bufB, L.ld1, d1, nvalid, rb, part2, tag, st1, sc, red, sg1, sgx1, 25);
    DN_STAMP(27);
    d2_store_rows<RB>(bufA, L.ld1, a.B, rb, d1, a.dz1);
    // dx = dz1 . H1^T
    if (a.dx) {
        d2_gemm<RB>(bufA, L.ld1, h, pb, d0, nullptr, red, bufB, L.ldx);
        DN_STAMP(28);
        d2_store_rows<RB>(bufB, L.ldx, a.B, rb, d0, a.dx);
    }
    DN_STAMP(29);
}

// where d2_bwd_body keeps the norms' mean | invstd | w | b rows (st1; st2 = st1 + 4 D2_MAXW): the one-launch form fills them itself
template <int RB>
__device__ __forceinline__ float *d2_bwd_stats(const DnArgs &a, float *sm) {
    constexpr int R = 16 * RB;
    const D2Lds L = d2_lds(RB, a.d0, a.d1, a.d2, a.d3, a.T);
    return sm + L.off_small + r4(a.d3 * a.d2) + r4(a.T * a.d3) + r4(R * a.T) + 2 * r4(R * a.d3);
}

// workgroup 0's leftovers of the backward (the norms' affine gradients, the dense head's, the generation bump), off every other
// workgroup's path
__device__ __forceinline__ void d2_bwd_tail(const DnArgs &a, const unsigned long long *partial, unsigned *sync, unsigned tag, float sg1, float sgx1,
                                            float sg2, float sgx2) {
    const int tid = threadIdx.x;
    if (a.dwb2 && tid < a.d2) { a.dwb2[tid] = sgx2; a.dwb2[a.d2 + tid] = sg2; }
    if (a.dwb1 && tid < a.d1) { a.dwb1[tid] = sgx1; a.dwb1[a.d1 + tid] = sg1; }
    if (a.dWr) d2_merge_extras(partial, a.G, tag, a.T * a.d3 + a.T, a.dWr, a.err);
    if (tid == 0) __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int RB>
__global__ __launch_bounds__(D2_THREADS) void downstream2_bwd_kernel(DnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const unsigned tag = __hip_atomic_load(a.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    float sg1 = 0.f, sgx1 = 0.f, sg2 = 0.f, sgx2 = 0.f;
    d2_bwd_body<RB, false>(a, a.partial, sm, tag, 0.f, sg1, sgx1, sg2, sgx2);
    if (blockIdx.x == 0) d2_bwd_tail(a, a.partial, a.sync, tag, sg1, sgx1, sg2, sgx2);
}

// Forward AND backward of the head in ONE launch (round 6): a training step whose loss is taken inside the forward launch
// (DnArgs.loss_y) knows d loss / d logits of its rows as soon as it knows the labeled count — every workgroup counts the labels
// itself (<= 4096 entries, read at kernel entry) — so, for an upstream gradient of 1, the backward can start where the forward
// ends: no launch boundary (its 8 us floor, the 6 us first dependent load of the backward's workgroups), the operands' lines warm.
// The two parts keep their own exchange buffers and generation words (partial / sync, partial_b / sync_b): the protocols are the
// separate launches'.  The parts' leftovers come behind the backward part (on two workgroups).  Same arithmetic as the two launches:
// bit-identical outputs.
template <int RB>
__global__ __launch_bounds__(D2_THREADS) void downstream2_fb_kernel(DnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int R = 16 * RB;
    const int tid = threadIdx.x, rb = blockIdx.x * R;
    const unsigned tag_f = __hip_atomic_load(a.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    const unsigned tag_b = __hip_atomic_load(a.sync_b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    const int total = a.B * a.T;
    const int mine = rb * a.T + tid;                          // (entry tid of this workgroup's rows: tid < R T)
    const float my_y = a.loss_y[min(mine, total - 1)];
    D2Keep k1 = {}, k2 = {};
    float my_logit = 0.f, c = 0.f;
    d2_fwd_body<RB, true>(a, sm, tag_f, k1, k2, my_logit, &c);
    // the labeled count (an integer in a float: the same number whatever the order) and this thread's d loss / d logit, as
    // d2_loss_tail writes it for an upstream gradient of 1
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) c += __shfl_xor(c, off, 64);
    const D2Lds L = d2_lds(RB, a.d0, a.d1, a.d2, a.d3, a.T);
    float *scratch = sm + L.off_red;
    float *stash = sm + L.total;                              // [14][256] behind both parts' carve-ups: the norms' statistics of column tid
    __syncthreads();                                          // the forward part's LDS is dead, its stores (z1, z2, z3: re-read below) complete
    if ((tid & 63) == 0) scratch[tid >> 6] = c;
    if (tid < D2_MAXW) {                                      // (workgroup 0's forward leftovers come behind the backward part: 14 registers less across it)
        stash[0 * D2_MAXW + tid] = k1.mean; stash[1 * D2_MAXW + tid] = k1.invstd; stash[2 * D2_MAXW + tid] = k1.w; stash[3 * D2_MAXW + tid] = k1.b;
        stash[4 * D2_MAXW + tid] = k1.unbiased; stash[5 * D2_MAXW + tid] = k1.rm_old; stash[6 * D2_MAXW + tid] = k1.rv_old;
        stash[7 * D2_MAXW + tid] = k2.mean; stash[8 * D2_MAXW + tid] = k2.invstd; stash[9 * D2_MAXW + tid] = k2.w; stash[10 * D2_MAXW + tid] = k2.b;
        stash[11 * D2_MAXW + tid] = k2.unbiased; stash[12 * D2_MAXW + tid] = k2.rm_old; stash[13 * D2_MAXW + tid] = k2.rv_old;
        stash[14 * D2_MAXW + tid] = __int_as_float(k1.valid | (k2.valid << 1));
        // ... and the backward part's mean | invstd | w | b rows of both norms, where d2_bwd_body reads them
        float *st1 = d2_bwd_stats<RB>(a, sm), *st2 = st1 + 4 * D2_MAXW;
        if (tid < a.d1) { st1[tid] = k1.mean; st1[D2_MAXW + tid] = k1.invstd; st1[2 * D2_MAXW + tid] = k1.w; st1[3 * D2_MAXW + tid] = k1.b; }
        if (tid < a.d2) { st2[tid] = k2.mean; st2[D2_MAXW + tid] = k2.invstd; st2[2 * D2_MAXW + tid] = k2.w; st2[3 * D2_MAXW + tid] = k2.b; }
    }
    d2_lds_barrier();
    c = 0.f;
#pragma unroll
    for (int w = 0; w < D2_WAVES; ++w) c += scratch[w];
    d2_lds_barrier();
    float my_dlogit = 0.f;
    if (mine < total && my_y == my_y) {
        const float scale = 1.f / c;
        if (a.loss_kind == 0) {
            my_dlogit = (1.f / (1.f + expf(-my_logit)) - my_y) * scale;
        } else {
            const float d = my_logit - my_y;
            my_dlogit = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * scale;
        }
    }
    float sg1 = 0.f, sgx1 = 0.f, sg2 = 0.f, sgx2 = 0.f;
    d2_bwd_body<RB, true>(a, a.partial_b, sm, tag_b, my_dlogit, sg1, sgx1, sg2, sgx2);
    // the two parts' leftovers on two different workgroups: the forward's (the norms' state, the loss: a collect of all logits) on the
    // LAST one — fewest rows, first through its backward part —, the backward's on workgroup 0; on one workgroup they were one behind
    // the other at the end of the launch's longest path
    if (blockIdx.x == gridDim.x - 1) {
        if (tid < D2_MAXW) {
            k1.mean = stash[0 * D2_MAXW + tid]; k1.invstd = stash[1 * D2_MAXW + tid]; k1.w = stash[2 * D2_MAXW + tid]; k1.b = stash[3 * D2_MAXW + tid];
            k1.unbiased = stash[4 * D2_MAXW + tid]; k1.rm_old = stash[5 * D2_MAXW + tid]; k1.rv_old = stash[6 * D2_MAXW + tid];
            k2.mean = stash[7 * D2_MAXW + tid]; k2.invstd = stash[8 * D2_MAXW + tid]; k2.w = stash[9 * D2_MAXW + tid]; k2.b = stash[10 * D2_MAXW + tid];
            k2.unbiased = stash[11 * D2_MAXW + tid]; k2.rm_old = stash[12 * D2_MAXW + tid]; k2.rv_old = stash[13 * D2_MAXW + tid];
            const int v = __float_as_int(stash[14 * D2_MAXW + tid]);
            k1.valid = v & 1; k2.valid = (v >> 1) & 1;
        }
        d2_fwd_tail<RB>(a, sm, tag_f, k1, k2);
    }
    if (blockIdx.x == 0) d2_bwd_tail(a, a.partial_b, a.sync_b, tag_b, sg1, sgx1, sg2, sgx2);
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

// ---- v2 (16-row workgroups, k4 operands) ----
static inline bool dn2_enabled() {
    static const bool off = getenv("PHC_DOWNSTREAM_V2") && atoi(getenv("PHC_DOWNSTREAM_V2")) == 0;
    return !off;
}
// rows per workgroup: 16 (RB = 1) up to 1280 rows, 32 above (at most D2_MAX_WG workgroups).
static inline int dn2_rb(int B) {
    constexpr int forced = 0;
    if (forced == 1 && (B + 15) / 16 <= D2_MAX_WG) return 1;
    // (1280, not 1024: the graph-replay batch of configs[2] is 1024 + 32 padding graphs — on 33 workgroups of 32 rows its head took
    // 65 + 63 us of the step, on 66 of 16 rows zinc_b1024 is 1.870 -> 1.848 ms)
    return (B > 1280 || forced == 2) ? 2 : 1;
}
static bool dn2_shape_ok(int B, int d0, int d1, int d2, int d3, int T) {
    auto wide = [](int d) { return d >= 16 && d <= D2_MAXW && d % 4 == 0; };
    if (!(dn2_enabled() && B >= 1 && wide(d0) && wide(d1) && wide(d2) && d3 >= 1 && d3 <= DN_MAXD3 && T >= 1 && T <= DN_MAXT)) return false;
    const int RB = dn2_rb(B);
    if ((B + 16 * RB - 1) / (16 * RB) > D2_MAX_WG) return false;
    return sizeof(float) * (size_t)d2_lds(RB, d0, d1, d2, d3, T).total <= 150 * 1024;
}
template <typename K>
static int dn2_capacity_of(K kernel, int slot, size_t lds_bytes) {
    static const bool masked = getenv("HSA_CU_MASK") || getenv("ROC_GLOBAL_CU_MASK") || getenv("HSA_CU_MASK_SKIP_INIT");
    if (masked) return 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    struct Slot { size_t lds; int cap; };
    static Slot cache[64][6] = {};
    Slot &sl = cache[dev][slot];
    if (sl.lds == lds_bytes && sl.cap > 0) return sl.cap;
    if (!ensure_dynamic_lds(kernel, 150 * 1024)) return 0;
    int per_cu = 0;
    hipDeviceProp_t prop;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, D2_THREADS, lds_bytes) != hipSuccess ||
        hipGetDeviceProperties(&prop, dev) != hipSuccess || per_cu < 1 || prop.multiProcessorCount < 1) {
        (void)hipGetLastError();
        return 0;
    }
    sl.lds = lds_bytes;
    sl.cap = per_cu * prop.multiProcessorCount;
    return sl.cap;
}
static bool dn2_device_ok(int B, int d0, int d1, int d2, int d3, int T) {
    const int RB = dn2_rb(B), G = (B + 16 * RB - 1) / (16 * RB);
    const size_t lds = sizeof(float) * (size_t)d2_lds(RB, d0, d1, d2, d3, T).total;
    if (RB == 1)
        return dn2_capacity_of(&downstream2_fwd_kernel<1>, 0, lds) >= G && dn2_capacity_of(&downstream2_bwd_kernel<1>, 1, lds) >= G;
    return dn2_capacity_of(&downstream2_fwd_kernel<2>, 2, lds) >= G && dn2_capacity_of(&downstream2_bwd_kernel<2>, 3, lds) >= G;
}

// ... and of the one-launch forward + backward
static bool dn2_fb_device_ok(int B, int d0, int d1, int d2, int d3, int T) {
    static const bool off = getenv("PHC_HEAD_FUSED_BWD") && atoi(getenv("PHC_HEAD_FUSED_BWD")) == 0;
    if (off) return false;
    const int RB = dn2_rb(B), G = (B + 16 * RB - 1) / (16 * RB);
    const size_t lds = sizeof(float) * ((size_t)d2_lds(RB, d0, d1, d2, d3, T).total + D2_FB_STASH);
    if (lds > 150 * 1024) return false;
    return RB == 1 ? dn2_capacity_of(&downstream2_fb_kernel<1>, 4, lds) >= G : dn2_capacity_of(&downstream2_fb_kernel<2>, 5, lds) >= G;
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

// v2 serves a call when its k4 operands are there and the shape / device allow it; *v2 says which kernels the caller launches
static int dn_common(const char *who, const phc_downstream_desc *d, DnArgs *a, bool *v2) {
    PHC_REQUIRE(d, PHC_ERR_INVALID_ARG, "%s: null descriptor", who);
    PHC_REQUIRE(d->B >= 1 && d->B <= (1 << 20), PHC_ERR_UNSUPPORTED, "%s: B=%lld", who, (long long)d->B);
    PHC_REQUIRE(d->act >= PHC_ACT_IDENTITY && d->act <= PHC_ACT_SELU, PHC_ERR_INVALID_ARG, "%s: unknown activation %d", who, d->act);
    *v2 = d->H1k4 && d->H2k4 && d->H1Tk4 && d->H2Tk4 && dn2_shape_ok((int)d->B, d->d0, d->d1, d->d2, d->d3, d->T) &&
          dn2_device_ok((int)d->B, d->d0, d->d1, d->d2, d->d3, d->T) &&
          ((reinterpret_cast<uintptr_t>(d->H1k4) | reinterpret_cast<uintptr_t>(d->H2k4) | reinterpret_cast<uintptr_t>(d->H1Tk4) |
            reinterpret_cast<uintptr_t>(d->H2Tk4)) & 15) == 0;
    if (!*v2) {
        PHC_REQUIRE(dn_shape_ok((int)d->B, d->d0, d->d1, d->d2, d->d3, d->T), PHC_ERR_UNSUPPORTED,
                    "%s: B=%lld widths %d-%d-%d-%d-%d outside the fused head's range (use the per-layer path)", who, (long long)d->B, d->d0,
                    d->d1, d->d2, d->d3, d->T);
        PHC_REQUIRE(dn_device_ok((int)d->B, d->d0, d->d1, d->d2, d->d3, d->T), PHC_ERR_UNSUPPORTED,
                    "%s: the %lld workgroups of the fused head cannot all be resident on this device (partitioned device or CU mask): use "
                    "the per-layer path", who, (long long)((d->B + DN_ROWS - 1) / DN_ROWS));
    }
    PHC_REQUIRE(d->partial && d->sync, PHC_ERR_WORKSPACE, "%s: partial / sync buffers missing", who);
    a->B = (int)d->B; a->d0 = d->d0; a->d1 = d->d1; a->d2 = d->d2; a->d3 = d->d3; a->T = d->T; a->act = d->act;
    a->G = *v2 ? (a->B + 16 * dn2_rb(a->B) - 1) / (16 * dn2_rb(a->B)) : (a->B + DN_ROWS - 1) / DN_ROWS;
    a->H1k4 = d->H1k4; a->H2k4 = d->H2k4; a->H1Tk4 = d->H1Tk4; a->H2Tk4 = d->H2Tk4;
    {
        const bool dropping = d->drop_p1 > 0.f || d->drop_p2 > 0.f;
        PHC_REQUIRE(!dropping || *v2, PHC_ERR_UNSUPPORTED, "%s: dropout inside the fused head needs the 16-row kernels (k4 operands, their shapes)", who);
        PHC_REQUIRE(!dropping || (d->rng_state && d->drop_p1 < 1.f && d->drop_p2 < 1.f && d->drop_p1 >= 0.f && d->drop_p2 >= 0.f),
                    PHC_ERR_INVALID_ARG, "%s: dropout p = %g / %g needs 0 <= p < 1 and an rng state", who, (double)d->drop_p1, (double)d->drop_p2);
        PHC_REQUIRE(!dropping || ((d->drop_same_k1 == 0 || (d->drop_same_k1 % 4 == 0 && d->d1 % d->drop_same_k1 == 0)) &&
                                  (d->drop_same_k2 == 0 || (d->drop_same_k2 % 4 == 0 && d->d2 % d->drop_same_k2 == 0))),
                    PHC_ERR_UNSUPPORTED, "%s: a shared dropout mask needs component widths that are multiples of 4", who);
        a->drop_p1 = d->drop_p1; a->drop_p2 = d->drop_p2;
        a->drop_same1 = d->drop_same_k1 > 0; a->drop_same2 = d->drop_same_k2 > 0;
        a->drop_k1 = d->drop_same_k1 > 0 ? d->drop_same_k1 : d->d1;
        a->drop_k2 = d->drop_same_k2 > 0 ? d->drop_same_k2 : d->d2;
        a->drop_site1 = d->drop_site1; a->drop_site2 = d->drop_site2;
        a->rng = dropping ? reinterpret_cast<const unsigned long long *>(d->rng_state) : nullptr;
        a->a1 = *v2 ? d->a1 : nullptr; a->a2 = *v2 ? d->a2 : nullptr;
        PHC_REQUIRE(((reinterpret_cast<uintptr_t>(d->a1) | reinterpret_cast<uintptr_t>(d->a2)) & 15) == 0, PHC_ERR_UNSUPPORTED,
                    "%s: 16-byte aligned a1 / a2 required", who);
    }
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
    const int Bc = (int)(B > (1LL << 20) ? 1 << 20 : B);
    return (dn_shape_ok(Bc, d0, d1, d2, d3, T) && dn_device_ok(Bc, d0, d1, d2, d3, T)) ? 1 : 0;
}
/* with the k4 operand forms at hand (phc_downstream_desc.H1k4 ...): the wider range of the 16-row kernels */
int phc_downstream_supported_k4(int64_t B, int d0, int d1, int d2, int d3, int T) {
    const int Bc = (int)(B > (1LL << 20) ? 1 << 20 : B);
    if (dn2_shape_ok(Bc, d0, d1, d2, d3, T) && dn2_device_ok(Bc, d0, d1, d2, d3, T)) return 1;
    return phc_downstream_supported(B, d0, d1, d2, d3, T);
}

/* dropout inside the fused head (phc_downstream_desc.drop_*): served by the 16-row kernels only */
int phc_downstream_dropout_supported(int64_t B, int d0, int d1, int d2, int d3, int T) {
    const int Bc = (int)(B > (1LL << 20) ? 1 << 20 : B);
    return (d1 % 4 == 0 && d2 % 4 == 0 && dn2_shape_ok(Bc, d0, d1, d2, d3, T) && dn2_device_ok(Bc, d0, d1, d2, d3, T)) ? 1 : 0;
}

/* ABI 8: phc_downstream_desc.loss_* are served (the 16-row kernels' range and at most 4096 logits) */
int phc_downstream_loss_supported(int64_t B, int d0, int d1, int d2, int d3, int T) {
    const int Bc = (int)(B > (1LL << 20) ? 1 << 20 : B);
    return ((long long)B * T <= DN_LOSS_MAX && dn2_shape_ok(Bc, d0, d1, d2, d3, T) && dn2_device_ok(Bc, d0, d1, d2, d3, T)) ? 1 : 0;
}

/* ABI 8: phc_downstream_desc.fused_bwd is served: forward (with the loss) AND backward for an upstream gradient of 1 in ONE launch */
int phc_downstream_fused_bwd_supported(int64_t B, int d0, int d1, int d2, int d3, int T) {
    const int Bc = (int)(B > (1LL << 20) ? 1 << 20 : B);
    return (phc_downstream_loss_supported(B, d0, d1, d2, d3, T) && 16 * dn2_rb(Bc) * T <= 256 && dn2_fb_device_ok(Bc, d0, d1, d2, d3, T)) ? 1 : 0;
}

size_t phc_downstream_partial_floats(int64_t B) {
    if (B < 1) return 0;
    const size_t v1 = 2 * 2 * (size_t)((B + DN_ROWS - 1) / DN_ROWS) * DN_PSTRIDE;      // 8-byte {value, tag} granules
    const size_t v2 = 2 * 2 * (size_t)((B + 15) / 16) * D2_PSTRIDE;
    return v1 > v2 ? v1 : v2;
}

int phc_downstream_fwd(const phc_downstream_desc *d, void *stream) {
    DnArgs a = {};
    bool v2 = false;
    int rc = dn_common("phc_downstream_fwd", d, &a, &v2);
    if (rc) return rc;
    rc = fill_norm("phc_downstream_fwd", &a.bn1, d->norm1, d->d1);
    if (rc) return rc;
    rc = fill_norm("phc_downstream_fwd", &a.bn2, d->norm2, d->d2);
    if (rc) return rc;
    a.x = d->x; a.H1 = d->H1; a.b1 = d->b1; a.H2 = d->H2; a.b2 = d->b2; a.H3 = d->H3; a.b3 = d->b3; a.logits = d->logits;
    PHC_REQUIRE(a.x && a.H1 && a.H2 && a.H3 && a.logits, PHC_ERR_INVALID_ARG, "phc_downstream_fwd: null pointer");
    if (d->loss_y) {
        PHC_REQUIRE(v2 && (long long)d->B * d->T <= DN_LOSS_MAX, PHC_ERR_UNSUPPORTED,
                    "phc_downstream_fwd: the loss inside the launch needs the 16-row kernels and B T <= %d (phc_downstream_loss_supported)", DN_LOSS_MAX);
        PHC_REQUIRE(d->loss_out && d->loss_dlogits && d->loss_granules && (d->loss_kind == 0 || d->loss_kind == 1) &&
                        (reinterpret_cast<uintptr_t>(d->loss_granules) & 7) == 0,
                    PHC_ERR_INVALID_ARG, "phc_downstream_fwd: loss_out / loss_dlogits / loss_granules / loss_kind");
        a.loss_y = d->loss_y; a.loss_out = d->loss_out; a.loss_dz = d->loss_dlogits; a.loss_kind = d->loss_kind;
        a.loss_gran = reinterpret_cast<unsigned long long *>(d->loss_granules);
    }
    PHC_REQUIRE(((reinterpret_cast<uintptr_t>(a.x) | reinterpret_cast<uintptr_t>(a.H1) | reinterpret_cast<uintptr_t>(a.H2) |
                  reinterpret_cast<uintptr_t>(a.z1) | reinterpret_cast<uintptr_t>(a.z2)) & 15) == 0,
                PHC_ERR_UNSUPPORTED, "phc_downstream_fwd: 16-byte aligned rows required");
    hipStream_t s = (hipStream_t)stream;
    if (d->fused_bwd) {
        // forward + backward in ONE launch: the backward's operands and outputs as phc_downstream_bwd takes them (dlogits is not read:
        // the launch derives it from its own logits and the labels), its own exchange buffers
        PHC_REQUIRE(d->loss_y && phc_downstream_fused_bwd_supported(d->B, d->d0, d->d1, d->d2, d->d3, d->T), PHC_ERR_UNSUPPORTED,
                    "phc_downstream_fwd: fused_bwd needs the loss inside the launch and phc_downstream_fused_bwd_supported");
        a.H1T = d->H1T; a.H2T = d->H2T; a.H3T = d->H3T;
        a.dx = d->dx; a.dz1 = d->dz1; a.dz2 = d->dz2; a.dz3 = d->dz3; a.dwb1 = d->dwb1; a.dwb2 = d->dwb2; a.dWr = d->dWr;
        a.dbr = d->dWr ? d->dWr + d->T * d->d3 : nullptr;
        a.partial_b = reinterpret_cast<unsigned long long *>(d->partial_bwd);
        a.sync_b = reinterpret_cast<unsigned *>(d->sync_bwd);
        PHC_REQUIRE(a.H1T && a.H2T && a.H3T && a.dz1 && a.dz2 && a.dz3 && a.dWr && a.partial_b && a.sync_b && a.partial_b != a.partial &&
                        a.sync_b != a.sync,
                    PHC_ERR_INVALID_ARG, "phc_downstream_fwd: fused_bwd: null backward pointer, or the backward's partial / sync are the forward's");
        PHC_REQUIRE(((reinterpret_cast<uintptr_t>(a.H1T) | reinterpret_cast<uintptr_t>(a.H2T) | reinterpret_cast<uintptr_t>(a.dz1) |
                      reinterpret_cast<uintptr_t>(a.dz2) | reinterpret_cast<uintptr_t>(a.dx)) & 15) == 0,
                    PHC_ERR_UNSUPPORTED, "phc_downstream_fwd: fused_bwd: 16-byte aligned rows required");
        ProfScope prof(K_DOWNSTREAM_FWD, s);
        const int RB = dn2_rb(a.B);
        const size_t lds2 = sizeof(float) * ((size_t)d2_lds(RB, a.d0, a.d1, a.d2, a.d3, a.T).total + D2_FB_STASH);
        if (RB == 1)
            PHC_LAUNCH(downstream2_fb_kernel<1>, dim3(a.G), dim3(D2_THREADS), lds2, s, a);
        else
            PHC_LAUNCH(downstream2_fb_kernel<2>, dim3(a.G), dim3(D2_THREADS), lds2, s, a);
        return check_launch("downstream2_fb");
    }
    // (hipFuncAttributeMaxDynamicSharedMemorySize was set, per device and checked, by the capacity queries of dn_common)
    ProfScope prof(K_DOWNSTREAM_FWD, s);
    if (v2) {
        const int RB = dn2_rb(a.B);
        const size_t lds2 = sizeof(float) * (size_t)d2_lds(RB, a.d0, a.d1, a.d2, a.d3, a.T).total;
        if (RB == 1)
            PHC_LAUNCH(downstream2_fwd_kernel<1>, dim3(a.G), dim3(D2_THREADS), lds2, s, a);
        else
            PHC_LAUNCH(downstream2_fwd_kernel<2>, dim3(a.G), dim3(D2_THREADS), lds2, s, a);
        return check_launch("downstream2_fwd");
    }
    const size_t lds = sizeof(float) * dn_lds_floats(a.d0, a.d1, a.d2, a.d3, a.T, false);
    PHC_LAUNCH(downstream_fwd_kernel, dim3(a.G), dim3(DN_THREADS), lds, s, a);
    return check_launch("downstream_fwd");
}

int phc_downstream_bwd(const phc_downstream_desc *d, void *stream) {
    DnArgs a = {};
    bool v2 = false;
    int rc = dn_common("phc_downstream_bwd", d, &a, &v2);
    if (rc) return rc;
    a.H1T = d->H1T; a.H2T = d->H2T; a.H3T = d->H3T; a.dlogits = d->dlogits;
    a.dx = d->dx; a.dz1 = d->dz1; a.dz2 = d->dz2; a.dz3 = d->dz3; a.dwb1 = d->dwb1; a.dwb2 = d->dwb2; a.dWr = d->dWr; a.dbr = d->dWr ? d->dWr + d->T * d->d3 : nullptr;
    PHC_REQUIRE(a.H1T && a.H2T && a.H3T && a.dlogits && a.dz1 && a.dz2 && a.dz3 && a.dWr, PHC_ERR_INVALID_ARG, "phc_downstream_bwd: null pointer");
    PHC_REQUIRE(((reinterpret_cast<uintptr_t>(a.H1T) | reinterpret_cast<uintptr_t>(a.H2T) | reinterpret_cast<uintptr_t>(a.z2) |
                  reinterpret_cast<uintptr_t>(a.dz1) | reinterpret_cast<uintptr_t>(a.dz2) | reinterpret_cast<uintptr_t>(a.dx)) & 15) == 0,
                PHC_ERR_UNSUPPORTED, "phc_downstream_bwd: 16-byte aligned rows required");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_DOWNSTREAM_BWD, s);
    if (v2) {
        const int RB = dn2_rb(a.B);
        const size_t lds2 = sizeof(float) * (size_t)d2_lds(RB, a.d0, a.d1, a.d2, a.d3, a.T).total;
        if (RB == 1)
            PHC_LAUNCH(downstream2_bwd_kernel<1>, dim3(a.G), dim3(D2_THREADS), lds2, s, a);
        else
            PHC_LAUNCH(downstream2_bwd_kernel<2>, dim3(a.G), dim3(D2_THREADS), lds2, s, a);
        return check_launch("downstream2_bwd");
    }
    const size_t lds = sizeof(float) * dn_lds_floats(a.d0, a.d1, a.d2, a.d3, a.T, true);
    PHC_LAUNCH(downstream_bwd_kernel, dim3(a.G), dim3(DN_THREADS), lds, s, a);
    return check_launch("downstream_bwd");
}

}  // extern "C"
