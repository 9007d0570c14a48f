// The downstream head at widths beyond the 16-row fused kernels of downstream.hip (phc_downstream_fwd / _bwd stop at 256-wide layers,
// d3 <= 32 and T <= 8): BASELINE configs[3]'s molpcba head 256 -> 768 -> 256 -> 512 -> 128 (downstream.py:90-120 + layers.py:399-414:
// three PHMLinear, two NaivePHMNorm + activation, the dense map to the 128 targets) as ONE launch per direction.
//
// A launch is a CHAIN of up to four stages   out_s = post_s( in_s . M_s^T + bias_s ),   in_{s+1} = out_s   over B rows (graphs):
//     post 0: nothing                               (the third PHM layer, the dense head; dz3 and dx in the backward chain)
//     post 1: training batch-norm over ALL rows (per column; norm.py:30-35) + activation       (forward; stages 0 and 1 only)
//     post 2: activation' gate + batch-norm backward against the forward's z and save_stat     (backward)
// The forward chain is (H1^T, H2^T, H3^T, Wr) with posts (1, 1, 0, 0); the backward chain (Wr^T, H3, H2, H1) with posts (0, 2, 2, 0): every
// operand is a matrix the step's operand build already holds in plain row-major form, read as M[column][k] (16 bytes per lane).
//
// B <= 1024 rows make 2.5 us of fp32 matrix work on the whole device, so the launch is latency, and the 16-row slabs of downstream.hip
// would put it on 64 CUs only (10 us of MFMA issue per 768-wide layer).  Here the grid is 2-D: workgroup (rb, cb) owns 16 rows x one
// column block (N / CB columns, CB in {8, 4, 2, 1}: 256 workgroups at B = 1024) of EVERY stage:
//   * its operand slice M[cb's columns][all k] goes straight into registers (at most 2 x 16 float4 per lane and pass: a "unit" is one
//     16-column tile x one part of the contraction; 16 units per pass over the 8 wavefronts, partial sums meet in LDS in a fixed order);
//   * the 16 x K input slab of stages >= 1 comes from the CB workgroups of the same row block: every workgroup PUBLISHES its output
//     tile as 8-byte {value, generation tag} granules (write-through stores), the readers poll 16-byte pairs with sc1 loads until every
//     tag is this launch's — the rendezvous of downstream.hip, per row block instead of grid-wide, no flags and no fences;
//   * the batch-norm sums travel the same way between the RB workgroups of a column block (Welford merges in row-block order).
// All workgroups must be resident together (one per CU at these LDS sizes: phc_head_chain_supported asks the occupancy calculator);
// every poll loop is bounded (PHC_ERR_RENDEZVOUS + NaN results, as downstream.hip).  Reductions run in fixed orders: a launch is
// bit-reproducible.  Rows >= *n_valid are padding graphs: out of the statistics, zero after a norm, zero gradient.
#include <stddef.h>
#include <stdlib.h>

#include <mutex>

#include "phc_common.h"

namespace phc {

using f32x4_t = __attribute__((ext_vector_type(4))) float;
using u32x4_t = __attribute__((ext_vector_type(4))) unsigned;

constexpr int HC_THREADS = 512;
constexpr int HC_MAX_STAGES = PHC_HEAD_MAX_STAGES;
constexpr int HC_MAX_NC = 192;                       // columns of one stage per workgroup
constexpr int HC_MAX_K = 1024;
constexpr int HC_MAX_UNITS = 32;                     // two passes of 16 units
constexpr int HC_SPIN_MAX = 1 << 21;


struct HcNorm {
    const float *weight[PHC_MAX_COMPONENTS];
    const float *bias[PHC_MAX_COMPONENTS];
    float *rmean[PHC_MAX_COMPONENTS];
    float *rvar[PHC_MAX_COMPONENTS];
    long long *nbt[PHC_MAX_COMPONENTS];
    int n_comp, k, affine;
    float momentum, eps;
};
struct HcStage {
    const float *M, *bias, *z;
    float *out, *stat, *dwb;
    float *a_out;                                    // post 1: the (dropped) activation [B][N] for the parameter-gradient job of the next layer, or null
    float drop_p;                                    // post 1 / 2: phm_dropout behind the activation (0: none); mask = drop_factors_v(...)
    int drop_same, drop_k, drop_site;
    int K, N, post;
    // filled by the host from (K, N, CB): the column block, its GEMM plan (hc_plan) and the stage's records in the work buffer (granules)
    int Nc, tiles, kg, ksplit, U;
    unsigned xch_off, stat_off;
};
struct HcArgs {
    int B, RBn, CB, n_stages, act, valid_max;
    const int *n_valid;
    const float *in;
    HcStage st[HC_MAX_STAGES];
    HcNorm bn[2];                                    // the norms of stages 0 and 1 (post 1)
    unsigned long long *work;
    unsigned *sync;                                  // [0] generation, [1] arrivals of the running launch
    const unsigned long long *rng;                   // device {seed, step} of the dropout masks, or null
    int *err;
    int lda, ldn, off_o, off_z, off_red, off_sc, off_st, off_bias;      // LDS carve-up (hc_lds), floats
};

#ifdef PHC_HC_TIMING
__device__ unsigned long long g_hc_ts[128];
#define HC_STAMP(i)                                                                              \
    do {                                                                                         \
        if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1))                \
            g_hc_ts[(blockIdx.x == 0 ? 0 : 64) + (i)] = wall_clock64();                          \
    } while (0)
#define HC_STAMP_LOADS(i)                            \
    do {                                             \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
        HC_STAMP(i);                                 \
    } while (0)
#else
#define HC_STAMP(i)
#define HC_STAMP_LOADS(i)
#endif

__device__ __forceinline__ void hc_barrier() {       // orders LDS traffic only (downstream.hip: d2_lds_barrier)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// ---- how one GEMM  [16, K] . [K, Nc]  is dealt: tiles of 16 columns x ksplit parts of the contraction = U units ----
struct HcPlan {
    int tiles, kg, ksplit, U;
};
__host__ __device__ inline HcPlan hc_plan(int K, int Nc) {
    HcPlan p;
    p.tiles = Nc >> 4;
    p.kg = K >> 4;
    int ks = (p.kg + 15) >> 4;                       // at most 16 k-groups (of 16) per unit: the register slice
    const int fill = 16 / p.tiles;                   // ... and enough parts to give every wavefront two units
    if (fill > ks) ks = fill;
    if (ks > p.kg) ks = p.kg;
    p.ksplit = ks;
    p.U = p.tiles * ks;
    return p;
}
struct HcUnit {
    bool live;
    int tile, kg0, kgn;
};
__device__ __forceinline__ HcUnit hc_unit(const HcPlan &p, int u) {
    HcUnit q;
    q.live = u < p.U;
    const int uu = q.live ? u : 0;
    const int ks = uu / p.tiles;
    q.tile = __builtin_amdgcn_readfirstlane(uu - ks * p.tiles);
    q.kg0 = __builtin_amdgcn_readfirstlane(ks * p.kg / p.ksplit);
    q.kgn = __builtin_amdgcn_readfirstlane(q.live ? (ks + 1) * p.kg / p.ksplit - q.kg0 : 0);
    return q;
}

struct HcRegs {
    f32x4_t v[2][16];                                // [unit of this wavefront][k-group of the unit]
};

// the operand slices of this wavefront's two units of pass `pass`: lane (c = lane % 16, kq = lane / 16), register gi, element q
// = M[c0 + 16 tile + c][16 (kg0 + gi) + 4 kq + q].  Whole blocks of four k-groups are skipped by wave-uniform branches (hc_gemm_pass
// skips the same ones, their registers are never read); inside a block the loads are unconditional (clamped address, selected value).
__device__ __forceinline__ void hc_load_operand(const float *__restrict__ M, int K, int c0, const HcPlan &p, int pass, int wave, int lane,
                                                HcRegs &h) {
    const int c = lane & 15, kq = lane >> 4;
    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const HcUnit un = hc_unit(p, 16 * pass + wave + 8 * t);
        const float *base = M + (size_t)(c0 + 16 * un.tile + c) * K + 16 * un.kg0 + 4 * kq;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            if (4 * g4 < un.kgn) {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int gi = 4 * g4 + gq;
                    const bool ok = gi < un.kgn;
                    const f32x4_t v = *reinterpret_cast<const f32x4_t *>(base + (ok ? 16 * gi : 0));
                    h.v[t][gi] = ok ? v : zero;
                }
            } else {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) h.v[t][4 * g4 + gq] = zero;      // (a full definition: the previous slice is dead from here on)
            }
        }
    }
}

// pass 0 of a stage's operand (column block cb of CB)
__device__ __forceinline__ HcPlan hc_plan_of(const HcStage &st) {
    HcPlan p;
    p.tiles = st.tiles; p.kg = st.kg; p.ksplit = st.ksplit; p.U = st.U;
    return p;
}
// (`live` false: behind the last stage — the registers are DEFINED all the same, or the previous slice would count as live across
// the exchange loops of the stage that has just used it)
__device__ __forceinline__ void hc_load_first(const HcStage &st, int cb, int wave, int lane, HcRegs &h, bool live = true) {
    if (live) {
        hc_load_operand(st.M, st.K, cb * st.Nc, hc_plan_of(st), 0, wave, lane, h);
    } else {
        const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int gi = 0; gi < 16; ++gi) h.v[t][gi] = zero;
    }
}

// partial products of the pass's units -> red[unit][16 rows][20]; MFMA roles as downstream.hip (A := operand slice, B := slab rows):
// lane (c, kq) ends with rows c, columns 4 kq .. 4 kq + 3 of the unit's tile
__device__ __forceinline__ void hc_gemm_pass(const float *A, int lda, const HcRegs &h, const HcPlan &p, int pass, int wave, int lane, float *red) {
    const int c = lane & 15, kq = lane >> 4;
    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int u = __builtin_amdgcn_readfirstlane(16 * pass + wave + 8 * t);
        const HcUnit un = hc_unit(p, u);
        if (un.live) {
            f32x4_t acc0 = zero, acc1 = zero;            // two independent accumulator chains
            const float *arow = A + c * lda + 16 * un.kg0 + 4 * kq;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                if (4 * g4 < un.kgn) {
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const int gi = 4 * g4 + gq;
                        const f32x4_t fr = *reinterpret_cast<const f32x4_t *>(arow + 16 * min(gi, un.kgn - 1));      // (beyond kgn the operand registers are zero)
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(h.v[t][gi][0], fr[0], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(h.v[t][gi][1], fr[1], acc1, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(h.v[t][gi][2], fr[2], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(h.v[t][gi][3], fr[3], acc1, 0, 0, 0);
                    }
                }
            }
            *reinterpret_cast<f32x4_t *>(red + (size_t)u * 320 + c * 20 + 4 * kq) = acc0 + acc1;
        }
    }
}

// out[row][col] = bias[col] + the units of the column's tile in ascending part order; ends with a barrier
__device__ __forceinline__ void hc_reduce(const float *red, const HcPlan &p, int Nc, const float *bias_s, float *out, int ldo) {
    for (int e = threadIdx.x; e < 16 * Nc; e += HC_THREADS) {
        const int row = e / Nc, col = e - row * Nc;
        const int tile = col >> 4, cc = col & 15;
        float v = bias_s[col];
        for (int ks = 0; ks < p.ksplit; ++ks) v += red[(size_t)(ks * p.tiles + tile) * 320 + row * 20 + cc];
        out[row * ldo + col] = v;
    }
    hc_barrier();
}

__device__ __forceinline__ int hc_rows_counted(int nvalid, int rb) { return max(0, min(16, nvalid - 16 * rb)); }

// rows [r0, r0 + 16) of a [B, w] matrix, columns [c0, c0 + wc) -> LDS [16][ld]; rows >= B read as 0
__device__ __forceinline__ void hc_rows_to_lds(const float *__restrict__ src, int B, int r0, int w, int c0, int wc, float *dst, int ld) {
    const int w4 = wc >> 2;
    for (int e = threadIdx.x; e < 16 * w4; e += HC_THREADS) {
        const int row = e / w4, c4 = e - row * w4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r0 + row < B) v = *reinterpret_cast<const float4 *>(src + (size_t)(r0 + row) * w + c0 + 4 * c4);
        *reinterpret_cast<float4 *>(dst + row * ld + 4 * c4) = v;
    }
}
__device__ __forceinline__ void hc_store_tile(const float *tile, int ld, int B, int r0, int w, int c0, int wc, float *__restrict__ dst) {
    const int w4 = wc >> 2;
    for (int e = threadIdx.x; e < 16 * w4; e += HC_THREADS) {
        const int row = e / w4, c4 = e - row * w4;
        if (r0 + row < B)
            *reinterpret_cast<float4 *>(dst + (size_t)(r0 + row) * w + c0 + 4 * c4) = *reinterpret_cast<const float4 *>(tile + row * ld + 4 * c4);
    }
}

// one 16-byte sc1 store of two granules {a, tag, b, tag}: it leaves the XCD's L2 for memory, where every XCD's sc1 loads find it; each
// half carries its own tag, so a torn pair is "not there yet".  (16-byte stores: a scalar sc1 store is one fabric write whatever its size,
// MI355X_MICROARCH.md "stores of each flavour"; plain stores for the slab exchange between same-XCD workgroups measured 8 us SLOWER per
// step than these.)  `base`: wave-uniform — a buffer descriptor lives in scalar registers, a per-lane one is rebuilt lane by lane, which
// made every poll and every store of the exchange a 64-trip loop; `g`: this lane's granule.
__device__ __forceinline__ void hc_put_pair(unsigned long long *base, size_t g, float va, float vb, unsigned tag) {
    const u32x4_t q = {__float_as_uint(va), tag, __float_as_uint(vb), tag};
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, 0x7ffffff0, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(q, rsrc, (int)(g * 8), 0, 16);           // aux 16 = sc1
}

// this workgroup's tile as granules of its row block's record [16][N]
__device__ __forceinline__ void hc_publish(const float *tile, int ld, int Nc, unsigned long long *rec, int N, int c0, unsigned tag) {
    const int h2 = Nc >> 1;
    for (int e = threadIdx.x; e < 16 * h2; e += HC_THREADS) {
        const int row = e / h2, col = 2 * (e - row * h2);
        const float2 v = *reinterpret_cast<const float2 *>(tile + row * ld + col);
        hc_put_pair(rec, (size_t)row * N + c0 + col, v.x, v.y, tag);
    }
}

// the row block's record [16][K] (written by its CB workgroups of the previous stage) -> slab [16][lda]: all 16-byte pair loads of a
// thread (K / 64, at most twelve per round) in flight together, repeated until every tag is this launch's (sc1 loads: past this CU's L1)
__device__ __forceinline__ void hc_gather_slab(const unsigned long long *rec, int K, unsigned tag, float *slab, int lda, int *err) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)rec, 0, 0x7ffffff0, 0x00020000);
    const int pairs = 8 * K, tid = threadIdx.x, K2 = K >> 1;          // 16 K granules
    constexpr int R = 12;                                              // pair loads per thread and round: one round up to K = 768
    for (int e0 = 0; e0 < pairs; e0 += R * HC_THREADS) {
        u32x4_t q[R];
        bool ok;
        int spins = 0;
        do {
            ok = true;
#pragma unroll
            for (int u = 0; u < R; ++u) {
                if (e0 + u * HC_THREADS < pairs) {           // wave-uniform
                    const int e = min(e0 + tid + u * HC_THREADS, pairs - 1);
                    q[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, e * 16, 0, 16);          // aux 16 = sc1
                    ok = ok && q[u][1] == tag && q[u][3] == tag;
                }
            }
            if (!ok) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > HC_SPIN_MAX) {
                    if (err) atomicOr(err, PHC_ERR_RENDEZVOUS);
#pragma unroll
                    for (int u = 0; u < R; ++u) q[u][0] = q[u][2] = 0x7fc00000u;            // a timed-out exchange must not look like a result
                    break;
                }
            }
        } while (!ok);
#pragma unroll
        for (int u = 0; u < R; ++u) {
            if (e0 + u * HC_THREADS < pairs) {
                const int e = e0 + tid + u * HC_THREADS;
                if (e < pairs) {
                    const int row = e / K2, col = 2 * (e - row * K2);
                    *reinterpret_cast<float2 *>(slab + row * lda + col) = make_float2(__uint_as_float(q[u][0]), __uint_as_float(q[u][2]));
                }
            }
        }
    }
}

// LDS carve-up shared by host and device (floats)
struct HcLds {
    int lda, ldn, off_slab, off_o, off_z, off_red, off_sc, off_st, off_bias, total;
};
static inline HcLds hc_lds(int Kmax, int Ncmax, int Umax) {
    HcLds l;
    l.lda = Kmax + 4;
    l.ldn = Ncmax + 4;
    l.off_slab = 0;
    l.off_o = l.off_slab + 16 * l.lda;
    l.off_z = l.off_o + 16 * l.ldn;
    l.off_red = l.off_z + 16 * l.ldn;
    l.off_sc = l.off_red + Umax * 320;
    l.off_st = l.off_sc + 2 * HC_MAX_NC;
    l.off_bias = l.off_st + 4 * HC_MAX_NC;
    l.total = l.off_bias + HC_MAX_NC + 16;
    return l;
}

// granule offsets of the work buffer: per stage the activation record [16 RBn][N] (stages in front of another one) and the
// statistics records [RBn][N][2] (partial pairs) + [N][2] (final pairs) (stages with a norm)
struct HcWork {
    size_t xch[HC_MAX_STAGES], stat[HC_MAX_STAGES], total;
};
static inline HcWork hc_work(int RBn, int n_stages, const int *N, const int *post) {
    HcWork w;
    size_t t = 0;
    for (int s = 0; s < HC_MAX_STAGES; ++s) {
        w.xch[s] = w.stat[s] = t;
        if (s < n_stages) {
            if (s + 1 < n_stages) t += (size_t)16 * RBn * N[s];
            w.stat[s] = t;
            if (post[s]) t += (size_t)2 * RBn * N[s] + (size_t)2 * N[s];
        }
    }
    w.total = t;
    return w;
}

// ---- the statistics exchange of a norm stage (forward: Welford partials -> scale | shift; backward: sum g | sum g x_hat -> their means) ----
// Two hops, both a handful of granules per workgroup.  The Nc columns of a column block are dealt over its RBn row-block workgroups
// that hold valid rows (cols_per = ceil(Nc / RBv) each): the OWNER of a column polls the column's partial pairs (one per lane), merges them with a
// fixed butterfly, derives the column's final pair and publishes it; every workgroup then polls the final pairs of its Nc columns.
// (One hop — every workgroup merging all RBn partials of all its columns — moved RBn x 16 x Nc bytes per workgroup through the
// fabric with sc1 loads: 12 us of the 82 us forward launch at B = 1024.)  The owner also writes what the norm leaves behind.
struct HcOwn {
    int cols_per, rounds;
};
__device__ __forceinline__ HcOwn hc_own(int Nc, int RBn) {
    HcOwn o;
    o.cols_per = (Nc + RBn - 1) / RBn;
    o.rounds = (o.cols_per + 7) >> 3;
    return o;
}
// granule pair `g` of a record (`base` wave-uniform); spins until the tag is this launch's
__device__ __forceinline__ void hc_poll_pair(const unsigned long long *base, size_t g, unsigned tag, float &va, float &vb, int *err) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, 0x7ffffff0, 0x00020000);
    int spins = 0;
    for (;;) {
        const u32x4_t q = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(g * 8), 0, 16);          // aux 16 = sc1
        if (q[1] == tag && q[3] == tag) {
            va = __uint_as_float(q[0]);
            vb = __uint_as_float(q[2]);
            return;
        }
        __builtin_amdgcn_s_sleep(1);
        if (++spins > HC_SPIN_MAX) {
            if (err) atomicOr(err, PHC_ERR_RENDEZVOUS);
            va = vb = __int_as_float(0x7fc00000);            // a timed-out exchange must not look like a result
            return;
        }
    }
}

// 8-byte word `idx` (wave-uniform) of the kernarg segment by a SCALAR load: the norms' per-component pointer tables are indexed by a
// run-time component there, and an index into the by-value struct would have it copied to scratch
__device__ __forceinline__ unsigned long long hc_kernarg_word(int idx) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const unsigned long long __attribute__((address_space(4))) *karg_t;
    return ((karg_t)__builtin_amdgcn_kernarg_segment_ptr())[__builtin_amdgcn_readfirstlane(idx)];
#else
    return 0;
#endif
}
static_assert(offsetof(HcArgs, bn) % 8 == 0 && sizeof(HcNorm) % 8 == 0 && offsetof(HcNorm, weight) == 0 &&
              offsetof(HcNorm, nbt) == 4 * 8 * PHC_MAX_COMPONENTS, "the norm tables are read as one array of 8-byte words");

template <bool BWD>
__global__ __launch_bounds__(HC_THREADS) void head_chain_kernel(HcArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // (blocks go to the 8 XCDs round-robin: with CB = 4 an XCD sees one column block's operand slice only — it stays in that L2; the
    // padding row blocks of a fixed-shape batch are the last block ids)
    const int CB = a.CB, rbi = blockIdx.x / CB, cb = blockIdx.x - rbi * CB, r0 = 16 * rbi;
    float *slab = sm, *tileO = sm + a.off_o, *tileZ = sm + a.off_z, *red = sm + a.off_red, *sc = sm + a.off_sc, *st4 = sm + a.off_st,
          *bias_s = sm + a.off_bias;
    const int lda = a.lda, ldn = a.ldn;
    const unsigned tag = __hip_atomic_load(a.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    const int nvalid = a.n_valid ? min(*a.n_valid, a.B) : a.B;
    const int nv = hc_rows_counted(nvalid, rbi);
    const unsigned long long rng_seed = a.rng ? a.rng[0] : 0ull, rng_step = a.rng ? a.rng[1] : 0ull;      // (uniform: scalar loads)
    const int RBv = max((nvalid + 15) >> 4, 1);      // row blocks with valid rows: they deal the norms' columns among themselves
    if (blockIdx.x == 0 && tid == 0 && nvalid > a.valid_max && a.err) atomicOr(a.err, PHC_ERR_RENDEZVOUS);      // (the caller's bound does not hold)

    HC_STAMP(0);
    if (r0 >= nvalid) {
        // a row block of padding graphs only: nobody waits for it (the owners skip empty row blocks, a slab is gathered by its own row
        // block), so it need not be resident with the others — its rows of every output are zero, the arrival is counted, done
#pragma unroll
        for (int s = 0; s < HC_MAX_STAGES; ++s) {
            if (s < a.n_stages) {
                const HcStage &st = a.st[s];
                const int N = st.N, Nc = st.Nc, c0 = cb * Nc, w4 = Nc >> 2;
                for (int e = tid; e < 16 * w4; e += HC_THREADS) {
                    const int row = e / w4, c4 = e - row * w4;
                    if (r0 + row < a.B) {
                        *reinterpret_cast<float4 *>(st.out + (size_t)(r0 + row) * N + c0 + 4 * c4) = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (st.a_out) *reinterpret_cast<float4 *>(st.a_out + (size_t)(r0 + row) * N + c0 + 4 * c4) = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
            }
        }
    } else {
    // the first stage's operand slice goes in flight before anything else
    HcRegs h;
    hc_load_first(a.st[0], cb, wave, lane, h);
    HC_STAMP_LOADS(51);

#pragma unroll
    for (int s = 0; s < HC_MAX_STAGES; ++s) {
        if (s < a.n_stages) {
            const HcStage &st = a.st[s];
            const int K = st.K, N = st.N, Nc = st.Nc, c0 = cb * Nc, post = st.post;
            const HcPlan p = hc_plan_of(st);
            const bool last = s + 1 == a.n_stages;
            HC_STAMP(8 * s + 1);
            // ---- what the stage reads besides its slab: bias, and the forward's z tile + statistics (post 2) ----
            if (tid < Nc) bias_s[tid] = st.bias ? st.bias[c0 + tid] : 0.f;
            if (BWD && post == 2) {
                hc_rows_to_lds(st.z, a.B, r0, N, c0, Nc, tileZ, ldn);
                if (tid < Nc) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) st4[q * HC_MAX_NC + tid] = st.stat[(size_t)(q < 2 ? q : q + 2) * N + c0 + tid];      // mean, invstd, w, b
                }
            }
            // ---- the slab: rows of the chain's input, or the previous stage's tiles of this row block ----
            if (s == 0)
                hc_rows_to_lds(a.in, a.B, r0, K, 0, K, slab, lda);
            else
                hc_gather_slab(a.work + a.st[s > 0 ? s - 1 : 0].xch_off + (size_t)r0 * K, K, tag, slab, lda, a.err);
            hc_barrier();
            HC_STAMP(8 * s + 2);
            // ---- GEMM ----
            hc_gemm_pass(slab, lda, h, p, 0, wave, lane, red);
            if (p.U > 16) {
                hc_load_operand(st.M, K, c0, p, 1, wave, lane, h);
                hc_gemm_pass(slab, lda, h, p, 1, wave, lane, red);
            }
            hc_barrier();
            hc_reduce(red, p, Nc, bias_s, tileO, ldn);
            HC_STAMP(8 * s + 3);
            unsigned long long *spart = a.work + st.stat_off;              // [RBn][N] partial pairs, then [N] final pairs
            unsigned long long *sfin = spart + (size_t)2 * a.RBn * N;
            if (post == 0) {
                // (every loop of the stage in front of the next operand's loads: hipcc waits for ALL loads in flight at a loop header)
                if (!last) hc_publish(tileO, ldn, Nc, a.work + st.xch_off + (size_t)r0 * N, N, c0, tag);
                hc_store_tile(tileO, ldn, a.B, r0, N, c0, Nc, st.out);
                hc_load_first(a.st[(s + 1) % HC_MAX_STAGES], cb, wave, lane, h, !last);
                hc_barrier();
            } else {
                const int j = s < 2 ? s : 0;
                if (!BWD) {
                    // ---- forward norm: z out, this row block's Welford partials of its columns ----
                    hc_store_tile(tileO, ldn, a.B, r0, N, c0, Nc, st.out);
                    if (tid < Nc) {
                        float sum = 0.f;
#pragma unroll
                        for (int r = 0; r < 16; ++r) sum += r < nv ? tileO[r * ldn + tid] : 0.f;
                        const float mean_l = nv > 0 ? sum / (float)nv : 0.f;
                        float m2 = 0.f;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float d = tileO[r * ldn + tid] - mean_l;
                            m2 = r < nv ? fmaf(d, d, m2) : m2;
                        }
                        hc_put_pair(spart, ((size_t)rbi * N + c0 + tid) * 2, sum, m2, tag);
                    }
                } else {
                    // ---- backward norm: g = da act'(x_hat w + b), x_hat = (z - mean) invstd; this row block's column sums ----
                    const bool relu = a.act == PHC_ACT_RELU;
                    for (int e = tid; e < 16 * Nc; e += HC_THREADS) {
                        const int row = e / Nc, c = e - row * Nc;
                        const float xh = (tileZ[row * ldn + c] - st4[c]) * st4[HC_MAX_NC + c];
                        const float pre = fmaf(xh, st4[2 * HC_MAX_NC + c], st4[3 * HC_MAX_NC + c]);
                        const float da = tileO[row * ldn + c];
                        float gv = relu ? (pre > 0.f ? da : 0.f) : da * act_grad_rt(a.act, pre);
                        if (r0 + row >= nvalid) gv = 0.f;
                        if (st.drop_p > 0.f) {                       // the forward's mask on the upstream gradient (uniform branch)
                            const float4 f = drop_factors_v(st.drop_p, st.drop_same, st.drop_k, st.drop_site, rng_seed, rng_step, r0 + row,
                                                            (c0 + c) & ~3, N);
                            const int q = c & 3;
                            gv *= q == 0 ? f.x : (q == 1 ? f.y : (q == 2 ? f.z : f.w));
                        }
                        tileO[row * ldn + c] = gv;
                        tileZ[row * ldn + c] = xh;
                    }
                    hc_barrier();
                    if (tid < Nc) {
                        float sg = 0.f, sgx = 0.f;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float gv = tileO[r * ldn + tid];
                            sg += gv;
                            sgx = fmaf(gv, tileZ[r * ldn + tid], sgx);
                        }
                        hc_put_pair(spart, ((size_t)rbi * N + c0 + tid) * 2, sg, sgx, tag);
                    }
                }
                HC_STAMP(8 * s + 4);
                // ---- hop 1: the owned columns ----
                const HcOwn own = hc_own(Nc, RBv);
                for (int r = 0; r < own.rounds; ++r) {
                    const int ci = 8 * r + wave, lc = rbi * own.cols_per + ci;      // wave-uniform
                    if (ci < own.cols_per && lc < Nc) {
                        const int col = c0 + lc;
                        if (!BWD) {
                            // the column's parameters: pointers by scalar loads (the component is wave-uniform), values in flight under the poll
                            const int comp = __builtin_amdgcn_readfirstlane(col / a.bn[j].k), kk = col - comp * a.bn[j].k;
                            const int tab = (int)((offsetof(HcArgs, bn) + j * sizeof(HcNorm)) / 8) + comp;
                            float *rm = reinterpret_cast<float *>(hc_kernarg_word(tab + 2 * PHC_MAX_COMPONENTS));
                            float *rv = reinterpret_cast<float *>(hc_kernarg_word(tab + 3 * PHC_MAX_COMPONENTS));
                            long long *nbt = reinterpret_cast<long long *>(hc_kernarg_word(tab + 4 * PHC_MAX_COMPONENTS));
                            float w = 1.f, b = 0.f;
                            const float rm_old = rm[kk], rv_old = rv[kk];
                            if (a.bn[j].affine) {
                                w = reinterpret_cast<const float *>(hc_kernarg_word(tab))[kk];
                                b = reinterpret_cast<const float *>(hc_kernarg_word(tab + PHC_MAX_COMPONENTS))[kk];
                            }
                            float n = 0.f, mean = 0.f, M2 = 0.f;
                            for (int g = lane; g < RBv; g += 64) {
                                const float ng = (float)hc_rows_counted(nvalid, g);
                                if (ng > 0.f) {                              // (an empty row block publishes nothing)
                                    float ps, pm;
                                    hc_poll_pair(spart, ((size_t)g * N + col) * 2, tag, ps, pm, a.err);
                                    const float delta = ps / ng - mean, tot = n + ng;
                                    mean += delta * (ng / tot);
                                    M2 += pm + delta * delta * (n * ng / tot);
                                    n = tot;
                                }
                            }
#pragma unroll
                            for (int off = 1; off < 64; off <<= 1) {             // fixed butterfly; both lanes of a pair merge (lower, upper)
                                const float n2 = __shfl_xor(n, off, 64), mean2 = __shfl_xor(mean, off, 64), M22 = __shfl_xor(M2, off, 64);
                                const bool up = (lane & off) != 0;
                                const float na = up ? n2 : n, ma = up ? mean2 : mean, Ma = up ? M22 : M2;
                                const float nb2 = up ? n : n2, mb = up ? mean : mean2, Mb = up ? M2 : M22;
                                const float tot = na + nb2;
                                if (nb2 > 0.f && na > 0.f) {
                                    const float delta = mb - ma;
                                    mean = ma + delta * (nb2 / tot);
                                    M2 = Ma + Mb + delta * delta * (na * nb2 / tot);
                                } else {
                                    mean = na > 0.f ? ma : mb;
                                    M2 = na > 0.f ? Ma : Mb;
                                }
                                n = tot;
                            }
                            if (lane == 0) {
                                const float cnt = fmaxf(n, 1.f);
                                const float var = fmaxf(M2 / cnt, 0.f);
                                const float invstd = 1.f / sqrtf(var + a.bn[j].eps);
                                const float scale = invstd * w, shift = b - mean * invstd * w;
                                hc_put_pair(sfin, (size_t)col * 2, scale, shift, tag);
                                if (mean == mean && M2 == M2) {                   // (NaN: the exchange timed out — nothing persistent is touched)
                                    const HcNorm &bn = a.bn[j];
                                    st.stat[col] = mean;
                                    st.stat[N + col] = invstd;
                                    st.stat[2 * (size_t)N + col] = scale;
                                    st.stat[3 * (size_t)N + col] = shift;
                                    st.stat[4 * (size_t)N + col] = w;
                                    st.stat[5 * (size_t)N + col] = b;
                                    const float unbiased = n > 1.f ? M2 / (n - 1.f) : var;
                                    rm[kk] = (1.f - bn.momentum) * rm_old + bn.momentum * mean;
                                    rv[kk] = (1.f - bn.momentum) * rv_old + bn.momentum * unbiased;
                                    if (kk == 0 && nbt) *nbt += 1;
                                }
                            }
                        } else {
                            float SG = 0.f, SGX = 0.f;
                            for (int g = lane; g < RBv; g += 64) {
                                if (hc_rows_counted(nvalid, g) > 0) {
                                    float pa, pb;
                                    hc_poll_pair(spart, ((size_t)g * N + col) * 2, tag, pa, pb, a.err);
                                    SG += pa;
                                    SGX += pb;
                                }
                            }
#pragma unroll
                            for (int off = 1; off < 64; off <<= 1) {
                                SG += __shfl_xor(SG, off, 64);
                                SGX += __shfl_xor(SGX, off, 64);
                            }
                            if (lane == 0) {
                                const float cnt = (float)max(nvalid, 1);
                                hc_put_pair(sfin, (size_t)col * 2, SG / cnt, SGX / cnt, tag);
                                if (st.dwb) {
                                    st.dwb[col] = SGX;                            // dweight
                                    st.dwb[N + col] = SG;                         // dbias
                                }
                            }
                        }
                    }
                }
                HC_STAMP(8 * s + 5);
                // the next stage's operand streams in under the second hop (loads return in order: issued in front of the first hop, its
                // polls came back 6 us later; a hop is ~3 us of store-to-load latency through memory whatever it carries)
                hc_load_first(a.st[(s + 1) % HC_MAX_STAGES], cb, wave, lane, h, !last);
                // ---- hop 2: the final pairs of this workgroup's columns ----
                if (tid < Nc) {
                    float fa, fb;
                    hc_poll_pair(sfin, (size_t)(c0 + tid) * 2, tag, fa, fb, a.err);
                    sc[tid] = fa;
                    sc[HC_MAX_NC + tid] = fb;
                }
                hc_barrier();
                if (!BWD) {
                    const int n4 = Nc >> 2;
                    const bool relu = a.act == PHC_ACT_RELU;
                    for (int e = tid; e < 16 * n4; e += HC_THREADS) {
                        const int row = e / n4, c = (e - row * n4) << 2;
                        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (r0 + row < nvalid) {
                            const float4 z = *reinterpret_cast<const float4 *>(tileO + row * ldn + c);
                            const float4 s4 = *reinterpret_cast<const float4 *>(sc + c), h4 = *reinterpret_cast<const float4 *>(sc + HC_MAX_NC + c);
                            if (relu) {
                                o.x = fmaxf(fmaf(z.x, s4.x, h4.x), 0.f); o.y = fmaxf(fmaf(z.y, s4.y, h4.y), 0.f);
                                o.z = fmaxf(fmaf(z.z, s4.z, h4.z), 0.f); o.w = fmaxf(fmaf(z.w, s4.w, h4.w), 0.f);
                            } else {
                                o.x = act_fwd_rt(a.act, fmaf(z.x, s4.x, h4.x)); o.y = act_fwd_rt(a.act, fmaf(z.y, s4.y, h4.y));
                                o.z = act_fwd_rt(a.act, fmaf(z.z, s4.z, h4.z)); o.w = act_fwd_rt(a.act, fmaf(z.w, s4.w, h4.w));
                            }
                            if (st.drop_p > 0.f) {                   // (uniform)
                                const float4 f = drop_factors_v(st.drop_p, st.drop_same, st.drop_k, st.drop_site, rng_seed, rng_step, r0 + row,
                                                                c0 + c, N);
                                o.x *= f.x; o.y *= f.y; o.z *= f.z; o.w *= f.w;
                            }
                        }
                        *reinterpret_cast<float4 *>(tileO + row * ldn + c) = o;
                    }
                } else {
                    for (int e = tid; e < 16 * Nc; e += HC_THREADS) {
                        const int row = e / Nc, c = e - row * Nc;
                        float v = 0.f;
                        if (r0 + row < nvalid)
                            v = st4[2 * HC_MAX_NC + c] * st4[HC_MAX_NC + c] * (tileO[row * ldn + c] - sc[c] - tileZ[row * ldn + c] * sc[HC_MAX_NC + c]);
                        tileO[row * ldn + c] = v;
                    }
                }
                hc_barrier();
                HC_STAMP(8 * s + 6);
                if (!last) hc_publish(tileO, ldn, Nc, a.work + st.xch_off + (size_t)r0 * N, N, c0, tag);
                if (BWD) hc_store_tile(tileO, ldn, a.B, r0, N, c0, Nc, st.out);
                if (!BWD && st.a_out) hc_store_tile(tileO, ldn, a.B, r0, N, c0, Nc, st.a_out);
                hc_barrier();
            }
        }
    }
    }
    HC_STAMP(40);
    // the generation moves on when the LAST workgroup is through (every one read it at its start)
    __syncthreads();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(a.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1u == gridDim.x) {
            __hip_atomic_store(a.sync + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(a.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    HC_STAMP(41);
}

// ---- host ----
struct HcShape {
    bool ok;
    int CB, RBn, Kmax, Ncmax, Umax;
    size_t lds_bytes;
};

static int hc_capacity_of(bool bwd, size_t lds_bytes) {
    static const bool masked = getenv("HSA_CU_MASK") || getenv("ROC_GLOBAL_CU_MASK") || getenv("HSA_CU_MASK_SKIP_INIT");
    if (masked) return 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    struct Slot { size_t lds; int cap; };
    static Slot cache[64][2][4] = {};                     // (a handful of LDS sizes per device and direction: one per chain shape)
    static int next[64][2] = {};
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    Slot *row = cache[dev][bwd ? 1 : 0];
    for (int i = 0; i < 4; ++i)
        if (row[i].cap > 0 && row[i].lds == lds_bytes) return row[i].cap;
    const bool lds_ok = bwd ? ensure_dynamic_lds(&head_chain_kernel<true>, 150 * 1024) : ensure_dynamic_lds(&head_chain_kernel<false>, 150 * 1024);
    if (!lds_ok) return 0;
    int per_cu = 0;
    hipDeviceProp_t prop;
    const hipError_t e = bwd ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, head_chain_kernel<true>, HC_THREADS, lds_bytes)
                             : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, head_chain_kernel<false>, HC_THREADS, lds_bytes);
    if (e != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess || per_cu < 1 || prop.multiProcessorCount < 1) {
        (void)hipGetLastError();
        return 0;
    }
    row[next[dev][bwd ? 1 : 0]++ & 3] = {lds_bytes, per_cu * prop.multiProcessorCount};
    return per_cu * prop.multiProcessorCount;
}
// (a forward chain and its backward chain share one column split: the smaller capacity of the two kernels decides)
static int hc_capacity(size_t lds_bytes) {
    const int f = hc_capacity_of(false, lds_bytes), b = hc_capacity_of(true, lds_bytes);
    return f < b ? f : b;
}

// the widest column split whose workgroups fit the device together; `device`: ask the occupancy calculator (needs a GPU)
static HcShape hc_shape(long long B, long long valid_max, int n_stages, const int *K, const int *N, bool device) {
    HcShape r = {};
    static const bool off = getenv("PHC_HEAD_CHAIN") && atoi(getenv("PHC_HEAD_CHAIN")) == 0;
    constexpr int forced_cb = 0;
    if (off || B < 1 || B > (1 << 20) || n_stages < 1 || n_stages > HC_MAX_STAGES) return r;
    for (int s = 0; s < n_stages; ++s) {
        if (K[s] < 16 || K[s] > HC_MAX_K || K[s] % 16 || N[s] < 16 || N[s] % 16) return r;
        if (s && K[s] != N[s - 1]) return r;
    }
    r.RBn = (int)((B + 15) / 16);
    for (int CB = 8; CB >= 1; CB >>= 1) {
        if (forced_cb && CB != forced_cb) continue;
        bool fits = true;
        int Kmax = 16, Ncmax = 16, Umax = 16;
        for (int s = 0; s < n_stages && fits; ++s) {
            if (N[s] % (16 * CB) || N[s] / CB > HC_MAX_NC) { fits = false; break; }
            const HcPlan p = hc_plan(K[s], N[s] / CB);
            if (p.U > HC_MAX_UNITS) fits = false;
            Kmax = K[s] > Kmax ? K[s] : Kmax;
            Ncmax = N[s] / CB > Ncmax ? N[s] / CB : Ncmax;
            Umax = p.U > Umax ? p.U : Umax;
        }
        if (!fits) continue;
        const size_t lds = sizeof(float) * (size_t)hc_lds(Kmax, Ncmax, Umax).total;
        if (lds > 150 * 1024) continue;
        if ((long long)r.RBn * CB > (1 << 16)) continue;
        // (only the row blocks with valid rows wait for one another: the others exit at once, whenever they are scheduled)
        const long long need = (valid_max > 0 && valid_max < B ? (valid_max + 15) / 16 : r.RBn) * CB;
        if (device && hc_capacity(lds) < need) continue;
        r.ok = true; r.CB = CB; r.Kmax = Kmax; r.Ncmax = Ncmax; r.Umax = Umax; r.lds_bytes = lds;
        return r;
    }
    return r;
}

static int fill_hc_norm(HcNorm *o, const phc_downstream_norm &n, int width) {
    PHC_REQUIRE(n.n_comp >= 1 && n.n_comp <= PHC_MAX_COMPONENTS && width % n.n_comp == 0, PHC_ERR_INVALID_ARG,
                "phc_head_chain: norm with %d components on width %d", n.n_comp, width);
    PHC_REQUIRE(n.running_mean && n.running_var, PHC_ERR_INVALID_ARG, "phc_head_chain: running statistics are required");
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
        PHC_REQUIRE(o->rmean[c] && o->rvar[c] && (!o->affine || (o->weight[c] && o->bias[c])), PHC_ERR_INVALID_ARG, "phc_head_chain: null norm pointer");
    }
    return PHC_OK;
}

}  // namespace phc

using namespace phc;

extern "C" {

#ifdef PHC_HC_TIMING
int phc_debug_hc_timestamps(unsigned long long *out128) {
    return hipMemcpyFromSymbol(out128, HIP_SYMBOL(g_hc_ts), sizeof(unsigned long long) * 128) == hipSuccess ? 0 : -1;
}
#endif

int phc_head_chain_supported(int64_t B, int64_t valid_rows_max, int n_stages, const int32_t *K, const int32_t *N) {
    if (!K || !N) return 0;
    return hc_shape(B, valid_rows_max, n_stages, K, N, true).ok ? 1 : 0;
}

size_t phc_head_chain_work_floats(int64_t B, int n_stages, const int32_t *N) {
    if (B < 1 || !N || n_stages < 1 || n_stages > HC_MAX_STAGES) return 0;
    int post[HC_MAX_STAGES] = {1, 1, 1, 1};              // (sized as if every stage had a norm)
    return 2 * hc_work((int)((B + 15) / 16), n_stages, N, post).total + 8;
}

int phc_head_chain(const phc_head_chain_desc *d, void *stream) {
    PHC_REQUIRE(d, PHC_ERR_INVALID_ARG, "phc_head_chain: null descriptor");
    PHC_REQUIRE(d->n_stages >= 1 && d->n_stages <= HC_MAX_STAGES, PHC_ERR_INVALID_ARG, "phc_head_chain: %d stages", d->n_stages);
    PHC_REQUIRE(d->act >= PHC_ACT_IDENTITY && d->act <= PHC_ACT_SELU, PHC_ERR_INVALID_ARG, "phc_head_chain: unknown activation %d", d->act);
    int K[HC_MAX_STAGES], N[HC_MAX_STAGES];
    for (int s = 0; s < d->n_stages; ++s) {
        K[s] = d->stage[s].K;
        N[s] = d->stage[s].N;
    }
    const HcShape sh = hc_shape(d->B, d->valid_rows_max, d->n_stages, K, N, true);
    PHC_REQUIRE(sh.ok, PHC_ERR_UNSUPPORTED, "phc_head_chain: B=%lld with %d stages (first %d -> %d) is outside the chain kernel's range or its "
                "workgroups cannot all be resident on this device: use the per-layer path", (long long)d->B, d->n_stages, K[0], N[0]);
    PHC_REQUIRE(d->in && d->work && d->sync, PHC_ERR_INVALID_ARG, "phc_head_chain: null input / work / sync pointer");
    HcArgs a = {};
    a.valid_max = (int)(d->valid_rows_max > 0 && d->valid_rows_max < d->B ? d->valid_rows_max : d->B);
    a.B = (int)d->B; a.RBn = sh.RBn; a.CB = sh.CB; a.n_stages = d->n_stages; a.act = d->act;
    a.n_valid = d->n_valid_dev;
    a.in = d->in;
    a.work = reinterpret_cast<unsigned long long *>(d->work);
    a.sync = d->sync;
    a.err = err_word();
    uintptr_t align = reinterpret_cast<uintptr_t>(d->in) | reinterpret_cast<uintptr_t>(d->work);
    bool bwd = false, fwd = false, dropping = false;
    int posts[HC_MAX_STAGES] = {0, 0, 0, 0};
    for (int s = 0; s < d->n_stages; ++s) {
        const phc_head_stage &q = d->stage[s];
        PHC_REQUIRE(q.M && q.out, PHC_ERR_INVALID_ARG, "phc_head_chain: stage %d: null operand / output", s);
        PHC_REQUIRE(q.post >= 0 && q.post <= 2, PHC_ERR_INVALID_ARG, "phc_head_chain: stage %d: post %d", s, q.post);
        PHC_REQUIRE(q.post != 1 || s < 2, PHC_ERR_UNSUPPORTED, "phc_head_chain: a forward norm sits on stage 0 or 1 only (stage %d)", s);
        PHC_REQUIRE(q.post != 1 || q.stat, PHC_ERR_INVALID_ARG, "phc_head_chain: stage %d: save_stat missing", s);
        PHC_REQUIRE(q.post != 2 || (q.z && q.stat), PHC_ERR_INVALID_ARG, "phc_head_chain: stage %d: the forward's z / save_stat missing", s);
        a.st[s].M = q.M; a.st[s].bias = q.bias; a.st[s].z = q.z; a.st[s].out = q.out; a.st[s].stat = q.stat; a.st[s].dwb = q.dwb;
        a.st[s].K = q.K; a.st[s].N = q.N; a.st[s].post = q.post;
        a.st[s].a_out = q.post == 1 ? q.a_out : nullptr;
        a.st[s].drop_p = 0.f; a.st[s].drop_same = 0; a.st[s].drop_k = q.N; a.st[s].drop_site = q.drop_site;
        if (q.post != 0 && q.drop_p > 0.f) {
            PHC_REQUIRE(q.drop_p < 1.f && d->rng_state, PHC_ERR_INVALID_ARG, "phc_head_chain: stage %d: dropout p = %g needs p < 1 and an rng state", s,
                        (double)q.drop_p);
            PHC_REQUIRE(q.drop_same_k == 0 || (q.drop_same_k % 4 == 0 && q.N % q.drop_same_k == 0), PHC_ERR_UNSUPPORTED,
                        "phc_head_chain: stage %d: a shared dropout mask needs a component width that is a multiple of 4", s);
            a.st[s].drop_p = q.drop_p;
            a.st[s].drop_same = q.drop_same_k > 0;
            a.st[s].drop_k = q.drop_same_k > 0 ? q.drop_same_k : q.N;
            dropping = true;
        }
        align |= reinterpret_cast<uintptr_t>(a.st[s].a_out);
        const HcPlan pl = hc_plan(q.K, q.N / sh.CB);
        a.st[s].Nc = q.N / sh.CB; a.st[s].tiles = pl.tiles; a.st[s].kg = pl.kg; a.st[s].ksplit = pl.ksplit; a.st[s].U = pl.U;
        posts[s] = q.post;
        align |= reinterpret_cast<uintptr_t>(q.M) | reinterpret_cast<uintptr_t>(q.out) | reinterpret_cast<uintptr_t>(q.z);
        fwd = fwd || q.post == 1;
        bwd = bwd || q.post == 2;
        if (q.post == 1) {
            const int rc = fill_hc_norm(&a.bn[s], q.norm, q.N);
            if (rc != PHC_OK) return rc;
        }
    }
    {
        const HcWork w = hc_work(sh.RBn, d->n_stages, N, posts);
        for (int s = 0; s < d->n_stages; ++s) {
            a.st[s].xch_off = (unsigned)w.xch[s];
            a.st[s].stat_off = (unsigned)w.stat[s];
        }
        PHC_REQUIRE(w.total < (1ull << 31), PHC_ERR_UNSUPPORTED, "phc_head_chain: work buffer beyond 2^31 granules");
        const HcLds l = hc_lds(sh.Kmax, sh.Ncmax, sh.Umax);
        a.lda = l.lda; a.ldn = l.ldn; a.off_o = l.off_o; a.off_z = l.off_z; a.off_red = l.off_red; a.off_sc = l.off_sc; a.off_st = l.off_st;
        a.off_bias = l.off_bias;
    }
    a.rng = dropping ? reinterpret_cast<const unsigned long long *>(d->rng_state) : nullptr;
    PHC_REQUIRE(!(fwd && bwd), PHC_ERR_INVALID_ARG, "phc_head_chain: a chain is forward (post 0 / 1) or backward (post 0 / 2)");
    PHC_REQUIRE((align & 15) == 0, PHC_ERR_UNSUPPORTED, "phc_head_chain: 16-byte aligned rows required");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (bwd) {
        PHC_REQUIRE(ensure_dynamic_lds(&head_chain_kernel<true>, 150 * 1024), PHC_ERR_LAUNCH, "phc_head_chain: the device refused the dynamic LDS size");
        PHC_LAUNCH(head_chain_kernel<true>, dim3(sh.RBn * sh.CB), dim3(HC_THREADS), sh.lds_bytes, s, a);
    } else {
        PHC_REQUIRE(ensure_dynamic_lds(&head_chain_kernel<false>, 150 * 1024), PHC_ERR_LAUNCH, "phc_head_chain: the device refused the dynamic LDS size");
        PHC_LAUNCH(head_chain_kernel<false>, dim3(sh.RBn * sh.CB), dim3(HC_THREADS), sh.lds_bytes, s, a);
    }
    return check_launch("head_chain");
}

}  // extern "C"
