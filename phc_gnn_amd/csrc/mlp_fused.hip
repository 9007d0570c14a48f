// The backward of  norm_o(PHMMLP(s))  at hidden 128 on molecule batches as ONE launch (round 6; VERDICT r5 "the backward mirror").
//
// Reference: autograd of  PHMMLP.forward (phc/hypercomplex/layers.py:349-355: linear2(act(norm(linear1(s)))))  followed by the
// model-level norm + activation (undirectional/models.py:208-215; norm.py:30-35).  Three launches so far:
//     bn_bwd_partial  (column sums of g_o = dout act'(.) and g_o x_hat over ALL rows)             5.6 us
//     gemm_hs16<.., BA | BW>  (outer norm backward applied to dout, da = dz_o . H2^T, inner norm's column sums)   17.2 us
//     gemm_hs16<.., BA>       (inner norm backward applied to da, ds = dz_i . H1^T)                              13.2 us
// for 2 x 2.7 us of matrix work: each launch re-reads its rows from memory behind a fresh prologue, and each boundary is a grid-wide
// dependency (a norm backward needs its column sums over all rows before any row can be transformed).  Here a workgroup OWNS its
// rows (at most MAXS 16-row steps) for the whole chain and keeps them in LDS; the two grid-wide dependencies are two barriers
// INSIDE the launch, and the only data that crosses them are the integer column-sum accumulators (phc_common.h: 64-bit integer
// atomics, read back with agent-scope loads) — no cache write-back or invalidate is needed, the rows never leave the CU, and `da`
// never reaches memory at all.
//     P0   rows of dout, y2, z1 -> LDS;  g_o = dout act_o'(x_hat_o w + b), x_hat_o;  column sums -> acc_o        | barrier 1
//     P1   dz_o = w invstd (g_o - mean g_o - x_hat_o mean(g_o x_hat_o)) -> memory (the dH job of linear2 reads it);
//          da = dz_o . H2^T (v_mfma_f32_16x16x4_f32, operand in registers as in gemm_hs_device.h);
//          g_i = da act_i'(x_hat_i w + b), x_hat_i;  column sums -> acc_i                                       | barrier 2
//     P2   dz_i -> memory (linear1's dH job);  ds = dz_i . H1^T -> memory
// Rows >= *n_valid are padding rows of a fixed-shape batch: zero gradient everywhere, as in the separate launches.
// All workgroups must be co-resident (they wait for each other): the host launches this form only when ceil(steps / MAXS)
// workgroups fit the device's real capacity for the kernel, and the barrier's poll is bounded (PHC_ERR_RENDEZVOUS in the device
// error word; the launch then runs to its end with garbage instead of hanging the GPU).
// Arithmetic: x_hat, the pre-activation and dz exactly as the BA transform of gemm_hs_device.h rounds them ((x - mean) invstd, then
// fma: an element on the ReLU kink falls on the same side in every implementation); the MFMA order is gemm_hs16_kernel's; the
// column sums are taken in a fixed order (thread's rows, the wavefront's two row groups, the four wavefronts) and meet in integer
// accumulators: bit-reproducible, and within fp32 round-off of the separate launches' sums.
#include <stdlib.h>

#include "phc_common.h"

namespace phc {

using f32x4_t = __attribute__((ext_vector_type(4))) float;

constexpr int MF_H = 128;                        // layer width (in = out)
constexpr int MF_THREADS = 256, MF_WAVES = 4;
constexpr int MF_LD = MF_H + 4;                  // LDS row stride (floats): the 16 lanes of a fragment read hit 16 different 16-byte bank groups
constexpr int MF_TILE = 16 * MF_LD;              // one 16-row tile
constexpr int MF_SPIN_MAX = 1 << 21;

struct MlpBwdArgs {
    const float *dout, *y2, *z1;                 // [N, 128]
    const float *save_o, *save_i;                // [6 x 128] mean | invstd | scale | shift | w | b
    const float *H2Tk4, *H1Tk4;                  // k4 forms of H2^T, H1^T
    float *dy2, *dz1, *ds;                       // [N, 128] out
    float *dwb_o, *dwb_i;                        // [2 x 128] out: dweight | dbias
    unsigned long long *acc_o, *acc_i;           // zeroed accumulators of width 128
    const int *n_valid;
    long long N;
    int act_o, act_i;
    unsigned *bar;                               // barrier words (mf_grid_barrier): zero before the first use, owned by the kernel
    int *err;
    unsigned long long *stamps;                  // diagnostics only (NULL in the product): 8 wall-clock stamps of workgroups 0 and G - 1
};

template <int MAXS>
constexpr size_t mf_lds_floats() {
    return (size_t)MAXS * 3 * MF_TILE + 8 * MF_H + 2 * MF_H + MF_WAVES * 2 * MF_H;
}

// grid-wide barrier, two levels: 416 arrivals at ONE address serialise at the memory side (~18 ns each: 7 us per barrier, measured
// through the launch: 48 us against the three launches' 36), so a workgroup arrives at the counter of its group g % MF_SUB, the last
// arrival of a group arrives at the top counter, and the last of those moves every group's generation word (each on its own
// 128-byte line: a group's workgroups poll their own line).  Words (uint32, stride MF_BSTRIDE): [0, MF_SUB) group counters,
// [MF_SUB] top counter, [MF_SUB + 1 + k] generation of group k.
constexpr int MF_SUB = 16, MF_BSTRIDE = 32, MF_BAR_WORDS = (2 * MF_SUB + 1) * MF_BSTRIDE;
__device__ __forceinline__ void mf_grid_barrier(unsigned *bar, unsigned &gen, int *err) {
    __syncthreads();                                         // (s_waitcnt vmcnt(0): this workgroup's atomics have been performed)
    if (threadIdx.x == 0) {
        const unsigned G = gridDim.x, k = blockIdx.x % MF_SUB;
        const unsigned members = (G - k + MF_SUB - 1) / MF_SUB, groups = G < MF_SUB ? G : MF_SUB;
        unsigned *mine = bar + (MF_SUB + 1 + k) * MF_BSTRIDE;
        bool released = false;
        if (__hip_atomic_fetch_add(bar + k * MF_BSTRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1) {
            __hip_atomic_store(bar + k * MF_BSTRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__hip_atomic_fetch_add(bar + MF_SUB * MF_BSTRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == groups - 1) {
                __hip_atomic_store(bar + MF_SUB * MF_BSTRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (unsigned q = 0; q < groups; ++q)
                    __hip_atomic_store(bar + (MF_SUB + 1 + q) * MF_BSTRIDE, gen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                released = true;
            }
        }
        if (!released) {
            int spins = 0;
            while (__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > MF_SPIN_MAX) {
                    if (err) atomicOr(err, PHC_ERR_RENDEZVOUS);
                    break;
                }
            }
        }
    }
    gen += 1u;
    __syncthreads();
}

// a barrier that orders LDS traffic only (__syncthreads() also waits for every global load / store in flight)
__device__ __forceinline__ void mf_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// operand slice of this wavefront (its 32 output columns, the whole contraction) -> registers: hreg[t][g] = H[16 g + 4 kq .. +3][col0 + 16 t + c]
__device__ __forceinline__ void mf_load_operand(const float *__restrict__ Hk4, int col0, int c, int kq, f32x4_t (&hreg)[2][8]) {
    const f32x4_t *src = reinterpret_cast<const f32x4_t *>(Hk4);
#pragma unroll
    for (int g = 0; g < 8; ++g)
#pragma unroll
        for (int t = 0; t < 2; ++t) hreg[t][g] = src[(size_t)(4 * g + kq) * MF_H + col0 + c + 16 * t];
}

// acc[t] = rows of `tile` (LDS, [16][MF_LD]) . operand: lane (c, kq) ends with y[row c][col0 + 16 t + 4 kq .. +3]
__device__ __forceinline__ void mf_gemm_tile(const float *tile, const f32x4_t (&hreg)[2][8], int c, int kq, f32x4_t (&acc)[2]) {
    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
    acc[0] = acc[1] = zero;
    f32x4_t fr[2];
    fr[0] = *reinterpret_cast<const f32x4_t *>(tile + c * MF_LD + 4 * kq);
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        if (g + 1 < 8) fr[(g + 1) & 1] = *reinterpret_cast<const f32x4_t *>(tile + c * MF_LD + 16 * (g + 1) + 4 * kq);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(hreg[t][g][q], fr[g & 1][q], acc[t], 0, 0, 0);
    }
}

// g = up act'(x_hat w + b), x_hat = (x - mean) invstd for this thread's two float4 of every tile; column sums of g and g x_hat over
// the workgroup's valid rows -> the accumulator.  UP / X: LDS tiles (in place: UP := g, X := x_hat); cst: mean | invstd | w | b [4][128]
template <int MAXS>
__device__ __forceinline__ void mf_gate_and_sums(float *UP, float *X, int tile_stride, const float *cst, int act, int steps, long long step0,
                                                 long long gstride, long long n_valid, float *red, unsigned long long *acc, int rep) {
    const int tid = threadIdx.x, c4 = tid & 31, rg = tid >> 5, lane = tid & 63, wave = tid >> 6;
    const f32x4_t mean = *reinterpret_cast<const f32x4_t *>(cst + 4 * c4), inv = *reinterpret_cast<const f32x4_t *>(cst + MF_H + 4 * c4);
    const f32x4_t w = *reinterpret_cast<const f32x4_t *>(cst + 2 * MF_H + 4 * c4), b = *reinterpret_cast<const f32x4_t *>(cst + 3 * MF_H + 4 * c4);
    f32x4_t sg = {0.f, 0.f, 0.f, 0.f}, sgx = {0.f, 0.f, 0.f, 0.f};
    const bool relu = act == PHC_ACT_RELU;
#pragma unroll
    for (int s = 0; s < MAXS; ++s) {
        if (s < steps) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int r = rg + 8 * u;
                const long long row = (step0 + s * gstride) * 16 + r;
                float *pu = UP + s * tile_stride + r * MF_LD + 4 * c4, *px = X + s * tile_stride + r * MF_LD + 4 * c4;
                const f32x4_t up = *reinterpret_cast<const f32x4_t *>(pu), xv = *reinterpret_cast<const f32x4_t *>(px);
                f32x4_t gq, xh;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    xh[q] = (xv[q] - mean[q]) * inv[q];
                    const float pre = fmaf(xh[q], w[q], b[q]);
                    float gv = up[q];
                    if (relu)
                        gv = pre > 0.f ? gv : 0.f;
                    else
                        gv *= act_grad_rt(act, pre);
                    gq[q] = row < n_valid ? gv : 0.f;
                    xh[q] = row < n_valid ? xh[q] : 0.f;     // (a padding row may hold anything: 0 x NaN must not reach the sums)
                    sg[q] += gq[q];
                    sgx[q] = fmaf(gq[q], xh[q], sgx[q]);
                }
                *reinterpret_cast<f32x4_t *>(pu) = gq;
                *reinterpret_cast<f32x4_t *>(px) = xh;
            }
        }
    }
    // the wavefront's two row groups (lanes l, l + 32), then the four wavefronts in order
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        sg[q] += __shfl_xor(sg[q], 32, 64);
        sgx[q] += __shfl_xor(sgx[q], 32, 64);
    }
    if (lane < 32) {
        *reinterpret_cast<f32x4_t *>(red + (wave * 2 + 0) * MF_H + 4 * c4) = sg;
        *reinterpret_cast<f32x4_t *>(red + (wave * 2 + 1) * MF_H + 4 * c4) = sgx;
    }
    mf_lds_barrier();
    if (tid < MF_H) {
        float a = 0.f, bq = 0.f;
#pragma unroll
        for (int wv = 0; wv < MF_WAVES; ++wv) {
            a += red[(wv * 2 + 0) * MF_H + tid];
            bq += red[(wv * 2 + 1) * MF_H + tid];
        }
        stat_acc_add(acc, MF_H, rep, 0, tid, (double)a);
        stat_acc_add(acc, MF_H, rep, 1, tid, (double)bq);
    }
}

// the accumulator's totals of column `col` (agent-scope loads: the words were written by other CUs' atomics inside this launch)
__device__ __forceinline__ void mf_totals(const unsigned long long *acc, int col, double &t0, double &t1) {
    long long hi[2] = {0, 0}, lo[2] = {0, 0};
#pragma unroll
    for (int rep = 0; rep < STAT_ACC_REP; ++rep)
#pragma unroll
        for (int qty = 0; qty < 2; ++qty) {
            const unsigned long long *p = acc + ((size_t)(rep * 2 + qty) * 2) * MF_H + col;
            hi[qty] += (long long)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lo[qty] += (long long)__hip_atomic_load(p + MF_H, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    t0 = (double)hi[0] * (1.0 / 4096.0) + (double)lo[0] * (1.0 / 4611686018427387904.0);
    t1 = (double)hi[1] * (1.0 / 4096.0) + (double)lo[1] * (1.0 / 4611686018427387904.0);
}

// dz = w invstd (g - mean g - x_hat mean(g x_hat)) in place over G (rows >= n_valid: 0), written to `out` too
template <int MAXS>
__device__ __forceinline__ void mf_norm_bwd(float *G, const float *XH, int tile_stride, const float *cst, const float *mm, int steps, long long step0,
                                            long long gstride, long long n_valid, long long N, float *__restrict__ out) {
    const int tid = threadIdx.x, c4 = tid & 31, rg = tid >> 5;
    const f32x4_t inv = *reinterpret_cast<const f32x4_t *>(cst + MF_H + 4 * c4), w = *reinterpret_cast<const f32x4_t *>(cst + 2 * MF_H + 4 * c4);
    const f32x4_t cng = *reinterpret_cast<const f32x4_t *>(mm + 4 * c4), cngx = *reinterpret_cast<const f32x4_t *>(mm + MF_H + 4 * c4);
#pragma unroll
    for (int s = 0; s < MAXS; ++s) {
        if (s < steps) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int r = rg + 8 * u;
                const long long row = (step0 + s * gstride) * 16 + r;
                float *pg = G + s * tile_stride + r * MF_LD + 4 * c4;
                const f32x4_t gq = *reinterpret_cast<const f32x4_t *>(pg), xh = *reinterpret_cast<const f32x4_t *>(XH + s * tile_stride + r * MF_LD + 4 * c4);
                f32x4_t o = {0.f, 0.f, 0.f, 0.f};
                if (row < n_valid) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) o[q] = (w[q] * inv[q]) * (gq[q] + fmaf(xh[q], cngx[q], cng[q]));
                }
                *reinterpret_cast<f32x4_t *>(pg) = o;
                if (row < N) *reinterpret_cast<f32x4_t *>(out + (size_t)row * MF_H + 4 * c4) = o;
            }
        }
    }
}

#define MF_STAMP(i)                                                                                        \
    do {                                                                                                   \
        if (a.stamps && threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1))              \
            a.stamps[(blockIdx.x == 0 ? 0 : 8) + (i)] = __builtin_amdgcn_s_memrealtime();                  \
    } while (0)

template <int MAXS>
__global__ __launch_bounds__(MF_THREADS) void mlp_bwd_fused_kernel(MlpBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int TS = 3 * MF_TILE;                          // per step: [G | X | Z] tiles
    float *tG = sm, *tX = sm + MF_TILE, *tZ = sm + 2 * MF_TILE;
    float *cst_o = sm + (size_t)MAXS * TS, *cst_i = cst_o + 4 * MF_H, *mm = cst_i + 4 * MF_H, *red = mm + 2 * MF_H;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, kq = lane >> 4, col0 = 32 * wave;
    const int c4 = tid & 31, rg = tid >> 5;
    const long long N = a.N, G = gridDim.x, g = blockIdx.x;
    const long long num_steps = (N + 15) / 16;
    const int steps = (int)min((long long)MAXS, g < num_steps ? (num_steps - g + G - 1) / G : 0LL);
    unsigned gen = 0;
    if (tid == 0) gen = __hip_atomic_load(a.bar + (MF_SUB + 1 + g % MF_SUB) * MF_BSTRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const long long n_valid = a.n_valid ? min((long long)*a.n_valid, N) : N;
    const int rep = (int)(g % STAT_ACC_REP);

    MF_STAMP(0);
    // ---- P0: everything this launch reads from memory before the first barrier goes in flight now ----
    // both operand slices live in registers for the whole launch (2 x 64): nothing is loaded between the barriers
    f32x4_t hreg[2][8], hreg1[2][8];
    f32x4_t ld[MAXS][2][3];
#pragma unroll
    for (int s = 0; s < MAXS; ++s)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long long row = min((g + (long long)min(s, max(steps - 1, 0)) * G) * 16 + rg + 8 * u, N - 1);
            ld[s][u][0] = *reinterpret_cast<const f32x4_t *>(a.dout + (size_t)row * MF_H + 4 * c4);
            ld[s][u][1] = *reinterpret_cast<const f32x4_t *>(a.y2 + (size_t)row * MF_H + 4 * c4);
            ld[s][u][2] = *reinterpret_cast<const f32x4_t *>(a.z1 + (size_t)row * MF_H + 4 * c4);
        }
    if (tid < MF_H) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int slot = q < 2 ? q : q + 2;              // mean, invstd, w, b = save_stat slots 0, 1, 4, 5
            cst_o[q * MF_H + tid] = a.save_o[slot * MF_H + tid];
            cst_i[q * MF_H + tid] = a.save_i[slot * MF_H + tid];
        }
    }
    mf_load_operand(a.H2Tk4, col0, c, kq, hreg);             // (behind the rows: the first barrier does not need them)
    mf_load_operand(a.H1Tk4, col0, c, kq, hreg1);
#pragma unroll
    for (int s = 0; s < MAXS; ++s)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int r = rg + 8 * u;
            *reinterpret_cast<f32x4_t *>(tG + s * TS + r * MF_LD + 4 * c4) = ld[s][u][0];
            *reinterpret_cast<f32x4_t *>(tX + s * TS + r * MF_LD + 4 * c4) = ld[s][u][1];
            *reinterpret_cast<f32x4_t *>(tZ + s * TS + r * MF_LD + 4 * c4) = ld[s][u][2];
        }
    mf_lds_barrier();
    MF_STAMP(1);
    mf_gate_and_sums<MAXS>(tG, tX, TS, cst_o, a.act_o, steps, g, G, n_valid, red, a.acc_o, rep);
    MF_STAMP(2);
    mf_grid_barrier(a.bar, gen, a.err);
    MF_STAMP(3);

    // ---- P1 ----
    const float cnt = (float)max(n_valid, 1LL);
    if (tid < MF_H) {
        double t0, t1;
        mf_totals(a.acc_o, tid, t0, t1);
        mm[tid] = -((float)t0 / cnt);                        // - mean g
        mm[MF_H + tid] = -((float)t1 / cnt);                 // - mean g x_hat
        if (g == 0 && a.dwb_o) { a.dwb_o[tid] = (float)t1; a.dwb_o[MF_H + tid] = (float)t0; }
    }
    mf_lds_barrier();
    mf_norm_bwd<MAXS>(tG, tX, TS, cst_o, mm, steps, g, G, n_valid, N, a.dy2);
    mf_lds_barrier();
    {
        f32x4_t acc[MAXS][2];
#pragma unroll
        for (int s = 0; s < MAXS; ++s)
            if (s < steps) mf_gemm_tile(tG + s * TS, hreg, c, kq, acc[s]);
        // da -> the X tiles (x_hat_o is consumed; every wavefront writes its own 32 columns)
#pragma unroll
        for (int s = 0; s < MAXS; ++s)
            if (s < steps) {
#pragma unroll
                for (int t = 0; t < 2; ++t) *reinterpret_cast<f32x4_t *>(tX + s * TS + c * MF_LD + col0 + 16 * t + 4 * kq) = acc[s][t];
            }
    }
    mf_lds_barrier();
    MF_STAMP(4);
    mf_gate_and_sums<MAXS>(tX, tZ, TS, cst_i, a.act_i, steps, g, G, n_valid, red, a.acc_i, rep);
    MF_STAMP(5);
    mf_grid_barrier(a.bar, gen, a.err);
    MF_STAMP(6);

    // ---- P2 ----
    if (tid < MF_H) {
        double t0, t1;
        mf_totals(a.acc_i, tid, t0, t1);
        mm[tid] = -((float)t0 / cnt);
        mm[MF_H + tid] = -((float)t1 / cnt);
        if (g == 0 && a.dwb_i) { a.dwb_i[tid] = (float)t1; a.dwb_i[MF_H + tid] = (float)t0; }
    }
    mf_lds_barrier();
    mf_norm_bwd<MAXS>(tX, tZ, TS, cst_i, mm, steps, g, G, n_valid, N, a.dz1);
    mf_lds_barrier();
#pragma unroll
    for (int s = 0; s < MAXS; ++s)
        if (s < steps) {
            f32x4_t acc[2];
            mf_gemm_tile(tX + s * TS, hreg1, c, kq, acc);
            const long long row = (g + (long long)s * G) * 16 + c;
            if (row < N) {
#pragma unroll
                for (int t = 0; t < 2; ++t) *reinterpret_cast<f32x4_t *>(a.ds + (size_t)row * MF_H + col0 + 16 * t + 4 * kq) = acc[t];
            }
        }
    MF_STAMP(7);
}

// ---- host ----------------------------------------------------------------------------------------------------
constexpr int MF_MAXS = 2;

static int mf_capacity() {
    static int cap[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (cap[dev] == 0) {
        const size_t lds = sizeof(float) * mf_lds_floats<MF_MAXS>();
        int per_cu = 0;
        hipDeviceProp_t prop;
        static const bool off = getenv("PHC_MLP_BWD_FUSED") && atoi(getenv("PHC_MLP_BWD_FUSED")) == 0;
        if (off || !ensure_dynamic_lds(&mlp_bwd_fused_kernel<MF_MAXS>, lds) ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mlp_bwd_fused_kernel<MF_MAXS>, MF_THREADS, lds) != hipSuccess ||
            hipGetDeviceProperties(&prop, dev) != hipSuccess || getenv("HSA_CU_MASK") || getenv("ROC_GLOBAL_CU_MASK"))
            cap[dev] = -1;
        else
            cap[dev] = (per_cu > 2 ? 2 : per_cu) * prop.multiProcessorCount;      // (two per CU is all the launch asks for)
        if (cap[dev] == 0) cap[dev] = -1;
    }
    return cap[dev] > 0 ? cap[dev] : 0;
}

}  // namespace phc

using namespace phc;

static unsigned long long *g_mf_stamps = nullptr;

extern "C" {

/* diagnostics only: 16 device words that workgroups 0 and G - 1 of every phc_mlp_norm_bwd_fused launch stamp (100 MHz clock) */
int phc_debug_mlp_bwd_stamps(unsigned long long *dev_words) {
    g_mf_stamps = dev_words;
    return PHC_OK;
}
size_t phc_mlp_norm_bwd_fused_barrier_words(void) { return MF_BAR_WORDS; }

int phc_mlp_norm_bwd_fused_supported(int64_t N, int h) {
    if (h != MF_H || N < 1) return 0;
    const long long steps = (N + 15) / 16, groups = (steps + MF_MAXS - 1) / MF_MAXS;
    return groups <= mf_capacity() ? 1 : 0;
}

int phc_mlp_norm_bwd_fused(const phc_mlp_norm_bwd_desc *d, void *stream) {
    PHC_REQUIRE(d, PHC_ERR_INVALID_ARG, "phc_mlp_norm_bwd_fused: null descriptor");
    PHC_REQUIRE(phc_mlp_norm_bwd_fused_supported(d->N, d->h), PHC_ERR_UNSUPPORTED,
                "phc_mlp_norm_bwd_fused: N=%lld h=%d outside the one-launch form's range (h = 128, ceil(N / 32) co-resident workgroups)",
                (long long)d->N, d->h);
    PHC_REQUIRE(d->dout && d->y2 && d->z1 && d->save_o && d->save_i && d->H2Tk4 && d->H1Tk4 && d->dy2 && d->dz1 && d->ds && d->acc_o &&
                    d->acc_i && d->barrier,
                PHC_ERR_INVALID_ARG, "phc_mlp_norm_bwd_fused: null pointer");
    PHC_REQUIRE(d->act_o >= PHC_ACT_IDENTITY && d->act_o <= PHC_ACT_SELU && d->act_i >= PHC_ACT_IDENTITY && d->act_i <= PHC_ACT_SELU,
                PHC_ERR_INVALID_ARG, "phc_mlp_norm_bwd_fused: unknown activation");
    PHC_REQUIRE(((reinterpret_cast<uintptr_t>(d->dout) | reinterpret_cast<uintptr_t>(d->y2) | reinterpret_cast<uintptr_t>(d->z1) |
                  reinterpret_cast<uintptr_t>(d->H2Tk4) | reinterpret_cast<uintptr_t>(d->H1Tk4) | reinterpret_cast<uintptr_t>(d->dy2) |
                  reinterpret_cast<uintptr_t>(d->dz1) | reinterpret_cast<uintptr_t>(d->ds)) & 15) == 0 &&
                    ((reinterpret_cast<uintptr_t>(d->acc_o) | reinterpret_cast<uintptr_t>(d->acc_i)) & 7) == 0,
                PHC_ERR_UNSUPPORTED, "phc_mlp_norm_bwd_fused: 16-byte aligned rows / 8-byte aligned accumulators required");
    MlpBwdArgs a;
    a.dout = d->dout; a.y2 = d->y2; a.z1 = d->z1; a.save_o = d->save_o; a.save_i = d->save_i; a.H2Tk4 = d->H2Tk4; a.H1Tk4 = d->H1Tk4;
    a.dy2 = d->dy2; a.dz1 = d->dz1; a.ds = d->ds; a.dwb_o = d->dwb_o; a.dwb_i = d->dwb_i;
    a.acc_o = reinterpret_cast<unsigned long long *>(d->acc_o); a.acc_i = reinterpret_cast<unsigned long long *>(d->acc_i);
    a.n_valid = d->n_valid_dev; a.N = d->N; a.act_o = d->act_o; a.act_i = d->act_i;
    a.bar = d->barrier; a.err = err_word();
    a.stamps = g_mf_stamps;
    const long long steps = (d->N + 15) / 16, groups = (steps + MF_MAXS - 1) / MF_MAXS;
    const size_t lds = sizeof(float) * mf_lds_floats<MF_MAXS>();
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_PHM_GEMM_DX, s);
    PHC_LAUNCH(mlp_bwd_fused_kernel<MF_MAXS>, dim3((unsigned)groups), dim3(MF_THREADS), lds, s, a);
    return check_launch("mlp_norm_bwd_fused");
}

}  // extern "C"
