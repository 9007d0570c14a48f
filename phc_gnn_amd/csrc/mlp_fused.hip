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
// AGGT (second half of round 6): `dout` itself is built in P0 — it is the dx of the sum aggregation of the layer ABOVE (autograd of
// messagepassing.py:132-146, identity message function): row pointers + the row's first four out-edges in one round trip, the rows of
// that layer's ds in the next, aggregate_dx_rows_kernel's own additions; the rows are written once (the skip's share reads them).
// Rows >= *n_valid are padding rows of a fixed-shape batch: zero gradient everywhere, as in the separate launches.
// All workgroups must be co-resident (they wait for each other): the host launches this form only when ceil(steps / MAXS)
// workgroups fit the device's real capacity for the kernel, and the barrier's poll is bounded (PHC_ERR_RENDEZVOUS in the device
// error word; the launch then runs to its end with garbage instead of hanging the GPU).
// Arithmetic: x_hat, the pre-activation and dz exactly as the BA transform of gemm_hs_device.h rounds them ((x - mean) invstd, then
// fma: an element on the ReLU kink falls on the same side in every implementation); the MFMA order is gemm_hs16_kernel's; the
// column sums are taken in a fixed order (thread's rows, the wavefront's two row groups, the four wavefronts) and meet in integer
// accumulators: bit-reproducible, and within fp32 round-off of the separate launches' sums.
#include <stdlib.h>

#include <type_traits>

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
    unsigned long long *stamps_all;              // diagnostics only: 8 stamps of EVERY workgroup ([8 g + k])
    // ---- AGGT: `dout` is not there yet — it is the dx of the sum aggregation of the layer ABOVE (messagepassing.py:132-146 backward,
    // identity message function): dout[r] = sum over the out-edges (r -> d) of gsrc[d] (+ gsrc[r] for the self term) (+ residual[r]),
    // gsrc = that layer's ds.  The rows are built in P0 by aggregate_dx_rows_kernel's own additions (encoder.hip: same edge order,
    // same association) and written to dx_out once (the skip's share of the layer below / the encoder's gradient read them) ----
    const float *gsrc;                           // [N, 128]
    const float *residual;                       // [N, 128] or NULL
    const int *rowptr_t, *col_t;                 // source-sorted view (CSC of the destination-sorted one)
    const int *ell_t;                            // [N][4]: the first four out-edges of every row (destinations), csr.hip
    float *dx_out;                               // [N, 128] out
    int add_self;
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
// (Tried: monotonic group counters that every workgroup polls directly — arrive, then see: two round trips instead of four.  With
// 416 workgroups sweeping all 16 counter lines per poll the arrivals themselves slowed down: 5 - 8 us per barrier instead of 2.5 - 4,
// molhiv_b512 0.51 -> 0.55 ms.  The pollers stay on lines nobody adds to.)
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

// the same with the accumulators starting from the bias (gemm_hs16_kernel's order: y = b, then the k-groups in ascending order)
__device__ __forceinline__ void mf_gemm_tile_bias(const float *tile, const f32x4_t (&hreg)[2][8], int c, int kq, const f32x4_t (&bias)[2],
                                                  f32x4_t (&acc)[2]) {
    acc[0] = bias[0];
    acc[1] = bias[1];
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

// a thread's column sums (its rows, four columns) -> the accumulator: the wavefront's two row groups (lanes l, l + 32), then the four
// wavefronts in order through LDS, then one integer add per column and quantity
__device__ __forceinline__ void mf_publish_sums(f32x4_t sg, f32x4_t sgx, float *red, unsigned long long *acc, int rep) {
    const int tid = threadIdx.x, c4 = tid & 31, lane = tid & 63, wave = tid >> 6;
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

// g = up act'(x_hat w + b), x_hat = (x - mean) invstd for this thread's two float4 of every tile; column sums of g and g x_hat over
// the workgroup's valid rows -> the accumulator.  UP / X: LDS tiles (in place: UP := g, X := x_hat); cst: mean | invstd | w | b [4][128]
// (the activation switch stays OUTSIDE the element loops — one straight-line body per kind, RELU a template parameter: as a
// per-element select the run-time switch over six activations was expanded at every element, 160 times in the aggregating forward:
// 20 000 lines of ISA, more than the instruction cache holds, 5 us of arithmetic for two rows)
template <int MAXS, bool RELU>
__device__ __forceinline__ void mf_gate_and_sums_k(float *UP, float *X, int tile_stride, const float *cst, int act, int steps, long long step0,
                                                   long long gstride, long long n_valid, float *red, unsigned long long *acc, int rep) {
    const int tid = threadIdx.x, c4 = tid & 31, rg = tid >> 5;
    const f32x4_t mean = *reinterpret_cast<const f32x4_t *>(cst + 4 * c4), inv = *reinterpret_cast<const f32x4_t *>(cst + MF_H + 4 * c4);
    const f32x4_t w = *reinterpret_cast<const f32x4_t *>(cst + 2 * MF_H + 4 * c4), b = *reinterpret_cast<const f32x4_t *>(cst + 3 * MF_H + 4 * c4);
    f32x4_t sg = {0.f, 0.f, 0.f, 0.f}, sgx = {0.f, 0.f, 0.f, 0.f};
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
                    if constexpr (RELU)
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
    mf_publish_sums(sg, sgx, red, acc, rep);
}
template <int MAXS>
__device__ __forceinline__ void mf_gate_and_sums(float *UP, float *X, int tile_stride, const float *cst, int act, int steps, long long step0,
                                                 long long gstride, long long n_valid, float *red, unsigned long long *acc, int rep) {
    if (act == PHC_ACT_RELU)
        mf_gate_and_sums_k<MAXS, true>(UP, X, tile_stride, cst, act, steps, step0, gstride, n_valid, red, acc, rep);
    else
        mf_gate_and_sums_k<MAXS, false>(UP, X, tile_stride, cst, act, steps, step0, gstride, n_valid, red, acc, rep);
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
            a.stamps[(blockIdx.x == 0 ? 0 : 16) + (i)] = __builtin_amdgcn_s_memrealtime();                 \
        if ((i) < 8 && a.stamps_all && threadIdx.x == 0)                                                   \
            a.stamps_all[8 * blockIdx.x + (i)] = __builtin_amdgcn_s_memrealtime();                         \
    } while (0)

template <int MAXS, bool AGGT>
__global__ __launch_bounds__(MF_THREADS, 2) void mlp_bwd_fused_kernel(MlpBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int TS = 3 * MF_TILE;                          // per step: [G | X | Z] tiles
    float *tG = sm, *tX = sm + MF_TILE, *tZ = sm + 2 * MF_TILE;
    float *cst_o = sm + (size_t)MAXS * TS, *cst_i = cst_o + 4 * MF_H, *mm = cst_i + 4 * MF_H, *red = mm + 2 * MF_H;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, kq = lane >> 4, col0 = 32 * wave;
    const int c4 = tid & 31, rg = tid >> 5;
    // workgroup -> steps.  Plain: round-robin (g, g + G).  AGGT: MAXS consecutive steps, the workgroups of one XCD a contiguous range of
    // them (xcd_contiguous_block; the grid is a multiple of 8) — the forward kernel's map: a molecule's bonds stay inside it, so the
    // rows of the layer above that the gather reads were written through this XCD's L2
    const long long N = a.N, g = blockIdx.x;
    const long long G = AGGT ? 1 : (long long)gridDim.x;     // step stride
    const long long num_steps = (N + 15) / 16;
    const long long g0 = AGGT ? (long long)xcd_contiguous_block(blockIdx.x, gridDim.x) * MAXS : g;      // first step
    const int steps = AGGT ? (int)max(0LL, min((long long)MAXS, num_steps - g0))
                           : (int)min((long long)MAXS, g < num_steps ? (num_steps - g + G - 1) / G : 0LL);
    unsigned gen = 0;
    if (tid == 0) gen = __hip_atomic_load(a.bar + (MF_SUB + 1 + g % MF_SUB) * MF_BSTRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const long long n_valid = a.n_valid ? min((long long)*a.n_valid, N) : N;
    const int rep = (int)(g % STAT_ACC_REP);

    MF_STAMP(0);
    // ---- P0: everything this launch reads from memory before the first barrier goes in flight now ----
    // both operand slices live in registers for the whole launch (2 x 64): nothing is loaded between the barriers
    f32x4_t hreg[2][8], hreg1[2][8];
    f32x4_t ld[MAXS][2][3];
    // the norms' constants: column tid's for the LDS tables (stored behind the rows), and the outer norm's of this thread's four
    // columns straight into registers — the first gate pass runs on the loaded rows BEFORE they go to LDS (one LDS round trip less in
    // front of the first barrier).  AGGT: issued between the gather's two round trips (behind the gather they were a third)
    float cso[4], csi[4];
    f32x4_t o_mean, o_inv, o_w, o_b;
    auto load_consts = [&]() __attribute__((always_inline)) {
        if (tid < MF_H) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int slot = q < 2 ? q : q + 2;          // mean, invstd, w, b = save_stat slots 0, 1, 4, 5
                cso[q] = a.save_o[slot * MF_H + tid];
                csi[q] = a.save_i[slot * MF_H + tid];
            }
        }
        o_mean = *reinterpret_cast<const f32x4_t *>(a.save_o + 4 * c4);
        o_inv = *reinterpret_cast<const f32x4_t *>(a.save_o + MF_H + 4 * c4);
        o_w = *reinterpret_cast<const f32x4_t *>(a.save_o + 4 * MF_H + 4 * c4);
        o_b = *reinterpret_cast<const f32x4_t *>(a.save_o + 5 * MF_H + 4 * c4);
    };
    if constexpr (!AGGT) {
#pragma unroll
        for (int s = 0; s < MAXS; ++s)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const long long row = min((g0 + (long long)min(s, max(steps - 1, 0)) * G) * 16 + rg + 8 * u, N - 1);
                ld[s][u][0] = *reinterpret_cast<const f32x4_t *>(a.dout + (size_t)row * MF_H + 4 * c4);
                ld[s][u][1] = *reinterpret_cast<const f32x4_t *>(a.y2 + (size_t)row * MF_H + 4 * c4);
                ld[s][u][2] = *reinterpret_cast<const f32x4_t *>(a.z1 + (size_t)row * MF_H + 4 * c4);
            }
        load_consts();
    } else {
        // One row per 32-lane group (this thread: columns 4 c4 .. + 3 of rows rg, rg + 8 of every step).  Round trip 1: row pointers
        // and the row's first four out-edges (ell_t: side by side per row) — with the rows of y2 and z1 behind them; round trip 2:
        // the row of gsrc itself, the residual's and the four destinations' (slots past the degree name the row itself: a line it
        // loads anyway), all four rows of the thread in flight together (the operand slices wait until the gather is through).
        const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
        long long dj[MAXS * 2];
        int eb[MAXS * 2], deg[MAXS * 2], dst[MAXS * 2][4];
#pragma unroll
        for (int j = 0; j < MAXS * 2; ++j) {
            const int sI = j >> 1, u = j & 1;
            const long long d = (g0 + sI) * 16 + rg + 8 * u;
            const bool ok = sI < steps && d < N;
            dj[j] = ok ? d : -1;
            const long long dc = ok ? d : 0;
            eb[j] = a.rowptr_t[dc];
            deg[j] = ok ? a.rowptr_t[dc + 1] - eb[j] : 0;
            const int4 ed = *reinterpret_cast<const int4 *>(a.ell_t + dc * 4);
            dst[j][0] = ed.x; dst[j][1] = ed.y; dst[j][2] = ed.z; dst[j][3] = ed.w;
        }
#pragma unroll
        for (int j = 0; j < MAXS * 2; ++j) {
            const long long dc = dj[j] >= 0 ? dj[j] : 0;
            ld[j >> 1][j & 1][1] = *reinterpret_cast<const f32x4_t *>(a.y2 + (size_t)dc * MF_H + 4 * c4);
            ld[j >> 1][j & 1][2] = *reinterpret_cast<const f32x4_t *>(a.z1 + (size_t)dc * MF_H + 4 * c4);
        }
        load_consts();
        const float *res = a.residual ? a.residual : a.gsrc;  // (without a residual the load re-reads the row itself: unconditional)
        f32x4_t sv[MAXS * 2], rv[MAXS * 2], gv[MAXS * 2][4];
#pragma unroll
        for (int j = 0; j < MAXS * 2; ++j) {
            const long long dc = dj[j] >= 0 ? dj[j] : 0;
            sv[j] = *reinterpret_cast<const f32x4_t *>(a.gsrc + (size_t)dc * MF_H + 4 * c4);
            rv[j] = *reinterpret_cast<const f32x4_t *>(res + (size_t)dc * MF_H + 4 * c4);
#pragma unroll
            for (int i = 0; i < 4; ++i) gv[j][i] = *reinterpret_cast<const f32x4_t *>(a.gsrc + (size_t)dst[j][i] * MF_H + 4 * c4);
        }
        // aggregate_dx_rows_kernel's additions: acc = 0 + (g0 + g1); + g2; + g3; ...; + self; + residual
#pragma unroll
        for (int j = 0; j < MAXS * 2; ++j) {
            f32x4_t acc = zero;
            if (deg[j] > 0) acc += gv[j][0] + (deg[j] > 1 ? gv[j][1] : zero);
            if (deg[j] > 2) acc += gv[j][2];
            if (deg[j] > 3) acc += gv[j][3];
            for (int e = eb[j] + 4; e < eb[j] + deg[j]; ++e)     // a row with more than four out-edges: one at a time (rare on molecules)
                acc += *reinterpret_cast<const f32x4_t *>(a.gsrc + (size_t)a.col_t[e] * MF_H + 4 * c4);
            if (a.add_self) acc += sv[j];
            if (a.residual) acc += rv[j];
            ld[j >> 1][j & 1][0] = dj[j] >= 0 ? acc : zero;
        }
        // (every store behind every load: hipcc cannot prove dx_out distinct from the inputs)
#pragma unroll
        for (int j = 0; j < MAXS * 2; ++j)
            if (dj[j] >= 0) *reinterpret_cast<f32x4_t *>(a.dx_out + (size_t)dj[j] * MF_H + 4 * c4) = ld[j >> 1][j & 1][0];
    }
    if (tid < MF_H) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            cst_o[q * MF_H + tid] = cso[q];
            cst_i[q * MF_H + tid] = csi[q];
        }
    }
    mf_load_operand(a.H2Tk4, col0, c, kq, hreg);             // (behind the rows: the first barrier does not need them)
    mf_load_operand(a.H1Tk4, col0, c, kq, hreg1);
    {
        f32x4_t sg = {0.f, 0.f, 0.f, 0.f}, sgx = {0.f, 0.f, 0.f, 0.f};
        auto gate0 = [&](auto relu_c) __attribute__((always_inline)) {
            constexpr bool RELU = decltype(relu_c)::value;
#pragma unroll
        for (int s = 0; s < MAXS; ++s)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int r = rg + 8 * u;
                const long long row = (g0 + (long long)s * G) * 16 + r;
                const bool live = s < steps && row < n_valid;
                f32x4_t gq, xh;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    xh[q] = (ld[s][u][1][q] - o_mean[q]) * o_inv[q];
                    const float pre = fmaf(xh[q], o_w[q], o_b[q]);
                    float gv = ld[s][u][0][q];
                    if constexpr (RELU)
                        gv = pre > 0.f ? gv : 0.f;
                    else
                        gv *= act_grad_rt(a.act_o, pre);
                    gq[q] = live ? gv : 0.f;
                    xh[q] = live ? xh[q] : 0.f;              // (a padding row may hold anything: 0 x NaN must not reach the sums)
                    sg[q] += gq[q];
                    sgx[q] = fmaf(gq[q], xh[q], sgx[q]);
                }
                *reinterpret_cast<f32x4_t *>(tG + s * TS + r * MF_LD + 4 * c4) = gq;
                *reinterpret_cast<f32x4_t *>(tX + s * TS + r * MF_LD + 4 * c4) = xh;
                *reinterpret_cast<f32x4_t *>(tZ + s * TS + r * MF_LD + 4 * c4) = ld[s][u][2];
            }
        };
        if (a.act_o == PHC_ACT_RELU)
            gate0(std::true_type{});
        else
            gate0(std::false_type{});
        MF_STAMP(1);
        mf_publish_sums(sg, sgx, red, a.acc_o, rep);
    }
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
    mf_norm_bwd<MAXS>(tG, tX, TS, cst_o, mm, steps, g0, G, n_valid, N, a.dy2);
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
    mf_gate_and_sums<MAXS>(tX, tZ, TS, cst_i, a.act_i, steps, g0, G, n_valid, red, a.acc_i, rep);
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
    mf_norm_bwd<MAXS>(tX, tZ, TS, cst_i, mm, steps, g0, G, n_valid, N, a.dz1);
    mf_lds_barrier();
#pragma unroll
    for (int s = 0; s < MAXS; ++s)
        if (s < steps) {
            f32x4_t acc[2];
            mf_gemm_tile(tX + s * TS, hreg1, c, kq, acc);
            const long long row = (g0 + (long long)s * G) * 16 + c;
            if (row < N) {
#pragma unroll
                for (int t = 0; t < 2; ++t) *reinterpret_cast<f32x4_t *>(a.ds + (size_t)row * MF_H + col0 + 16 * t + 4 * kq) = acc[t];
            }
        }
    MF_STAMP(7);
}

// =====================================================================================================================
// The FORWARD of the same node behind the aggregation:  z1 = s . H1 + b1;  a = act(norm_i(z1));  y2 = a . H2 + b2, with the
// statistics of y2 left in an accumulator for whoever applies the model-level norm (ops.PendingNorm) — layers.py:349-355 in
// training mode (norm.py:30-35: batch statistics over the valid rows, running statistics updated).  Two launches so far
// (gemm_hs16<.., ST> or the aggregation-carrying gemm_hs_ag_kernel, then gemm_hs16<.., XF | ST>: 21.6 + 16.4 us at molhiv_b512);
// here one: the rows of z1 stay in the wavefronts' REGISTERS across the one grid-wide dependency (the norm's statistics over all
// rows), only the accumulator crosses the barrier, and the activations `a` never reach memory.
//     (AGG = 1 / 2: P0 builds the rows of s itself — the layer's sum / softmax aggregation over x or over the un-applied
//      act(norm(y)) (+ skip) of the layer in front: MlpFwdArgs)
//     P0   rows of s -> LDS;  z1 = s . H1 + b1 -> registers (memory behind the barrier: the backward reads it);  column sum | sum of squares of
//          z1 over the valid rows (double, per wavefront: it owns its 32 columns for every row) -> acc1          | barrier
//     P1   scale / shift from the totals (bn_from_totals: one definition for every kernel); workgroup 0 leaves save_stat and the
//          running statistics behind;  a = act(z1 scale + shift) -> LDS (rows >= *n_valid: 0);  y2 = a . H2 + b2 -> memory;
//          statistics of y2 -> acc2
// MFMA order, bias handling and the statistics' pivot-free double sums: z1 and y2 are bit-identical to gemm_hs16_kernel's for the
// same operands; the totals agree to double round-off.
struct MlpFwdArgs {
    const float *s;                              // [N, 128]
    const float *H1k4, *b1, *H2k4, *b2;          // k4 operands; biases [128] or NULL
    float *z1, *y2;                              // [N, 128] out
    unsigned long long *acc1, *acc2;             // zeroed accumulators of width 128 (acc1: this launch's own exchange)
    BnParams p;                                  // the inner norm's parameter tables
    float momentum, eps;
    float *save_i;                               // [6 x 128] out (workgroup 0)
    int act_i;
    const int *n_valid;
    long long N;
    unsigned *bar;
    int *err;
    unsigned long long *stamps, *stamps_all;
    // ---- AGG: the layer's sum aggregation builds the rows of s inside the launch (messagepassing.py:132-146, identity message
    // function; encoder.py:31-34): s[d] = sum_{e: dst_e = d} (x[src_e] + sum_f table_f[code_f(e)]) (+ x[d]), x either a tensor or the
    // un-applied  act(norm_o(y)) (+ skip)  of the layer in front (nacc != NULL; models.py:208-215) — what aggregate_encp_direct_fwd /
    // aggregate_norm_direct_fwd (encoder.hip) compute in a launch of their own, same edge order and additions: s is bit-identical ----
    const float *x;                              // [N, 128] gathered tensor, or the pending norm's input y
    const unsigned *codes;                       // [E] packed bond codes, destination-sorted
    const float *const *tables;                  // device [n F] table pointers
    const float *tp[16];                         // ... the same pointers by value (n F <= 16): no pointer hop in front of the table rows
    const int *rowptr, *col;
    const int *ell;                              // [N][8]: the first four in-edges of every row (sources | codes), csr.hip
    int F, R, off[5], n_cmp, k, add_self;
    float *s_out;                                // [N, 128] out
    const unsigned long long *nacc;              // pending norm: the accumulator its producing GEMM filled; NULL: x is a tensor
    BnParams po;
    float mom_o, eps_o;
    int act_o;
    const float *skip;                           // [N, 128] or NULL
    float *stat_o, *x_out;                       // [6 x 128], [N, 128] out
    // ---- AGG == 2: the SOFTMAX aggregation instead of the sum (messagepassing.py:211-214, 297-300; softmax_agg.hip's walk): per
    // destination and column  out = sum_e softmax_e(beta m_e) m_e,  m_e = x[src_e] + (t0 + t1 + ...) — the online softmax of
    // softmax_agg_norm_fwd_kernel in its edge order, the table rows added in ITS association: out, lse and s = out (+ x) are
    // bit-identical to that launch's for in-degrees <= 16 ----
    const float *beta;                           // device scalar
    float *agg_out, *lse;                        // [N, 128] out: the plain aggregate and max + log(sum + eps) (the backward walk reads them)
};
constexpr int MF_AGG_MAX_R = 32;                 // bond-table rows staged in LDS

// sum over the 16 lanes that share kq (xor offsets 1, 2, 4, 8 stay inside the group)
__device__ __forceinline__ double mf_sum16(double v) {
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// column sums / sums of squares (double) of this wavefront's 32 columns over the workgroup's valid rows -> accumulator.  A lane holds
// one row (c) of 8 columns per step; the 16 rows meet through LDS (wred: [16 rows][64 (column, quantity)] doubles of this wavefront):
// as 4 x 16 dependent cross-lane shuffles of doubles the reduction took 3.2 us of the launch (scripts/dbg/mlp_fwd_phases.py).
// Lane L then owns (column L / 2, quantity L % 2) and adds its 16 rows in row order: one fixed tree.
template <int MAXS>
__device__ __forceinline__ void mf_column_stats(const f32x4_t (&v)[MAXS][2], int steps, long long step0, long long gstride, long long n_valid,
                                                int c, int kq, int col0, double *wred, unsigned long long *acc, int rep) {
    double S[2][4], Q[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) S[t][q] = Q[t][q] = 0.0;
#pragma unroll
    for (int s = 0; s < MAXS; ++s) {
        const long long row = (step0 + s * gstride) * 16 + c;
        if (s < steps && row < n_valid) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const double d = (double)v[s][t][q];
                    S[t][q] += d;
                    Q[t][q] = fma(d, d, Q[t][q]);
                }
        }
    }
    // wred[c][2 * lc + qty], lc = 16 t + 4 kq + q the wavefront's local column; row stride 66 doubles (the 16 lanes of a kq group
    // write 16 different rows: 66 x 8 bytes apart = 2 banks off per row)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int lc = 16 * t + 4 * kq + q;
            wred[c * 66 + 2 * lc] = S[t][q];
            wred[c * 66 + 2 * lc + 1] = Q[t][q];
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // (the wavefront's own LDS writes: no workgroup barrier needed)
    const int lane = 16 * kq + c;
    double mine = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) mine += wred[r * 66 + lane];
    stat_acc_add(acc, MF_H, rep, lane & 1, col0 + (lane >> 1), mine);
}

template <int MAXS, int AGG>                     // AGG: 0 rows of s from memory, 1 sum aggregation, 2 softmax aggregation
__global__ __launch_bounds__(MF_THREADS, 2) void mlp_fwd_fused_kernel(MlpFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *tS = sm;                                          // [MAXS] tiles: s, then a
    float *ss = sm + (size_t)MAXS * MF_TILE;                 // scale | shift [2][128]
    double *wred = reinterpret_cast<double *>(ss + 2 * MF_H) + (size_t)(threadIdx.x >> 6) * (16 * 66);   // per wavefront [16][66] doubles
    float *ss_o = reinterpret_cast<float *>(reinterpret_cast<double *>(ss + 2 * MF_H) + (size_t)MF_WAVES * 16 * 66);   // AGG: the pending norm's scale | shift
    float *tbl = ss_o + 2 * MF_H;                            // AGG: bond tables [R][128]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, kq = lane >> 4, col0 = 32 * wave;
    const int c4 = tid & 31, rg = tid >> 5;
    // workgroup -> rows: MAXS consecutive 16-row steps, and the workgroups of one XCD (block ids congruent mod 8) a CONTIGUOUS range of
    // them (phc_common.h: xcd_contiguous_block; the grid is a multiple of 8) — a molecule's atoms are consecutive rows and its bonds
    // stay inside it, so the rows the aggregation gathers come from that XCD's L2 (round-robin steps made every XCD touch every row:
    // the gather rounds ran at Infinity-Cache bandwidth, 19 us in front of the first GEMM)
    const long long N = a.N, g = blockIdx.x;
    const long long lb = xcd_contiguous_block(blockIdx.x, gridDim.x), G = 1;
    const long long num_steps = (N + 15) / 16;
    const long long gs = lb * MAXS;                          // first step of this workgroup
    const int steps = (int)max(0LL, min((long long)MAXS, num_steps - gs));
    unsigned gen = 0;
    if (tid == 0) gen = __hip_atomic_load(a.bar + (MF_SUB + 1 + g % MF_SUB) * MF_BSTRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const long long n_valid = a.n_valid ? min((long long)*a.n_valid, N) : N;
    const int rep = (int)(g % STAT_ACC_REP);
    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
    MF_STAMP(0);
    // ---- P0 ----
    f32x4_t ld[MAXS][2];
    BnFromTotals bt_o = {};
    float ow = 1.f, ob = 0.f;
    if constexpr (AGG == 0) {
#pragma unroll
        for (int s = 0; s < MAXS; ++s)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const long long row = min((gs + (long long)min(s, max(steps - 1, 0)) * G) * 16 + rg + 8 * u, N - 1);
                ld[s][u] = *reinterpret_cast<const f32x4_t *>(a.s + (size_t)row * MF_H + 4 * c4);
            }
    } else {
        // One row per 32-lane group (this thread: columns 4 c4 .. + 3 of rows rg, rg + 8 of every step).  Dependent chain:
        // row pointers -> sources | codes of the first four edges -> rows (two rounds of two edges, all four rows of the thread in
        // flight together); meanwhile the bond tables go to LDS and the pending norm's scale / shift are derived from its totals.
        const bool normed = a.nacc != nullptr;
        const long long count = n_valid;
        long long dj[MAXS * 2];
        int eb[MAXS * 2], deg[MAXS * 2];
        int src[MAXS * 2][4];
        unsigned code[MAXS * 2][4];
        // round trip 1: row pointers AND the row's first four edges (ell: csr.hip writes them side by side per row)
#pragma unroll
        for (int j = 0; j < MAXS * 2; ++j) {
            const int sI = j >> 1, u = j & 1;
            const long long d = (gs + (long long)sI * G) * 16 + rg + 8 * u;
            const bool ok = sI < steps && d < N;
            dj[j] = ok ? d : -1;
            const long long dc = ok ? d : 0;
            eb[j] = a.rowptr[dc];
            deg[j] = ok ? a.rowptr[dc + 1] - eb[j] : 0;
            const int4 es = *reinterpret_cast<const int4 *>(a.ell + dc * 8), ec = *reinterpret_cast<const int4 *>(a.ell + dc * 8 + 4);
            src[j][0] = es.x; src[j][1] = es.y; src[j][2] = es.z; src[j][3] = es.w;
            code[j][0] = (unsigned)ec.x; code[j][1] = (unsigned)ec.y; code[j][2] = (unsigned)ec.z; code[j][3] = (unsigned)ec.w;
        }
        const float *skp = a.skip ? a.skip : a.x;            // (without a skip the second load re-reads y: the loads stay unconditional)
        // the raw rows of one step's two rows: the row itself and its first four sources
        auto issue_rows = [&](int sI, f32x4_t (&ys)[2], f32x4_t (&ks)[2], f32x4_t (&ye)[2][4], f32x4_t (&ke)[2][4]) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int j = 2 * sI + u;
                const long long dc = dj[j] >= 0 ? dj[j] : 0;
                ys[u] = *reinterpret_cast<const f32x4_t *>(a.x + (size_t)dc * MF_H + 4 * c4);
                ks[u] = *reinterpret_cast<const f32x4_t *>(skp + (size_t)dc * MF_H + 4 * c4);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ye[u][i] = *reinterpret_cast<const f32x4_t *>(a.x + (size_t)src[j][i] * MF_H + 4 * c4);
                    ke[u][i] = *reinterpret_cast<const f32x4_t *>(skp + (size_t)src[j][i] * MF_H + 4 * c4);
                }
            }
        };
        // Everything else the prologue reads goes in flight in the SAME round trip as the row pointers and edge slots — the bond
        // table entries (addresses out of the kernel arguments), the pending norm's totals and affine parameters — BEFORE the first
        // use of an edge slot makes the wavefront wait: behind issue_rows they were a second round trip in front of the LDS barrier
        // (prologue 5.4 us, scripts/dbg/mlp_fwd_agg_phases.py).  bond tables -> LDS: row r of feature f(r), component cc = column / k.
        // All of a thread's entries travel together (clamped, unrolled; pointers out of the kernel arguments by constant index: as a
        // run-time loop with a pointer hop per entry the staging alone was four dependent round trips, 16 us in front of the gather)
        f32x4_t tv[MF_AGG_MAX_R * 32 / MF_THREADS];
#pragma unroll
        for (int it = 0; it < MF_AGG_MAX_R * 32 / MF_THREADS; ++it) {
            const int e = min(tid + it * MF_THREADS, a.R * 32 - 1);
            const int r = e >> 5, q = e & 31, cc = (4 * q) / a.k, kk = 4 * q - cc * a.k;
            int f = 0, o = 0;
#pragma unroll
            for (int u = 1; u < 4; ++u) {
                const bool past = u < a.F && r >= a.off[u];
                f = past ? u : f;
                o = past ? a.off[u] : o;
            }
            const int ti = cc * a.F + f;
            const float *base = a.tp[0];
#pragma unroll
            for (int u = 1; u < 16; ++u) base = ti == u ? a.tp[u] : base;
            tv[it] = *reinterpret_cast<const f32x4_t *>(base + (size_t)(r - o) * a.k + kk);
        }
        unsigned long long tw[STAT_ACC_REP][2][2];           // (the totals of the launch in front: plain loads across a kernel boundary)
        const bool norm_lane = normed && tid < MF_H;         // wave-uniform (wavefronts 0 and 1)
        if (norm_lane) {
#pragma unroll
            for (int rp = 0; rp < STAT_ACC_REP; ++rp)
#pragma unroll
                for (int qty = 0; qty < 2; ++qty) {
                    const unsigned long long *pw = a.nacc + ((size_t)(rp * 2 + qty) * 2) * MF_H + tid;
                    tw[rp][qty][0] = pw[0];
                    tw[rp][qty][1] = pw[MF_H];
                }
            const int comp = tid / a.po.k, kk = tid - comp * a.po.k;
            if (a.po.has_weight) ow = *bn_col_ptr(a.po.weight, a.po.flat, comp, kk, tid);
            if (a.po.has_bias) ob = *bn_col_ptr(a.po.bias, a.po.flat, comp, kk, tid);
        }
        // the first step's rows go in flight as soon as the edge slots are there: they travel under the table staging, the pending
        // norm's arithmetic and the barrier below
        f32x4_t ys0[2], ks0[2], ye0[2][4], ke0[2][4];
        issue_rows(0, ys0, ks0, ye0, ke0);
#pragma unroll
        for (int it = 0; it < MF_AGG_MAX_R * 32 / MF_THREADS; ++it) {
            const int e = tid + it * MF_THREADS;
            if (e < a.R * 32) *reinterpret_cast<f32x4_t *>(tbl + (e >> 5) * MF_H + 4 * (e & 31)) = tv[it];
        }
        if (norm_lane) {
            double S = 0.0, Q = 0.0;
#pragma unroll
            for (int rp = 0; rp < STAT_ACC_REP; ++rp) {      // (stat_acc_part's arithmetic, in its order)
                S += (double)(long long)tw[rp][0][0] * (1.0 / 4096.0) + (double)(long long)tw[rp][0][1] * (1.0 / 4611686018427387904.0);
                Q += (double)(long long)tw[rp][1][0] * (1.0 / 4096.0) + (double)(long long)tw[rp][1][1] * (1.0 / 4611686018427387904.0);
            }
            bt_o = bn_from_totals(S, Q, count, ow, ob, a.eps_o);
            ss_o[tid] = bt_o.scale;
            ss_o[MF_H + tid] = bt_o.shift;
        }
        MF_STAMP(8);
        mf_lds_barrier();                                    // tables and scale / shift are in LDS
        MF_STAMP(9);
        const f32x4_t sc = normed ? *reinterpret_cast<const f32x4_t *>(ss_o + 4 * c4) : zero;
        const f32x4_t sh = normed ? *reinterpret_cast<const f32x4_t *>(ss_o + MF_H + 4 * c4) : zero;
        // one normalised row slice: act(y scale + shift) (+ skip), 0 for a padding row; the plain tensor otherwise
        // KIND 0: a plain tensor; 1: relu(norm) (+ skip); 2: any other activation
        auto xrow = [&](auto kind_c, long long r, const f32x4_t yv, const f32x4_t kv) __attribute__((always_inline)) {
            constexpr int KIND = decltype(kind_c)::value;
            if constexpr (KIND == 0) {
                return yv;
            } else {
                f32x4_t o;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float pre = fmaf(yv[q], sc[q], sh[q]);
                    o[q] = KIND == 1 ? fmaxf(pre, 0.f) : act_fwd_rt(a.act_o, pre);
                    if (a.skip) o[q] += kv[q];
                    if (r >= count) o[q] = 0.f;
                }
                return o;
            }
        };
        // softmax_agg.hip's sm_enc_rows: the edge's table rows summed FIRST (t0 + t1 + ...), then added to the row
        auto table_sum = [&](unsigned cd) __attribute__((always_inline)) {
            f32x4_t t = *reinterpret_cast<const f32x4_t *>(tbl + (cd & 255u) * MF_H + 4 * c4);
#pragma unroll
            for (int f = 1; f < 4; ++f)
                if (f < a.F) t += *reinterpret_cast<const f32x4_t *>(tbl + ((cd >> (8 * f)) & 255u) * MF_H + 4 * c4);
            return t;
        };
        const float beta = AGG == 2 ? *a.beta : 0.f;
        float *keep2 = reinterpret_cast<float *>(reinterpret_cast<double *>(ss + 2 * MF_H));   // AGG == 2: [2][16 MAXS][MF_LD] over the statistics' scratch
        static_assert((size_t)2 * 16 * MAXS * MF_LD * sizeof(float) <= sizeof(double) * (size_t)MF_WAVES * 16 * 66, "aggregate + lse rows fit the scratch");
        auto add_tables = [&](f32x4_t &v, unsigned cd) __attribute__((always_inline)) {
#pragma unroll
            for (int f = 0; f < 4; ++f)
                if (f < a.F) v += *reinterpret_cast<const f32x4_t *>(tbl + ((cd >> (8 * f)) & 255u) * MF_H + 4 * c4);
        };
        // round trips 2 and 3: the row itself and its first four sources (slots past the degree name the row itself: lines it loads
        // anyway), one STEP's two rows at a time — all four rows in flight are 160 registers beside the sources and codes: 13 spilled
        // values whose scratch traffic sat in the middle of the gather (9 us between issuing the loads and the sums).  Additions in the
        // row walker's order (acc0: edges 0, 2, 3, ...; acc1: edge 1)
        f32x4_t self_keep[MAXS * 2];
        auto gather = [&](auto kind_c) __attribute__((always_inline)) {
#pragma unroll
        for (int sI = 0; sI < MAXS; ++sI) {
            f32x4_t ys[2], ks[2], ye[2][4], ke[2][4];
            if (sI == 0) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    ys[u] = ys0[u]; ks[u] = ks0[u];
#pragma unroll
                    for (int i = 0; i < 4; ++i) { ye[u][i] = ye0[u][i]; ke[u][i] = ke0[u][i]; }
                }
            } else {
                issue_rows(sI, ys, ks, ye, ke);
            }
            if (sI == 0) MF_STAMP(10);
            if (a.stamps) {                                  // diagnostics: the loads' arrival apart from the arithmetic behind it
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (sI == 0) MF_STAMP(12); else MF_STAMP(14);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int j = 2 * sI + u;
                const f32x4_t selfv = xrow(kind_c, dj[j] >= 0 ? dj[j] : 0, ys[u], ks[u]);
                f32x4_t a4;
                if constexpr (AGG == 2) {
                    // the online softmax of softmax_agg_norm_fwd_kernel: edges in order, one (max, sum, weighted sum) per column
                    float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, sN[4] = {0.f, 0.f, 0.f, 0.f}, wN[4] = {0.f, 0.f, 0.f, 0.f};
                    auto update = [&](const f32x4_t xv, unsigned cd) __attribute__((always_inline)) {
                        const f32x4_t m = xv + table_sum(cd);
#pragma unroll
                        for (int q = 0; q < 4; ++q) sm_update(mx[q], sN[q], wN[q], m[q], beta);
                    };
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (deg[j] > i) update(xrow(kind_c, src[j][i], ye[u][i], ke[u][i]), code[j][i]);
                    for (int e = eb[j] + 4; e < eb[j] + deg[j]; ++e) {   // more than four in-edges: one at a time (rare on molecules)
                        const int sJ = a.col[e];
                        update(xrow(kind_c, sJ, *reinterpret_cast<const f32x4_t *>(a.x + (size_t)sJ * MF_H + 4 * c4),
                                    *reinterpret_cast<const f32x4_t *>(skp + (size_t)sJ * MF_H + 4 * c4)), a.codes[e]);
                    }
                    f32x4_t o, l;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float z = sN[q] + SM_EPS;
                        o[q] = deg[j] > 0 ? wN[q] / z : 0.f;
                        l[q] = deg[j] > 0 ? mx[q] + __logf(z) : 0.f;
                    }
                    // (the aggregate and its log-sum-exp wait in LDS — the statistics' scratch, idle until the first GEMM is through — for
                    // the stores behind the gather: in registers they were 32 more across the second step's loads)
                    const int Lr = 16 * sI + rg + 8 * u;
                    *reinterpret_cast<f32x4_t *>(keep2 + Lr * MF_LD + 4 * c4) = o;
                    *reinterpret_cast<f32x4_t *>(keep2 + (16 * MAXS + Lr) * MF_LD + 4 * c4) = l;
                    a4 = o;
                } else {
                    f32x4_t v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        v[i] = xrow(kind_c, src[j][i], ye[u][i], ke[u][i]);
                        add_tables(v[i], code[j][i]);
                    }
                    f32x4_t acc0 = deg[j] > 0 ? v[0] : zero;
                    const f32x4_t acc1 = deg[j] > 1 ? v[1] : zero;
                    if (deg[j] > 2) acc0 += v[2];
                    if (deg[j] > 3) acc0 += v[3];
                    for (int e = eb[j] + 4; e < eb[j] + deg[j]; ++e) {   // a row with more than four in-edges: one at a time (rare on molecules)
                        const int sJ = a.col[e];
                        const unsigned cd = a.codes[e];
                        f32x4_t vv = xrow(kind_c, sJ, *reinterpret_cast<const f32x4_t *>(a.x + (size_t)sJ * MF_H + 4 * c4),
                                          *reinterpret_cast<const f32x4_t *>(skp + (size_t)sJ * MF_H + 4 * c4));
                        add_tables(vv, cd);
                        acc0 += vv;
                    }
                    a4 = acc0 + acc1;
                }
                if (a.add_self) a4 += selfv;
                if (dj[j] < 0) a4 = zero;
                ld[sI][u] = a4;
                self_keep[j] = selfv;
            }
            if (sI == 0) MF_STAMP(13);
        }
        MF_STAMP(15);
        };
        if (!normed)
            gather(std::integral_constant<int, 0>{});
        else if (a.act_o == PHC_ACT_RELU)
            gather(std::integral_constant<int, 1>{});
        else
            gather(std::integral_constant<int, 2>{});
        // every store BEHIND every load: hipcc cannot prove x_out / s_out distinct from x / skip and drains the stores (a full
        // s_waitcnt vmcnt(0): two microseconds of store completion) in front of every later load — per row that was 9 us of the gather
#pragma unroll
        for (int j = 0; j < MAXS * 2; ++j)
            if (dj[j] >= 0) {
                if (normed) *reinterpret_cast<f32x4_t *>(a.x_out + (size_t)dj[j] * MF_H + 4 * c4) = self_keep[j];   // x itself, once
                *reinterpret_cast<f32x4_t *>(a.s_out + (size_t)dj[j] * MF_H + 4 * c4) = ld[j >> 1][j & 1];
                if constexpr (AGG == 2) {
                    const int Lr = 16 * (j >> 1) + rg + 8 * (j & 1);
                    *reinterpret_cast<f32x4_t *>(a.agg_out + (size_t)dj[j] * MF_H + 4 * c4) = *reinterpret_cast<const f32x4_t *>(keep2 + Lr * MF_LD + 4 * c4);
                    *reinterpret_cast<f32x4_t *>(a.lse + (size_t)dj[j] * MF_H + 4 * c4) =
                        *reinterpret_cast<const f32x4_t *>(keep2 + (16 * MAXS + Lr) * MF_LD + 4 * c4);
                }
            }
        MF_STAMP(11);
    }
    // the norm's affine parameters of column tid (unconditional loads; without a table the running mean's address is read)
    float nw = 1.f, nb = 0.f;
    if (tid < MF_H) {
        const int comp = tid / a.p.k, kk = tid - comp * a.p.k;
        if (a.p.has_weight) nw = *bn_col_ptr(a.p.weight, a.p.flat, comp, kk, tid);
        if (a.p.has_bias) nb = *bn_col_ptr(a.p.bias, a.p.flat, comp, kk, tid);
    }
    f32x4_t bias1[2], bias2[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        bias1[t] = a.b1 ? *reinterpret_cast<const f32x4_t *>(a.b1 + col0 + 16 * t + 4 * kq) : zero;
        bias2[t] = a.b2 ? *reinterpret_cast<const f32x4_t *>(a.b2 + col0 + 16 * t + 4 * kq) : zero;
    }
    // (AGG: the operand slices go in flight behind the gather — its rounds need the registers)
    f32x4_t hreg[2][8], hreg2[2][8];
    mf_load_operand(a.H1k4, col0, c, kq, hreg);
    mf_load_operand(a.H2k4, col0, c, kq, hreg2);
#pragma unroll
    for (int s = 0; s < MAXS; ++s)
#pragma unroll
        for (int u = 0; u < 2; ++u) *reinterpret_cast<f32x4_t *>(tS + s * MF_TILE + (rg + 8 * u) * MF_LD + 4 * c4) = ld[s][u];
    mf_lds_barrier();
    MF_STAMP(1);
    f32x4_t z[MAXS][2];
#pragma unroll
    for (int s = 0; s < MAXS; ++s) {
        z[s][0] = z[s][1] = zero;
        if (s < steps) {
            mf_gemm_tile_bias(tS + s * MF_TILE, hreg, c, kq, bias1, z[s]);
        }
    }
    MF_STAMP(2);
    mf_column_stats<MAXS>(z, steps, gs, G, n_valid, c, kq, col0, wred, a.acc1, rep);
    MF_STAMP(3);
    mf_grid_barrier(a.bar, gen, a.err);
    MF_STAMP(4);
    // z1 goes to memory BEHIND the barrier (the backward reads it; the rows stay in registers across the barrier anyway): in front of
    // it the arrival waited for these stores to complete (the barrier's wait covers everything the workgroup has in flight)
#pragma unroll
    for (int s = 0; s < MAXS; ++s) {
        const long long row = (gs + (long long)s * G) * 16 + c;
        if (s < steps && row < N) {
#pragma unroll
            for (int t = 0; t < 2; ++t) *reinterpret_cast<f32x4_t *>(a.z1 + (size_t)row * MF_H + col0 + 16 * t + 4 * kq) = z[s][t];
        }
    }
    // ---- P1 ----
    BnFromTotals bt = {};
    if (tid < MF_H) {
        double S, Q;
        mf_totals(a.acc1, tid, S, Q);
        bt = bn_from_totals(S, Q, n_valid, nw, nb, a.eps);
        ss[tid] = bt.scale;
        ss[MF_H + tid] = bt.shift;
    }
    mf_lds_barrier();
    {
        auto xf = [&](auto relu_c) __attribute__((always_inline)) {
        constexpr bool RELU = decltype(relu_c)::value;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f32x4_t sc = *reinterpret_cast<const f32x4_t *>(ss + col0 + 16 * t + 4 * kq);
            const f32x4_t sh = *reinterpret_cast<const f32x4_t *>(ss + MF_H + col0 + 16 * t + 4 * kq);
#pragma unroll
            for (int s = 0; s < MAXS; ++s) {
                if (s < steps) {
                    const long long row = (gs + (long long)s * G) * 16 + c;
                    f32x4_t v;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float pre = fmaf(z[s][t][q], sc[q], sh[q]);
                        v[q] = row < n_valid ? (RELU ? fmaxf(pre, 0.f) : act_fwd_rt(a.act_i, pre)) : 0.f;
                    }
                    *reinterpret_cast<f32x4_t *>(tS + s * MF_TILE + c * MF_LD + col0 + 16 * t + 4 * kq) = v;
                }
            }
        }
        };
        if (a.act_i == PHC_ACT_RELU)
            xf(std::true_type{});
        else
            xf(std::false_type{});
    }
    mf_lds_barrier();
    MF_STAMP(5);
#pragma unroll
    for (int s = 0; s < MAXS; ++s) {
        z[s][0] = z[s][1] = zero;
        if (s < steps) {
            mf_gemm_tile_bias(tS + s * MF_TILE, hreg2, c, kq, bias2, z[s]);
            const long long row = (gs + (long long)s * G) * 16 + c;
            if (row < N) {
#pragma unroll
                for (int t = 0; t < 2; ++t) *reinterpret_cast<f32x4_t *>(a.y2 + (size_t)row * MF_H + col0 + 16 * t + 4 * kq) = z[s][t];
            }
        }
    }
    MF_STAMP(6);
    mf_column_stats<MAXS>(z, steps, gs, G, n_valid, c, kq, col0, wred, a.acc2, rep);
    MF_STAMP(7);
    // what the norm leaves behind (save_stat, the running statistics' read-modify-write): by the LAST workgroup — the one with the
    // fewest rows — behind everything else it does; on workgroup 0's path in front of the second GEMM this was 4.4 us of the launch
    if (lb == (long long)gridDim.x - 1) {
        if (tid < MF_H) {
            float *st = a.save_i;
            st[tid] = bt.mean; st[MF_H + tid] = bt.invstd; st[2 * MF_H + tid] = bt.scale; st[3 * MF_H + tid] = bt.shift;
            st[4 * MF_H + tid] = nw; st[5 * MF_H + tid] = nb;
            if (a.p.has_running) {
                const int comp = tid / a.p.k, kk = tid - comp * a.p.k;
                float *rm = bn_col_ptr(a.p.running_mean, a.p.flat, comp, kk, tid), *rv = bn_col_ptr(a.p.running_var, a.p.flat, comp, kk, tid);
                *rm = bn_running_update(*rm, a.momentum, bt.mean);
                *rv = bn_running_update(*rv, a.momentum, bt.unbiased);
            }
        }
        if (a.p.has_nbt && tid < a.p.n_comp)
            for (int q = 0; q < a.p.n_comp; ++q)
                if (q == tid) *a.p.num_batches_tracked[q] += 1;
        if constexpr (AGG != 0) {
            if (a.nacc) {                                    // ... and the pending norm of the layer in front
                if (tid < MF_H) {
                    float *st = a.stat_o;
                    st[tid] = bt_o.mean; st[MF_H + tid] = bt_o.invstd; st[2 * MF_H + tid] = bt_o.scale; st[3 * MF_H + tid] = bt_o.shift;
                    st[4 * MF_H + tid] = ow; st[5 * MF_H + tid] = ob;
                    if (a.po.has_running) {
                        const int comp = tid / a.po.k, kk = tid - comp * a.po.k;
                        float *rm = bn_col_ptr(a.po.running_mean, a.po.flat, comp, kk, tid), *rv = bn_col_ptr(a.po.running_var, a.po.flat, comp, kk, tid);
                        *rm = bn_running_update(*rm, a.mom_o, bt_o.mean);
                        *rv = bn_running_update(*rv, a.mom_o, bt_o.unbiased);
                    }
                }
                if (a.po.has_nbt && tid < a.po.n_comp)
                    for (int q = 0; q < a.po.n_comp; ++q)
                        if (q == tid) *a.po.num_batches_tracked[q] += 1;
            }
        }
    }
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
        if (off || !ensure_dynamic_lds(&mlp_bwd_fused_kernel<MF_MAXS, false>, lds) || !ensure_dynamic_lds(&mlp_bwd_fused_kernel<MF_MAXS, true>, lds) ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mlp_bwd_fused_kernel<MF_MAXS, true>, MF_THREADS, lds) != hipSuccess ||
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

static unsigned long long *g_mf_stamps = nullptr, *g_mf_stamps_all = nullptr;

static size_t mf_fwd_lds_bytes(bool agg = false) {
    return sizeof(float) * ((size_t)MF_MAXS * MF_TILE + 2 * MF_H) + sizeof(double) * (size_t)MF_WAVES * 16 * 66 +
           (agg ? sizeof(float) * (size_t)(2 * MF_H + MF_AGG_MAX_R * MF_H) : 0);
}
template <typename K>
static int mf_capacity_of(K kernel, int *cache, size_t lds, const char *switch_env) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (cache[dev] == 0) {
        int per_cu = 0;
        hipDeviceProp_t prop;
        const bool off = getenv(switch_env) && atoi(getenv(switch_env)) == 0;
        if (off || !ensure_dynamic_lds(kernel, lds) || hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, MF_THREADS, lds) != hipSuccess ||
            hipGetDeviceProperties(&prop, dev) != hipSuccess || getenv("HSA_CU_MASK") || getenv("ROC_GLOBAL_CU_MASK"))
            cache[dev] = -1;
        else
            cache[dev] = (per_cu > 2 ? 2 : per_cu) * prop.multiProcessorCount;
        if (cache[dev] == 0) cache[dev] = -1;
    }
    return cache[dev] > 0 ? cache[dev] : 0;
}
static int mf_fwd_capacity(bool agg) {
    static int cap[3][64];
    if (!agg) return mf_capacity_of(&mlp_fwd_fused_kernel<MF_MAXS, 0>, cap[0], mf_fwd_lds_bytes(false), "PHC_MLP_FWD_FUSED");
    // (the sum and the softmax form: the smaller of the two answers serves both)
    const int c1 = mf_capacity_of(&mlp_fwd_fused_kernel<MF_MAXS, 1>, cap[1], mf_fwd_lds_bytes(true), "PHC_MLP_FWD_FUSED");
    const int c2 = mf_capacity_of(&mlp_fwd_fused_kernel<MF_MAXS, 2>, cap[2], mf_fwd_lds_bytes(true), "PHC_MLP_FWD_FUSED");
    return c1 < c2 ? c1 : c2;
}

extern "C" {

/* diagnostics only: 32 device words that workgroups 0 and G - 1 of every phc_mlp_norm_bwd_fused launch stamp (100 MHz clock) */
int phc_debug_mlp_bwd_stamps(unsigned long long *dev_words) {
    g_mf_stamps = dev_words;
    return PHC_OK;
}
/* diagnostics only: 8 words per workgroup of every one-launch MLP kernel (phases 0 .. 7 of EVERY workgroup) */
int phc_debug_mlp_all_stamps(unsigned long long *dev_words) {
    g_mf_stamps_all = dev_words;
    return PHC_OK;
}
size_t phc_mlp_norm_bwd_fused_barrier_words(void) { return MF_BAR_WORDS; }

int phc_mlp_norm_bwd_fused_supported(int64_t N, int h) {
    if (h != MF_H || N < 1) return 0;
    const long long steps = (N + 15) / 16, groups = (steps + MF_MAXS - 1) / MF_MAXS;
    return groups <= mf_capacity() ? 1 : 0;
}

/* ... with the dx of the layer above's sum aggregation building the rows of dout inside the launch (desc.agg_src != NULL) */
int phc_mlp_norm_bwd_fused_agg_supported(int64_t N, int h) {
    if (h != MF_H || N < 1 || N >= (1LL << 31)) return 0;
    const long long steps = (N + 15) / 16;
    return xcd_grid((steps + MF_MAXS - 1) / MF_MAXS) <= mf_capacity() ? 1 : 0;
}

int phc_mlp_norm_bwd_fused(const phc_mlp_norm_bwd_desc *d, void *stream) {
    PHC_REQUIRE(d, PHC_ERR_INVALID_ARG, "phc_mlp_norm_bwd_fused: null descriptor");
    const bool agg = d->agg_src != nullptr;
    PHC_REQUIRE(agg ? phc_mlp_norm_bwd_fused_agg_supported(d->N, d->h) : phc_mlp_norm_bwd_fused_supported(d->N, d->h), PHC_ERR_UNSUPPORTED,
                "phc_mlp_norm_bwd_fused: N=%lld h=%d outside the one-launch form's range (h = 128, ceil(N / 32) co-resident workgroups)",
                (long long)d->N, d->h);
    PHC_REQUIRE((agg || d->dout) && d->y2 && d->z1 && d->save_o && d->save_i && d->H2Tk4 && d->H1Tk4 && d->dy2 && d->dz1 && d->ds && d->acc_o &&
                    d->acc_i && d->barrier,
                PHC_ERR_INVALID_ARG, "phc_mlp_norm_bwd_fused: null pointer");
    PHC_REQUIRE(d->act_o >= PHC_ACT_IDENTITY && d->act_o <= PHC_ACT_SELU && d->act_i >= PHC_ACT_IDENTITY && d->act_i <= PHC_ACT_SELU,
                PHC_ERR_INVALID_ARG, "phc_mlp_norm_bwd_fused: unknown activation");
    PHC_REQUIRE(((reinterpret_cast<uintptr_t>(d->dout) | reinterpret_cast<uintptr_t>(d->y2) | reinterpret_cast<uintptr_t>(d->z1) |
                  reinterpret_cast<uintptr_t>(d->H2Tk4) | reinterpret_cast<uintptr_t>(d->H1Tk4) | reinterpret_cast<uintptr_t>(d->dy2) |
                  reinterpret_cast<uintptr_t>(d->dz1) | reinterpret_cast<uintptr_t>(d->ds)) & 15) == 0 &&
                    ((reinterpret_cast<uintptr_t>(d->acc_o) | reinterpret_cast<uintptr_t>(d->acc_i)) & 7) == 0,
                PHC_ERR_UNSUPPORTED, "phc_mlp_norm_bwd_fused: 16-byte aligned rows / 8-byte aligned accumulators required");
    MlpBwdArgs a = {};
    a.dout = d->dout; a.y2 = d->y2; a.z1 = d->z1; a.save_o = d->save_o; a.save_i = d->save_i; a.H2Tk4 = d->H2Tk4; a.H1Tk4 = d->H1Tk4;
    a.dy2 = d->dy2; a.dz1 = d->dz1; a.ds = d->ds; a.dwb_o = d->dwb_o; a.dwb_i = d->dwb_i;
    a.acc_o = reinterpret_cast<unsigned long long *>(d->acc_o); a.acc_i = reinterpret_cast<unsigned long long *>(d->acc_i);
    a.n_valid = d->n_valid_dev; a.N = d->N; a.act_o = d->act_o; a.act_i = d->act_i;
    a.bar = d->barrier; a.err = err_word();
    a.stamps = g_mf_stamps; a.stamps_all = g_mf_stamps_all;
    const long long steps = (d->N + 15) / 16;
    long long groups = (steps + MF_MAXS - 1) / MF_MAXS;
    if (agg) {
        PHC_REQUIRE(d->agg_rowptr_t && d->agg_col_t && d->agg_ell_t && d->dx_out, PHC_ERR_INVALID_ARG,
                    "phc_mlp_norm_bwd_fused: the aggregation's dx needs rowptr_t, col_t, ell_t (phc_csr_build_pair_codes_resident_ell) and dx_out");
        PHC_REQUIRE(((reinterpret_cast<uintptr_t>(d->agg_src) | reinterpret_cast<uintptr_t>(d->agg_residual) | reinterpret_cast<uintptr_t>(d->dx_out) |
                      reinterpret_cast<uintptr_t>(d->agg_ell_t)) & 15) == 0,
                    PHC_ERR_UNSUPPORTED, "phc_mlp_norm_bwd_fused: 16-byte aligned agg_src / agg_residual / dx_out / ell_t required");
        a.gsrc = d->agg_src; a.residual = d->agg_residual; a.rowptr_t = d->agg_rowptr_t; a.col_t = d->agg_col_t; a.ell_t = d->agg_ell_t;
        a.dx_out = d->dx_out; a.add_self = d->agg_add_self ? 1 : 0;
        groups = xcd_grid(groups);                           // (a multiple of 8: xcd_contiguous_block)
    }
    const size_t lds = sizeof(float) * mf_lds_floats<MF_MAXS>();
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_PHM_GEMM_DX, s);
    if (agg)
        PHC_LAUNCH((mlp_bwd_fused_kernel<MF_MAXS, true>), dim3((unsigned)groups), dim3(MF_THREADS), lds, s, a);
    else
        PHC_LAUNCH((mlp_bwd_fused_kernel<MF_MAXS, false>), dim3((unsigned)groups), dim3(MF_THREADS), lds, s, a);
    return check_launch("mlp_norm_bwd_fused");
}

static long long mf_fwd_groups(long long N) {
    const long long steps = (N + 15) / 16;
    return xcd_grid((steps + MF_MAXS - 1) / MF_MAXS);        // (a multiple of 8: xcd_contiguous_block)
}
int phc_mlp_norm_fwd_fused_supported(int64_t N, int h) {
    if (h != MF_H || N < 1) return 0;
    return mf_fwd_groups(N) <= mf_fwd_capacity(false) ? 1 : 0;
}
/* ... with the layer's aggregation inside the launch (phc_mlp_norm_fwd_desc.codes != NULL): F bond features, vocab_rows = sum(vocab) */
int phc_mlp_norm_fwd_fused_agg_supported(int64_t N, int h, int n, int F, int vocab_rows) {
    if (h != MF_H || N < 1 || N >= (1LL << 31) || n < 1 || h % n || (h / n) % 4 || F < 1 || F > 4 || vocab_rows < 1 || vocab_rows > MF_AGG_MAX_R) return 0;
    return mf_fwd_groups(N) <= mf_fwd_capacity(true) ? 1 : 0;
}

int phc_mlp_norm_fwd_fused(const phc_mlp_norm_fwd_desc *d, void *stream) {
    PHC_REQUIRE(d, PHC_ERR_INVALID_ARG, "phc_mlp_norm_fwd_fused: null descriptor");
    const bool agg = d->codes != nullptr;
    MlpFwdArgs a = {};
    if (agg) {
        PHC_REQUIRE(d->vocab && d->F >= 1 && d->F <= 4 && d->n >= 1, PHC_ERR_INVALID_ARG, "phc_mlp_norm_fwd_fused: aggregation needs vocab, 1 <= F <= 4");
        a.F = d->F;
        a.off[0] = 0;
        for (int f = 0; f < d->F; ++f) {
            PHC_REQUIRE(d->vocab[f] >= 1, PHC_ERR_INVALID_ARG, "phc_mlp_norm_fwd_fused: vocab[%d] = %d", f, d->vocab[f]);
            a.off[f + 1] = a.off[f] + d->vocab[f];
        }
        a.R = a.off[d->F];
        PHC_REQUIRE(phc_mlp_norm_fwd_fused_agg_supported(d->N, d->h, d->n, d->F, a.R), PHC_ERR_UNSUPPORTED,
                    "phc_mlp_norm_fwd_fused: N=%lld h=%d n=%d F=%d sum(vocab)=%d outside the aggregating form's range", (long long)d->N, d->h, d->n,
                    d->F, a.R);
        PHC_REQUIRE(d->x && d->tables && d->rowptr && d->col && d->s_out, PHC_ERR_INVALID_ARG, "phc_mlp_norm_fwd_fused: null aggregation pointer");
        const bool normed = d->norm_acc != nullptr;
        PHC_REQUIRE(!normed || (d->save_stat_o && d->x_out && d->norm_act >= PHC_ACT_IDENTITY && d->norm_act <= PHC_ACT_SELU), PHC_ERR_INVALID_ARG,
                    "phc_mlp_norm_fwd_fused: a pending norm needs save_stat_o, x_out and an activation");
        PHC_REQUIRE(((reinterpret_cast<uintptr_t>(d->x) | reinterpret_cast<uintptr_t>(d->s_out) |
                      (normed ? (reinterpret_cast<uintptr_t>(d->x_out) | reinterpret_cast<uintptr_t>(d->skip)) : 0)) & 15) == 0 &&
                        (reinterpret_cast<uintptr_t>(d->norm_acc) & 7) == 0,
                    PHC_ERR_UNSUPPORTED, "phc_mlp_norm_fwd_fused: 16-byte aligned rows / 8-byte aligned accumulator required");
        if (normed) {
            int rc = fill_params("phc_mlp_norm_fwd_fused(outer)", &a.po, d->weight_o, d->bias_o, d->running_mean_o, d->running_var_o,
                                 d->num_batches_tracked_o, d->n_comp_o, d->h);
            if (rc) return rc;
            a.nacc = reinterpret_cast<const unsigned long long *>(d->norm_acc);
            a.mom_o = d->momentum_o; a.eps_o = d->eps_o; a.act_o = d->norm_act; a.skip = d->skip; a.stat_o = d->save_stat_o; a.x_out = d->x_out;
        }
        PHC_REQUIRE(d->ell && (reinterpret_cast<uintptr_t>(d->ell) & 15) == 0, PHC_ERR_INVALID_ARG,
                    "phc_mlp_norm_fwd_fused: the aggregation needs the row-major edge slots (ell, phc_csr_build_pair_codes_resident_ell)");
        a.ell = d->ell;
        PHC_REQUIRE(d->tables_host && d->n * d->F <= 16, PHC_ERR_INVALID_ARG,
                    "phc_mlp_norm_fwd_fused: the aggregation takes the table pointers by value (tables_host, n F <= 16)");
        for (int i = 0; i < 16; ++i) a.tp[i] = i < d->n * d->F ? d->tables_host[i] : d->tables_host[0];
        a.x = d->x; a.codes = d->codes; a.tables = d->tables; a.rowptr = d->rowptr; a.col = d->col; a.n_cmp = d->n; a.k = d->h / d->n;
        a.add_self = d->add_self ? 1 : 0; a.s_out = d->s_out;
    } else {
        PHC_REQUIRE(phc_mlp_norm_fwd_fused_supported(d->N, d->h), PHC_ERR_UNSUPPORTED,
                    "phc_mlp_norm_fwd_fused: N=%lld h=%d outside the one-launch form's range (h = 128, ceil(N / 32) co-resident workgroups)",
                    (long long)d->N, d->h);
        PHC_REQUIRE(d->s && (reinterpret_cast<uintptr_t>(d->s) & 15) == 0, PHC_ERR_INVALID_ARG, "phc_mlp_norm_fwd_fused: s");
    }
    PHC_REQUIRE(d->H1k4 && d->H2k4 && d->z1 && d->y2 && d->acc1 && d->acc2 && d->save_stat && d->barrier, PHC_ERR_INVALID_ARG,
                "phc_mlp_norm_fwd_fused: null pointer");
    PHC_REQUIRE(d->act >= PHC_ACT_IDENTITY && d->act <= PHC_ACT_SELU, PHC_ERR_INVALID_ARG, "phc_mlp_norm_fwd_fused: unknown activation %d", d->act);
    PHC_REQUIRE(((reinterpret_cast<uintptr_t>(d->H1k4) | reinterpret_cast<uintptr_t>(d->H2k4) | reinterpret_cast<uintptr_t>(d->z1) |
                  reinterpret_cast<uintptr_t>(d->y2) | reinterpret_cast<uintptr_t>(d->b1) | reinterpret_cast<uintptr_t>(d->b2)) & 15) == 0 &&
                    ((reinterpret_cast<uintptr_t>(d->acc1) | reinterpret_cast<uintptr_t>(d->acc2)) & 7) == 0,
                PHC_ERR_UNSUPPORTED, "phc_mlp_norm_fwd_fused: 16-byte aligned rows / biases, 8-byte aligned accumulators required");
    int rc = fill_params("phc_mlp_norm_fwd_fused", &a.p, d->weight, d->bias, d->running_mean, d->running_var, d->num_batches_tracked, d->n_comp, d->h);
    if (rc) return rc;
    a.s = d->s; a.H1k4 = d->H1k4; a.b1 = d->b1; a.H2k4 = d->H2k4; a.b2 = d->b2; a.z1 = d->z1; a.y2 = d->y2;
    a.acc1 = reinterpret_cast<unsigned long long *>(d->acc1); a.acc2 = reinterpret_cast<unsigned long long *>(d->acc2);
    a.momentum = d->momentum; a.eps = d->eps; a.save_i = d->save_stat; a.act_i = d->act; a.n_valid = d->n_valid_dev; a.N = d->N;
    a.bar = d->barrier; a.err = err_word(); a.stamps = g_mf_stamps; a.stamps_all = g_mf_stamps_all;
    const long long groups = mf_fwd_groups(d->N);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(agg ? K_PHM_GEMM_AG : K_PHM_GEMM_FWD, s);
    if (agg && d->softmax_beta) {
        PHC_REQUIRE(d->agg_out && d->lse && ((reinterpret_cast<uintptr_t>(d->agg_out) | reinterpret_cast<uintptr_t>(d->lse)) & 15) == 0,
                    PHC_ERR_INVALID_ARG, "phc_mlp_norm_fwd_fused: the softmax aggregation needs agg_out and lse (16-byte aligned)");
        a.beta = d->softmax_beta; a.agg_out = d->agg_out; a.lse = d->lse;
        PHC_LAUNCH((mlp_fwd_fused_kernel<MF_MAXS, 2>), dim3((unsigned)groups), dim3(MF_THREADS), mf_fwd_lds_bytes(true), s, a);
    } else if (agg) {
        PHC_LAUNCH((mlp_fwd_fused_kernel<MF_MAXS, 1>), dim3((unsigned)groups), dim3(MF_THREADS), mf_fwd_lds_bytes(true), s, a);
    } else {
        PHC_REQUIRE(!d->softmax_beta, PHC_ERR_INVALID_ARG, "phc_mlp_norm_fwd_fused: softmax_beta without an aggregation (codes)");
        PHC_LAUNCH((mlp_fwd_fused_kernel<MF_MAXS, 0>), dim3((unsigned)groups), dim3(MF_THREADS), mf_fwd_lds_bytes(false), s, a);
    }
    return check_launch("mlp_norm_fwd_fused");
}

}  // extern "C"
