// Device bodies and job tables of the PHM parameter-gradient kernels (dH = x^T.dy as fixed-order split-K slabs).
// Shared by phm_linear.hip (phc_phm_linear_bwd, phc_phm_param_grads) and encoder.hip (phc_step_tail: the same
// workgroup bodies launched side by side with the table-gradient workgroups).
#pragma once
#include "phc_common.h"

namespace phc {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

// ------------------------------------------------------------------------------------------
// dH = x^T . dy  as split-K slabs.  Workgroup (tile, split) accumulates the 64 x 64 tile
// (i0.., j0..) of dH over its row range and writes it to slab[split].  Workgroups with
// tile_i == 0 also produce the column sums of dy (db) for their rows.
// ------------------------------------------------------------------------------------------
constexpr int DH_T = 64;            // tile edge
constexpr int DH_RB = 64;           // rows staged per step
constexpr int DH_STRIDE = DH_T;     // lanes read consecutive columns: conflict-free without padding

__device__ __forceinline__ void dH_partial_body(const float *__restrict__ X, const float *__restrict__ DY,
                                                float *__restrict__ slabs, long long N, int in, int out, int tiles_j,
                                                long long rows_per_split, int vec4, int tile, long long split,
                                                float *__restrict__ smem, const float *__restrict__ xf_stat = nullptr,
                                                int xf_act = 0, long long xf_count = 0) {
    // xf_stat != NULL: the X operand is a batch-norm that was never applied in HBM (phc_gemm_prebuilt_xf):
    // x := act(x * scale + shift) per column, rows >= xf_count read as 0, evaluated while the tile is staged
    float *xs = smem;                                    // [DH_RB][DH_STRIDE]
    float *ds = smem + DH_RB * DH_STRIDE;                // [DH_RB][DH_STRIDE]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int r = lane & 31, hh = lane >> 5;
    const int ti = tile / tiles_j, tj = tile - ti * tiles_j;
    const int i0 = ti * DH_T, j0 = tj * DH_T;
    const long long r_begin = split * rows_per_split;
    const long long r_end = min(N, r_begin + rows_per_split);
    const size_t slab_elems = (size_t)in * out + out;
    float *slab = slabs + (size_t)split * slab_elems;

#ifdef PHC_EXP_DH_STAGGER
    {   // experiment: offset the phases of the workgroups that share a CU (they start together and otherwise stay in step)
        const unsigned bid = blockIdx.x;
        const unsigned q = ((bid >> 8) + (bid >> 3)) & 3u;
        for (unsigned t = 0; t < q; ++t) __builtin_amdgcn_s_sleep(PHC_EXP_DH_STAGGER);
    }
#endif
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    // column sums of dy (db): in tile_i == 0 workgroups wavefront q adds rows q*16 .. q*16+15 of every staged block for
    // column (tid & 63); the four partials meet in fixed order after the loop
    float colsum = 0.f;
    f32x4 xf_sc = {1.f, 1.f, 1.f, 1.f}, xf_sh = {0.f, 0.f, 0.f, 0.f};
    if (xf_stat && vec4) {                               // this thread always stages float4 column (tid & 15) of the X tile
        const int ci = min(i0 + (tid & 15) * 4, in - 4);
        xf_sc = *reinterpret_cast<const f32x4 *>(xf_stat + 2 * (size_t)in + ci);
        xf_sh = *reinterpret_cast<const f32x4 *>(xf_stat + 3 * (size_t)in + ci);
    }

    // vec4 path: the 64 x 64 block of each operand is 4 float4 per thread; the NEXT block's eight loads are issued
    // (clamped addresses, no branches) before the current block's MFMAs, so their latency hides behind the matrix work
    f32x4 xv[4], dv[4];
    const int c4 = tid & 15;
    const int ci4 = min(i0 + c4 * 4, in - 4), cj4 = min(j0 + c4 * 4, out - 4);
    const bool x_col_ok = i0 + c4 * 4 < in, d_col_ok = j0 + c4 * 4 < out;
    if (vec4) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const long long grow = min(r_begin + (tid >> 4) + 16 * t, r_end - 1);
            xv[t] = *reinterpret_cast<const f32x4 *>(X + (size_t)grow * in + ci4);
            dv[t] = *reinterpret_cast<const f32x4 *>(DY + (size_t)grow * out + cj4);
        }
    }

    for (long long rb = r_begin; rb < r_end; rb += DH_RB) {
        __syncthreads();
        if (vec4) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int row = (tid >> 4) + 16 * t;
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                const bool rok = rb + row < r_end;
                if (xf_stat) {
                    const bool live = rb + row < xf_count;
                    if (xf_act == PHC_ACT_RELU) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) xv[t][q] = live ? fmaxf(fmaf(xv[t][q], xf_sc[q], xf_sh[q]), 0.f) : 0.f;
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) xv[t][q] = live ? act_fwd_rt(xf_act, fmaf(xv[t][q], xf_sc[q], xf_sh[q])) : 0.f;
                    }
                }
                *reinterpret_cast<f32x4 *>(xs + row * DH_STRIDE + c4 * 4) = (rok && x_col_ok) ? xv[t] : z;
                *reinterpret_cast<f32x4 *>(ds + row * DH_STRIDE + c4 * 4) = (rok && d_col_ok) ? dv[t] : z;
            }
        } else {
            for (int e = tid; e < DH_RB * DH_T; e += 256) {
                const int row = e / DH_T, c = e - row * DH_T;
                float xe = 0.f, de = 0.f;
                if (rb + row < r_end) {
                    if (i0 + c < in) {
                        xe = X[(size_t)(rb + row) * in + i0 + c];
                        if (xf_stat)
                            xe = rb + row < xf_count
                                     ? act_fwd_rt(xf_act, fmaf(xe, xf_stat[2 * (size_t)in + i0 + c], xf_stat[3 * (size_t)in + i0 + c]))
                                     : 0.f;
                    }
                    if (j0 + c < out) de = DY[(size_t)(rb + row) * out + j0 + c];
                }
                xs[row * DH_STRIDE + c] = xe;
                ds[row * DH_STRIDE + c] = de;
            }
        }
        __syncthreads();
#ifdef PHC_EXP_DH_NO_LOAD
        if (tid == 100000)
#endif
        if (vec4) {                                      // prefetch the next block (the last iteration re-reads row r_end-1)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const long long grow = min(rb + DH_RB + (tid >> 4) + 16 * t, r_end - 1);
                xv[t] = *reinterpret_cast<const f32x4 *>(X + (size_t)grow * in + ci4);
                dv[t] = *reinterpret_cast<const f32x4 *>(DY + (size_t)grow * out + cj4);
            }
        }
        // A operand = x^T: lane (r, hh) supplies x[row rr+hh][i0 + wi*32 + r]; B = dy[row rr+hh][j0 + wj*32 + r]
        const float *xa = xs + hh * DH_STRIDE + wi * 32 + r;
        const float *db_ = ds + hh * DH_STRIDE + wj * 32 + r;
#ifdef PHC_EXP_DH_NO_MFMA
        if (tid == 100000)
#endif
#pragma unroll 8
        for (int rr = 0; rr < DH_RB; rr += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[rr * DH_STRIDE], db_[rr * DH_STRIDE], acc, 0, 0, 0);
        if (ti == 0) {
            const float *dc = ds + (wave * 16) * DH_STRIDE + lane;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) colsum += dc[rr * DH_STRIDE];
        }
    }

    const int col = j0 + wj * 32 + r;
    if (col < out) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = i0 + wi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
            if (row < in) slab[(size_t)row * out + col] = acc[reg];
        }
    }
    if (ti == 0) {                                       // uniform per workgroup
        __syncthreads();
        xs[wave * DH_T + lane] = colsum;
        __syncthreads();
        if (tid < DH_T && j0 + tid < out)
            slab[(size_t)in * out + j0 + tid] = (xs[tid] + xs[DH_T + tid]) + (xs[2 * DH_T + tid] + xs[3 * DH_T + tid]);
    }
}

// 128 x 128 tile variant (in, out >= 128): each wavefront owns a 64 x 64 quadrant = 2 x 2 MFMA tiles, rows staged
// 32 at a time.  Twice the MFMAs per staged byte and per LDS read of the 64 x 64 form, and when in = out = 128 the
// tile is the whole matrix, so x and dy are read exactly once (the 64 x 64 form reads each twice: its 258 MB per
// batched launch at config 2 made it HBM/L2-bound at 56 % of the MFMA peak).
constexpr int DH_T2 = 128;
constexpr int DH_RB2 = 32;

__device__ __forceinline__ void dH_partial_body128(const float *__restrict__ X, const float *__restrict__ DY,
                                                   float *__restrict__ slabs, long long N, int in, int out, int tiles_j,
                                                   long long rows_per_split, int vec4, int tile, long long split,
                                                   float *__restrict__ smem, const float *__restrict__ xf_stat = nullptr,
                                                   int xf_act = 0, long long xf_count = 0) {
    float *xs = smem;                                    // [DH_RB2][DH_T2]
    float *ds = smem + DH_RB2 * DH_T2;                   // [DH_RB2][DH_T2]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int r = lane & 31, hh = lane >> 5;
    const int ti = tile / tiles_j, tj = tile - ti * tiles_j;
    const int i0 = ti * DH_T2, j0 = tj * DH_T2;
    const long long r_begin = split * rows_per_split;
    const long long r_end = min(N, r_begin + rows_per_split);
    const size_t slab_elems = (size_t)in * out + out;
    float *slab = slabs + (size_t)split * slab_elems;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    float colsum = 0.f;                                  // threads 0..127 of tile_i == 0 workgroups
    f32x4 xf_sc = {1.f, 1.f, 1.f, 1.f}, xf_sh = {0.f, 0.f, 0.f, 0.f};
    if (xf_stat && vec4) {
        const int ci = min(i0 + (tid & 31) * 4, in - 4);
        xf_sc = *reinterpret_cast<const f32x4 *>(xf_stat + 2 * (size_t)in + ci);
        xf_sh = *reinterpret_cast<const f32x4 *>(xf_stat + 3 * (size_t)in + ci);
    }

    for (long long rb = r_begin; rb < r_end; rb += DH_RB2) {
        __syncthreads();
        if (vec4) {
            // 32 rows x 32 float4 per operand = 4 float4 per thread and operand, all issued before the LDS stores
            f32x4 xv[4], dv[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int e = tid + 256 * t;
                const int row = e >> 5, c4 = e & 31;
                const long long grow = min(rb + row, r_end - 1);
                const int ci = min(i0 + c4 * 4, in - 4), cj = min(j0 + c4 * 4, out - 4);
                xv[t] = *reinterpret_cast<const f32x4 *>(X + (size_t)grow * in + ci);
                dv[t] = *reinterpret_cast<const f32x4 *>(DY + (size_t)grow * out + cj);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int e = tid + 256 * t;
                const int row = e >> 5, c4 = e & 31;
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                const bool rok = rb + row < r_end;
                if (xf_stat) {
                    const f32x4 sc = xf_sc, sh = xf_sh;      // c4 = tid & 31 for every element of this thread
                    const bool live = rb + row < xf_count;
                    if (xf_act == PHC_ACT_RELU) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) xv[t][q] = live ? fmaxf(fmaf(xv[t][q], sc[q], sh[q]), 0.f) : 0.f;
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) xv[t][q] = live ? act_fwd_rt(xf_act, fmaf(xv[t][q], sc[q], sh[q])) : 0.f;
                    }
                }
                *reinterpret_cast<f32x4 *>(xs + row * DH_T2 + c4 * 4) = (rok && i0 + c4 * 4 < in) ? xv[t] : z;
                *reinterpret_cast<f32x4 *>(ds + row * DH_T2 + c4 * 4) = (rok && j0 + c4 * 4 < out) ? dv[t] : z;
            }
        } else {
            for (int e = tid; e < DH_RB2 * DH_T2; e += 256) {
                const int row = e / DH_T2, c = e - row * DH_T2;
                float xv = 0.f, dv = 0.f;
                if (rb + row < r_end) {
                    if (i0 + c < in) {
                        xv = X[(size_t)(rb + row) * in + i0 + c];
                        if (xf_stat)
                            xv = rb + row < xf_count
                                     ? act_fwd_rt(xf_act, fmaf(xv, xf_stat[2 * (size_t)in + i0 + c], xf_stat[3 * (size_t)in + i0 + c]))
                                     : 0.f;
                    }
                    if (j0 + c < out) dv = DY[(size_t)(rb + row) * out + j0 + c];
                }
                xs[row * DH_T2 + c] = xv;
                ds[row * DH_T2 + c] = dv;
            }
        }
        __syncthreads();
        const float *xa = xs + hh * DH_T2 + wi * 64 + r;
        const float *db_ = ds + hh * DH_T2 + wj * 64 + r;
#pragma unroll 4
        for (int rr = 0; rr < DH_RB2; rr += 2) {
            const float a0 = xa[rr * DH_T2], a1 = xa[rr * DH_T2 + 32];
            const float b0 = db_[rr * DH_T2], b1 = db_[rr * DH_T2 + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (ti == 0 && tid < DH_T2) {
#pragma unroll 8
            for (int rr = 0; rr < DH_RB2; ++rr) colsum += ds[rr * DH_T2 + tid];
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = j0 + wj * 64 + b * 32 + r;
            if (col < out) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int row = i0 + wi * 64 + a * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
                    if (row < in) slab[(size_t)row * out + col] = acc[a][b][reg];
                }
            }
        }
    if (ti == 0 && tid < DH_T2 && j0 + tid < out) slab[(size_t)in * out + j0 + tid] = colsum;
}

// ------------------------------------------------------------------------------------------
// Register-resident form (in, out multiples of 128, 16-byte aligned operands): every WAVEFRONT owns a whole 128 x 128
// tile of dH for its own row range and feeds the MFMAs straight from global memory — no LDS staging, no barrier in the
// main loop.  It works because the contraction index of dH = x^T . dy is the ROW: in v_mfma_f32_32x32x2 lane (r, hh)
// supplies A[m = r][k = hh] and B[k = hh][n = r], so with row pair (rr, rr+1) <-> k = (0, 1) the half-wave hh reads row
// rr + hh, and lane r reads the float4 x[row][4r..4r+3] / dy[row][4r..4r+3] (one fully coalesced 512-byte row per half-wave
// and operand).  Component q of the x load is the A operand of "M block q" = columns {4m + q}, component p of the dy load
// the B operand of "N block p" = columns {4n + p}: 16 MFMAs per row pair from two 16-byte loads per lane, and
// acc[q][p][reg] = dH[i0 + 4m + q][j0 + 4r + p] with m = (reg & 3) + 8 (reg >> 2) + 4 hh — the four p of one (q, reg) are
// one float4 of the slab.  256 accumulator registers per lane (one wavefront per SIMD, 512-register budget); the loads of
// the next DH_PF row pairs are in flight while the current ones are consumed.
// A workgroup = 4 wavefronts on 4 consecutive row quarters of one split; they meet once, at the end, through LDS (fixed
// summation order 0, 1, 2, 3) and write one slab per WORKGROUP.
// ------------------------------------------------------------------------------------------
constexpr int DH_D = 128;                               // tile edge of the register-resident form
// columns per lane of the 256-register form (t128 == 3) for an output width: 128 x 64 tiles, or 128 x 32 ones when out is an
// odd multiple of 32; any other EVEN width takes 128 x 64 tiles with a partial last one (two columns of dy per lane: 8-byte loads), an
// ODD width 128 x 32 tiles (one column per lane)
__host__ __device__ inline int dh3_np(int out) { return out % 64 == 0 ? 2 : ((out % 32 == 0 || out % 2 != 0) ? 1 : 2); }
#ifndef PHC_DH_PF
#define PHC_DH_PF 4
#endif
constexpr int DH_PF = PHC_DH_PF;                        // row pairs per load batch (two batches alternate)
constexpr int DH_DIRECT_LDS_BYTES = 4 * 16 * 64 * 16;   // exchange slots: [wave][reg][lane] float4 = 64 KB

using f32x2 = __attribute__((ext_vector_type(2))) float;
// NP = "N blocks" of a register-resident tile: 4 -> 128 columns of dy per wavefront (float4 per lane), 2 -> 64 (float2), 1 -> 32
template <int NP> struct DhVec;
template <> struct DhVec<4> { using T = f32x4; };
template <> struct DhVec<2> { using T = f32x2; };
using f32x1 = __attribute__((ext_vector_type(1))) float;
template <> struct DhVec<1> { using T = f32x1; };

// One batch of DH_PF row pairs.  The x operand is always x := max(x * scale + shift, floor) per column: a plain operand is
// (1, 0, -inf) — exact —, an unapplied batch-norm its (scale, shift) with floor -inf / 0 for identity / relu (the other
// activations stay on the LDS forms).  MASKED = false: every row of the batch is a live row of the wavefront's range — no
// selects; MASKED = true: rows at or past `lim` contribute zero.
// (fp32 MFMAs run on the vector ALU's own multipliers: every VALU instruction between them costs its full issue time, nothing
// hides in their shadow — measured 36.6 ns per MFMA with 2 VALU instructions per MFMA against 30 with none — so the interior
// loop carries only the operand transform and the column sums.)
template <bool MASKED, int NP>
__device__ __forceinline__ void dH_direct_pairs(f32x16 (&acc)[4][NP], typename DhVec<NP>::T &colsum, const f32x4 (&XV)[DH_PF],
                                                const typename DhVec<NP>::T (&DV)[DH_PF],
                                                long long row0, long long lim, int hh, const f32x4 &xf_sc, const f32x4 &xf_sh,
                                                float floor_) {
#pragma unroll
    for (int t = 0; t < DH_PF; ++t) {
        f32x4 xv = XV[t];
        typename DhVec<NP>::T dv = DV[t];
        bool live = true;
        if constexpr (MASKED) live = row0 + 2 * t + hh < lim;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float a = fmaxf(fmaf(xv[q], xf_sc[q], xf_sh[q]), floor_);
            xv[q] = MASKED ? (live ? a : 0.f) : a;
        }
        if constexpr (MASKED) {
#pragma unroll
            for (int q = 0; q < NP; ++q) dv[q] = live ? dv[q] : 0.f;
        }
        colsum += dv;
#ifdef PHC_EXP_DIRECT_NO_MFMA
        colsum[0] += xv[0] + xv[1] + xv[2] + xv[3];
#else
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int p2 = 0; p2 < NP; ++p2) acc[q][p2] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[q], dv[p2], acc[q][p2], 0, 0, 0);
#endif
    }
}

// Loads of one batch.  EXACT = false: the pair base (wave-uniform: scalar registers) is clamped to the last whole pair of the
// range, the lane part is a constant byte offset — no per-load vector arithmetic; a pair past the range reads a neighbour pair,
// which is fine for prefetches whose data is discarded.  EXACT = true: every lane's own row clamped to `last_row` (the masked tail).
template <bool EXACT, int NP>
__device__ __forceinline__ void dH_direct_load(f32x4 (&XV)[DH_PF], typename DhVec<NP>::T (&DV)[DH_PF], const float *__restrict__ X,
                                               const float *__restrict__ DY, int in, int out, unsigned xoff, unsigned doff,
                                               unsigned xrow, unsigned drow, long long row0, long long last_row) {
#pragma unroll
    for (int t = 0; t < DH_PF; ++t) {
#ifdef PHC_EXP_DIRECT_NO_LOAD
        const long long g = 2 * t;                        // experiment: the same few (cached) rows every time
#else
        const long long g = row0 + 2 * t;
#endif
        if constexpr (EXACT) {
            const long long gc = min(g, last_row);
            const bool up = g < last_row;                // the upper half-wave's row exists (else it re-reads row gc: masked)
            XV[t] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(X + (size_t)gc * in) + (xoff + (up ? xrow : 0u)));
            DV[t] = *reinterpret_cast<const typename DhVec<NP>::T *>(reinterpret_cast<const char *>(DY + (size_t)gc * out) + (doff + (up ? drow : 0u)));
        } else {
            // X / DY point at the wavefront's first row here and row0 counts from it: 32-bit scalar byte offsets (the planner
            // keeps a wavefront's rows below 2^31 bytes), clamped to the last whole pair of the range
            const unsigned pr = min((unsigned)g, (unsigned)last_row - 1u);
            XV[t] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(X) + (pr * ((unsigned)in * 4u) + (xoff + xrow)));
            DV[t] = *reinterpret_cast<const typename DhVec<NP>::T *>(reinterpret_cast<const char *>(DY) + (pr * ((unsigned)out * 4u) + (doff + drow)));
        }
    }
}

// Rows [w_begin, w_end) of one wavefront; rows at or past xf_count are padding of the x operand (read as zero).
//   1. the LAST 4 * DH_PF-row iteration of the live rows, masked, straight-line, FIRST (it only initialises the accumulators:
//      placed after the loop, or as a second body inside it, it made the register allocator shuffle or spill accumulators);
//   2. the interior iterations, select-free: two register sets alternate, the set being loaded is consumed one half-iteration
//      later.  The scheduling fences keep each batch of loads where it is written — issued BEFORE the MFMAs of the other set
//      (left alone, the scheduler sinks every load to just in front of its first use and the wavefront, the only one on its
//      SIMD, waits out the full memory latency once per row pair);
//   3. padding rows: only the column sums of dy.
template <int NP>
__device__ __forceinline__ void dH_direct_rows(f32x16 (&acc)[4][NP], typename DhVec<NP>::T &colsum, const float *__restrict__ X,
                                               const float *__restrict__ DY, int in, int out, unsigned xoff, unsigned doff,
                                               unsigned xrow, unsigned drow, long long w_begin, long long w_end, long long last,
                                               int hh, const f32x4 &xf_sc, const f32x4 &xf_sh, float floor_, long long xf_count) {
    constexpr long long IT = 4 * DH_PF;
    f32x4 xa[DH_PF], xb[DH_PF];
    typename DhVec<NP>::T da[DH_PF], db[DH_PF];
    const long long lim = max(min(w_end, xf_count), w_begin);
    const long long n_it = (lim - w_begin + IT - 1) / IT;
    const long long tail0 = w_begin + (n_it > 0 ? n_it - 1 : 0) * IT;
    dH_direct_load<true, NP>(xa, da, X, DY, in, out, xoff, doff, xrow, drow, tail0, last);
    dH_direct_load<true, NP>(xb, db, X, DY, in, out, xoff, doff, xrow, drow, tail0 + 2 * DH_PF, last);
    __builtin_amdgcn_sched_barrier(0);
    dH_direct_pairs<true, NP>(acc, colsum, xa, da, tail0, lim, hh, xf_sc, xf_sh, floor_);
    dH_direct_pairs<true, NP>(acc, colsum, xb, db, tail0 + 2 * DH_PF, lim, hh, xf_sc, xf_sh, floor_);
    __builtin_amdgcn_sched_barrier(0);
    if (tail0 > w_begin) {
        // interior: addresses relative to the wavefront's first row (32-bit arithmetic, scalar)
        const float *Xw = X + (size_t)w_begin * in, *DYw = DY + (size_t)w_begin * out;
        const long long lastw = last - w_begin;          // >= 1: the interior exists only when the range has > 4 * DH_PF rows
        const int n_rows = (int)(tail0 - w_begin);
        dH_direct_load<false, NP>(xa, da, Xw, DYw, in, out, xoff, doff, xrow, drow, 0, lastw);
        for (int row = 0; row < n_rows; row += (int)IT) {
            dH_direct_load<false, NP>(xb, db, Xw, DYw, in, out, xoff, doff, xrow, drow, row + 2 * DH_PF, lastw);
            __builtin_amdgcn_sched_barrier(0);
            dH_direct_pairs<false, NP>(acc, colsum, xa, da, row, lim, hh, xf_sc, xf_sh, floor_);
            __builtin_amdgcn_sched_barrier(0);
            dH_direct_load<false, NP>(xa, da, Xw, DYw, in, out, xoff, doff, xrow, drow, row + (int)IT, lastw);
            __builtin_amdgcn_sched_barrier(0);
            dH_direct_pairs<false, NP>(acc, colsum, xb, db, row + 2 * DH_PF, lim, hh, xf_sc, xf_sh, floor_);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    for (long long g = lim + hh; g < w_end; g += 2)
        colsum += *reinterpret_cast<const typename DhVec<NP>::T *>(reinterpret_cast<const char *>(DY + (size_t)g * out) + doff);
}

#ifdef PHC_KTIMING
// experiment build only: wave 0 of the first and the last workgroup stamp the phases of the launch (100 MHz wall clock)
static __device__ unsigned long long g_dh_ts[64];
#define DH_STAMP(i)                                                                                    \
    do {                                                                                               \
        if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1))                      \
            g_dh_ts[(blockIdx.x == 0 ? 0 : 32) + (i)] = wall_clock64();                                \
    } while (0)
#else
#define DH_STAMP(i) do { } while (0)
#endif

template <int NP>
__device__ __forceinline__ void dH_direct_body(const float *__restrict__ X, const float *__restrict__ DY, float *__restrict__ slabs,
                                               long long N, int in, int out, int tiles_j, long long rows_per_split, int tile,
                                               long long split, float *__restrict__ smem, const float *__restrict__ xf_stat,
                                               int xf_act, long long xf_count) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // scalar: the row range of the wavefront is uniform
    const int r = lane & 31, hh = lane >> 5;
    const int ti = tile / tiles_j, tj = tile - ti * tiles_j;
    const int i0 = ti * DH_D, j0 = tj * (32 * NP);
    using DVT = typename DhVec<NP>::T;
    const long long r_begin = split * rows_per_split;
    const long long r_end = min(N, r_begin + rows_per_split);
    // the rows of THIS split in four wavefront quarters of whole 16-row iterations (a full split: rows_per_split / 4, as ever; the
    // shorter last split of a job is dealt evenly too — its workgroup may go on to the last split of another tile: GradJobDev.pair)
    const long long rq = ((max(r_end - r_begin, 0LL) + 63) >> 6) << 4;
    const long long w_begin = min(r_begin + wave * rq, r_end), w_end = min(w_begin + rq, r_end);
    const size_t slab_elems = (size_t)in * out + out;
    float *slab = slabs + (size_t)split * slab_elems;

    DH_STAMP(0);
    f32x16 acc[4][NP];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int p2 = 0; p2 < NP; ++p2)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[q][p2][i] = 0.f;
    DVT colsum;
#pragma unroll
    for (int p2 = 0; p2 < NP; ++p2) colsum[p2] = 0.f;
    f32x4 xf_sc = {1.f, 1.f, 1.f, 1.f}, xf_sh = {0.f, 0.f, 0.f, 0.f};
    float floor_ = -__builtin_inff();
    // partial edge tiles (in / out not multiples of the tile: the published 180- and 200-wide models): a lane whose columns lie
    // past the matrix reads the tile's first columns instead (in bounds) — its accumulator rows / columns are never stored
    const bool x_live = i0 + 4 * r < in, d_live = j0 + NP * r < out;
    const int xc = x_live ? i0 + 4 * r : i0, dc = d_live ? j0 + NP * r : j0;
    if (xf_stat) {
        xf_sc = *reinterpret_cast<const f32x4 *>(xf_stat + 2 * (size_t)in + xc);
        xf_sh = *reinterpret_cast<const f32x4 *>(xf_stat + 3 * (size_t)in + xc);
        if (xf_act == PHC_ACT_RELU) floor_ = 0.f;
    } else {
        xf_count = N;                                    // a plain operand has no padding rows of its own
    }
    // lane part of every address: this lane's four columns; the upper half-wave reads the odd row of each pair
    const unsigned xoff = (unsigned)xc * 4u, doff = (unsigned)dc * 4u;
    const unsigned xrow = hh ? (unsigned)in * 4u : 0u, drow = hh ? (unsigned)out * 4u : 0u;
    const long long last = max(r_end - 1, 0LL);          // clamped rows: reads past the range stay in bounds

    DH_STAMP(1);
    if (w_begin < w_end)
        dH_direct_rows<NP>(acc, colsum, X, DY, in, out, xoff, doff, xrow, drow, w_begin, w_end, last, hh, xf_sc, xf_sh, floor_, xf_count);

    DH_STAMP(2);
    // ---- the four wavefronts meet.  One phase per "M block" q: every wavefront parks the block in LDS ([wave][reg][lane] float4 =
    // 64 KB, which leaves room for other kernels' workgroups on the CU), then wavefront w sums registers 4w .. 4w+3 of all four
    // wavefronts in the fixed order 0, 1, 2, 3 and writes those rows of the slab.  Nothing here indexes the accumulators by the
    // wavefront number (a rotation that did cost 10 us of selects and spills).
    DVT *slots = reinterpret_cast<DVT *>(smem);          // [wave][reg][lane] of NP floats: 16 KB * NP
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (q) __syncthreads();
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            DVT v;
#pragma unroll
            for (int p2 = 0; p2 < NP; ++p2) v[p2] = acc[q][p2][reg];
            slots[(wave * 16 + reg) * 64 + lane] = v;
        }
        __syncthreads();
        DVT part[4][4];
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2)
#pragma unroll
            for (int e = 0; e < 4; ++e) part[w2][e] = slots[(w2 * 16 + 4 * wave + e) * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const DVT v = ((part[0][e] + part[1][e]) + part[2][e]) + part[3][e];
            const int m = e + 8 * wave + 4 * hh;         // reg = 4 wave + e: (reg & 3) + 8 (reg >> 2) + 4 hh
            if (d_live && i0 + 4 * m + q < in) *reinterpret_cast<DVT *>(slab + (size_t)(i0 + 4 * m + q) * out + j0 + NP * r) = v;
        }
    }
    DH_STAMP(3);
    if (ti == 0) {                                       // db: column sums of dy over the split (uniform per workgroup)
        __syncthreads();
        slots[tid] = colsum;                             // [wave][hh][r]
        __syncthreads();
        if (tid < 32) {
            DVT v = slots[tid];
#pragma unroll
            for (int e = 1; e < 8; ++e) v += slots[e * 32 + tid];           // fixed order: (wave 0: hh 0, hh 1), wave 1, ...
            if (j0 + NP * tid < out) *reinterpret_cast<DVT *>(slab + (size_t)in * out + j0 + NP * tid) = v;
        }
    }
    DH_STAMP(4);
}

// dH[idx] (and the db tail) = sum over splits.  64 elements x 4 split lanes per workgroup; every lane
// walks its splits in order and the 4 lane sums are added in order: one fixed summation tree.
__device__ __forceinline__ void dH_reduce_body(const float *__restrict__ slabs, float *__restrict__ dH, size_t slab_elems, int splits,
                                               unsigned blk) {
    __shared__ float part[4][64];
    const int e = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const size_t idx = (size_t)blk * 64 + e;
    float s = 0.f;
    if (idx < slab_elems) {
        // eight of this lane's slabs per round trip (clamped, selected; same ascending order as a plain loop)
        for (int k0 = sl; k0 < splits; k0 += 32) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = slabs[(size_t)min(k0 + 4 * q, splits - 1) * slab_elems + idx];
#pragma unroll
            for (int q = 0; q < 8; ++q) s += k0 + 4 * q < splits ? v[q] : 0.f;
        }
    }
    part[sl][e] = s;
    __syncthreads();
    if (sl == 0 && idx < slab_elems) dH[idx] = ((part[0][e] + part[1][e]) + part[2][e]) + part[3][e];
}

// ------------------------------------------------------------------------------------------
// Batched parameter gradients (phc_phm_param_grads): the dW / dA / db chains of MANY layers in
// three launches.  Parameter gradients are off the critical path of the backward pass (only the
// optimizer reads them), and one layer's chain cannot fill 256 CUs at molecule scale; batching
// the layers fills the chip and needs ~30 split-K slabs per layer instead of ~130.
// The job table travels by value in the kernarg segment; workgroup -> job by a scalar scan.
// ------------------------------------------------------------------------------------------
constexpr int GRAD_JOBS_PER_LAUNCH = 16;

struct GradJobDev {
    const float *x, *dy, *A, *W;
    float *dW, *dA, *db;
    float *slabs, *dH;
    long long N, rows_per_split;
    int in, out, n, kin, kout;
    int tiles_j, tiles, splits, vec4;
    int dA_blocks, t128, dw_transposed;                      // t128: tile form 0 = 64 x 64 (LDS), 1 = 128 x 128 (LDS), 2 / 3 = register-resident
                                                             // 128 x 128 (own launch) / 128 x 64 (shares the 64 x 64 form's launch)
    const float *xf_stat;                                    // X operand = act(x * scale + shift) (see dH_partial_body)
    const int *xf_n_valid;
    int xf_act;
    unsigned blk0_partial, blk0_reduce, blk0_contract;      // first workgroup of this job in each launch
    int pair;                                                // form 3, > 1: the last (shorter) splits of `pair` tiles share ONE workgroup
    const float *reg_add;                                    // != NULL: added to dW as it is written (the weight regulariser's gradient)
};
struct GradBatch {
    int num;
    GradJobDev j[GRAD_JOBS_PER_LAUNCH];
};

// XCD-aware (tile, split) of the `local`-th workgroup of a job whose workgroup has the launch-wide id `gid`: workgroups go to the
// 8 XCDs round-robin by block id, and the tiles of one split read the same x / dy rows.  The job's workgroups that land on one XCD
// (ids congruent mod 8) take a CONTIGUOUS range of the split-major order L = split * tiles + tile, so a split's tiles run on one
// XCD (two where a range boundary cuts it) and its rows come from memory once, from that XCD's L2 afterwards — whatever the
// number of splits (the first form of this map needed whole runs of 8 splits: six splits of 256-wide layers at 23 k rows had
// every tile of a split on a different XCD and fetched the rows eight times).  Which workgroup computes a (tile, split) never
// changes what is computed: results are bit-identical with the map on or off.
__device__ __forceinline__ void xcd_tile_split(unsigned gid, unsigned local, unsigned tiles, unsigned splits, int &tile, long long &split) {
    const unsigned C = tiles * splits;
    unsigned L = local;
    if (PHC_XCD_MAP && tiles > 1 && C >= 16u) {
        const unsigned base = gid - local, x = gid & 7u;
        unsigned before = 0;
#pragma unroll
        for (unsigned xp = 0; xp < 8u; ++xp) {
            const unsigned f = (xp - base) & 7u;                 // the job's first workgroup on XCD xp
            const unsigned cnt = f < C ? (C - f + 7u) >> 3 : 0u;
            before += xp < x ? cnt : 0u;
        }
        L = before + ((local - ((x - base) & 7u)) >> 3);
    }
    tile = (int)(L % tiles);
    split = L / tiles;
}

// ---- the explicit weight regulariser inside the step tail (regularization.py:15-23, train_hiv.py:181-196) ----
// reg = coef * sum over PHM layers of mean_{kappa, pi} || W[:, kappa, pi] ||_p depends on the weights only.  Its workgroups ride in
// the tail's REDUCE launch (a workgroup = 256 (kappa, pi) cells of one layer): d reg / d W goes to a side buffer, the
// workgroup's share of the value to `shares[block]`; the tail's LAST launch adds the side buffer into dW as it writes it
// (GradJobDev.reg_add, GateGradArgs.reg_add) and one extra workgroup sums the shares in block order — plain stores and loads
// across a kernel boundary, no launch of its own (the stand-alone form, phc_phm_weight_reg, is one more launch per step).
constexpr int TAIL_REG_MAX_JOBS = 24;
constexpr int TAIL_REG_MAX_N = 16;
struct TailReg {
    const float *W[TAIL_REG_MAX_JOBS];
    float *grad[TAIL_REG_MAX_JOBS];                          // side buffers [n][cells]
    int n[TAIL_REG_MAX_JOBS], cells[TAIL_REG_MAX_JOBS], first_block[TAIL_REG_MAX_JOBS + 1];
    int num, p;
    float coef;
    float *shares;                                           // [first_block[num]]
    float *value, *loss_inout;                               // written / incremented by the last launch
};

__device__ __forceinline__ void tail_reg_body(const TailReg &r, unsigned b) {
    __shared__ float reg_red[4];
    const int tid = threadIdx.x;
    int j = 0;
    for (int q = 1; q < r.num; ++q) j = (int)b >= r.first_block[q] ? q : j;        // uniform scan of the by-value table
    const float *W = nullptr;
    float *G = nullptr;
    int n = 1, cells = 1;
    for (int q = 0; q < r.num; ++q)
        if (q == j) { W = r.W[q]; G = r.grad[q]; n = r.n[q]; cells = r.cells[q]; }
    const int c = ((int)b - r.first_block[j]) * 256 + tid;
    const bool live = c < cells;
    const int cc = live ? c : 0;
    float w[TAIL_REG_MAX_N];
#pragma unroll
    for (int i = 0; i < TAIL_REG_MAX_N; ++i) w[i] = W[(size_t)min(i, n - 1) * cells + cc];      // one round of unconditional loads
    float nrm = 0.f;
#pragma unroll
    for (int i = 0; i < TAIL_REG_MAX_N; ++i)
        if (i < n) nrm += r.p == 1 ? fabsf(w[i]) : w[i] * w[i];
    if (r.p != 1) nrm = sqrtf(nrm);
    if (live) {
        const float scale = r.coef / (float)cells;
#pragma unroll
        for (int i = 0; i < TAIL_REG_MAX_N; ++i)
            if (i < n) {
                // d ||v||_1 / dv = sign(v);  d ||v||_2 / dv = v / ||v||  (0 at the origin, as torch's norm backward)
                const float gq = r.p == 1 ? (w[i] > 0.f ? 1.f : (w[i] < 0.f ? -1.f : 0.f)) : (nrm > 0.f ? w[i] / nrm : 0.f);
                G[(size_t)i * cells + c] = scale * gq;
            }
    }
    float v = live ? nrm * (r.coef / (float)cells) : 0.f;          // this cell's share of coef * mean
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((tid & 63) == 0) reg_red[tid >> 6] = v;
    __syncthreads();
    if (tid == 0) r.shares[b] = (reg_red[0] + reg_red[1]) + (reg_red[2] + reg_red[3]);
}

// the value: all shares in block order (fixed summation order), by one wavefront-sized loop of the last launch's extra workgroup
__device__ __forceinline__ void tail_reg_finish(const TailReg &r) {
    if (threadIdx.x != 0) return;
    const int nb = r.first_block[r.num];
    float total = 0.f;
    for (int q0 = 0; q0 < nb; q0 += 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = r.shares[min(q0 + u, nb - 1)];
#pragma unroll
        for (int u = 0; u < 16; ++u) total += q0 + u < nb ? v[u] : 0.f;
    }
    if (r.value) *r.value = total;
    if (r.loss_inout) *r.loss_inout += total;
}

template <int STAGE>
__device__ __forceinline__ int grad_job_of_block(const GradBatch &b, unsigned bx) {
    int job = 0;
    for (int k = 1; k < b.num; ++k) {
        const unsigned start = STAGE == 0 ? b.j[k].blk0_partial : (STAGE == 1 ? b.j[k].blk0_reduce : b.j[k].blk0_contract);
        if (bx >= start) job = k;
    }
    return job;
}

// Host: plans, slab placement and the by-value job table of one launch (<= GRAD_JOBS_PER_LAUNCH jobs); defined in
// phm_linear.hip.  Returns PHC_OK or an error code (phc_last_error has the text).
struct GradLaunch {
    GradBatch batch;                                        // blk0_partial counts inside the job's own tile form
    unsigned nb_partial, nb_partial128, nb_direct, nb_reduce, nb_contract;
};
size_t grad_chunk_ws_floats(const phc_phm_grad_job *jobs, int num_jobs, int direct_form = 2);
int prepare_grad_launch(const char *who, const phc_phm_grad_job *jobs, int num_jobs, float *ws_base, GradLaunch *out, int direct_form = 2,
                        int slots3 = 512);
int launch_grad_rest(const char *who, const GradLaunch &gl, bool partial64_too, hipStream_t s);
int launch_grad_contract(const char *who, const GradLaunch &gl, hipStream_t s);
int launch_grad_partial128(const char *who, const GradLaunch &gl, hipStream_t s);    // the t128 == 1 jobs of a prepared batch, if any
// the tail's last launch with the gate map's parameter gradients (gate != NULL, see GateGradHost) and the weight regulariser's
// value (reg != NULL) riding in it as extra workgroups
struct GateGradHost {
    int job;                                                 // the dense job (x, dg) among the batch's jobs whose dH holds dM | dbg
    const float *H, *gate_w, *phm_b, *A, *W, *reg_add;
    int n, kin, kout, kg;
    float *dA, *dW, *db, *dWg, *dbg;
};
int launch_tail_finish(const char *who, const GradLaunch &gl, const GateGradHost *gate, const TailReg *reg, hipStream_t s);
// the register-resident jobs' partial launch (if any); profile = false: the caller already holds a launch-profiler scope
int launch_grad_direct(const char *who, const GradLaunch &gl, hipStream_t s, bool profile = true);
int check_grad_jobs(const char *who, const phc_phm_grad_job *jobs, int num_jobs);

}  // namespace phc
