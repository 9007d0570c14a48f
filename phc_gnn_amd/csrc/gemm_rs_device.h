// Register-stationary streaming GEMM for the PHM layers (y = x . H + b, dx = dy . H^T) on gfx950.
//
// Replaces matvec_product_new's matmul (phc/hypercomplex/layers.py:198-219) for operand matrices prebuilt once per step.
//
// Layout idea.  The fp32 matrix instruction v_mfma_f32_16x16x4_f32 wants, per lane (c = lane % 16, kq = lane / 16), ONE
// element x[row c][k] of the left operand and ONE element H[k][col] of the right one.  The sum over k may run in any order,
// so the k index of MFMA step (g, q) on lane group kq is chosen as  k = 16 g + 4 kq + q : then
//   * a lane's left operands of four consecutive steps are x[row c][16 g + 4 kq .. +3] — ONE 16-byte load from global memory,
//     straight into the registers the MFMAs read (no LDS staging of x, no transposition, no barrier): a whole 16-row x K slab
//     is K / 16 float4 registers per lane and stays in registers for every output column of the panel;
//   * a lane's right operands of the same four steps are H[16 g + 4 kq .. +3][col] — ONE ds_read_b128 if the panel is kept
//     k-interleaved in LDS: float4 slot (k / 4) * BN + col holds H[4 (k / 4) .. +3][col].  The operand matrix is prebuilt in
//     exactly that layout (phc_phm_build_H_batched, "k4" form), so the panel fill is a straight 16-byte copy.
// With the roles of the two MFMA operands swapped (A := H, B := x) the result tile comes out transposed: lane (c, kq) holds
// y[row c][col0 + 4 kq .. +3] — one 16-byte store per 16 x 16 tile, bias / residual as float4.
//
// Work split.  One workgroup per CU holds a [K x BN] panel in LDS (K = 256, BN = 128: 128 KB).  A task = (16-row slab, panel):
// (K / 16) * (BN / 16) * 4 MFMAs of 32 cycles (512 at 256 x 128 = 6.8 us).  Tasks of a panel go to SIMDs round-robin
// (slot = t mod P), the two wavefronts of a SIMD alternate: with ~3 tasks per SIMD at 24 k rows the chip finishes within one
// task of the ideal.  While a task's MFMAs run, the NEXT task's slab is loaded into the registers the MFMAs have finished with
// (rolling prefetch: slab register g is reloaded right after k-group g), so the loads have a whole task of lead time.
#pragma once
#include <hip/hip_runtime.h>

namespace phc {

using f32x4_t = __attribute__((ext_vector_type(4))) float;

struct GemmRsArgs {
    const float *x;            // [M, K] row-major, 16-byte aligned rows
    const float *hk4;          // operand matrix, k4 form: float4 slot (k / 4) * Nout + j = H[4 (k / 4) .. +3][j]
    const float *bias;         // [Nout] or null
    const float *residual;     // [M, Nout] or null
    float *y;                  // [M, Nout]
    long long M;
    int Nout;
    int panels;                // ceil(Nout / BN)
    int groups;                // workgroups per panel (gridDim.x = panels * groups)
};

// XCD-aware (panel, group) of a linear block id: blocks b, b + 8, b + 16, ... share an XCD (round-robin dispatch), so the
// `panels` workgroups that read the same x slabs (same group, different panel) are placed 8 ids apart.
__device__ __forceinline__ void rs_block_role(unsigned b, unsigned panels, unsigned groups, unsigned &panel, unsigned &group) {
    const unsigned run = 8u * panels, chunk = b / run, rem = b - chunk * run;
    if ((chunk + 1u) * 8u <= groups) {
        panel = rem >> 3;
        group = chunk * 8u + (rem & 7u);
    } else {                                              // ragged tail: plain order
        const unsigned tail0 = (groups / 8u) * 8u, r = b - tail0 * panels;
        panel = r % panels;
        group = tail0 + r / panels;
    }
}

template <int K, int BN, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void gemm_rs16_kernel(GemmRsArgs a) {
    static_assert(K % 16 == 0 && BN % 16 == 0, "K, BN multiples of 16");
    constexpr int G = K / 16, CT = BN / 16, THREADS = 64 * WAVES;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4_t *hs = reinterpret_cast<f32x4_t *>(smem);              // [K / 4][BN] float4

    unsigned panel, group;
    rs_block_role(blockIdx.x, (unsigned)a.panels, (unsigned)a.groups, panel, group);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, kq = lane >> 4;
    const int j0 = (int)panel * BN;
    const int Nout = a.Nout;

    // tasks of this panel: slabs t = slot + P * (half + 2 j), slot = this wave's SIMD among the P = 4 * groups SIMDs of the panel
    // (waves w and w + 4 of an 8-wave workgroup share a SIMD)
    const long long num_slabs = (a.M + 15) >> 4;
    const long long P = 4LL * a.groups;
    const long long slot = (long long)group * 4 + (wave & 3);
    const int half = wave >> 2;
    constexpr int HALVES = WAVES / 4;
    long long slab = slot + P * half;
    const long long slab_step = P * HALVES;

    // first slab -> registers (clamped rows: loads are unconditional)
    f32x4_t xa[G];
    auto load_slab = [&](long long s, int g) {
        const long long row = min(s * 16 + c, a.M - 1);
        return *reinterpret_cast<const f32x4_t *>(a.x + (size_t)row * K + 16 * g + 4 * kq);
    };
    if (slab < num_slabs) {
#pragma unroll
        for (int g = 0; g < G; ++g) xa[g] = load_slab(slab, g);
    }
    // panel fill: straight 16-byte copies (the operand is stored k4-interleaved); columns past Nout read as zero
    {
        const f32x4_t *src = reinterpret_cast<const f32x4_t *>(a.hk4);
        constexpr int TOTAL = (K / 4) * BN;
        static_assert(TOTAL % THREADS == 0, "panel slots divide among the threads");
        constexpr int PER = TOTAL / THREADS, BATCH = PER < 8 ? PER : 8;
        static_assert(PER % BATCH == 0, "");
#pragma unroll 1
        for (int i0 = 0; i0 < PER; i0 += BATCH) {
            f32x4_t v[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int idx = (i0 + u) * THREADS + tid;
                const int k4 = idx / BN, col = idx - k4 * BN;
                v[u] = src[(size_t)k4 * Nout + min(j0 + col, Nout - 1)];
            }
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int idx = (i0 + u) * THREADS + tid;
                const int k4 = idx / BN, col = idx - k4 * BN;
                const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
                hs[idx] = j0 + col < Nout ? v[u] : z;
            }
        }
    }
    // bias of the 4 columns this lane stores per tile: col = j0 + ct * 16 + 4 kq
    f32x4_t bias4[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int col = min(j0 + ct * 16 + 4 * kq, Nout - 4);
        const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
        bias4[ct] = a.bias ? *reinterpret_cast<const f32x4_t *>(a.bias + col) : z;
    }
    __syncthreads();

    const f32x4_t *hl = hs + kq * BN + c;                          // + (4 g) * BN + ct * 16
    for (; slab < num_slabs; slab += slab_step) {
        f32x4_t acc[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[ct] = bias4[ct];
        const long long next = slab + slab_step;
        const long long nrow = min((next < num_slabs ? next : slab) * 16 + c, a.M - 1);
        const float *nx = a.x + (size_t)nrow * K + 4 * kq;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            f32x4_t b[CT];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) b[ct] = hl[(4 * g) * BN + ct * 16];
            const f32x4_t av = xa[g];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[ct][q], av[q], acc[ct], 0, 0, 0);
            // rolling prefetch: this register is free now; the next task's k-group g goes in flight
            xa[g] = *reinterpret_cast<const f32x4_t *>(nx + 16 * g);
        }
        // epilogue: lane (c, kq) holds y[row c][j0 + ct * 16 + 4 kq .. +3]
        const long long row = slab * 16 + c;
        if (row < a.M) {
            float *yr = a.y + (size_t)row * Nout + j0 + 4 * kq;
            const float *rr = a.residual ? a.residual + (size_t)row * Nout + j0 + 4 * kq : nullptr;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                if (j0 + ct * 16 + 4 * kq < Nout) {
                    f32x4_t v = acc[ct];
                    if (rr) v += *reinterpret_cast<const f32x4_t *>(rr + ct * 16);
                    *reinterpret_cast<f32x4_t *>(yr + ct * 16) = v;
                }
            }
        }
    }
}

}  // namespace phc
