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
    unsigned long long *stamps;   // diagnostics only (null in the product): per wave {entry, loop start, loop end} x {shader clock, 100 MHz clock}
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

// MODE (experiments): bit 0 = no panel reads in the loop, bit 1 = no slab prefetch loads
template <int K, int BN, int WAVES, int MODE = 0>
__global__ __launch_bounds__(64 * WAVES) void gemm_rs16_kernel(GemmRsArgs a) {
    static_assert(K % 16 == 0 && BN % 16 == 0, "K, BN multiples of 16");
    constexpr int G = K / 16, CT = BN / 16, THREADS = 64 * WAVES;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4_t *hs = reinterpret_cast<f32x4_t *>(smem);              // [K / 4][BN] float4

    unsigned long long st_t[3] = {0, 0, 0}, st_r[3] = {0, 0, 0};
    if (a.stamps) { st_t[0] = __builtin_amdgcn_s_memtime(); st_r[0] = __builtin_amdgcn_s_memrealtime(); }
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
    __syncthreads();
    if (a.stamps) { st_t[1] = __builtin_amdgcn_s_memtime(); st_r[1] = __builtin_amdgcn_s_memrealtime(); }

    const f32x4_t *hl = hs + kq * BN + c;                          // + (4 g) * BN + ct * 16
    const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
    for (; slab < num_slabs; slab += slab_step) {
        f32x4_t acc[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[ct] = zero4;
        const long long next = slab + slab_step;
        const long long nrow = min((next < num_slabs ? next : slab) * 16 + c, a.M - 1);
        const float *nx = a.x + (size_t)nrow * K + 4 * kq;
        // software pipeline over the k-groups: the panel reads of group g + 1 are issued before the MFMAs of group g (two
        // register sets, renamed by the full unroll); scheduling fences keep hipcc from hoisting every read of the task
        f32x4_t b[2][CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) b[0][ct] = hl[ct * 16];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (g + 1 < G && !(MODE & 1)) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) b[(g + 1) & 1][ct] = hl[(4 * (g + 1)) * BN + ct * 16];
            }
            if (MODE & 1) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) b[(g + 1) & 1][ct] = b[g & 1][ct];
            }
            __builtin_amdgcn_sched_barrier(0);
            const f32x4_t av = xa[g];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[g & 1][ct][q], av[q], acc[ct], 0, 0, 0);
            // rolling prefetch: this register is free now; the next task's k-group g goes in flight
            if (!(MODE & 2)) xa[g] = *reinterpret_cast<const f32x4_t *>(nx + 16 * g);
            __builtin_amdgcn_sched_barrier(0);
        }
        // epilogue: lane (c, kq) holds y[row c][j0 + ct * 16 + 4 kq .. +3]
        const long long row = slab * 16 + c;
        if (row < a.M) {
            float *yr = a.y + (size_t)row * Nout + j0 + 4 * kq;
            const float *rr = a.residual ? a.residual + (size_t)row * Nout + j0 + 4 * kq : nullptr;
            const float *bb = a.bias ? a.bias + j0 + 4 * kq : nullptr;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                if (j0 + ct * 16 + 4 * kq < Nout) {
                    f32x4_t v = acc[ct];
                    if (bb) v += *reinterpret_cast<const f32x4_t *>(bb + ct * 16);
                    if (rr) v += *reinterpret_cast<const f32x4_t *>(rr + ct * 16);
                    *reinterpret_cast<f32x4_t *>(yr + ct * 16) = v;
                }
            }
        }
    }
    if (a.stamps) {
        st_t[2] = __builtin_amdgcn_s_memtime(); st_r[2] = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) {
            unsigned long long *o = a.stamps + ((size_t)blockIdx.x * WAVES + wave) * 6;
#pragma unroll
            for (int i = 0; i < 3; ++i) { o[i] = st_t[i]; o[3 + i] = st_r[i]; }
        }
    }
}



// ==========================================================================================================================
// H-stationary form: the OPERAND MATRIX lives in registers, x streams through LDS.
//
// Every wavefront of a workgroup keeps its own 16 TPW columns of H for the WHOLE contraction in registers (K / 16 float4 per
// tile and lane, loaded once per launch from the k4-interleaved operand: 256-byte runs), and the workgroup streams x in steps
// of RS = 16 RB rows: a step's rows are brought into a ring of LDS slots by LDS-DMA (global_load_lds_dwordx4: whole 1-KiB
// lines, no VGPR round trip, issued once per WORKGROUP — every wavefront reads the same rows), and each wavefront reads its
// B fragments x[row c][16 g + 4 kq .. +3] with one ds_read_b128 per k-group for all of its tiles.  Against the x-stationary
// form above: x is read from memory once (not once per column panel), in full lines; no 128 KB operand panel has to be filled
// before the first MFMA; half the LDS reads per MFMA; the per-column batch statistics of the output are register sums that
// live across the whole launch (a wavefront owns its columns for every row).
//   LDS image of a slot: one DMA instruction = 1 KiB = RPI = 256 / K rows, instruction stride 1 KiB + 32 B (the pad makes the
//   fragment reads conflict-free: the b128 lane groups then cover 16 distinct 16-byte bank groups); for K = 128 the odd row of
//   a pair is stored with its 16-byte chunks XOR 1 (applied on the SOURCE address: the DMA's LDS side is lane-linear).
//   Ring protocol (one raw s_barrier per step): wait for the own DMA pieces of step i (counted vmcnt: stores and younger DMAs
//   stay in flight) -> barrier -> issue the DMA of step i + R - 1 into the slot step i - 1 was read from -> MFMAs of step i.
// ==========================================================================================================================
struct GemmHsArgs {
    const float *x;            // [M, K]
    const float *hk4;          // k4 form of the operand, [K / 4][Nout] float4
    const float *bias;         // [Nout] or null
    const float *residual;     // [M, Nout] or null
    float *y;                  // [M, Nout]
    long long M;
    int Nout;                  // multiple of 16 * TPW
    int panels;                // ceil(Nout / (16 * TPW * WAVES))
    int groups;                // workgroups per panel
    unsigned long long *stamps;
};

template <int N>
__device__ __forceinline__ void hs_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

typedef __attribute__((address_space(3))) void hs_lds_void;

// One LDS-DMA instruction: every lane's 16 bytes at `gsrc` land at LDS byte address lds_dst + 16 * lane (lds_dst wave-uniform).
// Inline asm on purpose: hipcc orders a __builtin_amdgcn_global_load_lds against every later ds_read with s_waitcnt vmcnt(0),
// which would drain the prefetch ring at each step; the ring's own counted waits (hs_wait_vm) are the ordering here.  An asm
// VMEM operation only makes hipcc's waits for its own loads stricter (more operations outstanding), never too weak.  M0 is the
// DMA's LDS base; it is saved and restored inside the statement (cdna_hip_programming.md, "What hipcc does not do").
__device__ __forceinline__ void hs_dma16(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

template <int K, int TPW, int WAVES, int RB, int R>
__global__ __launch_bounds__(64 * WAVES) void gemm_hs16_kernel(GemmHsArgs a) {
    static_assert(K == 128 || K == 256, "K = 128 or 256 (one DMA instruction = 1 KiB = 256 / K rows)");
    constexpr int G = K / 16, RS = 16 * RB, RPI = 256 / K, INSTR = RS / RPI, IPW = INSTR / WAVES;
    static_assert(INSTR % WAVES == 0 && IPW >= 1, "the DMA instructions of a step divide among the wavefronts");
    constexpr int ISTRIDE = 66;                                   // float4 slots per DMA instruction (64 + 2 pad)
    constexpr int SLOT_F4 = INSTR * ISTRIDE;
    static_assert(R >= 3 && R <= 8, "ring depth");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4_t *ring = reinterpret_cast<f32x4_t *>(smem);            // [R][SLOT_F4]
    const unsigned ring_lds = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(hs_lds_void *)smem);   // LDS byte address

    unsigned long long st_t[3] = {0, 0, 0}, st_r[3] = {0, 0, 0};
    if (a.stamps) { st_t[0] = __builtin_amdgcn_s_memtime(); st_r[0] = __builtin_amdgcn_s_memrealtime(); }
    unsigned panel, group;
    rs_block_role(blockIdx.x, (unsigned)a.panels, (unsigned)a.groups, panel, group);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, kq = lane >> 4;
    const int Nout = a.Nout;
    const int col0 = ((int)panel * WAVES + wave) * 16 * TPW;      // this wavefront's first column
    const bool active = col0 < Nout;                              // (a ragged last panel: idle wavefronts still move x)

    const long long num_steps = (a.M + RS - 1) / RS;
    const long long my_steps = group < num_steps ? (num_steps - group + a.groups - 1) / a.groups : 0;
    if (my_steps == 0) return;                                    // whole workgroup: uniform

    // ---- operand slice -> registers (once per launch) ----
    f32x4_t hreg[TPW][G];
    {
        const f32x4_t *src = reinterpret_cast<const f32x4_t *>(a.hk4);
        const int cc = min(col0, Nout - 16 * TPW) + c;
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int g = 0; g < G; ++g) hreg[t][g] = src[(size_t)(4 * g + kq) * Nout + cc + 16 * t];
    }
    // ---- DMA of one step: instruction ii = wave + WAVES * u covers rows ii * RPI .. of the step ----
    auto issue_step = [&](long long step_local, int slot) {
        const long long step = group + min(step_local, my_steps - 1) * a.groups;       // clamped: the tail re-loads valid rows
#pragma unroll
        for (int u = 0; u < IPW; ++u) {
            const int ii = wave + WAVES * u;
            const int sub = RPI == 1 ? 0 : (lane >> 5);                               // row of the pair (K = 128)
            const int p = RPI == 1 ? lane : (lane & 31);                              // 16-byte chunk slot within the row
            const long long row = min(step * RS + ii * RPI + sub, a.M - 1);
            const float *src = a.x + (size_t)row * K + 4 * (p ^ sub);
            hs_dma16(src, ring_lds + 16u * (unsigned)(slot * SLOT_F4 + ii * ISTRIDE));   // wave-uniform destination
        }
    };
#pragma unroll
    for (int s = 0; s < R - 1; ++s) issue_step(s, s);

    // fragment base of this lane per row block (float4 slots): row = rb * 16 + c
    int fbase[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int row = rb * 16 + c;
        fbase[rb] = (row / RPI) * ISTRIDE + (row % RPI) * (K / 4) + (kq ^ (row % RPI));
    }
    f32x4_t bias4[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
        bias4[t] = a.bias ? *reinterpret_cast<const f32x4_t *>(a.bias + min(col0, Nout - 16 * TPW) + 16 * t + 4 * kq) : z;
    }
    // The operand slice and the bias must be COMPLETE in hipcc's own bookkeeping before the loop: a load it still counts as
    // pending at the loop header is waited for with vmcnt(0) at its first use in EVERY iteration — which would also drain the
    // DMA ring.  An empty asm that reads the registers makes hipcc place that wait here, once.
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
#pragma unroll
        for (int g = 0; g < G; g += 8)
            asm volatile("" ::"v"(hreg[t][g]), "v"(hreg[t][g + 1]), "v"(hreg[t][g + 2]), "v"(hreg[t][g + 3]), "v"(hreg[t][g + 4]),
                         "v"(hreg[t][g + 5]), "v"(hreg[t][g + 6]), "v"(hreg[t][g + 7]));
        asm volatile("" ::"v"(bias4[t]));
    }
    if (a.stamps) { st_t[1] = __builtin_amdgcn_s_memtime(); st_r[1] = __builtin_amdgcn_s_memrealtime(); }

    // Counted wait for DMA(i): the (R - 2) * IPW DMA instructions of steps i + 1 .. i + R - 2 are younger than it in every
    // iteration (the tail re-issues clamped steps), so "at most that many operations outstanding" implies DMA(i) has landed
    // whatever number of stores was issued in between (stores and loads share the counter; a masked-off store may be skipped).
    constexpr int VM_KEEP = (R - 2) * IPW;
    int slot = 0;
    for (long long i = 0; i < my_steps; ++i) {
        hs_wait_vm<VM_KEEP>();
        // ... and after the barrier everyone's: the slot of step i is complete, the slot of step i - 1 is free
        __builtin_amdgcn_s_barrier();
        {
            int nslot = slot + R - 1;
            nslot = nslot >= R ? nslot - R : nslot;
            issue_step(i + R - 1, nslot);
        }
        const f32x4_t *sl = ring + (size_t)slot * SLOT_F4;
        f32x4_t acc[RB][TPW];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int t = 0; t < TPW; ++t) acc[rb][t] = bias4[t];
        f32x4_t fr[2][RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) fr[0][rb] = sl[fbase[rb]];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (g + 1 < G) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) fr[(g + 1) & 1][rb] = sl[fbase[rb] + 4 * (g + 1)];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int t = 0; t < TPW; ++t)
                        acc[rb][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(hreg[t][g][q], fr[g & 1][rb][q], acc[rb][t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // epilogue: lane (c, kq) holds y[row][col0 + 16 t + 4 kq .. +3]
        const long long step = group + i * a.groups;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const long long row = step * RS + rb * 16 + c;
            if (active && row < a.M) {
                float *yr = a.y + (size_t)row * Nout + col0 + 4 * kq;
#pragma unroll
                for (int t = 0; t < TPW; ++t) {
                    f32x4_t v = acc[rb][t];
                    if (a.residual) v += *reinterpret_cast<const f32x4_t *>(a.residual + (size_t)row * Nout + col0 + 4 * kq + 16 * t);
                    *reinterpret_cast<f32x4_t *>(yr + 16 * t) = v;
                }
            }
        }
        slot = slot + 1 == R ? 0 : slot + 1;
    }
    hs_wait_vm<0>();
    if (a.stamps) {
        st_t[2] = __builtin_amdgcn_s_memtime(); st_r[2] = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) {
            unsigned long long *o = a.stamps + ((size_t)blockIdx.x * WAVES + wave) * 6;
#pragma unroll
            for (int i = 0; i < 3; ++i) { o[i] = st_t[i]; o[3 + i] = st_r[i]; }
        }
    }
}

}  // namespace phc
