// H-stationary streaming GEMM of the PHM layers on gfx950:  y = x . H (+ b) (+ residual),  dx = dy . H^T
// (the matmul of matvec_product_new, phc/hypercomplex/layers.py:198-219, on an operand matrix prebuilt once per step).
//
// The OPERAND MATRIX lives in registers, x streams through LDS:
//   * v_mfma_f32_16x16x4_f32 takes from lane (c = lane % 16, kq = lane / 16) ONE element of each operand.  The sum over k may
//     run in any order, so the k of MFMA step (g, q) on lane group kq is chosen as k = 16 g + 4 kq + q: a lane's operands of
//     four consecutive steps are then 16 contiguous bytes of one row of x, and — with the operand matrix stored "k4-interleaved"
//     (float4 slot (k / 4) * Nout + j = H[4 (k / 4) .. +3][j], written by phc_phm_build_H_batched) — 16 contiguous bytes of H.
//   * every wavefront of a workgroup keeps its own 16 TPW columns of H for the WHOLE contraction in registers (K / 16 float4 per
//     tile and lane, loaded once per launch in 256-byte runs);
//   * the workgroup streams x in steps of RS = 16 RB rows: a step's rows are brought into a ring of LDS slots by LDS-DMA
//     (global_load_lds_dwordx4: whole 1-KiB lines, no VGPR round trip, issued once per WORKGROUP — all wavefronts read the same
//     rows), and each wavefront reads its fragments x[row c][16 g + 4 kq .. +3] with ONE ds_read_b128 per k-group for all of
//     its tiles;
//   * with the MFMA operand roles swapped (A := H, B := x) the result tile comes out transposed: lane (c, kq) holds
//     y[row c][col0 + 4 kq .. +3] — one 16-byte store per tile, bias / residual as float4, and per-column statistics of the
//     output are per-lane running sums that live across the whole launch (a wavefront owns its columns for every row).
// Against an LDS-resident operand panel (the previous design: 128 KB panel fill before the first MFMA, x re-read once per
// column panel through per-wavefront staging): x is read once, in full lines; no panel fill; a quarter of the LDS reads per MFMA.
//
// LDS image of a slot: one DMA instruction = 1 KiB = RPI = 256 / K rows; instruction stride 1 KiB + 32 B (the pad makes the
// fragment reads conflict-free: each ds_read_b128 lane group then covers 16 distinct 16-byte bank groups); for K = 128 the odd
// row of a pair is stored with its 16-byte chunks XOR 1 (applied on the SOURCE address: the DMA's LDS side is lane-linear).
// Ring protocol (one raw s_barrier per step): counted vmcnt wait for the own DMA pieces of step i (younger DMAs and stores
// stay in flight) -> [operand transform of the own pieces, in place] -> barrier -> issue the DMA of step i + R - 1 into the slot
// step i - 1 was read from -> MFMAs of step i -> epilogue.
//
// FLAGS: HS_ST  emit the batch-norm statistics (sum, sum of squares over the rows below *n_valid) of the OUTPUT columns into a
//               statistics accumulator (phc_common.h; 64-bit integer atomics, one add per column and workgroup);
//        HS_XF  the INPUT is a batch-norm that has not been applied: x := act(x * scale + shift), rows >= *n_valid := 0,
//               evaluated in LDS by the wavefront that fetched the rows (scale / shift from save_stat, or derived in the
//               prologue from an accumulator: XfAcc);
//        HS_BW  dz = dy . H^T of a layer whose input was act(batch_norm(bw_x)): besides dz the launch adds the norm backward's
//               column sums of g = dz * act'(.) and g * x_hat into a statistics accumulator.
//        HS_BA  the operand is a batch-norm BACKWARD that has not been applied: the rows of the upstream gradient (x) and of
//               the norm's input (ba_x) arrive side by side in the ring, and the wavefront that fetched them evaluates
//                   dz = w invstd (g - mean g - x_hat mean(g x_hat)),   g = dout * act'(x_hat w + b),   rows >= *n_valid := 0
//               in LDS (norm.py:30-35 backward; the per-column means come from the sums accumulator of the pass in front),
//               feeds it to the MFMAs and also writes it out (the parameter-gradient GEMM of the layer reads it later):
//               the norm backward's element-wise pass and its launch disappear into the dx GEMM behind it.
//        HS_DR  (with HS_BA) the norm's output went through phm_dropout (models.py:211-213: act -> dropout -> + skip): the upstream
//               gradient is multiplied by the keep-and-scale mask first, regenerated from (seed, step, site, row, column) exactly as
//               bn_bwd_apply_acc_kernel regenerates it (phc_common.h: drop_factors) — round 5: the reference's default and published
//               settings all train with dropout, and without this flag their layers fell off the fused MLP + norm node.
#pragma once
#include <hip/hip_runtime.h>

// Every lambda of the kernel below MUST be inlined: one that is not takes its captures through a closure object in memory, and a
// captured by-value kernel argument struct then lives in scratch for the whole launch (seen when the dropout mask's Philox rounds
// pushed `transform` past hipcc's inlining threshold: 432 - 720 bytes of private segment, 88 instead of 30 us per launch).
#define PHC_HS_INLINE __attribute__((always_inline))

#include <type_traits>

#include "phc_common.h"

namespace phc {

using f32x4_t = __attribute__((ext_vector_type(4))) float;

constexpr int HS_ST = 1, HS_XF = 2, HS_BW = 4, HS_BA = 8, HS_DR = 16;
constexpr int HS_BW_COLS = 256;                      // widest output the BW variants serve (their norm constants' LDS copy: [4][HS_BW_COLS])

// XF operand whose batch statistics are still in an ACCUMULATOR: the launch derives scale / shift itself, in its prologue
// (acc == NULL: xf_stat holds them, merged elsewhere).
struct XfAcc {
    const unsigned long long *acc = nullptr;
    BnParams p = {};
    float momentum = 0.f, eps = 0.f;
    float *stat_out = nullptr;                               // [6 K] mean | invstd | scale | shift | w | b, written by workgroup 0
};

struct GemmHsArgs {
    const float *x;            // [M, K]
    const float *hk4;          // k4 form of the operand, [K / 4][Nout] float4
    const float *bias;         // [Nout] or null
    const float *residual;     // [M, Nout] or null
    float *y;                  // [M, Nout]
    long long M;
    int K;                     // contraction depth = floats per row of x (<= the kernel's K template: the published 180 / 200 widths run
                               // on the 192 / 208 instantiations, their last k-group and column tiles partly empty)
    int Nout;                  // columns of the operand and of y (<= 16 * TPW * WAVES * panels)
    int panels;
    int groups;                // workgroups per panel (gridDim.x = panels * groups)
    const int *n_valid;        // device scalar or null (ST / XF / BW)
    unsigned long long *stat_acc;   // ST: accumulator of width Nout;  BW: accumulator of width Nout (sum g | sum g x_hat)
    const float *xf_stat;      // XF: [6 K] mean | invstd | scale | shift | w | b
    int xf_act;
    const float *bw_x;         // BW: the norm's input [M, Nout]
    const float *bw_stat;      // BW: [6 Nout]
    int bw_act;
    const float *ba_x;         // BA: the norm's input [M, K] (x holds the upstream gradient)
    const float *ba_stat;      // BA: [6 K] save_stat of that norm
    const unsigned long long *ba_acc;   // BA: accumulator of width K: sum g | sum g x_hat over the valid rows
    long long ba_count;        // BA: rows those sums were taken over when that is not this launch's valid rows (a row shard); 0 = n_valid
    int ba_act;
    float *ba_dz;              // BA: [M, K] out
    float *ba_dwb;             // BA: [2 K] out or null: dweight = sum g x_hat | dbias = sum g   (written by workgroup 0)
    BnDrop ba_drop;            // DR: the dropout behind the norm's activation (p, same, k, site, device {seed, step})
    unsigned long long *stamps;     // diagnostics only (null in the product): 12 words per wavefront, see the kernel's end
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
// DMA's LDS base; it is saved and restored inside the statement.
__device__ __forceinline__ void hs_dma16(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);           // wave-uniform by construction; hipcc must also know it
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

// Loads that hipcc's wait-count insertion does not see (the caller waits with a counted s_waitcnt and ties the registers to it).
__device__ __forceinline__ unsigned long long hs_ld64(const unsigned long long *p) {
    unsigned long long v;
    asm volatile("global_load_dwordx2 %0, %1, off" : "=&v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ float hs_ld32(const float *p) {
    float v;
    asm volatile("global_load_dword %0, %1, off" : "=&v"(v) : "v"(p) : "memory");
    return v;
}

// sum over the 16 lanes that share kq (lanes kq * 16 .. + 15): xor offsets 1, 2, 4, 8 stay inside the group
__device__ __forceinline__ double hs_sum16(double v) {
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Publish NV = 8 * TPW column totals that every lane of a 16-lane group holds (after hs_sum16): lane c takes total number c —
// (tile t = c / 8, column q = (c / 2) % 4, quantity c % 2) — so that the whole wavefront adds its 32 TPW values with TWO atomic
// wave-instructions (high and low word) instead of one pair per value from a single lane: the atomic instructions, not their
// bytes, are what a CU pays for (MI355X_MICROARCH.md, global atomics: ~120 cycles per wave-instruction).
template <int TPW>
__device__ __forceinline__ void hs_publish(unsigned long long *acc, int Nout, int rep, int col0, int kq, int c, const double (&v0)[TPW][4],
                                           const double (&v1)[TPW][4]) {
    double mine = 0.0;
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            mine = c == 8 * t + 2 * q ? v0[t][q] : mine;
            mine = c == 8 * t + 2 * q + 1 ? v1[t][q] : mine;
        }
    if (c < 8 * TPW && col0 + 16 * (c >> 3) + 4 * kq + ((c >> 1) & 3) < Nout)
        stat_acc_add(acc, Nout, rep, c & 1, col0 + 16 * (c >> 3) + 4 * kq + ((c >> 1) & 3), mine);
}

constexpr int HS_ISTRIDE = 66;                                    // float4 slots per DMA instruction (64 + 2 pad)

template <int K, int RB>
constexpr size_t hs_slot_bytes() {                                // one operand image of a ring slot (BA: two images per slot)
    return (size_t)(16 * RB / (256 / K)) * HS_ISTRIDE * 16;
}
template <int K, int RB, int R, int FLAGS>
constexpr size_t hs_lds_bytes() {
    return (size_t)R * hs_slot_bytes<K, RB>() * ((FLAGS & HS_BA) ? 2 : 1) + ((FLAGS & HS_XF) ? 2 * (size_t)K * sizeof(float) : 0) +
           ((FLAGS & HS_BA) ? 7 * (size_t)K * sizeof(float) : 0) + (((FLAGS & HS_BW) && K == 256) ? 4 * (size_t)HS_BW_COLS * sizeof(float) : 0);
}

template <int K, int TPW, int WAVES, int RB, int R, int FLAGS>
__global__ __launch_bounds__(64 * WAVES) void gemm_hs16_kernel(GemmHsArgs a, XfAcc xa) {
    static_assert(K == 128 || (K > 128 && K <= 256 && K % 16 == 0),
                  "K = 128 (two rows per 1-KiB DMA instruction) or a multiple of 16 in (128, 256] (one row per instruction)");
    // KR = the rows' REAL width a.K <= K (a multiple of 4; == K for 128 / 256).  Chunks of a row past KR are never fetched (their
    // lanes re-fetch chunk 0) and meet operand rows that read as zero; tiles / columns past a.Nout are loaded as zero and never stored.
    const int KR = a.K;
    constexpr bool ST = (FLAGS & HS_ST) != 0, XF = (FLAGS & HS_XF) != 0, BW = (FLAGS & HS_BW) != 0, BA = (FLAGS & HS_BA) != 0;
    constexpr bool DR = (FLAGS & HS_DR) != 0;
    constexpr bool BW_LDS = BW && K == 256;                      // the norm constants of the output columns parked in LDS behind the first step
    static_assert(!(BA && XF), "one operand transform per launch");
    static_assert(!DR || BA, "the dropout mask rides in the norm-backward transform");
    // DR: seed and step of the mask — a uniform address out of the kernel arguments: scalar loads (lgkmcnt), issued in front of
    // everything and invisible to the counted vmcnt waits below
    unsigned long long dr_seed = 0, dr_step = 0;
    float dr_p = 0.f;
    int dr_same = 0, dr_k = 4, dr_site = 0;                       // (copies: a reference to a member of `a` would put all of `a` in scratch)
    if constexpr (DR) {
        dr_seed = a.ba_drop.state[0];
        dr_step = a.ba_drop.state[1];
        dr_p = a.ba_drop.p; dr_same = a.ba_drop.same; dr_k = a.ba_drop.k; dr_site = a.ba_drop.site;
    }
    constexpr int G = K / 16, RS = 16 * RB, RPI = 256 / K, INSTR = RS / RPI, IPW = INSTR / WAVES;
    static_assert(INSTR % WAVES == 0 && IPW >= 1, "the DMA instructions of a step divide among the wavefronts");
    constexpr int ISTRIDE = HS_ISTRIDE;
    constexpr int IMG_F4 = INSTR * ISTRIDE;                        // one operand image
    constexpr int SLOT_F4 = IMG_F4 * (BA ? 2 : 1);                 // BA: [upstream gradient | norm input]
    constexpr int THREADS = 64 * WAVES;
    static_assert(R >= 3 && R <= 8, "ring depth");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4_t *ring = reinterpret_cast<f32x4_t *>(smem);            // [R][SLOT_F4]
    float *xf_lds = smem + (size_t)R * SLOT_F4 * 4;               // XF from an accumulator: scale[K] | shift[K]
    float *ba_lds = xf_lds;                                       // BA: invstd | mean | w | b | -mean(g x_hat) | -mean g | w invstd   [7][K]
    float *bw_lds = xf_lds + (XF ? 2 * K : 0) + (BA ? 7 * K : 0);   // BW: mean | invstd | w | b of the OUTPUT columns   [4][HS_BW_COLS]
    const unsigned ring_lds = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(hs_lds_void *)smem);   // LDS byte address

    unsigned long long st_t[3] = {0, 0, 0}, st_r[3] = {0, 0, 0}, dbg_t[3] = {0, 0, 0};
    if (a.stamps) { st_t[0] = __builtin_amdgcn_s_memtime(); st_r[0] = __builtin_amdgcn_s_memrealtime(); }
    const unsigned panel = blockIdx.x % (unsigned)a.panels, group = blockIdx.x / (unsigned)a.panels;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, kq = lane >> 4;
    const int Nout = a.Nout;
    const int col0 = ((int)panel * WAVES + wave) * 16 * TPW;      // this wavefront's first column (< Nout by construction)

    const long long num_steps = (a.M + RS - 1) / RS;
    const long long my_steps = group < num_steps ? (num_steps - group + a.groups - 1) / a.groups : 0;
    if (my_steps == 0) return;                                    // whole workgroup: uniform
    long long n_valid = a.M;
    // XF / BA launches: steps that hold padding rows ONLY (every row >= *n_valid: the tail of a fixed-shape batch) take no loads and no
    // matrix work.  Their operand rows read as zero by definition, so their output rows are the bias (0 for a dx GEMM) and dz = 0 —
    // written directly, bit-identical to computing them.  Whether a bucket's padded row count is a whole number of rounds of this
    // launch's step loop then no longer decides whether the launch pays a nearly empty extra pass (24 832 rows at 16 rows x 256
    // workgroups: seven passes against six).  The workgroup's FIRST step is always computed (the peeled step carries the prologue's
    // waits); padding rows stay out of every statistic and carry zero gradient either way.
    long long my_live = my_steps;
    if constexpr (ST || XF || BW || BA) {
        if (a.n_valid) n_valid = min((long long)*a.n_valid, a.M);
    }
    if constexpr (XF || BA) {
        if (a.n_valid && !a.residual) {
            const long long live_steps = (n_valid + RS - 1) / RS;
            const long long lv = group < live_steps ? (live_steps - group + a.groups - 1) / a.groups : 0;
            my_live = max(1LL, min(my_steps, lv));
        }
    }

    // ---- DMA of one step: instruction ii = wave + WAVES * u covers rows ii * RPI .. of the step ----
    const int dsub = RPI == 1 ? 0 : (lane >> 5);                  // row of the pair (K = 128)
    const int dchunk = (RPI == 1 ? lane : (lane & 31)) ^ dsub;    // the 16-byte chunk of its row this lane fetches
    const bool dlive = 4 * dchunk < KR;                           // (always true at K = 128 / 256)
    const int dfetch = dlive ? dchunk : 0;                        // a lane past the row's end re-fetches chunk 0: finite values x zero operand
    auto issue_step = [&](long long step_local, int slot) PHC_HS_INLINE {
        const long long step = group + min(step_local, my_steps - 1) * a.groups;       // clamped: the tail re-loads valid rows
#pragma unroll
        for (int u = 0; u < IPW; ++u) {
            const int ii = wave + WAVES * u;
            const long long row = min(step * RS + ii * RPI + dsub, a.M - 1);
            hs_dma16(a.x + (size_t)row * KR + 4 * dfetch, ring_lds + 16u * (unsigned)(slot * SLOT_F4 + ii * ISTRIDE));
            if constexpr (BA)
                hs_dma16(a.ba_x + (size_t)row * KR + 4 * dfetch, ring_lds + 16u * (unsigned)(slot * SLOT_F4 + IMG_F4 + ii * ISTRIDE));
        }
    };
    // BA / XF from an accumulator: the totals' words and the norm constants of column `tid` go in flight FIRST, as ASM loads with
    // a counted wait of their own (ACC_YOUNGER below).  As ordinary loads hipcc waited for them with a count that leaves the
    // ring's first fill out (it does not see the asm DMAs): the constants — and with them the first transform — came 5 us after
    // entry instead of one memory round trip (scripts/dbg/hs_phases.py, prologue detail).  Only the K / 64 wavefronts that own
    // a column load.
    // (the asm loads' destinations are never initialised, copied or merged with another value before `acc_arrived` has tied
    // them to the counted wait: a register copy hipcc placed in between would read them before the data is there)
    unsigned long long ba_raw[(BA || XF) ? 2 : 1][(BA || XF) ? STAT_ACC_REP : 1][2];
    float cst_raw[4], xfw_raw, xfb_raw;
    auto acc_loads = [&](const unsigned long long *acc) PHC_HS_INLINE {
        const unsigned long long *p0 = acc + tid;
#pragma unroll
        for (int qty = 0; qty < 2; ++qty)
#pragma unroll
            for (int rep = 0; rep < STAT_ACC_REP; ++rep) {
                const unsigned long long *p = p0 + ((size_t)(rep * 2 + qty) * 2) * KR;
                ba_raw[qty][rep][0] = hs_ld64(p);
                ba_raw[qty][rep][1] = hs_ld64(p + KR);
            }
    };
    if constexpr (XF) {                                           // an operand norm whose statistics are still totals
        static_assert(K <= THREADS, "one column per thread in the XF prologue");
        if (xa.acc && tid < KR) {
            acc_loads(xa.acc);
            const int comp = tid / xa.p.k, kk = tid - comp * xa.p.k;
            // (always two loads — the counted wait below relies on it; without affine parameters they read accumulator words)
            // (components laid out back to back — phc/hypercomplex/norm.py keeps them so —: the first table's pointer + the column; the
            // table walk costs a pointer round trip in front of the asm load, with a full drain of the ring's first fill)
            const float *pw = reinterpret_cast<const float *>(xa.acc), *pb = pw;
            if (xa.p.has_weight) pw = bn_col_ptr(xa.p.weight, xa.p.flat, comp, kk, tid);
            if (xa.p.has_bias) pb = bn_col_ptr(xa.p.bias, xa.p.flat, comp, kk, tid);
            xfw_raw = hs_ld32(pw);
            xfb_raw = hs_ld32(pb);
        }
    }
    if constexpr (BA) {
        static_assert(K <= THREADS, "one column per thread in the BA prologue");
        if (tid < KR) {
            acc_loads(a.ba_acc);
            cst_raw[0] = hs_ld32(a.ba_stat + tid);
            cst_raw[1] = hs_ld32(a.ba_stat + KR + tid);
            cst_raw[2] = hs_ld32(a.ba_stat + 4 * KR + tid);
            cst_raw[3] = hs_ld32(a.ba_stat + 5 * KR + tid);
        }
    }
    // what a wavefront issues between those loads and their first use: the ring's first fill (bias and operand slice follow
    // the constants)
    constexpr int ACC_YOUNGER = (R - 1) * IPW * (BA ? 2 : 1);
    static_assert(ACC_YOUNGER <= 63, "vmcnt is a 6-bit counter");
    auto acc_arrived = [&]() PHC_HS_INLINE {                                // counted wait + the registers tied behind it
        asm volatile("s_waitcnt vmcnt(%8)"
                     : "+v"(ba_raw[0][0][0]), "+v"(ba_raw[0][0][1]), "+v"(ba_raw[0][1][0]), "+v"(ba_raw[0][1][1]), "+v"(ba_raw[0][2][0]),
                       "+v"(ba_raw[0][2][1]), "+v"(ba_raw[0][3][0]), "+v"(ba_raw[0][3][1])
                     : "n"(ACC_YOUNGER)
                     : "memory");
        asm volatile("" : "+v"(ba_raw[1][0][0]), "+v"(ba_raw[1][0][1]), "+v"(ba_raw[1][1][0]), "+v"(ba_raw[1][1][1]), "+v"(ba_raw[1][2][0]),
                          "+v"(ba_raw[1][2][1]), "+v"(ba_raw[1][3][0]), "+v"(ba_raw[1][3][1]));
        if constexpr (BA) asm volatile("" : "+v"(cst_raw[0]), "+v"(cst_raw[1]), "+v"(cst_raw[2]), "+v"(cst_raw[3]));
        if constexpr (XF) asm volatile("" : "+v"(xfw_raw), "+v"(xfb_raw));
    };
    static_assert(STAT_ACC_REP == 4, "acc_arrived lists the accumulator's registers by name");
#pragma unroll
    for (int s = 0; s < R - 1; ++s) issue_step(s, s);

    const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
    // ---- XF: scale / shift of the 4 operand columns this lane fetches (fixed for the launch) ----
    f32x4_t xsc = {1.f, 1.f, 1.f, 1.f}, xsh = zero4;
    if constexpr (XF) {
        if (xa.acc) {
            // statistics still in an accumulator: column `tid` -> scale / shift in LDS (K <= THREADS or strided), workgroup 0
            // also leaves save_stat and the running statistics behind (the merge launch this replaces)
            const bool owner = blockIdx.x == 0;
            if (tid < KR) {
                const int col = tid;
                acc_arrived();
                const float xf_w = xa.p.has_weight ? xfw_raw : 1.f, xf_b = xa.p.has_bias ? xfb_raw : 0.f;
                double tq[2];
#pragma unroll
                for (int qty = 0; qty < 2; ++qty) {
                    long long hi = 0, lo = 0;
#pragma unroll
                    for (int rep = 0; rep < STAT_ACC_REP; ++rep) {
                        hi += (long long)ba_raw[qty][rep][0];
                        lo += (long long)ba_raw[qty][rep][1];
                    }
                    tq[qty] = (double)hi * (1.0 / 4096.0) + (double)lo * (1.0 / 4611686018427387904.0);
                }
                const double S = tq[0], Q = tq[1];
                const int comp = col / xa.p.k, kk = col - comp * xa.p.k;
                const float w = xf_w, bq = xf_b;
                const BnFromTotals bt = bn_from_totals(S, Q, n_valid, w, bq, xa.eps);
                xf_lds[col] = bt.scale;
                xf_lds[K + col] = bt.shift;
                if (owner) {
                    float *st = xa.stat_out;
                    st[col] = bt.mean; st[KR + col] = bt.invstd; st[2 * KR + col] = bt.scale; st[3 * KR + col] = bt.shift;
                    st[4 * KR + col] = w; st[5 * KR + col] = bq;
                    if (xa.p.has_running) {
                        float *rm = bn_col_ptr(xa.p.running_mean, xa.p.flat, comp, kk, col), *rv = bn_col_ptr(xa.p.running_var, xa.p.flat, comp, kk, col);
                        *rm = bn_running_update(*rm, xa.momentum, bt.mean);
                        *rv = bn_running_update(*rv, xa.momentum, bt.unbiased);
                    }
                }
            }
            if (owner && xa.p.has_nbt && tid < xa.p.n_comp)
                for (int q = 0; q < xa.p.n_comp; ++q)
                    if (q == tid) *xa.p.num_batches_tracked[q] += 1;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // (a raw barrier: no wait for the operand slice / ring fill)
            __builtin_amdgcn_s_barrier();
            xsc = *reinterpret_cast<const f32x4_t *>(xf_lds + 4 * dfetch);
            xsh = *reinterpret_cast<const f32x4_t *>(xf_lds + K + 4 * dfetch);
        } else {
            xsc = *reinterpret_cast<const f32x4_t *>(a.xf_stat + 2 * (size_t)KR + 4 * dfetch);
            xsh = *reinterpret_cast<const f32x4_t *>(a.xf_stat + 3 * (size_t)KR + 4 * dfetch);
        }
    }
    // ---- BA: per-column constants of the un-applied norm backward -> LDS (workgroup 0 also leaves dweight | dbias behind) ----
    if constexpr (BA) {
        if (tid < KR) {
            const int col = tid;
            acc_arrived();
            const float ba_cst[4] = {cst_raw[0], cst_raw[1], cst_raw[2], cst_raw[3]};
            double tot[2];
#pragma unroll
            for (int qty = 0; qty < 2; ++qty) {
                long long hi = 0, lo = 0;
#pragma unroll
                for (int rep = 0; rep < STAT_ACC_REP; ++rep) {
                    hi += (long long)ba_raw[qty][rep][0];
                    lo += (long long)ba_raw[qty][rep][1];
                }
                tot[qty] = (double)hi * (1.0 / 4096.0) + (double)lo * (1.0 / 4611686018427387904.0);
            }
            if (a.stamps) { asm volatile("" ::"v"(tot[0]), "v"(tot[1])); dbg_t[0] = __builtin_amdgcn_s_memrealtime(); }
            const float cnt = (float)max(a.ba_count > 0 ? a.ba_count : n_valid, 1LL);
            ba_lds[col] = ba_cst[1];
            ba_lds[K + col] = ba_cst[0];
            ba_lds[2 * K + col] = ba_cst[2];
            ba_lds[3 * K + col] = ba_cst[3];
            ba_lds[4 * K + col] = -((float)tot[1] / cnt);
            ba_lds[5 * K + col] = -((float)tot[0] / cnt);
            ba_lds[6 * K + col] = ba_cst[2] * ba_cst[1];
            if (blockIdx.x == 0 && a.ba_dwb) {
                a.ba_dwb[col] = (float)tot[1];
                a.ba_dwb[KR + col] = (float)tot[0];
            }
        }
        // (a raw barrier: __syncthreads() would also wait for the operand slice and the ring's first fill)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (a.stamps) dbg_t[1] = __builtin_amdgcn_s_memrealtime();
    }
    // ---- bias and operand slice: issued BEHIND the XF / BA constants (when there are any).  In front of them their 256 KB per
    // workgroup (every workgroup of the chip at once: the L2's bandwidth for 2-3 us) stood between the accumulator words and
    // their consumer: the constants' barrier came 7.5 us after entry (scripts/dbg/hs_phases.py, prologue detail) ----
    __builtin_amdgcn_sched_barrier(0);
    // (the accumulators start from the bias: its loads go in front of the operand slice's, or the first MFMA would wait for
    // the youngest load of the prologue and with it for all of them)
    f32x4_t bias4[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        // unconditional load (a null bias reads operand words and selects zero): the number of prologue loads is a constant
        // of the kernel — the first step's counted wait relies on it (PRO_LOADS below)
        const bool cok = col0 + 16 * t + 4 * kq < Nout;           // (a column tile past the operand's width: bias and operand read as 0)
        const f32x4_t v = *reinterpret_cast<const f32x4_t *>((a.bias ? a.bias : a.hk4) + (cok ? col0 + 16 * t + 4 * kq : 0));
        bias4[t] = (a.bias && cok) ? v : zero4;
    }
    __builtin_amdgcn_sched_barrier(0);                            // (hipcc would otherwise issue these loads last)
    // ---- operand slice -> registers (once per launch), k-group major: the first step consumes it in this order ----
    f32x4_t hreg[TPW][G];
    {
        const f32x4_t *src = reinterpret_cast<const f32x4_t *>(a.hk4);
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const bool ok = 4 * (4 * g + kq) < KR && col0 + c + 16 * t < Nout;     // always true at 128 / 256
                const f32x4_t v = src[ok ? (size_t)(4 * g + kq) * Nout + col0 + c + 16 * t : 0];
                hreg[t][g] = ok ? v : zero4;
            }
    }
    // fragment base of this lane per row block (float4 slots): row = rb * 16 + c
    int fbase[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int row = rb * 16 + c;
        fbase[rb] = (row / RPI) * ISTRIDE + (row % RPI) * (K / 4) + (kq ^ (row % RPI));
    }

    // ---- BW: norm constants of the 4 columns this lane owns per tile, running sums ----
    f32x4_t bw_mean[BW ? TPW : 1], bw_inv[BW ? TPW : 1], bw_w[BW ? TPW : 1], bw_b[BW ? TPW : 1];
    f32x4_t bw_sg[BW ? TPW : 1], bw_sgx[BW ? TPW : 1];
    if constexpr (BW) {
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int col = col0 + 16 * t + 4 * kq < Nout ? col0 + 16 * t + 4 * kq : 0;
            bw_mean[t] = *reinterpret_cast<const f32x4_t *>(a.bw_stat + col);
            bw_inv[t] = *reinterpret_cast<const f32x4_t *>(a.bw_stat + (size_t)Nout + col);
            bw_w[t] = *reinterpret_cast<const f32x4_t *>(a.bw_stat + 4 * (size_t)Nout + col);
            bw_b[t] = *reinterpret_cast<const f32x4_t *>(a.bw_stat + 5 * (size_t)Nout + col);
            bw_sg[t] = zero4;
            bw_sgx[t] = zero4;
        }
    }
    // ---- ST: per lane, per owned column: pivot (the lane's first counted value), sum and sum of squares of (v - pivot) ----
    f32x4_t st_p[ST ? TPW : 1], st_s1[ST ? TPW : 1], st_s2[ST ? TPW : 1];
    float st_n = 0.f;
    if constexpr (ST) {
#pragma unroll
        for (int t = 0; t < TPW; ++t) st_p[t] = st_s1[t] = st_s2[t] = zero4;
    }

    // The operand slice and the per-column constants must be COMPLETE in hipcc's own bookkeeping before the LOOP: a load it
    // still counts as pending at the loop header is waited for with vmcnt(0) at its first use in EVERY iteration — which would
    // also drain the DMA ring.  An empty asm that reads the registers makes hipcc place that wait where `settle` is called:
    // behind the FIRST step, which is peeled out of the loop and waits for each operand register where it first needs it —
    // so the slice's 256 KB per workgroup arrive under the first step's MFMAs instead of in front of them.
    auto settle = [&]() PHC_HS_INLINE {
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
#pragma unroll
            for (int g = 0; g < G; g += 8)
                asm volatile("" ::"v"(hreg[t][g]), "v"(hreg[t][g + 1]), "v"(hreg[t][g + 2]), "v"(hreg[t][g + 3]), "v"(hreg[t][g + 4]),
                             "v"(hreg[t][g + 5]), "v"(hreg[t][g + 6]), "v"(hreg[t][g + 7]));
            asm volatile("" ::"v"(bias4[t]));
            if constexpr (BW) asm volatile("" ::"v"(bw_mean[t]), "v"(bw_inv[t]), "v"(bw_w[t]), "v"(bw_b[t]));
        }
        if constexpr (XF) asm volatile("" ::"v"(xsc), "v"(xsh));
    };
    if (a.stamps) { st_t[1] = __builtin_amdgcn_s_memtime(); st_r[1] = __builtin_amdgcn_s_memrealtime(); }

    // Counted wait for DMA(j): the (R - 2) * IPW DMA instructions of the R - 2 steps behind it are younger than it at every
    // wait below (the tail re-issues clamped steps), so "at most that many operations outstanding" implies DMA(j) has landed
    // whatever number of stores / epilogue loads was issued in between (they share the counter and only make the wait stricter).
    constexpr int VM_KEEP = (R - 2) * IPW * (BA ? 2 : 1);
    // BW variants issue the DMA of step i + R - 1 at the END of step i instead of behind its barrier: their epilogue consumes
    // ordinary loads (the norm-input rows), and hipcc's wait for those covers every older operation — a DMA issued a moment ago
    // included.  A transform that waits for DMA(i + 1) in mid-step then has one step fewer behind it.
    constexpr bool LATE = BW;
    constexpr int VM_KEEP_MID = (R - 2 - (LATE ? 1 : 0)) * IPW * (BA ? 2 : 1);
    // The FIRST wait of the launch additionally has the prologue's own loads behind DMA(0): bias, operand slice (and the BW
    // constants) were issued after the ring's first fill — a lower bound of what is certainly younger is what the count needs.
    constexpr int PRO_LOADS = TPW + TPW * G + (BW ? 4 * TPW : 0);
    // (vmcnt is a 6-bit counter: a count past 63 saturates — a stricter wait, the first step then also waits for a few of the
    // prologue's own loads; the 192 x 3-tile configuration with BA | BW reaches 67)
    constexpr int FIRST_WAIT = VM_KEEP + PRO_LOADS > 63 ? 63 : VM_KEEP + PRO_LOADS;

    // Operand transform of THIS wavefront's own pieces of step `sl_step` (local index), in place in ring slot `tslot`
    // (XF: act(x * scale + shift); BA: the norm backward, also written out).  It runs one step AHEAD of the MFMAs: the pieces
    // of step i + 1 are transformed in front of the MFMAs of step i, so that their LDS round trip and the barrier that
    // publishes them never sit between two steps' matrix work.
    auto transform = [&](long long sl_step, int tslot) PHC_HS_INLINE {
        const long long step = group + sl_step * a.groups;
        if constexpr (XF) {
#pragma unroll
            for (int u = 0; u < IPW; ++u) {
                const int ii = wave + WAVES * u;
                const long long row = step * RS + ii * RPI + dsub;
                f32x4_t *p = ring + (size_t)tslot * SLOT_F4 + ii * ISTRIDE + lane;
                f32x4_t v = *p;
                const bool live = row < n_valid && dlive;
                if (a.xf_act == PHC_ACT_RELU) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = live ? fmaxf(fmaf(v[q], xsc[q], xsh[q]), 0.f) : 0.f;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = live ? act_fwd_rt(a.xf_act, fmaf(v[q], xsc[q], xsh[q])) : 0.f;
                }
                *p = v;
            }
        }
        if constexpr (BA) {
            const f32x4_t ci = *reinterpret_cast<const f32x4_t *>(ba_lds + 4 * dfetch), cnm = *reinterpret_cast<const f32x4_t *>(ba_lds + K + 4 * dfetch);
            const f32x4_t cw = *reinterpret_cast<const f32x4_t *>(ba_lds + 2 * K + 4 * dfetch), cb = *reinterpret_cast<const f32x4_t *>(ba_lds + 3 * K + 4 * dfetch);
            const f32x4_t cngx = *reinterpret_cast<const f32x4_t *>(ba_lds + 4 * K + 4 * dfetch), cng = *reinterpret_cast<const f32x4_t *>(ba_lds + 5 * K + 4 * dfetch);
            const f32x4_t c1 = *reinterpret_cast<const f32x4_t *>(ba_lds + 6 * K + 4 * dfetch);
            // The activation switch stays OUTSIDE the element loops (one straight-line body per kind): as a per-element select
            // chain hipcc turned it into ~20 scalar branches per element, 0.3-0.4 us of a 16-row step.
            auto rows = [&](auto kind) PHC_HS_INLINE {
                constexpr int KIND = decltype(kind)::value;           // 0 identity, 1 relu, 2 any other (runtime switch per element)
#pragma unroll
                for (int u = 0; u < IPW; ++u) {
                    const int ii = wave + WAVES * u;
                    const long long row = step * RS + ii * RPI + dsub;
                    f32x4_t *p = ring + (size_t)tslot * SLOT_F4 + ii * ISTRIDE + lane;
                    f32x4_t gv = *p;
                    const f32x4_t xv = p[IMG_F4];
                    f32x4_t o = zero4;
                    if (row < n_valid && dlive) {
                        if constexpr (DR) {                       // g := dout * mask (the forward's keep-and-scale factors)
                            const float4 f = drop_factors_v(dr_p, dr_same, dr_k, dr_site, dr_seed, dr_step, row, 4 * dchunk, KR);
                            gv[0] *= f.x; gv[1] *= f.y; gv[2] *= f.z; gv[3] *= f.w;
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            // dz = w invstd (g - mean g - x_hat mean(g x_hat)); x_hat and the pre-activation exactly as the separate
                            // norm backward (and the reference) round them — (x - mean) invstd, then fma — so that an element
                            // on the ReLU kink falls on the same side in every implementation of this pass
                            const float xh = (xv[q] - cnm[q]) * ci[q];
                            float g = gv[q];
                            if constexpr (KIND == 1)
                                g = fmaf(xh, cw[q], cb[q]) > 0.f ? g : 0.f;
                            else if constexpr (KIND == 2)
                                g *= act_grad_rt(a.ba_act, fmaf(xh, cw[q], cb[q]));
                            o[q] = c1[q] * (g + fmaf(xh, cngx[q], cng[q]));
                        }
                    }
                    *p = o;
                    if (row < a.M && dlive) *reinterpret_cast<f32x4_t *>(a.ba_dz + (size_t)row * KR + 4 * dchunk) = o;
                }
            };
            if (a.ba_act == PHC_ACT_RELU)
                rows(std::integral_constant<int, 1>{});
            else if (a.ba_act == PHC_ACT_IDENTITY)
                rows(std::integral_constant<int, 0>{});
            else
                rows(std::integral_constant<int, 2>{});
        }
    };
    if constexpr (XF || BA) {
        hs_wait_vm<FIRST_WAIT>();                        // DMA(0)
        if (a.stamps) dbg_t[2] = __builtin_amdgcn_s_memrealtime();
        transform(0, 0);
    }
    int slot = 0;
    unsigned long long ph[4] = {0, 0, 0, 0};                      // diagnostics: 100 MHz ticks in wait+barrier | issue+transform | MFMAs | epilogue
    auto step_body = [&](long long i, auto first) PHC_HS_INLINE {
        const long long step = group + i * a.groups;
        unsigned long long pt0 = 0, pt1 = 0, pt2 = 0, pt3 = 0;
        if (a.stamps) pt0 = __builtin_amdgcn_s_memrealtime();
        if constexpr (XF || BA)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the in-place writes of step i are done before the barrier releases readers
        else if constexpr (decltype(first)::value)
            hs_wait_vm<FIRST_WAIT>();                    // own DMA pieces of step 0 (the prologue's loads stay in flight)
        else
            hs_wait_vm<VM_KEEP>();                                // own DMA pieces of step i
        // after the barrier: every wavefront's pieces of step i are in place, the slot of step i - 1 is free
        __builtin_amdgcn_s_barrier();
        if (a.stamps) pt1 = __builtin_amdgcn_s_memrealtime();
        if constexpr (!LATE) {
            int nslot = slot + R - 1;
            nslot = nslot >= R ? nslot - R : nslot;
            issue_step(i + R - 1, nslot);
        }
        if constexpr (XF || BA) {
            if (i + 1 < my_live) {
                hs_wait_vm<VM_KEEP_MID>();                        // DMA(i + 1): the younger steps stay in flight behind it
                transform(i + 1, slot + 1 == R ? 0 : slot + 1);
            }
        }
        if (a.stamps) pt2 = __builtin_amdgcn_s_memrealtime();
        // BW: the norm-input rows the epilogue needs go in flight now, under the MFMAs (clamped, unconditional)
        f32x4_t bwx[BW ? RB : 1][BW ? TPW : 1];
        if constexpr (BW) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const long long rowc = min(step * RS + rb * 16 + c, a.M - 1);
#pragma unroll
                for (int t = 0; t < TPW; ++t)
                    bwx[rb][t] = *reinterpret_cast<const f32x4_t *>(a.bw_x + (size_t)rowc * Nout + (col0 + 16 * t + 4 * kq < Nout ? col0 + 16 * t + 4 * kq : 0));
            }
        }
        f32x4_t cw_mean[BW ? TPW : 1], cw_inv[BW ? TPW : 1], cw_w[BW ? TPW : 1], cw_b[BW ? TPW : 1];
        if constexpr (BW) {
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                if constexpr (decltype(first)::value || !BW_LDS) {
                    cw_mean[t] = bw_mean[t]; cw_inv[t] = bw_inv[t]; cw_w[t] = bw_w[t]; cw_b[t] = bw_b[t];
                } else {
                    const int lc = min(col0 + 16 * t + 4 * kq, HS_BW_COLS - 4);
                    cw_mean[t] = *reinterpret_cast<const f32x4_t *>(bw_lds + lc);
                    cw_inv[t] = *reinterpret_cast<const f32x4_t *>(bw_lds + HS_BW_COLS + lc);
                    cw_w[t] = *reinterpret_cast<const f32x4_t *>(bw_lds + 2 * HS_BW_COLS + lc);
                    cw_b[t] = *reinterpret_cast<const f32x4_t *>(bw_lds + 3 * HS_BW_COLS + lc);
                }
            }
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
        if (a.stamps) {
            asm volatile("s_nop 0" ::"v"(acc[0][0]));             // (the stamp waits for the last MFMA's result)
            pt3 = __builtin_amdgcn_s_memrealtime();
        }
        // ---- epilogue: lane (c, kq) holds y[row][col0 + 16 t + 4 kq .. +3] ----
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const long long row = step * RS + rb * 16 + c;
            if constexpr (ST) {
                // statistics of what is STORED: with a residual (PHMConv's self loop behind the transform, messagepassing.py:62-65)
                // it joins the accumulators in front of them
                if (a.residual && row < a.M) {
#pragma unroll
                    for (int t = 0; t < TPW; ++t)
                        if (col0 + 4 * kq + 16 * t < Nout)
                            acc[rb][t] += *reinterpret_cast<const f32x4_t *>(a.residual + (size_t)row * Nout + col0 + 4 * kq + 16 * t);
                }
                if (row < n_valid) {
                    if (st_n == 0.f) {
#pragma unroll
                        for (int t = 0; t < TPW; ++t) st_p[t] = acc[rb][t];
                    } else {
#pragma unroll
                        for (int t = 0; t < TPW; ++t) {
                            const f32x4_t d = acc[rb][t] - st_p[t];
                            st_s1[t] += d;
                            st_s2[t] += d * d;
                        }
                    }
                    st_n += 1.f;
                }
            }
            if constexpr (BW) {
                if (row < n_valid) {
                    auto sums = [&](auto kind) PHC_HS_INLINE {               // (the activation switch outside the element loops, as in `transform`)
                        constexpr int KIND = decltype(kind)::value;
#pragma unroll
                        for (int t = 0; t < TPW; ++t)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const float xh = (bwx[rb][t][q] - cw_mean[t][q]) * cw_inv[t][q];
                                const float z = fmaf(xh, cw_w[t][q], cw_b[t][q]);
                                const float v = acc[rb][t][q];
                                float gq;
                                if constexpr (KIND == 1)
                                    gq = z > 0.f ? v : 0.f;
                                else if constexpr (KIND == 0)
                                    gq = v;
                                else
                                    gq = v * act_grad_rt(a.bw_act, z);
                                bw_sg[t][q] += gq;
                                bw_sgx[t][q] = fmaf(gq, xh, bw_sgx[t][q]);
                            }
                    };
                    if (a.bw_act == PHC_ACT_RELU)
                        sums(std::integral_constant<int, 1>{});
                    else if (a.bw_act == PHC_ACT_IDENTITY)
                        sums(std::integral_constant<int, 0>{});
                    else
                        sums(std::integral_constant<int, 2>{});
                }
            }
            if (row < a.M) {
                float *yr = a.y + (size_t)row * Nout + col0 + 4 * kq;
#pragma unroll
                for (int t = 0; t < TPW; ++t) {
                    if (col0 + 4 * kq + 16 * t < Nout) {          // (a partial last tile: Nout % 4 == 0, so a lane's four columns stand or fall together)
                        f32x4_t v = acc[rb][t];
                        if constexpr (!ST)
                            if (a.residual) v += *reinterpret_cast<const f32x4_t *>(a.residual + (size_t)row * Nout + col0 + 4 * kq + 16 * t);
                        *reinterpret_cast<f32x4_t *>(yr + 16 * t) = v;
                    }
                }
            }
        }
        if constexpr (LATE) {                                     // into the slot step i - 1 was read from (free since this step's barrier)
            int nslot = slot + R - 1;
            nslot = nslot >= R ? nslot - R : nslot;
            issue_step(i + R - 1, nslot);
        }
        slot = slot + 1 == R ? 0 : slot + 1;
        if (a.stamps) {
            const unsigned long long pt4 = __builtin_amdgcn_s_memrealtime();
            ph[0] += pt1 - pt0; ph[1] += pt2 - pt1; ph[2] += pt3 - pt2; ph[3] += pt4 - pt3;
        }
    };
    step_body(0, std::true_type{});
    settle();
    // BW: from here on the norm constants of this lane's columns live in LDS (every lane parks its own copy; the 16 row lanes of a
    // column quad write the same words) and are re-read where a step's epilogue needs them.  Held in registers through the loop they
    // were 32 of 256 + 12 more: the <256, 2, 8, ..., BA | BW> instantiations reloaded three spilled values from scratch several times
    // per step — ordinary vector-memory loads in the middle of the ring's counted waits (zinc_b1024 1.845 -> 1.824 ms, molpcba_b1024
    // 2.076 -> 2.064 without them).  K = 256 only: the narrower instantiations have the registers (and at K = 128 the extra LDS
    // reads cost what they save: 0.5755 vs 0.5783 ms).
    if constexpr (BW_LDS) {
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int lc = min(col0 + 16 * t + 4 * kq, HS_BW_COLS - 4);
            *reinterpret_cast<f32x4_t *>(bw_lds + lc) = bw_mean[t];
            *reinterpret_cast<f32x4_t *>(bw_lds + HS_BW_COLS + lc) = bw_inv[t];
            *reinterpret_cast<f32x4_t *>(bw_lds + 2 * HS_BW_COLS + lc) = bw_w[t];
            *reinterpret_cast<f32x4_t *>(bw_lds + 3 * HS_BW_COLS + lc) = bw_b[t];
        }
    }
    for (long long i = 1; i < my_live; ++i) step_body(i, std::false_type{});
    hs_wait_vm<0>();                                              // the clamped tail DMAs must not outlive the workgroup's LDS
    // padding-only steps (XF / BA): y = bias, the norm backward's dz = 0
    for (long long i = my_live; i < my_steps; ++i) {
        const long long step = group + i * a.groups;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const long long row = step * RS + rb * 16 + c;
            if (row < a.M) {
#pragma unroll
                for (int t = 0; t < TPW; ++t)
                    if (col0 + 4 * kq + 16 * t < Nout) *reinterpret_cast<f32x4_t *>(a.y + (size_t)row * Nout + col0 + 4 * kq + 16 * t) = bias4[t];
            }
        }
        if constexpr (BA) {
            const int k4 = KR >> 2;
            for (int e = tid; e < RS * k4; e += THREADS) {
                const long long row = step * RS + e / k4;
                if (row < a.M) *reinterpret_cast<f32x4_t *>(a.ba_dz + (size_t)row * KR + 4 * (e % k4)) = zero4;
            }
        }
    }

    // ---- statistics: the 16 row lanes of a column group meet (fixed order), then two atomic instructions per wavefront ----
    if constexpr (ST) {
        double S[TPW][4], Q[TPW][4];
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // about zero, in double: sum = n p + s1 ; sum of squares = n p^2 + 2 p s1 + s2
                const double n = (double)st_n, p = (double)st_p[t][q], s1 = (double)st_s1[t][q], s2 = (double)st_s2[t][q];
                S[t][q] = hs_sum16(n * p + s1);
                Q[t][q] = hs_sum16(n * p * p + 2.0 * p * s1 + s2);
            }
        hs_publish<TPW>(a.stat_acc, Nout, (int)(group % STAT_ACC_REP), col0, kq, c, S, Q);
    }
    if constexpr (BW) {
        double sg[TPW][4], sgx[TPW][4];
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                sg[t][q] = hs_sum16((double)bw_sg[t][q]);
                sgx[t][q] = hs_sum16((double)bw_sgx[t][q]);
            }
        hs_publish<TPW>(a.stat_acc, Nout, (int)(group % STAT_ACC_REP), col0, kq, c, sg, sgx);
    }
    if (a.stamps) {
        st_t[2] = __builtin_amdgcn_s_memtime(); st_r[2] = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) {
            // [0..2] shader clock at entry / loop start / end, [3..5] the same in 100 MHz ticks, [6..9] per-phase tick sums,
            // [10] steps, [11] unused
            unsigned long long *o = a.stamps + ((size_t)blockIdx.x * WAVES + wave) * 12;
#pragma unroll
            for (int i = 0; i < 3; ++i) { o[i] = st_t[i]; o[3 + i] = st_r[i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i) o[6 + i] = ph[i];
            o[10] = (unsigned long long)my_steps;
            // [11] prologue detail (BA): ticks from entry until the accumulator words arrived (bits 0-19), the constants' barrier
            // (20-39), the first DMA image (40-59)
            o[11] = ((dbg_t[0] - st_r[0]) & 0xfffff) | (((dbg_t[1] - st_r[0]) & 0xfffff) << 20) | (((dbg_t[2] - st_r[0]) & 0xfffff) << 40);
        }
    }
}

}  // namespace phc
