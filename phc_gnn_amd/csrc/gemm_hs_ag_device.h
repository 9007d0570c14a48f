// Aggregation INSIDE the first GEMM of a message-passing layer on a molecule batch (round 4):
//     s[d] = sum_{e: dst_e = d} (x[src_e] + enc_e) (+ x[d])        messagepassing.py:132-146 (PHMGINEConv.propagate, GINE sum)
//     z    = s . H (+ b),  column statistics of z                   layers.py:198-219, 349-355 (PHMMLP.linear1 + its batch-norm's sums)
// with, optionally, the model-level norm of the layer in FRONT still un-applied:  x := act(norm(y)) (+ skip)   (models.py:208-215).
// Separately these were `aggregate_norm_direct_fwd_kernel` / `aggregate_encp_direct_fwd_kernel` (encoder.hip) followed by the
// H-stationary GEMM with the statistics epilogue (gemm_hs_device.h, HS_ST): two launches whose second streams straight back in what
// the first wrote.  Here the workgroup that owns a 16-row step BUILDS the step's operand image itself: every wavefront gathers two
// rows (one per half-wavefront: the row-per-lane-group walk of the aggregation kernels, same edge order, same additions — the rows
// are bit-identical), writes them into the LDS image in the layout the MFMA fragment reads expect, and also out to memory once (s is
// the input of the layer's parameter-gradient GEMM; x, when a norm was pending, is what every later reader of the layer input sees).
// Then the 16 x 128 x 128 product on the operand slice the wavefront holds in registers, exactly as gemm_hs16_kernel does it
// (v_mfma_f32_16x16x4_f32, swapped roles, k-permutation k = 16 g + 4 kq + q), and that kernel's statistics epilogue.
//
// Register budget decides the shape: 805 steps at 12.9 k rows want 512 workgroups resident (two per CU, 16 wavefronts per CU:
// 128 registers per lane), and a lane group's walk with a norm pending keeps 6 row loads + the norm constants in flight.  So
//   * 8 wavefronts x ONE 16-column tile (the 4 x 2 split of the streaming kernel would have a wavefront gather four rows);
//   * the operand matrix (64 KB, k4 form) sits in LDS — brought in by LDS-DMA at entry, no register round trip — instead of in the
//     wavefronts' registers: the fragment reads of both MFMA operands are ds_read_b128 (the operand's are conflict-free in the
//     k4 layout as it stands: the 16 lanes of a kq group read 256 contiguous bytes);
//   * the bond tables (13 rows on molecule data) are staged in LDS too: as global reads inside the walk they were 2 F more
//     float4 loads in flight per lane;
//   * rows with more than 16 in-edges (the aggregation kernels hand those to the whole wavefront, the lane groups taking alternate
//     edges) are walked by their own lane group in two strided passes whose partial sums are added in the same order: the same
//     bits, without a second inlined copy of the walk.
// The only hand-counted wait is the vmcnt(0) behind the operand's DMA, in front of the first barrier.
#pragma once
#include "gemm_hs_device.h"

namespace phc {

constexpr int HS_AG_MAX_F = 4;
constexpr int HS_AG_MAX_ROWS = 64;                             // bond-table rows staged in LDS (32 KB)
struct HsAgArgs {
    const float *y;                    // rows that are gathered: x itself (acc == null) or the input of the un-applied norm
    const float *skip;                 // residual behind act(norm(.)), or null (only with acc)
    const unsigned *codes;             // packed bond codes, CSR order (one byte per feature: table row id incl. the feature's offset)
    const float *const *tables;        // [n_comp * F] table pointers
    const int *rowptr, *col;           // destination-sorted view
    float *s_out;                      // [M, 128] the aggregate
    float *x_out;                      // [M, 128] the normalised rows (acc != null)
    float *stat;                       // [6 * 128] save_stat of the pending norm (acc != null)
    const unsigned long long *acc;     // the pending norm's statistics accumulator, or null
    BnParams p;
    float momentum, eps;
    int norm_act;
    int F, off[HS_AG_MAX_F + 1];       // features per edge, row offset of feature f in the packed row space
    int n, k, add_self;
};

// the wide layers' form of the same launch (gemm_hs_agw_device.h, compiled in gemm_agw.hip): widths 256, 180, 200 (square)
constexpr int HS_AGW_MAX_ROWS = 64;                               // bond-table rows it stages in LDS (64 KB at 256 columns)
bool gemm_agw_shape_ok(int h, int Nout, int vocab_rows);
int launch_gemm_agw(const GemmHsArgs &a, const HsAgArgs &ag, bool normed, hipStream_t s);

constexpr int HS_AG_THREADS = 512;
#ifndef PHC_AG_OCC
#define PHC_AG_OCC __attribute__((amdgpu_waves_per_eu(4, 4)))
#endif

template <int F, bool NORMED, bool RELU, bool FLAT>
__global__ __launch_bounds__(HS_AG_THREADS) PHC_AG_OCC void gemm_hs_ag_kernel(GemmHsArgs a, HsAgArgs ag) {
    constexpr int K = 128, G = K / 16, ISTRIDE = HS_ISTRIDE, INSTR = 8, H4 = K / 4, NOUT = 128;
    __shared__ f32x4_t image[INSTR * ISTRIDE];                   // one 16-row operand image (8 two-row "instructions", padded)
    __shared__ __attribute__((aligned(16))) float s_scale[K], s_shift[K];
    extern __shared__ __attribute__((aligned(16))) float ag_dyn[];       // [K / 4][128] float4 operand (k4 form) | [R][128] bond tables
    const f32x4_t *opnd = reinterpret_cast<const f32x4_t *>(ag_dyn);
    float *ag_tbl = ag_dyn + (size_t)K * NOUT;                   // row = packed code byte, columns component-major
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, kq = lane >> 4;
    const int col0 = wave * 16;
    const int group = (int)blockIdx.x;
    const int num_steps = (int)((a.M + 15) / 16);
    const int my_steps = group < num_steps ? (num_steps - group + a.groups - 1) / a.groups : 0;
    if (my_steps == 0) return;
    const int M = (int)a.M;
    // diagnostics (phc_debug_hs_stamps; null in the product): 100 MHz ticks at entry | first barrier passed | loop done | end, and per
    // phase summed over the steps: rows added up | image barrier | MFMAs + epilogue
    unsigned long long tk[4] = {0, 0, 0, 0}, ph[3] = {0, 0, 0}, dbg[4] = {0, 0, 0, 0};
    if (a.stamps) tk[0] = __builtin_amdgcn_s_memrealtime();
    const int count = a.n_valid ? min(*a.n_valid, M) : M;
    auto lds_barrier = [] {                                       // LDS traffic only: __syncthreads() would also wait for the s / x
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // stores of the step (vmcnt(0))
        __builtin_amdgcn_s_barrier();
    };

    // ---- this half-wavefront's row of a step: image instruction `wave`, row 2 wave + dsub; its 16-byte chunk (odd rows XOR 1) ----
    const int dsub = lane >> 5, li = lane & 31;
    const int c4 = li ^ dsub;
    auto row_of = [&](int sl) { return (group + sl * a.groups) * 16 + 2 * wave + dsub; };
    const float4 *Y4 = reinterpret_cast<const float4 *>(ag.y) + c4;
    const float4 *S4 = reinterpret_cast<const float4 *>(ag.skip ? ag.skip : ag.y) + c4;   // (a missing skip re-reads y and is not added)
    const bool has_skip = ag.skip != nullptr;
    // What a lane group needs of a step before it can add anything up: the row's edge range, the sources and codes of its first two
    // edges, and (RAW) the rows themselves as they lie in memory.  All of it is fetched BEFORE the launch's first barrier for the
    // first step, and down to the codes for the second: three dependent round trips that overlap the operand's DMA, the tables'
    // staging and the pending norm's prologue instead of following them.
    struct Raw {
        float4 v, rr;
    };
    auto load_raw = [&](int r) {
        Raw w;
        w.v = Y4[(size_t)r * H4];
        if constexpr (NORMED) w.rr = S4[(size_t)r * H4];
        else w.rr = w.v;
        return w;
    };
    constexpr int NPRE0 = NORMED ? 1 : 2;                         // steps whose pointers / sources / codes are fetched ahead
    int d_[2] = {0, 0}, e0_[2] = {0, 0}, e1_[2] = {0, 0}, s0_[2] = {0, 0}, s1_[2] = {0, 0};
    unsigned k0_[2] = {0, 0}, k1_[2] = {0, 0};
#pragma unroll
    for (int q = 0; q < NPRE0; ++q) {
        d_[q] = row_of(min(q, my_steps - 1));
        const int dc = d_[q] < M ? d_[q] : 0;
        e0_[q] = ag.rowptr[dc];
        e1_[q] = ag.rowptr[dc + 1];
    }
    // ---- bond tables: pointer hop now, rows below ----
    const int tblR = ag.off[F];
    const float *tsrc = nullptr;
    int tcomp = 0, tkk = 0, tr = 0, tf = 0, toff = 0;
    if (tid < tblR * H4) {                                       // (R <= 16 rows on molecule data: one float4 per thread; more rows: the loop below)
        tr = tid >> 5;
        const int q4 = tid & 31;
        tcomp = (q4 * 4) / ag.k;
        tkk = q4 * 4 - tcomp * ag.k;
#pragma unroll
        for (int t = 1; t < F; ++t) {                            // (constant indices into the by-value offset table: no scratch copy)
            const bool past = tr >= ag.off[t];
            tf = past ? t : tf;
            toff = past ? ag.off[t] : toff;
        }
        tsrc = ag.tables[tcomp * F + tf];
    }
    // pending norm: column ncol = tid / 4, accumulator replica tid % 4 (STAT_ACC_REP == 4) — every wavefront takes 16 columns, so the
    // accumulator's round trip and the double arithmetic are spread evenly instead of holding up the two wavefronts that would own
    // all 128 columns (they reached the first barrier 2 us behind the others, scripts/dbg/ag_phases.py).  The loads go out BEHIND
    // the sources / codes below: nothing on the row-pointer -> source -> row chain waits for them.
    static_assert(STAT_ACC_REP == 4 && HS_AG_THREADS == 4 * K, "one (column, replica) pair per thread");
    const int ncol = tid >> 2, nrep = tid & 3;
    const int ncomp = NORMED ? ncol / ag.p.k : 0, nkk = NORMED ? ncol - ncomp * ag.p.k : 0;
    const bool owner = blockIdx.x == 0;
    double nS = 0.0, nQ = 0.0;
    float nw = 1.f, nb = 0.f, rm_old = 0.f, rv_old = 1.f;
    // ---- second round trip: sources and codes of the first two edges of both steps (pointer-selected: an empty row reads row pointers).
    // With a norm pending only the first step's: at 128 registers per lane hipcc spilled the second step's as they arrived — a wait
    // per load, in the middle of the chain (scripts/dbg/ag_phases.py: 2 us) ----
    constexpr int NPRE = NORMED ? 1 : 2;
#pragma unroll
    for (int q = 0; q < NPRE; ++q) {
        const bool valid = d_[q] < M;
        e0_[q] = valid ? e0_[q] : 0;
        e1_[q] = valid ? e1_[q] : 0;
        const bool any = e0_[q] < e1_[q];
        const int stride = e1_[q] - e0_[q] > 16 ? 2 : 1;         // (AGG_HEAVY of encoder.hip: two strided passes)
        const int i0 = any ? min(e0_[q], e1_[q] - 1) : 0, i1 = any ? min(e0_[q] + stride, e1_[q] - 1) : 0;
        const int *cp = any ? ag.col : ag.rowptr;
        const unsigned *kp = any ? ag.codes : reinterpret_cast<const unsigned *>(ag.rowptr);
        s0_[q] = cp[i0];
        s1_[q] = cp[i1];
        k0_[q] = kp[i0];
        k1_[q] = kp[i1];
    }
    unsigned long long araw[4] = {0, 0, 0, 0};                   // this replica's words of (sum | sum of squares) x (hi | lo): converted below
    if constexpr (NORMED) {
        const unsigned long long *ap = ag.acc + (size_t)(nrep * 2) * 2 * K + ncol;      // (the layout of stat_acc_part, phc_common.h)
        araw[0] = ap[0]; araw[1] = ap[K]; araw[2] = ap[2 * K]; araw[3] = ap[3 * K];
        if (ag.p.has_weight) nw = FLAT ? ag.p.weight[0][ncol] : bn_pick(ag.p.weight, ncomp)[nkk];
        if (ag.p.has_bias) nb = FLAT ? ag.p.bias[0][ncol] : bn_pick(ag.p.bias, ncomp)[nkk];
        if (ag.p.has_running && owner) {
            rm_old = FLAT ? ag.p.running_mean[0][ncol] : bn_pick(ag.p.running_mean, ncomp)[nkk];
            rv_old = FLAT ? ag.p.running_var[0][ncol] : bn_pick(ag.p.running_var, ncomp)[nkk];
        }
    }
    if (a.stamps) dbg[1] = __builtin_amdgcn_s_memrealtime();
    float4 trow = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < tblR * H4) trow = *reinterpret_cast<const float4 *>(tsrc + (size_t)(tr - toff) * ag.k + tkk);
    // ---- the operand matrix -> LDS by LDS-DMA: 64 instructions of 1 KiB, eight per wavefront, a straight copy of the k4 form.
    // Issued HERE: memory operations of a wavefront complete in order, so everything issued behind the DMA waits for it — the two
    // dependent hops above would each have paid the 64 KB ----
    {
        const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(hs_lds_void *)ag_dyn);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int ii = wave * 8 + u;
            hs_dma16(a.hk4 + (size_t)ii * 256 + 4 * lane, lds0 + 1024u * (unsigned)ii);
        }
    }
    // ---- third round trip: the first step's rows ----
    const int self0 = ((NORMED || ag.add_self) && d_[0] < M) ? d_[0] : 0;
    float4 w0v, w0r, w1v, w1r, wsv, wsr;                         // (plain float4 variables: a struct modified in the step loop stayed in scratch)
    {
        const Raw a0 = load_raw(e0_[0] < e1_[0] ? s0_[0] : 0), a1 = load_raw(e0_[0] < e1_[0] ? s1_[0] : 0), as = load_raw(self0);
        w0v = a0.v; w0r = a0.rr; w1v = a1.v; w1r = a1.rr; wsv = as.v; wsr = as.rr;
    }
    // ---- LDS: tables, scale / shift ----
    if (tid < tblR * H4) reinterpret_cast<float4 *>(ag_tbl)[tid] = trow;
    for (int e = tid + HS_AG_THREADS; e < tblR * H4; e += HS_AG_THREADS) {       // (table rows past the first 16)
        const int r = e >> 5, q4 = e & 31;
        const int comp = (q4 * 4) / ag.k, kk = q4 * 4 - comp * ag.k;
        int f = 0, fo = 0;
#pragma unroll
        for (int t = 1; t < F; ++t) {
            const bool past = r >= ag.off[t];
            f = past ? t : f;
            fo = past ? ag.off[t] : fo;
        }
        reinterpret_cast<float4 *>(ag_tbl)[e] = *reinterpret_cast<const float4 *>(ag.tables[comp * F + f] + (size_t)(r - fo) * ag.k + kk);
    }
    // the norm's constants: their double arithmetic runs while the rows travel.  The four replicas' parts meet in the first lane of
    // the quad in replica order — ((p0 + p1) + p2) + p3, the order of the stand-alone kernels' loop: the same bits
    __builtin_amdgcn_sched_barrier(0);                           // (nothing below may be scheduled in front of the row loads above: the
                                                                 // conversion of the accumulator words would wait for them there)
    if constexpr (NORMED) {
        nS = (double)(long long)araw[0] * (1.0 / 4096.0) + (double)(long long)araw[1] * (1.0 / 4611686018427387904.0);
        nQ = (double)(long long)araw[2] * (1.0 / 4096.0) + (double)(long long)araw[3] * (1.0 / 4611686018427387904.0);
        const int qbase = lane & ~3;
        double S = __shfl(nS, qbase, 64), Q = __shfl(nQ, qbase, 64);
#pragma unroll
        for (int r = 1; r < 4; ++r) {
            S += __shfl(nS, qbase + r, 64);
            Q += __shfl(nQ, qbase + r, 64);
        }
        if (nrep == 0) {
            const BnFromTotals bt = bn_from_totals(S, Q, count, nw, nb, ag.eps);
            s_scale[ncol] = bt.scale;
            s_shift[ncol] = bt.shift;
            if (owner) {
                float *stat = ag.stat;
                stat[ncol] = bt.mean;
                stat[K + ncol] = bt.invstd;
                stat[2 * K + ncol] = bt.scale;
                stat[3 * K + ncol] = bt.shift;
                stat[4 * K + ncol] = nw;
                stat[5 * K + ncol] = nb;
                if (ag.p.has_running) {
                    *(FLAT ? ag.p.running_mean[0] + ncol : bn_pick(ag.p.running_mean, ncomp) + nkk) = bn_running_update(rm_old, ag.momentum, bt.mean);
                    *(FLAT ? ag.p.running_var[0] + ncol : bn_pick(ag.p.running_var, ncomp) + nkk) = bn_running_update(rv_old, ag.momentum, bt.unbiased);
                }
            }
        }
        if (owner && ag.p.has_nbt && tid < ag.p.n_comp)
            for (int q = 0; q < ag.p.n_comp; ++q)
                if (q == tid) *ag.p.num_batches_tracked[q] += 1;
    }
    if (a.stamps) dbg[2] = __builtin_amdgcn_s_memrealtime();
    hs_wait_vm<0>();                                             // the operand's DMA (hipcc does not see it) has landed
    if (a.stamps) dbg[3] = __builtin_amdgcn_s_memrealtime();
    lds_barrier();                                               // operand, tables (and scale / shift) are in LDS
    if (a.stamps) tk[1] = __builtin_amdgcn_s_memrealtime();
    const float4 *tb = reinterpret_cast<const float4 *>(ag_tbl) + c4;    // this lane's chunk of every table row
    // one row slice as the aggregation sees it: x, or act(y * scale + shift) (+ skip), 0 for a padding row of a normalised tensor
    auto xval = [&](const Raw &w, int r) {
        if constexpr (!NORMED) {
            return w.v;
        } else {
            const float4 v = w.v, rr = w.rr;
            const float4 sc = *reinterpret_cast<const float4 *>(s_scale + 4 * c4), sh = *reinterpret_cast<const float4 *>(s_shift + 4 * c4);
            float4 o;
            if constexpr (RELU) {                                // (one activation kind per instantiation: both bodies in one kernel cost registers)
                o.x = fmaxf(fmaf(v.x, sc.x, sh.x), 0.f); o.y = fmaxf(fmaf(v.y, sc.y, sh.y), 0.f);
                o.z = fmaxf(fmaf(v.z, sc.z, sh.z), 0.f); o.w = fmaxf(fmaf(v.w, sc.w, sh.w), 0.f);
            } else {
                o.x = act_fwd_rt(ag.norm_act, fmaf(v.x, sc.x, sh.x)); o.y = act_fwd_rt(ag.norm_act, fmaf(v.y, sc.y, sh.y));
                o.z = act_fwd_rt(ag.norm_act, fmaf(v.z, sc.z, sh.z)); o.w = act_fwd_rt(ag.norm_act, fmaf(v.w, sc.w, sh.w));
            }
            o.x = has_skip ? o.x + rr.x : o.x; o.y = has_skip ? o.y + rr.y : o.y;
            o.z = has_skip ? o.z + rr.z : o.z; o.w = has_skip ? o.w + rr.w : o.w;
            if (r >= count) o = make_float4(0.f, 0.f, 0.f, 0.f);
            return o;
        }
    };
    auto message = [&](const Raw &w, int s, unsigned code) {
        float4 v = xval(w, s);
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const float4 t = tb[((code >> (8 * f)) & 255u) * H4];
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        return v;
    };
    // Add up this wavefront's two rows of a step from the prefetched pieces: LDS image + s (+ x) in memory.  The walk is the
    // aggregation kernels' `row` body: edges e0 + first, + stride, ... of [e0, e1), the first two in one round trip, partial sums
    // acc0 / acc1.  A row with more than 16 in-edges takes two passes (first = 0, 1; stride 2) whose results are added — what the
    // two lane groups of a wavefront compute and exchange there.
    auto finish = [&](int d, int e0, int e1, int s0, int s1, unsigned k0, unsigned k1, const float4 r0v, const float4 r0r, const float4 r1v,
                      const float4 r1r, const float4 rsv, const float4 rsr) {
        const bool valid = d < M;
        const int stride = e1 - e0 > 16 ? 2 : 1;
        const int selfr = ((NORMED || ag.add_self) && valid) ? d : 0;
        const float4 selfv = xval(Raw{rsv, rsr}, selfr);
        float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int first = 0; first < stride; ++first) {
            float4 acc0 = make_float4(0.f, 0.f, 0.f, 0.f), acc1 = acc0;
            int e = e0 + first, nth = 0;
            if (first == 0 && e0 < e1) {                         // the prefetched pair: the first two edges travel together
                const float4 v0 = message(Raw{r0v, r0r}, s0, k0), v1 = message(Raw{r1v, r1r}, s1, k1);
                if (e < e1) { acc0.x += v0.x; acc0.y += v0.y; acc0.z += v0.z; acc0.w += v0.w; }
                if (e + stride < e1) { acc1.x += v1.x; acc1.y += v1.y; acc1.z += v1.z; acc1.w += v1.w; }
                e += 2 * stride;
                nth = 2;
            }
            // the rest, one at a time; the SECOND edge of a pass goes to acc1 (second pass of a long row: same partial sums as the
            // pair above would give)
            for (; e < e1; e += stride, ++nth) {
                const int t0 = ag.col[e];
                const unsigned c0 = ag.codes[e];
                const Raw q0 = load_raw(t0);
                const float4 v = message(q0, t0, c0);
                if (nth == 1) { acc1.x += v.x; acc1.y += v.y; acc1.z += v.z; acc1.w += v.w; }
                else { acc0.x += v.x; acc0.y += v.y; acc0.z += v.z; acc0.w += v.w; }
            }
            const float4 a4 = make_float4(acc0.x + acc1.x, acc0.y + acc1.y, acc0.z + acc1.z, acc0.w + acc1.w);
            // (one pass: tot = a4.  two passes: a4(first 0) + a4(first 1), the shuffle-add of the shared walk)
            tot = first == 0 ? a4 : make_float4(tot.x + a4.x, tot.y + a4.y, tot.z + a4.z, tot.w + a4.w);
        }
        if (ag.add_self) { tot.x += selfv.x; tot.y += selfv.y; tot.z += selfv.z; tot.w += selfv.w; }
        if (!valid) tot = make_float4(0.f, 0.f, 0.f, 0.f);
        image[wave * ISTRIDE + lane] = f32x4_t{tot.x, tot.y, tot.z, tot.w};
        if (valid) {
            // buffer stores: base in scalar registers, the lane's part a 32-bit offset (M <= 2^20 rows of 512 bytes).  As pointer
            // arithmetic hipcc kept `s_out + lane chunk` and `x_out + lane chunk` as two 64-bit register pairs through the loop,
            // spilled the second one (128 registers per lane: two workgroups per CU) and reloaded it in front of every row's store
            // behind `s_waitcnt vmcnt(0)` — which also drained the row gathers issued ahead.
            using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
            const int off = d * (H4 * 16) + c4 * 16;
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(tot.x), __float_as_uint(tot.y), __float_as_uint(tot.z), __float_as_uint(tot.w)},
                                                   __builtin_amdgcn_make_buffer_rsrc((void *)ag.s_out, 0, 0x7ffffff0, 0x00020000), off, 0, 0);
            if constexpr (NORMED)
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(selfv.x), __float_as_uint(selfv.y), __float_as_uint(selfv.z), __float_as_uint(selfv.w)},
                                                       __builtin_amdgcn_make_buffer_rsrc((void *)ag.x_out, 0, 0x7ffffff0, 0x00020000), off, 0, 0);
        }
    };

    const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4_t bias4 = zero4;
    if (a.bias) bias4 = *reinterpret_cast<const f32x4_t *>(a.bias + col0 + 4 * kq);
    const int fbase = (c >> 1) * ISTRIDE + (c & 1) * H4 + (kq ^ (c & 1));
    const f32x4_t *hfrag = opnd + kq * NOUT + col0 + c;          // operand fragment of k-group g: hfrag[4 g * NOUT]
    f32x4_t st_p = zero4, st_s1 = zero4, st_s2 = zero4;
    float st_n = 0.f;
    int cd = d_[0], ce0 = e0_[0], ce1 = e1_[0], cs0 = s0_[0], cs1 = s1_[0];
    unsigned ck0 = k0_[0], ck1 = k1_[0];
    for (int i = 0; i < my_steps; ++i) {
        const int step = group + i * a.groups;
        if (i > 0) {
            // the rows of a later step: sources / codes of the second step were fetched with the first's, a third one's are not
            cd = d_[1]; ce0 = e0_[1]; ce1 = e1_[1]; cs0 = s0_[1]; cs1 = s1_[1]; ck0 = k0_[1]; ck1 = k1_[1];
            if (i >= NPRE) {                                     // (not fetched ahead: a norm pending, or launches past 512 x 32 rows)
                cd = row_of(i);
                const bool valid = cd < M;
                const int dc = valid ? cd : 0;
                ce0 = valid ? ag.rowptr[dc] : 0;
                ce1 = valid ? ag.rowptr[dc + 1] : 0;
                const bool any = ce0 < ce1;
                const int stride = ce1 - ce0 > 16 ? 2 : 1;
                const int i0 = any ? min(ce0, ce1 - 1) : 0, i1 = any ? min(ce0 + stride, ce1 - 1) : 0;
                const int *cp = any ? ag.col : ag.rowptr;
                const unsigned *kp = any ? ag.codes : reinterpret_cast<const unsigned *>(ag.rowptr);
                cs0 = cp[i0]; cs1 = cp[i1]; ck0 = kp[i0]; ck1 = kp[i1];
            }
            const int selfr = ((NORMED || ag.add_self) && cd < M) ? cd : 0;
            const Raw a0 = load_raw(ce0 < ce1 ? cs0 : 0), a1 = load_raw(ce0 < ce1 ? cs1 : 0), as = load_raw(selfr);
            w0v = a0.v; w0r = a0.rr; w1v = a1.v; w1r = a1.rr; wsv = as.v; wsr = as.rr;
        }
        unsigned long long p0 = 0, p1 = 0, p2 = 0;
        if (a.stamps) p0 = __builtin_amdgcn_s_memrealtime();
        finish(cd, ce0, ce1, cs0, cs1, ck0, ck1, w0v, w0r, w1v, w1r, wsv, wsr);     // (ONE call site: the walk is a lot of code)
        if (a.stamps) p1 = __builtin_amdgcn_s_memrealtime();
        lds_barrier();                                           // the step's image is complete
        if (a.stamps) p2 = __builtin_amdgcn_s_memrealtime();
        f32x4_t acc = bias4;
        f32x4_t fr[2], hr[2];
        fr[0] = image[fbase];
        hr[0] = hfrag[0];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (g + 1 < G) {
                fr[(g + 1) & 1] = image[fbase + 4 * (g + 1)];
                hr[(g + 1) & 1] = hfrag[4 * (g + 1) * NOUT];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(hr[g & 1][q], fr[g & 1][q], acc, 0, 0, 0);
        }
        const int row = step * 16 + c;
        if (row < count) {
            if (st_n == 0.f) {
                st_p = acc;
            } else {
                const f32x4_t dlt = acc - st_p;
                st_s1 += dlt;
                st_s2 += dlt * dlt;
            }
            st_n += 1.f;
        }
        if (row < M) *reinterpret_cast<f32x4_t *>(a.y + (size_t)row * NOUT + col0 + 4 * kq) = acc;
        if (i + 1 < my_steps) lds_barrier();                     // every wavefront has read the image
        if (a.stamps) {
            asm volatile("s_nop 0" ::"v"(acc));
            const unsigned long long p3 = __builtin_amdgcn_s_memrealtime();
            ph[0] += p1 - p0; ph[1] += p2 - p1; ph[2] += p3 - p2;
        }
    }
    if (a.stamps) tk[2] = __builtin_amdgcn_s_memrealtime();
    // ---- statistics of the output columns (gemm_hs16_kernel's HS_ST finale) ----
    {
        double S[1][4], Q[1][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double n = (double)st_n, p = (double)st_p[q], s1 = (double)st_s1[q], s2 = (double)st_s2[q];
            S[0][q] = hs_sum16(n * p + s1);
            Q[0][q] = hs_sum16(n * p * p + 2.0 * p * s1 + s2);
        }
        hs_publish<1>(a.stat_acc, NOUT, (int)(group % STAT_ACC_REP), col0, kq, c, S, Q);
    }
    if (a.stamps) {
        tk[3] = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) {
            unsigned long long *o = a.stamps + ((size_t)blockIdx.x * 8 + wave) * 12;
            o[0] = tk[0]; o[1] = tk[1]; o[2] = tk[2]; o[3] = tk[3];
            o[4] = ph[0]; o[5] = ph[1]; o[6] = ph[2]; o[7] = (unsigned long long)my_steps;
            o[8] = dbg[0]; o[9] = dbg[1]; o[10] = dbg[2]; o[11] = dbg[3];
        }
    }
}

}  // namespace phc
