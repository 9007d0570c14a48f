// Read-out side of the step: NaN-masked BCE-with-logits and the gated per-graph sum of
// the soft-attention pooling, each as one forward and one backward kernel instead of a chain of
// elementwise / reduce launches.
//
// Reference: benchmarks/train_hiv.py:174-178 (mask = ~isnan(y); BCEWithLogits(logits[mask], y[mask])),
//            phc/hypercomplex/pooling.py:57-66 (x * sigmoid(gate) broadcast over components, global_add_pool).
#include "phc_common.h"

namespace phc {

constexpr int RO_THREADS = 1024;

// ---- masked BCE -----------------------------------------------------------------------------------
// partial[b] = (sum of per-entry losses, number of labeled entries) of workgroup b's strided share;
// with one workgroup the result goes straight to loss / stat.
__device__ __forceinline__ float bce_entry(float z, float y) {
    return fmaxf(z, 0.f) - z * y + log1pf(expf(-fabsf(z)));
}

__device__ __forceinline__ void block_reduce2(float &a, float &b, float *red) {
    const int tid = threadIdx.x;
    red[tid] = a;
    red[RO_THREADS + tid] = b;
    __syncthreads();
    for (int st = RO_THREADS / 2; st > 0; st >>= 1) {       // fixed tree: reproducible
        if (tid < st) {
            red[tid] += red[tid + st];
            red[RO_THREADS + tid] += red[RO_THREADS + tid + st];
        }
        __syncthreads();
    }
    a = red[0];
    b = red[RO_THREADS];
}

// KIND 0: BCE with logits (train_hiv.py:174-178); KIND 1: L1 (train_zinc.py: torch.nn.L1Loss) — both as the mean over the labeled
// (non-NaN) entries
template <int KIND>
__device__ __forceinline__ float loss_entry(float z, float yv) {
    return KIND == 0 ? bce_entry(z, yv) : fabsf(z - yv);
}
template <int KIND>
__device__ __forceinline__ float loss_grad(float z, float yv) {
    if (KIND == 0) return 1.f / (1.f + expf(-z)) - yv;
    const float d = z - yv;
    return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);              // torch's sign(0) = 0
}

template <int KIND>
__global__ __launch_bounds__(RO_THREADS) void bce_masked_fwd_kernel(const float *__restrict__ logits, const float *__restrict__ y,
                                                                  long long total, float *__restrict__ partial,
                                                                  float *__restrict__ loss, float *__restrict__ stat) {
    __shared__ float red[2 * RO_THREADS];
    float s = 0.f, c = 0.f;
    for (long long e = (long long)blockIdx.x * RO_THREADS + threadIdx.x; e < total; e += (long long)gridDim.x * RO_THREADS) {
        const float yv = y[e];
        if (yv == yv) {                                       // labeled (not NaN)
            s += loss_entry<KIND>(logits[e], yv);
            c += 1.f;
        }
    }
    block_reduce2(s, c, red);
    if (threadIdx.x == 0) {
        if (gridDim.x == 1) {
            stat[0] = s;
            stat[1] = c;
            *loss = s / c;                                    // no labeled entry: NaN, like the mean of an empty selection
        } else {
            partial[2 * blockIdx.x] = s;
            partial[2 * blockIdx.x + 1] = c;
        }
    }
}

// One workgroup (total <= 4096): the loss AND, from the same labeled count, d loss / d logits for an upstream gradient
// of exactly 1 — what the backward of a scalar training loss receives — so that backward needs no launch of its own.
template <int KIND>
__global__ __launch_bounds__(RO_THREADS) void bce_masked_fwd_unit_kernel(const float *__restrict__ logits, const float *__restrict__ y,
                                                                       long long total, float *__restrict__ loss,
                                                                       float *__restrict__ stat, float *__restrict__ dlogits_unit) {
    __shared__ float red[2 * RO_THREADS];
    float s = 0.f, c = 0.f;
    for (long long e = threadIdx.x; e < total; e += RO_THREADS) {
        const float yv = y[e];
        if (yv == yv) {
            s += loss_entry<KIND>(logits[e], yv);
            c += 1.f;
        }
    }
    block_reduce2(s, c, red);
    if (threadIdx.x == 0) {
        stat[0] = s;
        stat[1] = c;
        *loss = s / c;
    }
    const float scale = 1.f / c;
    for (long long e = threadIdx.x; e < total; e += RO_THREADS) {
        const float yv = y[e];
        dlogits_unit[e] = yv == yv ? loss_grad<KIND>(logits[e], yv) * scale : 0.f;
    }
}

__global__ __launch_bounds__(RO_THREADS) void bce_masked_finalize_kernel(const float *__restrict__ partial, int num_partials,
                                                                       float *__restrict__ loss, float *__restrict__ stat) {
    __shared__ float red[2 * RO_THREADS];
    float s = 0.f, c = 0.f;
    for (int b = threadIdx.x; b < num_partials; b += RO_THREADS) {
        s += partial[2 * b];
        c += partial[2 * b + 1];
    }
    block_reduce2(s, c, red);
    if (threadIdx.x == 0) {
        stat[0] = s;
        stat[1] = c;
        *loss = s / c;
    }
}

// dlogits = dloss * (sigmoid(z) - y) / count on labeled entries, 0 elsewhere (L1: sign(z - y) in place of the difference)
template <int KIND>
__global__ __launch_bounds__(256) void bce_masked_bwd_kernel(const float *__restrict__ logits, const float *__restrict__ y,
                                                           const float *__restrict__ dloss, const float *__restrict__ stat,
                                                           long long total, float *__restrict__ dlogits) {
    const float scale = *dloss / stat[1];
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const float yv = y[e];
        float g = 0.f;
        if (yv == yv) {
            g = loss_grad<KIND>(logits[e], yv) * scale;
        }
        dlogits[e] = g;
    }
}

// ---- gated per-graph sum --------------------------------------------------------------------------
// out[g, c*k + j] = sum_{r in segment g} x[r, c*k + j] * sigmoid(gate[r, j])      (rows grouped by graph: seg_ptr)
// One workgroup per graph: thread -> (column slot, row lane); 8 row lanes walk the graph's rows and are summed in
// lane order through LDS: fixed summation order.  h <= GP_MAX_H per pass (wider rows loop over column blocks).
constexpr int GP_THREADS = 256;

// First index in [0, n) of the SORTED int64 vector `v` with v[i] >= key (n if none), by the whole workgroup: 256
// probes per level (two levels cover 65 536 entries), each level one round trip — a per-thread binary search would
// chain ~14 dependent loads.  `slot` is one int of shared memory.
__device__ __forceinline__ int coop_lower_bound(const long long *__restrict__ v, int n, long long key, int *slot) {
    int lo = 0, hi = n;                                       // the answer lies in [lo, hi]
    while (hi > lo) {
        const int len = hi - lo, stride = (len + GP_THREADS - 1) / GP_THREADS;
        const int i = lo + (int)threadIdx.x * stride;
        const bool ge = i < hi && v[i] >= key;
        if (threadIdx.x == 0) *slot = GP_THREADS;
        __syncthreads();
        if (ge) atomicMin(slot, (int)threadIdx.x);
        __syncthreads();
        const int first = *slot;                              // first probe that is >= key (GP_THREADS: none)
        __syncthreads();
        const int probes = (len + stride - 1) / stride;       // probes that fell inside [lo, hi)
        const int nlo = first == 0 ? lo : lo + (min(first, probes) - 1) * stride + 1;
        const int nhi = first >= probes ? hi : lo + first * stride;
        if (stride == 1) return first >= probes ? hi : lo + first;
        lo = nlo;
        hi = nhi;
    }
    return lo;
}

template <int VEC>
__global__ __launch_bounds__(GP_THREADS) void gated_pool_fwd_kernel(const float *__restrict__ x, const float *__restrict__ gate,
                                                                  const int *__restrict__ seg_ptr, int h, int k,
                                                                  float *__restrict__ out,
                                                                  const long long *__restrict__ batch_sorted, int n_rows) {
    __shared__ float red[VEC * GP_THREADS];
    __shared__ int slot;
    const int g = blockIdx.x, tid = threadIdx.x;
    // rows of graph g: from the segment pointers, or (seg_ptr == NULL) found in the sorted `batch` vector itself — the
    // separate launch that turns `batch` into pointers is then not needed
    int r0, r1;
    if (seg_ptr) {
        r0 = seg_ptr[g];
        r1 = seg_ptr[g + 1];
    } else {
        r0 = coop_lower_bound(batch_sorted, n_rows, (long long)g, &slot);
        r1 = coop_lower_bound(batch_sorted, n_rows, (long long)g + 1, &slot);
    }
    const int hv = h / VEC;
    for (int cbase = 0; cbase < hv; cbase += GP_THREADS) {
        const int cols = min(GP_THREADS, hv - cbase);
        const int lanes = GP_THREADS / cols;
        const int cs = tid % cols, rl = tid / cols;
        const int col = (cbase + cs) * VEC;
        const int j = col % k;                               // VEC | k: the VEC columns share one component
        float acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
        if (rl < lanes && r0 < r1) {
            // four rows of this lane per round trip (clamped addresses, selected afterwards; same ascending row order)
            for (int rb = r0 + rl; rb < r1; rb += 4 * lanes) {
                float xv[4][VEC], gv[4][VEC];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = min(rb + q * lanes, r1 - 1);
                    if constexpr (VEC == 4) {
                        const float4 a = *reinterpret_cast<const float4 *>(x + (size_t)r * h + col);
                        const float4 b = *reinterpret_cast<const float4 *>(gate + (size_t)r * k + j);
                        xv[q][0] = a.x; xv[q][1] = a.y; xv[q][2] = a.z; xv[q][3] = a.w;
                        gv[q][0] = b.x; gv[q][1] = b.y; gv[q][2] = b.z; gv[q][3] = b.w;
                    } else {
                        xv[q][0] = x[(size_t)r * h + col];
                        gv[q][0] = gate[(size_t)r * k + j];
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int v = 0; v < VEC; ++v)
                        acc[v] = rb + q * lanes < r1 ? fmaf(xv[q][v], 1.f / (1.f + expf(-gv[q][v])), acc[v]) : acc[v];
            }
        }
        __syncthreads();
#pragma unroll
        for (int v = 0; v < VEC; ++v) red[v * GP_THREADS + tid] = acc[v];
        __syncthreads();
        for (int e = tid; e < cols * VEC; e += GP_THREADS) {
            const int cc = e / VEC, v = e - cc * VEC;
            float a = 0.f;
            for (int l = 0; l < lanes; ++l) a += red[v * GP_THREADS + l * cols + cc];          // fixed order
            out[(size_t)g * h + (cbase + cc) * VEC + v] = a;
        }
    }
}

// ---- the whole read-out forward as ONE launch (round 4) ---------------------------------------------------------------------------
//   x    = act(norm(y)) (+ skip)            the last layer's un-applied norm (models.py:208-215), or x as it is
//   gate = x . M + c                        the composed gate map Linear(PHMLinear(x)) (pooling.py:57-66), M [128, 32]
//   out[g, c k + j] = sum_{r in graph g} x[r, c k + j] sigmoid(gate[r, j])
// Separately: bn_apply_acc_kernel -> the gate GEMM -> gated_pool_fwd_kernel, three launches over the same 12.9 k x 128 rows (34 us of the
// molecule step).  Here a workgroup owns a GRAPH (its rows found in the sorted batch vector by two cooperative searches in lockstep,
// beside the norm's accumulator prologue): 16 rows at a time go into an LDS image in the MFMA fragment layout (gemm_hs_ag_device.h:
// two rows per 1 KiB instruction, odd row XOR 1), two wavefronts multiply the image with the gate operand they hold in registers
// (16 x 128 x 32 by v_mfma_f32_16x16x4_f32, swapped roles), sigmoid(gate) goes to LDS, and every thread adds its column's products.
// x and gate are written out once (the backward kernels read them, unchanged).  h = 128, k = 32.
constexpr int RF_THREADS = 512;
struct ReadoutFusedArgs {
    const float *y, *skip;
    const unsigned long long *acc;
    BnParams p;
    float momentum, eps;
    int norm_act;
    float *stat, *x_out;
    const int *n_valid;
    const long long *batch;
    int N, num_graphs;
    const float *M, *c;
    float *gate, *out;
    unsigned long long *stamps;      // diagnostics (phc_debug_readout_stamps): 8 words per workgroup, null in the product
};
using rf_f32x4 = __attribute__((ext_vector_type(4))) float;
// LDS traffic only: __syncthreads() also waits for every global load and store in flight (vmcnt(0)) — the next tile's rows, fetched
// ahead on purpose, and the x / gate rows on their way out
__device__ __forceinline__ void rf_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// K = 128: two rows per 1 KiB image "instruction" (half-wavefront per row, odd row XOR 1), one pass of the 8 wavefronts per tile;
// K = 256: one row per instruction (a wavefront per row), two passes.  KG = gate width = columns per component (k): 32 or 64.
template <int K, int KG, bool NORMED, bool RELU, bool FLAT>
__global__ __launch_bounds__(RF_THREADS) void readout_fused_fwd_kernel(ReadoutFusedArgs a) {
    static_assert((K == 128 || K == 256) && (KG == 32 || KG == 64), "read-out widths");
    constexpr int H4 = K / 4, ISTRIDE = 66, TILES = KG / 16;
    // the contraction of a gate tile is split over KSPLIT wavefronts (128 rows of the operand each: 32 registers per lane; all 256 in
    // one wavefront would be 64 and push the kernel past two workgroups per CU); the halves meet in LDS
    constexpr int KSPLIT = K / 128, G = 8;
    constexpr int RPI = 256 / K, PASSES = 2 / RPI;                  // rows per image instruction, gather passes per tile
    constexpr int PARTS = RF_THREADS / K, PAIRS = 4 * K / RF_THREADS;   // row parts of the column sums; (column, replica) pairs per thread
    __shared__ rf_f32x4 image[(16 / RPI) * ISTRIDE];
    __shared__ __attribute__((aligned(16))) float s_scale[K], s_shift[K];
    __shared__ float sgate[16][KG + 1];
    __shared__ rf_f32x4 khalf[KSPLIT == 2 ? TILES * 64 : 1];        // the second half's partial gate tile, one float4 per lane
    __shared__ float red[PARTS][K];
    __shared__ int slot[2];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, kq = lane >> 4;
    const int N = a.N;
    unsigned long long tk[6] = {0, 0, 0, 0, 0, 0};
    if (a.stamps) tk[0] = __builtin_amdgcn_s_memrealtime();
    // ---- first level of the search for this graph's rows: both keys probe the same positions, ONE load, in flight before everything
    // else (memory operations complete in order: behind the accumulator words and the gate operand it came back 9 us after entry) ----
    const int stride1 = max((N + RF_THREADS - 1) / RF_THREADS, 1);
    const int i1 = tid * stride1;
    const long long v1 = i1 < N ? a.batch[i1] : 0x7fffffffffffffffLL;
    const int nvr = a.n_valid ? *a.n_valid : N;                  // (behind the probe: the first thing waited for is the probe)
    // ---- pending norm: (column, replica) per thread, as gemm_hs_ag_kernel ----
    const int nrep = tid & 3;
    const bool owner = blockIdx.x == 0;
    int ncol[PAIRS], ncomp[PAIRS], nkk[PAIRS];
    unsigned long long araw[PAIRS][4];
    float nw[PAIRS], nb[PAIRS], rm_old[PAIRS], rv_old[PAIRS];
#pragma unroll
    for (int pp = 0; pp < PAIRS; ++pp) {
        ncol[pp] = (tid + RF_THREADS * pp) >> 2;
        ncomp[pp] = NORMED ? ncol[pp] / a.p.k : 0;
        nkk[pp] = NORMED ? ncol[pp] - ncomp[pp] * a.p.k : 0;
        araw[pp][0] = araw[pp][1] = araw[pp][2] = araw[pp][3] = 0;
        nw[pp] = 1.f; nb[pp] = 0.f; rm_old[pp] = 0.f; rv_old[pp] = 1.f;
        if constexpr (NORMED) {
            const unsigned long long *ap = a.acc + (size_t)(nrep * 2) * 2 * K + ncol[pp];
            araw[pp][0] = ap[0]; araw[pp][1] = ap[K]; araw[pp][2] = ap[2 * K]; araw[pp][3] = ap[3 * K];
            if (a.p.has_weight) nw[pp] = FLAT ? a.p.weight[0][ncol[pp]] : bn_pick(a.p.weight, ncomp[pp])[nkk[pp]];
            if (a.p.has_bias) nb[pp] = FLAT ? a.p.bias[0][ncol[pp]] : bn_pick(a.p.bias, ncomp[pp])[nkk[pp]];
            if (a.p.has_running && owner) {
                rm_old[pp] = FLAT ? a.p.running_mean[0][ncol[pp]] : bn_pick(a.p.running_mean, ncomp[pp])[nkk[pp]];
                rv_old[pp] = FLAT ? a.p.running_var[0][ncol[pp]] : bn_pick(a.p.running_var, ncomp[pp])[nkk[pp]];
            }
        }
    }
    // A workgroup takes graphs blockIdx.x, + gridDim.x, ...: the launch has at most as many workgroups as are resident at once (two per
    // CU), so nobody waits for a slot, and a workgroup's later graphs find the batch vector, the code and the TLB warm (the first
    // dependent load of a workgroup costs 6 us, a later search level 2).
    int g = (int)blockIdx.x;
    // ---- rows of graph g: lower bounds of g and g + 1 in the sorted batch vector, both searches in lockstep (RF_THREADS probes per
    // level and key: two levels cover 262 144 rows) ----
    int r0, r1;
    {
        int lo[2] = {0, 0}, hi[2] = {N, N};
        const long long key[2] = {(long long)g, (long long)g + 1};
        for (int level = 0;; ++level) {
            int stride[2], first[2], pi[2];
            bool ge[2];
            long long pv[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int len = hi[s] - lo[s];
                stride[s] = max((len + RF_THREADS - 1) / RF_THREADS, 1);
                pi[s] = lo[s] + tid * stride[s];
            }
            // both keys' probes in ONE round trip, unconditional (clamped): as `i < hi && batch[i] >= key` each load sits in a branch of
            // its own with a full drain (vmcnt(0): the accumulator words too) at the join — two serial round trips per level
            if (level == 0) {
                pv[0] = pv[1] = v1;                              // (lo = 0, hi = N for both keys: the probe fetched at entry)
            } else {
                pv[0] = a.batch[min(pi[0], N - 1)];
                pv[1] = a.batch[min(pi[1], N - 1)];
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) ge[s] = pi[s] < hi[s] && pv[s] >= key[s];
            if (tid < 2) slot[tid] = RF_THREADS;
            rf_lds_barrier();                                  // (not __syncthreads: that would drain the accumulator words' loads)
#pragma unroll
            for (int s = 0; s < 2; ++s)
                if (ge[s]) atomicMin(&slot[s], tid);
            rf_lds_barrier();
            first[0] = slot[0];
            first[1] = slot[1];
            if (a.stamps && level < 2) tk[4 + level] = __builtin_amdgcn_s_memrealtime();
            rf_lds_barrier();
            bool done = true;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int len = hi[s] - lo[s];
                if (len <= 0) continue;
                const int probes = (len + stride[s] - 1) / stride[s];
                if (stride[s] == 1) {
                    const int ans = first[s] >= probes ? hi[s] : lo[s] + first[s];
                    lo[s] = hi[s] = ans;
                } else {
                    const int nlo = first[s] == 0 ? lo[s] : lo[s] + (min(first[s], probes) - 1) * stride[s] + 1;
                    const int nhi = first[s] >= probes ? hi[s] : lo[s] + first[s] * stride[s];
                    lo[s] = nlo;
                    hi[s] = nhi;
                    done = done && nhi <= nlo;
                }
            }
            if (done) break;
        }
        r0 = lo[0];
        r1 = lo[1];
    }
    if (a.stamps) tk[1] = __builtin_amdgcn_s_memrealtime();
    const int rows_first = r1 - r0;
    const int count = min(nvr, N);
    // ---- the gate operand of this wavefront's 16-column tile (the first KG / 16 wavefronts): k-group gq, step q -> row 16 gq + 4 kq + q of M ----
    rf_f32x4 hreg[G];
    const int mtile = wave % TILES, mhalf = wave / TILES;         // (MFMA wavefronts: wave < TILES * KSPLIT)
    if (wave < TILES * KSPLIT) {
#pragma unroll
        for (int gq = 0; gq < G; ++gq)
#pragma unroll
            for (int q = 0; q < 4; ++q) hreg[gq][q] = a.M[(size_t)(128 * mhalf + 16 * gq + 4 * kq + q) * KG + 16 * mtile + c];
    }
    // ---- the norm's constants (arithmetic and order of the stand-alone kernels: bn_from_totals, replicas in order) ----
    if constexpr (NORMED) {
#pragma unroll
        for (int pp = 0; pp < PAIRS; ++pp) {
            const double pS = (double)(long long)araw[pp][0] * (1.0 / 4096.0) + (double)(long long)araw[pp][1] * (1.0 / 4611686018427387904.0);
            const double pQ = (double)(long long)araw[pp][2] * (1.0 / 4096.0) + (double)(long long)araw[pp][3] * (1.0 / 4611686018427387904.0);
            const int qbase = lane & ~3;
            double S = __shfl(pS, qbase, 64), Q = __shfl(pQ, qbase, 64);
#pragma unroll
            for (int r = 1; r < 4; ++r) {
                S += __shfl(pS, qbase + r, 64);
                Q += __shfl(pQ, qbase + r, 64);
            }
            if (nrep == 0) {
                const int nc = ncol[pp];
                const BnFromTotals bt = bn_from_totals(S, Q, count, nw[pp], nb[pp], a.eps);
                s_scale[nc] = bt.scale;
                s_shift[nc] = bt.shift;
                if (owner) {
                    float *stat = a.stat;
                    stat[nc] = bt.mean;
                    stat[K + nc] = bt.invstd;
                    stat[2 * K + nc] = bt.scale;
                    stat[3 * K + nc] = bt.shift;
                    stat[4 * K + nc] = nw[pp];
                    stat[5 * K + nc] = nb[pp];
                    if (a.p.has_running) {
                        *(FLAT ? a.p.running_mean[0] + nc : bn_pick(a.p.running_mean, ncomp[pp]) + nkk[pp]) = bn_running_update(rm_old[pp], a.momentum, bt.mean);
                        *(FLAT ? a.p.running_var[0] + nc : bn_pick(a.p.running_var, ncomp[pp]) + nkk[pp]) = bn_running_update(rv_old[pp], a.momentum, bt.unbiased);
                    }
                }
            }
        }
        if (owner && a.p.has_nbt && tid < a.p.n_comp)
            for (int q = 0; q < a.p.n_comp; ++q)
                if (q == tid) *a.p.num_batches_tracked[q] += 1;
        rf_lds_barrier();
    }
    if (a.stamps) tk[2] = __builtin_amdgcn_s_memrealtime();
    // this thread's place in the image: K = 128: row 2 wave + dsub of the tile, chunk li ^ dsub; K = 256: rows wave and wave + 8, chunk lane
    const int dsub = RPI == 2 ? lane >> 5 : 0, li = RPI == 2 ? lane & 31 : lane, c4 = li ^ dsub;
    const float4 *Y4 = reinterpret_cast<const float4 *>(a.y) + c4;
    const float4 *S4 = reinterpret_cast<const float4 *>(a.skip ? a.skip : a.y) + c4;
    const bool has_skip = a.skip != nullptr;
    const int fbase = RPI == 2 ? (c >> 1) * ISTRIDE + (c & 1) * H4 + (kq ^ (c & 1)) : c * ISTRIDE + kq;
    rf_f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (wave < TILES && a.c) bias4 = *reinterpret_cast<const rf_f32x4 *>(a.c + 16 * wave + 4 * kq);      // (first halves only: mtile == wave)
    const int pcol = tid & (K - 1), part = tid / K;            // the column this thread adds up, and its share of a tile's rows
    for (;;) {
        float pacc = 0.f;
        auto tile_row = [&](int u) { return RPI == 2 ? 2 * wave + dsub : wave + 8 * u; };
        // the next tile's rows travel while this tile is multiplied and added up (a 200-atom molecule is 13 tiles: its workgroup is the
        // launch's critical path)
        float4 v_next[PASSES], rr_next[PASSES];
    #pragma unroll
        for (int u = 0; u < PASSES; ++u) {
            v_next[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            rr_next[u] = v_next[u];
            if (r0 < r1) {
                const int rn = min(r0 + tile_row(u), r1 - 1);
                v_next[u] = Y4[(size_t)rn * H4];
                if constexpr (NORMED) rr_next[u] = S4[(size_t)rn * H4];
            }
        }
        for (int rb = r0; rb < r1; rb += 16) {
            // ---- this (half-)wavefront's rows of the tile -> LDS image (and out to memory once) ----
    #pragma unroll
            for (int u = 0; u < PASSES; ++u) {
                const int tr = tile_row(u);
                const int r = rb + tr;
                const bool valid = r < r1;
                const int rc = valid ? r : r1 - 1;
                const float4 v = v_next[u];
                const float4 rr = rr_next[u];
                if (rb + 16 < r1) {
                    const int rn = min(rb + 16 + tr, r1 - 1);
                    v_next[u] = Y4[(size_t)rn * H4];
                    if constexpr (NORMED) rr_next[u] = S4[(size_t)rn * H4];
                }
                float4 o = v;
                if constexpr (NORMED) {
                    const float4 sc = *reinterpret_cast<const float4 *>(s_scale + 4 * c4), sh = *reinterpret_cast<const float4 *>(s_shift + 4 * c4);
                    if constexpr (RELU) {
                        o.x = fmaxf(fmaf(v.x, sc.x, sh.x), 0.f); o.y = fmaxf(fmaf(v.y, sc.y, sh.y), 0.f);
                        o.z = fmaxf(fmaf(v.z, sc.z, sh.z), 0.f); o.w = fmaxf(fmaf(v.w, sc.w, sh.w), 0.f);
                    } else {
                        o.x = act_fwd_rt(a.norm_act, fmaf(v.x, sc.x, sh.x)); o.y = act_fwd_rt(a.norm_act, fmaf(v.y, sc.y, sh.y));
                        o.z = act_fwd_rt(a.norm_act, fmaf(v.z, sc.z, sh.z)); o.w = act_fwd_rt(a.norm_act, fmaf(v.w, sc.w, sh.w));
                    }
                    o.x = has_skip ? o.x + rr.x : o.x; o.y = has_skip ? o.y + rr.y : o.y;
                    o.z = has_skip ? o.z + rr.z : o.z; o.w = has_skip ? o.w + rr.w : o.w;
                    if (rc >= count) o = make_float4(0.f, 0.f, 0.f, 0.f);
                }
                if (!valid) o = make_float4(0.f, 0.f, 0.f, 0.f);
                image[(RPI == 2 ? wave : tr) * ISTRIDE + lane] = rf_f32x4{o.x, o.y, o.z, o.w};
                if constexpr (NORMED) {
                    if (valid) reinterpret_cast<float4 *>(a.x_out)[(size_t)r * H4 + c4] = o;
                }
            }
            rf_lds_barrier();
            // ---- gate tile: wavefront w < KG / 16 owns columns 16 w .. 16 w + 15 ----
            rf_f32x4 acc = bias4;
            if (wave < TILES * KSPLIT) {
    #pragma unroll
                for (int gq = 0; gq < G; ++gq) {
                    const rf_f32x4 fr = image[fbase + 4 * (gq + G * mhalf)];
    #pragma unroll
                    for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(hreg[gq][q], fr[q], acc, 0, 0, 0);
                }
            }
            if constexpr (KSPLIT == 2) {
                if (wave >= TILES && wave < 2 * TILES) khalf[mtile * 64 + lane] = acc;          // (bias4 is zero in the second halves)
                rf_lds_barrier();
                if (wave < TILES) acc += khalf[mtile * 64 + lane];
            }
            if (wave < TILES) {
                const int row = rb + c;                              // lane (c, kq) holds gate[row][16 w + 4 kq .. + 3]
                if (row < r1) *reinterpret_cast<rf_f32x4 *>(a.gate + (size_t)row * KG + 16 * wave + 4 * kq) = acc;
    #pragma unroll
                for (int q = 0; q < 4; ++q) sgate[c][16 * wave + 4 * kq + q] = 1.f / (1.f + expf(-acc[q]));
            }
            rf_lds_barrier();
            // ---- this thread's column: rows part, part + PARTS, ... of the tile ----
            const float *img = reinterpret_cast<const float *>(image);
    #pragma unroll
            for (int u = 0; u < 16 / PARTS; ++u) {
                const int rr_ = part + PARTS * u;
                const int slot4 = RPI == 2 ? (rr_ >> 1) * ISTRIDE + 32 * (rr_ & 1) + ((pcol >> 2) ^ (rr_ & 1)) : rr_ * ISTRIDE + (pcol >> 2);
                const float xv = img[slot4 * 4 + (pcol & 3)];
                pacc = rb + rr_ < r1 ? fmaf(xv, sgate[rr_][pcol & (KG - 1)], pacc) : pacc;
            }
            rf_lds_barrier();
        }
        if (a.stamps && tk[3] == 0) tk[3] = __builtin_amdgcn_s_memrealtime();
        red[part][pcol] = pacc;
        __syncthreads();
        if (part == 0) {
            float tot = red[0][pcol] + red[1][pcol];
            if constexpr (PARTS == 4) tot += red[2][pcol] + red[3][pcol];
            a.out[(size_t)g * K + pcol] = tot;
        }

        g += (int)gridDim.x;
        if (g >= a.num_graphs) break;
        rf_lds_barrier();                                        // (red / slot are reused)
        // ---- the next graph's rows: the same search, every level from memory ----
        {
            int lo[2] = {0, 0}, hi[2] = {N, N};
            const long long key[2] = {(long long)g, (long long)g + 1};
            for (;;) {
                int stride[2], first[2], pi[2];
                bool ge[2];
                long long pv[2];
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int len = hi[s] - lo[s];
                    stride[s] = max((len + RF_THREADS - 1) / RF_THREADS, 1);
                    pi[s] = lo[s] + tid * stride[s];
                }
                pv[0] = a.batch[min(pi[0], N - 1)];
                pv[1] = a.batch[min(pi[1], N - 1)];
#pragma unroll
                for (int s = 0; s < 2; ++s) ge[s] = pi[s] < hi[s] && pv[s] >= key[s];
                if (tid < 2) slot[tid] = RF_THREADS;
                rf_lds_barrier();
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    if (ge[s]) atomicMin(&slot[s], tid);
                rf_lds_barrier();
                first[0] = slot[0];
                first[1] = slot[1];
                rf_lds_barrier();
                bool done = true;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int len = hi[s] - lo[s];
                    if (len <= 0) continue;
                    const int probes = (len + stride[s] - 1) / stride[s];
                    if (stride[s] == 1) {
                        const int ans = first[s] >= probes ? hi[s] : lo[s] + first[s];
                        lo[s] = hi[s] = ans;
                    } else {
                        const int nlo = first[s] == 0 ? lo[s] : lo[s] + (min(first[s], probes) - 1) * stride[s] + 1;
                        const int nhi = first[s] >= probes ? hi[s] : lo[s] + first[s] * stride[s];
                        lo[s] = nlo;
                        hi[s] = nhi;
                        done = done && nhi <= nlo;
                    }
                }
                if (done) break;
            }
            r0 = lo[0];
            r1 = lo[1];
        }
    }
    if (a.stamps && tid == 0) {
        unsigned long long *o = a.stamps + (size_t)blockIdx.x * 8;        // (the workgroup's FIRST graph)
        o[0] = tk[0]; o[1] = tk[1]; o[2] = tk[2]; o[3] = tk[3]; o[4] = __builtin_amdgcn_s_memrealtime(); o[5] = (unsigned long long)rows_first;
        o[6] = tk[4]; o[7] = tk[5];
    }
}

// dx[r, c*k+j] = dout[g(r), c*k+j] * s_j ;  dgate[r, j] = s_j (1 - s_j) * sum_c dout[g(r), c*k+j] * x[r, c*k+j]
// Row-parallel: thread -> (row, VEC gate columns), graph id from the int64 `batch` vector.
template <int VEC>
__global__ __launch_bounds__(256) void gated_pool_bwd_kernel(const float *__restrict__ dout, const float *__restrict__ x,
                                                           const float *__restrict__ gate, const long long *__restrict__ batch,
                                                           long long N, int h, int k, int n_comp, float *__restrict__ dx,
                                                           float *__restrict__ dgate) {
    const int kv = k / VEC;
    const long long total = N * kv;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long r = e / kv;
        const int j = (int)(e - r * kv) * VEC;
        const long long g = batch[r];
        float s[VEC], dg[VEC];
        if constexpr (VEC == 4) {
            const float4 gv = *reinterpret_cast<const float4 *>(gate + (size_t)r * k + j);
            s[0] = 1.f / (1.f + expf(-gv.x)); s[1] = 1.f / (1.f + expf(-gv.y));
            s[2] = 1.f / (1.f + expf(-gv.z)); s[3] = 1.f / (1.f + expf(-gv.w));
        } else {
            s[0] = 1.f / (1.f + expf(-gate[(size_t)r * k + j]));
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) dg[v] = 0.f;
        for (int c = 0; c < n_comp; ++c) {
            const int col = c * k + j;
            if constexpr (VEC == 4) {
                const float4 d = *reinterpret_cast<const float4 *>(dout + (size_t)g * h + col);
                const float4 xv = *reinterpret_cast<const float4 *>(x + (size_t)r * h + col);
                dg[0] = fmaf(d.x, xv.x, dg[0]); dg[1] = fmaf(d.y, xv.y, dg[1]);
                dg[2] = fmaf(d.z, xv.z, dg[2]); dg[3] = fmaf(d.w, xv.w, dg[3]);
                if (dx) *reinterpret_cast<float4 *>(dx + (size_t)r * h + col) = make_float4(d.x * s[0], d.y * s[1], d.z * s[2], d.w * s[3]);
            } else {
                const float d = dout[(size_t)g * h + col];
                dg[0] = fmaf(d, x[(size_t)r * h + col], dg[0]);
                if (dx) dx[(size_t)r * h + col] = d * s[0];
            }
        }
        if (dgate) {
            if constexpr (VEC == 4)
                *reinterpret_cast<float4 *>(dgate + (size_t)r * k + j) =
                    make_float4(dg[0] * s[0] * (1.f - s[0]), dg[1] * s[1] * (1.f - s[1]), dg[2] * s[2] * (1.f - s[2]),
                                dg[3] * s[3] * (1.f - s[3]));
            else
                dgate[(size_t)r * k + j] = dg[0] * s[0] * (1.f - s[0]);
        }
    }
}

// The same with the gate layer's backward-data map folded in (pooling.py:57-66 computes gate = x . M + c with M the composed
// [h, k] operand of ops.gate_linear): the row's dgate values never leave the row's lane group before
//     dx[r, :] = dout[g(r), :] * s  +  dgate[r, :] . M^T                       (M^T [k, h] row-major, staged in LDS)
// is complete — one launch instead of this one and a K = k GEMM with the pooled share as its residual operand.
// KV = k / 4 lanes per row (8 or 16, adjacent and aligned): every lane owns 4 gate columns and the 4 n_comp columns of x they gate;
// the lane group all-gathers its dgate values by shuffles, each lane adds the k-term dot products of ITS columns.
template <int KV>
__global__ __launch_bounds__(256) void gated_pool_bwd_gate_kernel(const float *__restrict__ dout, const float *__restrict__ x,
                                                                const float *__restrict__ gate, const long long *__restrict__ batch,
                                                                long long N, int h, int n_comp, const float *__restrict__ MT,
                                                                float *__restrict__ dx, float *__restrict__ dgate) {
    extern __shared__ __attribute__((aligned(16))) float4 mt4[];          // [k][h / 4]
    constexpr int k = 4 * KV, RPB = 256 / KV;
    const int h4 = h >> 2, tid = threadIdx.x;
    for (int e = tid; e < k * h4; e += 256) mt4[e] = reinterpret_cast<const float4 *>(MT)[e];
    __syncthreads();
    const int jq = tid % KV, j = 4 * jq, lane = tid & 63, lbase = lane - (lane % KV);
    for (long long r0 = (long long)blockIdx.x * RPB; r0 < N; r0 += (long long)gridDim.x * RPB) {       // (uniform bound: every lane shuffles)
        const long long r = r0 + tid / KV;
        const bool live = r < N;
        const long long rc = live ? r : N - 1;
        const long long g = batch[rc];
        const float4 gv = *reinterpret_cast<const float4 *>(gate + (size_t)rc * k + j);
        const float s0 = 1.f / (1.f + expf(-gv.x)), s1 = 1.f / (1.f + expf(-gv.y)), s2 = 1.f / (1.f + expf(-gv.z)), s3 = 1.f / (1.f + expf(-gv.w));
        float4 acc[8];
        float dg0 = 0.f, dg1 = 0.f, dg2 = 0.f, dg3 = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (c < n_comp) {
                const int col = c * k + j;
                const float4 d = *reinterpret_cast<const float4 *>(dout + (size_t)g * h + col);
                const float4 xv = *reinterpret_cast<const float4 *>(x + (size_t)rc * h + col);
                dg0 = fmaf(d.x, xv.x, dg0); dg1 = fmaf(d.y, xv.y, dg1); dg2 = fmaf(d.z, xv.z, dg2); dg3 = fmaf(d.w, xv.w, dg3);
                acc[c] = make_float4(d.x * s0, d.y * s1, d.z * s2, d.w * s3);
            }
        }
        const float4 dgs = make_float4(dg0 * s0 * (1.f - s0), dg1 * s1 * (1.f - s1), dg2 * s2 * (1.f - s2), dg3 * s3 * (1.f - s3));
        if (live) *reinterpret_cast<float4 *>(dgate + (size_t)r * k + j) = dgs;
#pragma unroll
        for (int t = 0; t < KV; ++t) {
            const float u0 = __shfl(dgs.x, lbase + t, 64), u1 = __shfl(dgs.y, lbase + t, 64), u2 = __shfl(dgs.z, lbase + t, 64),
                        u3 = __shfl(dgs.w, lbase + t, 64);
            const float4 *mrow = mt4 + (size_t)(4 * t) * h4 + jq;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (c < n_comp) {
                    const int o = c * KV;                    // (c k + j) / 4 = c KV + jq
                    const float4 m0 = mrow[o], m1 = mrow[h4 + o], m2 = mrow[2 * h4 + o], m3 = mrow[3 * h4 + o];
                    acc[c].x += u0 * m0.x + u1 * m1.x + u2 * m2.x + u3 * m3.x;
                    acc[c].y += u0 * m0.y + u1 * m1.y + u2 * m2.y + u3 * m3.y;
                    acc[c].z += u0 * m0.z + u1 * m1.z + u2 * m2.z + u3 * m3.z;
                    acc[c].w += u0 * m0.w + u1 * m1.w + u2 * m2.w + u3 * m3.w;
                }
            }
        }
        if (live) {
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (c < n_comp) *reinterpret_cast<float4 *>(dx + (size_t)r * h + c * k + j) = acc[c];
        }
    }
}

// ---- tiny dense head ------------------------------------------------------------------------------
// y[r, o] = b[o] + sum_i x[r, i] w[o, i] for the last map of the read-out (RealTransformer 'linear' on
// n*target -> target columns, layers.py:399-414: 4 -> 1 for ogbg-molhiv): a BLAS call plus its two backward GEMMs,
// a bias reduction and a fill for 544 x 4 numbers.  in * out <= SL_MAX_W.
constexpr int SL_MAX_W = 64;

__global__ __launch_bounds__(256) void small_linear_fwd_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                             const float *__restrict__ b, long long N, int fi, int fo,
                                                             float *__restrict__ y) {
    __shared__ float ws[SL_MAX_W], bs[SL_MAX_W];
    for (int e = threadIdx.x; e < fi * fo; e += 256) ws[e] = w[e];
    for (int e = threadIdx.x; e < fo; e += 256) bs[e] = b ? b[e] : 0.f;
    __syncthreads();
    const long long total = N * fo;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long r = e / fo;
        const int o = (int)(e - r * fo);
        float acc = bs[o];
        for (int i = 0; i < fi; ++i) acc = fmaf(x[r * fi + i], ws[o * fi + i], acc);
        y[e] = acc;
    }
}

// blocks [0, gridDim.x - 1): dx[r, i] = sum_o dy[r, o] w[o, i];  last block: dw[o, i] = sum_r dy[r, o] x[r, i], db[o] = sum_r dy[r, o]
// (one workgroup, rows in a fixed strided order + fixed tree: reproducible)
__global__ __launch_bounds__(256) void small_linear_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                             const float *__restrict__ w, long long N, int fi, int fo,
                                                             float *__restrict__ dx, float *__restrict__ dw, float *__restrict__ db) {
    __shared__ float ws[SL_MAX_W];
    __shared__ float red[256];
    if (blockIdx.x + 1 < gridDim.x) {
        if (!dx) return;
        for (int e = threadIdx.x; e < fi * fo; e += 256) ws[e] = w[e];
        __syncthreads();
        const long long total = N * fi;
        for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)(gridDim.x - 1) * 256) {
            const long long r = e / fi;
            const int i = (int)(e - r * fi);
            float acc = 0.f;
            for (int o = 0; o < fo; ++o) acc = fmaf(dy[r * fo + o], ws[o * fi + i], acc);
            dx[e] = acc;
        }
        return;
    }
    const int outs = (dw ? fi * fo : 0) + (db ? fo : 0);
    for (int q = 0; q < fi * fo + fo; ++q) {
        const bool is_w = q < fi * fo;
        if ((is_w && !dw) || (!is_w && !db)) continue;
        const int o = is_w ? q / fi : q - fi * fo, i = is_w ? q - o * fi : 0;
        float acc = 0.f;
        for (long long r = threadIdx.x; r < N; r += 256) acc = is_w ? fmaf(dy[r * fo + o], x[r * fi + i], acc) : acc + dy[r * fo + o];
        red[threadIdx.x] = acc;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
            __syncthreads();
        }
        if (threadIdx.x == 0) (is_w ? dw[q] : db[o]) = red[0];
        __syncthreads();
    }
    (void)outs;
}

// ---- n-ary sum ------------------------------------------------------------------------------------
// out = src[0] + src[1] + ... in that order: the gradient of a tensor that fans out to several consumers (the
// encoded atoms feed the first conv AND every layer's skip connection, models.py:230-234) in one pass instead of
// n - 1 pairwise adds.
constexpr int SUMN_MAX = 16;
struct SumSrcs {
    const float *p[SUMN_MAX];
};

__global__ __launch_bounds__(256) void sum_n_kernel(SumSrcs srcs, int n, long long numel4, long long numel, float *__restrict__ out) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < numel4; e += (long long)gridDim.x * 256) {
        float4 acc = reinterpret_cast<const float4 *>(srcs.p[0])[e];
        for (int k = 1; k < n; ++k) {
            const float4 v = reinterpret_cast<const float4 *>(srcs.p[k])[e];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        reinterpret_cast<float4 *>(out)[e] = acc;
    }
    if (blockIdx.x == 0)
        for (long long e = numel4 * 4 + threadIdx.x; e < numel; e += 256) {
            float acc = srcs.p[0][e];
            for (int k = 1; k < n; ++k) acc += srcs.p[k][e];
            out[e] = acc;
        }
}

// ---- several partial-sum vectors to scalars: out[j] = sum(src[j][0 .. len[j])) ---------------------------------
// One workgroup per vector, fixed order (lane-strided partial sums, then a fixed tree): the d/d beta partials of all
// softmax-aggregation layers of a step in one launch instead of one framework reduction per layer.
struct VecSums {
    const float *p[SUMN_MAX];
    int len[SUMN_MAX];
    float *out[SUMN_MAX];
};

__global__ __launch_bounds__(256) void vec_sums_kernel(VecSums v) {
    __shared__ float red[256];
    const float *src = v.p[blockIdx.x];
    const int n = v.len[blockIdx.x];
    float s = 0.f;
    for (int e0 = 0; e0 < n; e0 += 8 * 256) {                // eight loads per round trip (clamped, selected)
        float t[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) t[q] = src[min(e0 + (int)threadIdx.x + 256 * q, n - 1)];
#pragma unroll
        for (int q = 0; q < 8; ++q) s += e0 + (int)threadIdx.x + 256 * q < n ? t[q] : 0.f;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) *v.out[blockIdx.x] = red[0];
}

}  // namespace phc

using namespace phc;

extern "C" {

int phc_vec_sums(const float *const *srcs, const int32_t *lens, float *const *outs, int num, void *stream) {
    PHC_REQUIRE(num >= 1 && num <= SUMN_MAX && srcs && lens && outs, PHC_ERR_INVALID_ARG, "phc_vec_sums: %d vectors (1..%d)", num, SUMN_MAX);
    VecSums v;
    for (int j = 0; j < SUMN_MAX; ++j) {
        v.p[j] = j < num ? srcs[j] : nullptr;
        v.len[j] = j < num ? lens[j] : 0;
        v.out[j] = j < num ? outs[j] : nullptr;
        PHC_REQUIRE(j >= num || (srcs[j] && outs[j] && lens[j] >= 1), PHC_ERR_INVALID_ARG, "phc_vec_sums: vector %d: null or empty", j);
    }
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_SUM_N, s);
    PHC_LAUNCH(vec_sums_kernel, dim3((unsigned)num), dim3(256), 0, s, v);
    return check_launch("vec_sums");
}

size_t phc_bce_masked_ws_bytes(int64_t total) {
    if (total < 0) return 0;
    return sizeof(float) * 2 * 256;
}

static int masked_loss_fwd(const char *who, int kind, const float *logits, const float *y, int64_t total, float *loss, float *stat, void *ws,
                           size_t ws_bytes, void *stream) {
    PHC_REQUIRE(total > 0 && logits && y && loss && stat, PHC_ERR_INVALID_ARG, "%s: bad arguments", who);
    hipStream_t s = (hipStream_t)stream;
    int blocks = (int)((total + (1 << 12) - 1) >> 12);          // 4k entries (4 per thread) per workgroup
    if (blocks > 256) blocks = 256;
    PHC_REQUIRE(blocks == 1 || (ws && ws_bytes >= phc_bce_masked_ws_bytes(total)), PHC_ERR_WORKSPACE, "%s: workspace too small", who);
    ProfScope prof(K_BCE_FWD, s);
    if (kind == 0)
        PHC_LAUNCH(bce_masked_fwd_kernel<0>, dim3(blocks), dim3(RO_THREADS), 0, s, logits, y, (long long)total, static_cast<float *>(ws), loss, stat);
    else
        PHC_LAUNCH(bce_masked_fwd_kernel<1>, dim3(blocks), dim3(RO_THREADS), 0, s, logits, y, (long long)total, static_cast<float *>(ws), loss, stat);
    if (blocks > 1)
        PHC_LAUNCH(bce_masked_finalize_kernel, dim3(1), dim3(RO_THREADS), 0, s, static_cast<const float *>(ws), blocks, loss, stat);
    return check_launch(who);
}

static int masked_loss_fwd_unit(const char *who, int kind, const float *logits, const float *y, int64_t total, float *loss, float *stat,
                                float *dlogits_unit, void *stream) {
    PHC_REQUIRE(total > 0 && total <= (1 << 12) && logits && y && loss && stat && dlogits_unit, PHC_ERR_INVALID_ARG,
                "%s: 1 <= total <= 4096 entries and non-null pointers", who);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_BCE_FWD, s);
    if (kind == 0)
        PHC_LAUNCH(bce_masked_fwd_unit_kernel<0>, dim3(1), dim3(RO_THREADS), 0, s, logits, y, (long long)total, loss, stat, dlogits_unit);
    else
        PHC_LAUNCH(bce_masked_fwd_unit_kernel<1>, dim3(1), dim3(RO_THREADS), 0, s, logits, y, (long long)total, loss, stat, dlogits_unit);
    return check_launch(who);
}

static int masked_loss_bwd(const char *who, int kind, const float *logits, const float *y, const float *dloss, const float *stat,
                           int64_t total, float *dlogits, void *stream) {
    PHC_REQUIRE(total > 0 && logits && y && dloss && stat && dlogits, PHC_ERR_INVALID_ARG, "%s: bad arguments", who);
    hipStream_t s = (hipStream_t)stream;
    long long blocks = (total + 1023) / 1024;
    if (blocks > 1024) blocks = 1024;
    ProfScope prof(K_BCE_BWD, s);
    if (kind == 0)
        PHC_LAUNCH(bce_masked_bwd_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, s, logits, y, dloss, stat, (long long)total, dlogits);
    else
        PHC_LAUNCH(bce_masked_bwd_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, s, logits, y, dloss, stat, (long long)total, dlogits);
    return check_launch(who);
}

int phc_bce_masked_fwd(const float *logits, const float *y, int64_t total, float *loss, float *stat, void *ws, size_t ws_bytes,
                       void *stream) {
    return masked_loss_fwd("phc_bce_masked_fwd", 0, logits, y, total, loss, stat, ws, ws_bytes, stream);
}
int phc_bce_masked_fwd_unit(const float *logits, const float *y, int64_t total, float *loss, float *stat, float *dlogits_unit,
                            void *stream) {
    return masked_loss_fwd_unit("phc_bce_masked_fwd_unit", 0, logits, y, total, loss, stat, dlogits_unit, stream);
}
int phc_bce_masked_bwd(const float *logits, const float *y, const float *dloss, const float *stat, int64_t total, float *dlogits,
                       void *stream) {
    return masked_loss_bwd("phc_bce_masked_bwd", 0, logits, y, dloss, stat, total, dlogits, stream);
}
// the same three for the L1 loss of the regression models (benchmarks/train_zinc.py: torch.nn.L1Loss), NaN labels left out
int phc_l1_masked_fwd(const float *pred, const float *y, int64_t total, float *loss, float *stat, void *ws, size_t ws_bytes,
                      void *stream) {
    return masked_loss_fwd("phc_l1_masked_fwd", 1, pred, y, total, loss, stat, ws, ws_bytes, stream);
}
int phc_l1_masked_fwd_unit(const float *pred, const float *y, int64_t total, float *loss, float *stat, float *dpred_unit, void *stream) {
    return masked_loss_fwd_unit("phc_l1_masked_fwd_unit", 1, pred, y, total, loss, stat, dpred_unit, stream);
}
int phc_l1_masked_bwd(const float *pred, const float *y, const float *dloss, const float *stat, int64_t total, float *dpred,
                      void *stream) {
    return masked_loss_bwd("phc_l1_masked_bwd", 1, pred, y, dloss, stat, total, dpred, stream);
}

static bool gp_vec4(int h, int k, const void *a, const void *b, const void *c, const void *d) {
    return (k % 4 == 0) && (h % 4 == 0) && ((reinterpret_cast<uintptr_t>(a) & 15) == 0) && ((reinterpret_cast<uintptr_t>(b) & 15) == 0) &&
           ((reinterpret_cast<uintptr_t>(c) & 15) == 0) && ((reinterpret_cast<uintptr_t>(d) & 15) == 0);
}

static int gated_pool_fwd_impl(const float *x, const float *gate, const int32_t *seg_ptr, const int64_t *batch_sorted, int64_t N,
                               int64_t num_graphs, int h, int n_comp, float *out, void *stream) {
    PHC_REQUIRE(num_graphs >= 0 && h >= 1 && n_comp >= 1 && h % n_comp == 0, PHC_ERR_INVALID_ARG, "phc_gated_pool_fwd: bad shape");
    if (num_graphs == 0) return PHC_OK;
    PHC_REQUIRE(x && gate && (seg_ptr || batch_sorted) && out, PHC_ERR_INVALID_ARG, "phc_gated_pool_fwd: null pointer");
    PHC_REQUIRE(seg_ptr || (N >= 0 && N < (1LL << 31)), PHC_ERR_UNSUPPORTED, "phc_gated_pool_fwd_sorted: N out of range");
    hipStream_t s = (hipStream_t)stream;
    const int k = h / n_comp;
    const long long *b = reinterpret_cast<const long long *>(batch_sorted);
    ProfScope prof(K_GATED_POOL_FWD, s);
    if (gp_vec4(h, k, x, gate, out, nullptr))
        PHC_LAUNCH((gated_pool_fwd_kernel<4>), dim3((unsigned)num_graphs), dim3(GP_THREADS), 0, s, x, gate, seg_ptr, h, k, out, b, (int)N);
    else
        PHC_LAUNCH((gated_pool_fwd_kernel<1>), dim3((unsigned)num_graphs), dim3(GP_THREADS), 0, s, x, gate, seg_ptr, h, k, out, b, (int)N);
    return check_launch("gated_pool_fwd");
}

int phc_gated_pool_fwd(const float *x, const float *gate, const int32_t *seg_ptr, int64_t num_graphs, int h, int n_comp, float *out,
                       void *stream) {
    PHC_REQUIRE(num_graphs == 0 || seg_ptr, PHC_ERR_INVALID_ARG, "phc_gated_pool_fwd: null pointer");
    return gated_pool_fwd_impl(x, gate, seg_ptr, nullptr, 0, num_graphs, h, n_comp, out, stream);
}

int phc_gated_pool_fwd_sorted(const float *x, const float *gate, const int64_t *batch_sorted_i64, int64_t N, int64_t num_graphs, int h,
                              int n_comp, float *out, void *stream) {
    PHC_REQUIRE(num_graphs == 0 || batch_sorted_i64, PHC_ERR_INVALID_ARG, "phc_gated_pool_fwd_sorted: null pointer");
    return gated_pool_fwd_impl(x, gate, nullptr, batch_sorted_i64, N, num_graphs, h, n_comp, out, stream);
}

static int ro_num_cu() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t pr;
        n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
    }
    return n;
}
static unsigned long long *g_readout_stamps = nullptr;
int phc_debug_readout_stamps(unsigned long long *dev_words) {
    g_readout_stamps = dev_words;
    return PHC_OK;
}

int phc_readout_fused_supported(int64_t N, int64_t num_graphs, int h, int n_comp, int kg) {
    static const bool off = getenv("PHC_READOUT_FUSED") && atoi(getenv("PHC_READOUT_FUSED")) == 0;
    return (!off && N >= 1 && N <= (1 << 18) && num_graphs >= 1 && num_graphs <= (1 << 20) && (h == 128 || h == 256) && n_comp >= 1 &&
            h % n_comp == 0 && ((h == 128 && kg == 32) || (h == 256 && (kg == 32 || kg == 64))) && h / n_comp == kg) ? 1 : 0;
}

int phc_readout_fused_fwd(const phc_readout_desc *d, void *stream) {
    PHC_REQUIRE(d, PHC_ERR_INVALID_ARG, "phc_readout_fused_fwd: null descriptor");
    PHC_REQUIRE(phc_readout_fused_supported(d->N, d->num_graphs, d->h, d->n_comp, d->kg), PHC_ERR_UNSUPPORTED,
                "phc_readout_fused_fwd: %lld rows, %lld graphs, h = %d, %d components, gate width %d outside the fused form's range",
                (long long)d->N, (long long)d->num_graphs, d->h, d->n_comp, d->kg);
    PHC_REQUIRE(d->x && d->batch_sorted_i64 && d->M && d->gate && d->out, PHC_ERR_INVALID_ARG, "phc_readout_fused_fwd: null pointer");
    const bool normed = d->norm_acc != nullptr;
    PHC_REQUIRE(!normed || (d->save_stat && d->x_out), PHC_ERR_INVALID_ARG, "phc_readout_fused_fwd: a pending norm needs save_stat and x_out");
    PHC_REQUIRE(!normed || (d->norm_act >= PHC_ACT_IDENTITY && d->norm_act <= PHC_ACT_SELU), PHC_ERR_INVALID_ARG,
                "phc_readout_fused_fwd: unknown activation %d", d->norm_act);
    PHC_REQUIRE(((reinterpret_cast<uintptr_t>(d->x) | reinterpret_cast<uintptr_t>(d->gate) | reinterpret_cast<uintptr_t>(d->c) |
                  (normed ? (reinterpret_cast<uintptr_t>(d->x_out) | reinterpret_cast<uintptr_t>(d->skip)) : 0)) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(d->norm_acc) & 7) == 0,
                PHC_ERR_UNSUPPORTED, "phc_readout_fused_fwd: 16-byte aligned rows / 8-byte aligned accumulator");
    ReadoutFusedArgs a = {};
    if (normed) {
        int rc = fill_params("phc_readout_fused_fwd", &a.p, d->weight, d->bias, d->running_mean, d->running_var, d->num_batches_tracked,
                             d->norm_n_comp, d->h);
        if (rc) return rc;
        a.acc = reinterpret_cast<const unsigned long long *>(d->norm_acc);
        a.skip = d->skip; a.x_out = d->x_out; a.stat = d->save_stat;
        a.momentum = d->momentum; a.eps = d->eps; a.norm_act = d->norm_act;
    }
    a.y = d->x; a.n_valid = d->n_valid_dev; a.batch = reinterpret_cast<const long long *>(d->batch_sorted_i64);
    a.N = (int)d->N; a.num_graphs = (int)d->num_graphs; a.M = d->M; a.c = d->c; a.gate = d->gate; a.out = d->out;
    a.stamps = g_readout_stamps;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_GATED_POOL_FWD, s);
    const long long slots = 2LL * ro_num_cu();                       // both instantiations fit two workgroups of 512 threads per CU
    const dim3 grid((unsigned)(d->num_graphs < slots ? d->num_graphs : slots)), block(RF_THREADS);
#define PHC_RF_LAUNCH(KC, GC)                                                                                  \
    do {                                                                                                      \
        if (!normed)                                                                                          \
            PHC_LAUNCH((readout_fused_fwd_kernel<KC, GC, false, true, true>), grid, block, 0, s, a);          \
        else if (a.norm_act == PHC_ACT_RELU && a.p.flat)                                                      \
            PHC_LAUNCH((readout_fused_fwd_kernel<KC, GC, true, true, true>), grid, block, 0, s, a);           \
        else if (a.norm_act == PHC_ACT_RELU)                                                                  \
            PHC_LAUNCH((readout_fused_fwd_kernel<KC, GC, true, true, false>), grid, block, 0, s, a);          \
        else if (a.p.flat)                                                                                    \
            PHC_LAUNCH((readout_fused_fwd_kernel<KC, GC, true, false, true>), grid, block, 0, s, a);          \
        else                                                                                                  \
            PHC_LAUNCH((readout_fused_fwd_kernel<KC, GC, true, false, false>), grid, block, 0, s, a);         \
    } while (0)
    if (d->h == 128)
        PHC_RF_LAUNCH(128, 32);
    else if (d->kg == 32)
        PHC_RF_LAUNCH(256, 32);
    else
        PHC_RF_LAUNCH(256, 64);
#undef PHC_RF_LAUNCH
    return check_launch("phc_readout_fused_fwd");
}

int phc_gated_pool_bwd(const float *dout, const float *x, const float *gate, const int64_t *batch, int64_t N, int64_t num_graphs,
                       int h, int n_comp, float *dx, float *dgate, void *stream) {
    PHC_REQUIRE(N >= 0 && num_graphs >= 0 && h >= 1 && n_comp >= 1 && h % n_comp == 0, PHC_ERR_INVALID_ARG,
                "phc_gated_pool_bwd: bad shape");
    if (N == 0 || (!dx && !dgate)) return PHC_OK;
    PHC_REQUIRE(dout && x && gate && batch, PHC_ERR_INVALID_ARG, "phc_gated_pool_bwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int k = h / n_comp;
    const bool v4 = gp_vec4(h, k, x, gate, dout, dx) && ((reinterpret_cast<uintptr_t>(dgate) & 15) == 0);
    long long blocks = (N * (long long)(k / (v4 ? 4 : 1)) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    ProfScope prof(K_GATED_POOL_BWD, s);
    if (v4)
        PHC_LAUNCH((gated_pool_bwd_kernel<4>), dim3((unsigned)blocks), dim3(256), 0, s, dout, x, gate,
                   reinterpret_cast<const long long *>(batch), (long long)N, h, k, n_comp, dx, dgate);
    else
        PHC_LAUNCH((gated_pool_bwd_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, s, dout, x, gate,
                   reinterpret_cast<const long long *>(batch), (long long)N, h, k, n_comp, dx, dgate);
    return check_launch("gated_pool_bwd");
}

int phc_gated_pool_bwd_gate_supported(int h, int n_comp) {
    static const bool off = getenv("PHC_POOL_BWD_GATE") && atoi(getenv("PHC_POOL_BWD_GATE")) == 0;
    if (off || n_comp < 1 || n_comp > 8 || h < 1 || h % n_comp) return 0;
    const int k = h / n_comp;
    // (k = 64 — configs[3] — is served by the kernel but measured 6 us per step slower than the two launches: 256 LDS reads per lane
    // behind a 64 KB operand; PHC_POOL_BWD_GATE=2 turns it on)
    static const bool k64 = getenv("PHC_POOL_BWD_GATE") && atoi(getenv("PHC_POOL_BWD_GATE")) == 2;
    return ((k == 32 || (k == 64 && k64)) && (size_t)k * h * sizeof(float) <= 96 * 1024) ? 1 : 0;
}

int phc_gated_pool_bwd_gate(const float *dout, const float *x, const float *gate, const int64_t *batch, int64_t N, int64_t num_graphs,
                            int h, int n_comp, const float *MT, float *dx, float *dgate, void *stream) {
    PHC_REQUIRE(N >= 0 && num_graphs >= 0 && phc_gated_pool_bwd_gate_supported(h, n_comp), PHC_ERR_UNSUPPORTED,
                "phc_gated_pool_bwd_gate: h = %d with %d components (k = 32 or 64, at most 8 components)", h, n_comp);
    if (N == 0) return PHC_OK;
    PHC_REQUIRE(dout && x && gate && batch && MT && dx && dgate, PHC_ERR_INVALID_ARG, "phc_gated_pool_bwd_gate: null pointer");
    PHC_REQUIRE(((reinterpret_cast<uintptr_t>(dout) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gate) |
                  reinterpret_cast<uintptr_t>(MT) | reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(dgate)) & 15) == 0,
                PHC_ERR_UNSUPPORTED, "phc_gated_pool_bwd_gate: 16-byte aligned rows required");
    hipStream_t s = (hipStream_t)stream;
    const int k = h / n_comp, KV = k / 4;
    const size_t lds = sizeof(float) * (size_t)k * h;
    long long blocks = (N * KV + 255) / 256;
    const long long cap = 2LL * ro_num_cu();
    if (blocks > cap) blocks = cap;
    ProfScope prof(K_GATED_POOL_BWD, s);
    if (KV == 8) {
        PHC_REQUIRE(ensure_dynamic_lds(&gated_pool_bwd_gate_kernel<8>, 96 * 1024), PHC_ERR_LAUNCH, "phc_gated_pool_bwd_gate: dynamic LDS refused");
        PHC_LAUNCH((gated_pool_bwd_gate_kernel<8>), dim3((unsigned)blocks), dim3(256), lds, s, dout, x, gate,
                   reinterpret_cast<const long long *>(batch), (long long)N, h, n_comp, MT, dx, dgate);
    } else {
        PHC_REQUIRE(ensure_dynamic_lds(&gated_pool_bwd_gate_kernel<16>, 96 * 1024), PHC_ERR_LAUNCH, "phc_gated_pool_bwd_gate: dynamic LDS refused");
        PHC_LAUNCH((gated_pool_bwd_gate_kernel<16>), dim3((unsigned)blocks), dim3(256), lds, s, dout, x, gate,
                   reinterpret_cast<const long long *>(batch), (long long)N, h, n_comp, MT, dx, dgate);
    }
    return check_launch("gated_pool_bwd_gate");
}

// ---- batched strided 2-D copies (phc_copy2d_batched): the flat [R, h] image of an encoder's [V_f, k] tables and back ----
__global__ __launch_bounds__(256) void copy2d_batched_kernel(const phc_copy2d_job *__restrict__ jobs, const float *__restrict__ src_base,
                                                            float *__restrict__ dst_base) {
    const phc_copy2d_job j = jobs[blockIdx.y];
    const float *src = (j.src ? j.src : src_base) + j.src_off;
    float *dst = (j.dst ? j.dst : dst_base) + j.dst_off;
    const long long total = (long long)j.rows * j.cols;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long r = e / j.cols;
        const int cc = (int)(e - r * j.cols);
        dst[r * j.dst_stride + cc] = src[r * j.src_stride + cc];
    }
}

int phc_copy2d_batched(const phc_copy2d_job *jobs_dev, int num_jobs, int64_t max_elems, const float *src_base, float *dst_base,
                       void *stream) {
    PHC_REQUIRE(num_jobs >= 0 && max_elems >= 0 && (num_jobs == 0 || jobs_dev), PHC_ERR_INVALID_ARG, "phc_copy2d_batched: bad arguments");
    if (num_jobs == 0 || max_elems == 0) return PHC_OK;
    PHC_REQUIRE(num_jobs <= 65535, PHC_ERR_UNSUPPORTED, "phc_copy2d_batched: at most 65535 jobs per call");
    long long bx = (max_elems + 255) / 256;
    if (bx > 64) bx = 64;
    hipStream_t s = (hipStream_t)stream;
    PHC_LAUNCH(copy2d_batched_kernel, dim3((unsigned)bx, (unsigned)num_jobs), dim3(256), 0, s, jobs_dev, src_base, dst_base);
    return check_launch("copy2d_batched");
}

int phc_sum_n(const float *const *srcs, int n, int64_t numel, float *out, void *stream) {
    PHC_REQUIRE(n >= 1 && n <= SUMN_MAX && numel >= 0 && srcs && (numel == 0 || out), PHC_ERR_INVALID_ARG, "phc_sum_n: n=%d (1..%d)", n,
                SUMN_MAX);
    if (numel == 0) return PHC_OK;
    SumSrcs ss;
    bool al = (reinterpret_cast<uintptr_t>(out) & 15) == 0;
    for (int k = 0; k < SUMN_MAX; ++k) {
        ss.p[k] = k < n ? srcs[k] : nullptr;
        PHC_REQUIRE(k >= n || srcs[k], PHC_ERR_INVALID_ARG, "phc_sum_n: source %d is null", k);
        if (k < n) al = al && (reinterpret_cast<uintptr_t>(srcs[k]) & 15) == 0;
    }
    hipStream_t s = (hipStream_t)stream;
    const long long numel4 = al ? numel / 4 : 0;
    long long blocks = (numel4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    ProfScope prof(K_SUM_N, s);
    PHC_LAUNCH(sum_n_kernel, dim3((unsigned)blocks), dim3(256), 0, s, ss, n, numel4, (long long)numel, out);
    return check_launch("sum_n");
}

int phc_small_linear_max_weights(void) { return SL_MAX_W; }

int phc_small_linear_fwd(const float *x, const float *w, const float *b, int64_t N, int fi, int fo, float *y, void *stream) {
    PHC_REQUIRE(N >= 0 && fi >= 1 && fo >= 1 && fi * fo <= SL_MAX_W, PHC_ERR_UNSUPPORTED,
                "phc_small_linear_fwd: in*out = %d (max %d)", fi * fo, SL_MAX_W);
    if (N == 0) return PHC_OK;
    PHC_REQUIRE(x && w && y, PHC_ERR_INVALID_ARG, "phc_small_linear_fwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    long long blocks = (N * fo + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    ProfScope prof(K_SMALL_LINEAR, s);
    PHC_LAUNCH(small_linear_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, w, b, (long long)N, fi, fo, y);
    return check_launch("small_linear_fwd");
}

int phc_small_linear_bwd(const float *dy, const float *x, const float *w, int64_t N, int fi, int fo, float *dx, float *dw, float *db,
                         void *stream) {
    PHC_REQUIRE(N >= 0 && fi >= 1 && fo >= 1 && fi * fo <= SL_MAX_W, PHC_ERR_UNSUPPORTED,
                "phc_small_linear_bwd: in*out = %d (max %d)", fi * fo, SL_MAX_W);
    PHC_REQUIRE(dy && x && w, PHC_ERR_INVALID_ARG, "phc_small_linear_bwd: null pointer");
    if (!dx && !dw && !db) return PHC_OK;
    hipStream_t s = (hipStream_t)stream;
    long long blocks = (N * fi + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    ProfScope prof(K_SMALL_LINEAR, s);
    PHC_LAUNCH(small_linear_bwd_kernel, dim3((unsigned)blocks + 1), dim3(256), 0, s, dy, x, w, (long long)N, fi, fo, dx, dw, db);
    return check_launch("small_linear_bwd");
}

}  // extern "C"
