// Destination-/source-sorted edge lists (CSR / CSC) built on the device.
//
// The reference hands PyG an unsorted int64 edge_index [2, E] and lets
// torch-scatter's atomics sort it out on every call (messagepassing.py:60,136).
// This engine groups the edges once per batch — stable counting sort by key:
//   histogram -> exclusive scan -> slot assignment -> per-row canonical order
// so that every later reduction walks contiguous segments in ONE fixed order
// (ascending original edge id).  All kernels are enqueue-only (graph-capturable).
#include "phc_common.h"

namespace phc {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_ELEMS = SCAN_THREADS * SCAN_ITEMS;      // elements per workgroup

__global__ void csr_hist_kernel(const long long *__restrict__ key, long long E, int *__restrict__ counts, long long num_keys,
                                int *__restrict__ err, int err_bit) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (long long)gridDim.x * blockDim.x)
        atomicAdd(&counts[checked_index(key[e], num_keys, err, err_bit)], 1);
}

// In-place exclusive scan of SCAN_ELEMS-element blocks; block totals go to sums[blockIdx.x].
__global__ __launch_bounds__(SCAN_THREADS) void scan_block_kernel(int *__restrict__ data, int *__restrict__ sums, long long n) {
    __shared__ int wave_tot[SCAN_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long base = (long long)blockIdx.x * SCAN_ELEMS + (long long)tid * SCAN_ITEMS;
    int v[SCAN_ITEMS];
    int tot = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        v[i] = (base + i < n) ? data[base + i] : 0;
        tot += v[i];
    }
    int incl = tot;                                      // inclusive scan of thread totals inside the wavefront
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int wave_off = 0;
    for (int w = 0; w < wave; ++w) wave_off += wave_tot[w];
    int run = wave_off + incl - tot;                     // exclusive prefix of this thread
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        if (base + i < n) data[base + i] = run;
        run += v[i];
    }
    if (tid == SCAN_THREADS - 1 && sums) sums[blockIdx.x] = run;
}

__global__ void scan_add_kernel(int *__restrict__ data, const int *__restrict__ sums, long long n) {
    const long long i = (long long)blockIdx.x * SCAN_ELEMS + threadIdx.x;
    const int off = sums[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const long long j = i + (long long)k * SCAN_THREADS;
        if (j < n) data[j] += off;
    }
}

// slot = rowptr[key] + (arrival order); arrival order is arbitrary, csr_canon fixes it.
__global__ void csr_fill_kernel(const long long *__restrict__ key, long long E, const int *__restrict__ rowptr,
                                int *__restrict__ cursor, int *__restrict__ tmp_perm, long long num_keys) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (long long)gridDim.x * blockDim.x) {
        const long long k = safe_index(key[e], num_keys);
        const int pos = rowptr[k] + atomicAdd(&cursor[k], 1);
        tmp_perm[pos] = (int)e;
    }
}

// Rank sort of every row's item ids (unique => ranks are a permutation), out of place.
// SUB lanes cooperate on one row; rows with degree outside (lo, hi] are left to the other launch.
template <int SUB>
__global__ __launch_bounds__(256) void csr_canon_kernel(const int *__restrict__ rowptr, const int *__restrict__ tmp_perm,
                                                       const long long *__restrict__ other, long long num_keys, int lo, int hi,
                                                       int *__restrict__ perm, int *__restrict__ col, long long num_other,
                                                       int *__restrict__ err) {
    const int li = threadIdx.x & (SUB - 1);
    const long long grp = ((long long)blockIdx.x * blockDim.x + threadIdx.x) / SUB;
    const long long n_grp = ((long long)gridDim.x * blockDim.x) / SUB;
    for (long long k = grp; k < num_keys; k += n_grp) {
        const int b = rowptr[k], deg = rowptr[k + 1] - b;
        if (deg <= lo || deg > hi) continue;
        for (int t = li; t < deg; t += SUB) {
            const int v = tmp_perm[b + t];
            int rank = 0;
            for (int j = 0; j < deg; ++j) rank += (tmp_perm[b + j] < v) ? 1 : 0;
            perm[b + rank] = v;
            col[b + rank] = other ? (int)checked_index(other[v], num_other, err, PHC_IDX_ERR_EDGE) : v;
        }
    }
}

// Rows too long for the rank sorts above (O(deg^2 / lanes): a 1e5-degree hub of a power-law graph would keep one lane group
// busy for seconds): one WORKGROUP per heavy row sorts the row's item ids with a bitonic network whose comparators all point
// the same way (the "flip" formulation), so a length that is not a power of two needs no padding — a partner index past the
// end is +infinity and never moves.  Rows of up to CSR_HEAVY_LDS items are sorted in LDS, longer ones in place in the scratch
// segment (tmp_perm) through L2.  O(deg log^2 deg / 256) per row.  Every workgroup scans its share of the row pointers and
// queues the heavy rows it finds (no host round trip, no extra buffer).
constexpr int CSR_HEAVY_MIN = 2048;                            // rows longer than this are left to csr_heavy_rows_kernel
constexpr int CSR_HEAVY_LDS = 8192;                            // items sorted in LDS (32 KB)
constexpr int CSR_HEAVY_GRID = 1024;

template <typename Arr>
__device__ __forceinline__ void bitonic_flip_sort(Arr a, int n, int tid, int nthreads) {
    int P = 1;
    while (P < n) P <<= 1;
    for (int k = 2; k <= P; k <<= 1) {
        for (int i = tid; i < n; i += nthreads) {              // flip: i <-> i ^ (k - 1)
            const int j = i ^ (k - 1);
            if (j > i && j < n) {
                const int x = a[i], y = a[j];
                if (x > y) { a[i] = y; a[j] = x; }
            }
        }
        __syncthreads();
        for (int st = k >> 2; st > 0; st >>= 1) {
            for (int i = tid; i < n; i += nthreads) {
                const int j = i ^ st;
                if (j > i && j < n) {
                    const int x = a[i], y = a[j];
                    if (x > y) { a[i] = y; a[j] = x; }
                }
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(256) void csr_heavy_rows_kernel(const int *__restrict__ rowptr, int *__restrict__ tmp_perm,
                                                            const long long *__restrict__ other, long long num_keys,
                                                            long long num_other, int *__restrict__ perm, int *__restrict__ col,
                                                            int *__restrict__ err) {
    __shared__ int items[CSR_HEAVY_LDS];
    __shared__ int queue[256];
    __shared__ int qn;
    const int tid = threadIdx.x;
    for (long long base = (long long)blockIdx.x * 256; base < num_keys; base += (long long)gridDim.x * 256) {
        if (tid == 0) qn = 0;
        __syncthreads();
        const long long k = base + tid;
        if (k < num_keys && rowptr[k + 1] - rowptr[k] > CSR_HEAVY_MIN) queue[atomicAdd(&qn, 1)] = tid;
        __syncthreads();
        const int nq = qn;
        for (int qi = 0; qi < nq; ++qi) {
            // (queue order is arbitrary; each row's result does not depend on it)
            const long long row = base + queue[qi];
            const int b = rowptr[row], deg = rowptr[row + 1] - b;
            if (deg <= CSR_HEAVY_LDS) {
                for (int t = tid; t < deg; t += 256) items[t] = tmp_perm[b + t];
                __syncthreads();
                bitonic_flip_sort(items, deg, tid, 256);
                for (int t = tid; t < deg; t += 256) {
                    const int v = items[t];
                    perm[b + t] = v;
                    col[b + t] = other ? (int)checked_index(other[v], num_other, err, PHC_IDX_ERR_EDGE) : v;
                }
            } else {
                __syncthreads();
                bitonic_flip_sort(tmp_perm + b, deg, tid, 256);
                for (int t = tid; t < deg; t += 256) {
                    const int v = tmp_perm[b + t];
                    perm[b + t] = v;
                    col[b + t] = other ? (int)checked_index(other[v], num_other, err, PHC_IDX_ERR_EDGE) : v;
                }
            }
            __syncthreads();
        }
        __syncthreads();
    }
}

// rowptr[s] = first position whose key is >= s, for NON-DECREASING keys (PyG's collated `batch` vector):
// one thread per boundary, binary search.  No histogram, no atomics, no sort.
__global__ void sorted_segment_ptr_kernel(const long long *__restrict__ key, long long n, long long num_keys, int *__restrict__ rowptr) {
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s > num_keys) return;
    long long lo = 0, hi = n;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (key[mid] < s) lo = mid + 1; else hi = mid;
    }
    rowptr[s] = (int)lo;
}

// ---- large graphs: two-level counting sort ---------------------------------------------------------------------------
// phc_csr_build's histogram / fill passes are 2 E random 4-byte atomics over num_keys counters and its canonicalising pass
// gathers through them again: at 10 M keys / 50 M items every access is its own DRAM sector (12.5 ms for both views,
// 2.8 % of HBM, profiles/r01_microbench_config5_1gpu.txt).  Here the items are first PARTITIONED by key >> shift into
// <= 16384 buckets of 2^shift consecutive keys (LDS histograms per 64 K-item chunk, one global atomic per non-empty
// (chunk, bucket) instead of one per item; the partition order inside a bucket is arbitrary), then ONE workgroup per bucket
// runs the same histogram -> scan -> fill -> per-row rank sort on its ~5 K items with LDS counters, streaming the bucket's
// (key, id) pairs from L2.  The result is identical to phc_csr_build's (the rank sort makes every row ascending in item id
// whatever the arrival order was).
constexpr int BK_MAX_BUCKETS = 16384;
constexpr int BK_CHUNK = 65536;                              // items per partition workgroup
constexpr int BK_THREADS = 256;
constexpr int BK_MAX_KEYS_PER_BUCKET = 4096;                 // LDS counters of the per-bucket sort (2 x 16 KB)

__global__ __launch_bounds__(BK_THREADS) void bk_count_kernel(const long long *__restrict__ key, long long E, long long num_keys, int shift,
                                                             int nb, int *__restrict__ bucket_count, int *__restrict__ err, int err_bit) {
    extern __shared__ int hist[];                            // [nb]
    for (int b = threadIdx.x; b < nb; b += BK_THREADS) hist[b] = 0;
    __syncthreads();
    const long long e0 = (long long)blockIdx.x * BK_CHUNK, e1 = min(E, e0 + BK_CHUNK);
    for (long long e = e0 + threadIdx.x; e < e1; e += BK_THREADS)
        atomicAdd(&hist[(int)(checked_index(key[e], num_keys, err, err_bit) >> shift)], 1);
    __syncthreads();
    for (int b = threadIdx.x; b < nb; b += BK_THREADS) {
        const int c = hist[b];
        if (c) atomicAdd(&bucket_count[b], c);
    }
}

// exclusive scan of bucket_count[0 .. nb] (nb <= 16384) by one workgroup; also zeroes the scatter cursors
__global__ __launch_bounds__(1024) void bk_scan_kernel(int *__restrict__ bucket_count, int *__restrict__ bucket_cursor, int nb) {
    __shared__ int wave_tot[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int ITEMS = (BK_MAX_BUCKETS + 1024) / 1024;     // 17 consecutive entries per thread
    int v[ITEMS], tot = 0;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int idx = tid * ITEMS + i;
        v[i] = idx < nb ? bucket_count[idx] : 0;
        tot += v[i];
    }
    int incl = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int run = incl - tot;
    for (int w = 0; w < wave; ++w) run += wave_tot[w];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int idx = tid * ITEMS + i;
        if (idx <= nb) bucket_count[idx] = run;              // entry nb = total
        if (idx < nb) bucket_cursor[idx] = 0;
        run += v[i];
    }
}

__global__ __launch_bounds__(BK_THREADS) void bk_scatter_kernel(const long long *__restrict__ key, long long E, long long num_keys, int shift,
                                                               int nb, const int *__restrict__ bucket_start, int *__restrict__ bucket_cursor,
                                                               uint2 *__restrict__ pairs) {
    extern __shared__ int hist[];                            // [nb]: count, then next free slot of (this chunk, bucket)
    for (int b = threadIdx.x; b < nb; b += BK_THREADS) hist[b] = 0;
    __syncthreads();
    const long long e0 = (long long)blockIdx.x * BK_CHUNK, e1 = min(E, e0 + BK_CHUNK);
    for (long long e = e0 + threadIdx.x; e < e1; e += BK_THREADS) atomicAdd(&hist[(int)(safe_index(key[e], num_keys) >> shift)], 1);
    __syncthreads();
    for (int b = threadIdx.x; b < nb; b += BK_THREADS) {
        const int c = hist[b];
        if (c) hist[b] = bucket_start[b] + atomicAdd(&bucket_cursor[b], c);     // ONE global atomic per non-empty (chunk, bucket)
    }
    __syncthreads();
    const unsigned mask = (1u << shift) - 1u;
    for (long long e = e0 + threadIdx.x; e < e1; e += BK_THREADS) {
        const long long k = safe_index(key[e], num_keys);
        const int pos = atomicAdd(&hist[(int)(k >> shift)], 1);
        pairs[pos] = make_uint2((unsigned)k & mask, (unsigned)e);
    }
}

// one workgroup per bucket: counting sort of its items by local key, rows made ascending in item id (rank sort, 8 lanes per row)
__global__ __launch_bounds__(BK_THREADS) void bk_sort_kernel(const int *__restrict__ bucket_start, const uint2 *__restrict__ pairs,
                                                            const long long *__restrict__ other, long long num_keys, long long num_other,
                                                            int shift, int nb, int *__restrict__ tmp_perm, int *__restrict__ rowptr,
                                                            int *__restrict__ col, int *__restrict__ perm, int *__restrict__ err) {
    __shared__ int cnt[BK_MAX_KEYS_PER_BUCKET + 1], cur[BK_MAX_KEYS_PER_BUCKET];
    __shared__ int wave_tot[BK_THREADS / 64];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int keys_here = (int)min((long long)(1 << shift), num_keys - ((long long)b << shift));
    const int s0 = bucket_start[b], s1 = bucket_start[b + 1];
    for (int k = tid; k <= keys_here; k += BK_THREADS) cnt[k] = 0;
    __syncthreads();
    for (int i = s0 + tid; i < s1; i += BK_THREADS) atomicAdd(&cnt[pairs[i].x], 1);
    __syncthreads();
    // exclusive scan of cnt[0 .. keys_here] (<= 4097 entries): 17 consecutive entries per thread
    constexpr int ITEMS = (BK_MAX_KEYS_PER_BUCKET + BK_THREADS) / BK_THREADS;
    int v[ITEMS], tot = 0;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int idx = tid * ITEMS + i;
        v[i] = idx < keys_here ? cnt[idx] : 0;
        tot += v[i];
    }
    int incl = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int run = incl - tot;
    for (int w = 0; w < wave; ++w) run += wave_tot[w];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int idx = tid * ITEMS + i;
        if (idx <= keys_here) cnt[idx] = run;
        if (idx < keys_here) cur[idx] = run;
        run += v[i];
    }
    __syncthreads();
    const long long key0 = (long long)b << shift;
    for (int k = tid; k < keys_here; k += BK_THREADS) rowptr[key0 + k] = s0 + cnt[k];
    if (b == nb - 1 && tid == 0) rowptr[num_keys] = s1;
    for (int i = s0 + tid; i < s1; i += BK_THREADS) {
        const uint2 pr = pairs[i];
        tmp_perm[s0 + atomicAdd(&cur[pr.x], 1)] = (int)pr.y;
    }
    __syncthreads();
    constexpr int SUB = 8;
    const int li = tid & (SUB - 1);
    for (int k = tid / SUB; k < keys_here; k += BK_THREADS / SUB) {
        const int rb = s0 + cnt[k], deg = cnt[k + 1] - cnt[k];
        if (deg > CSR_HEAVY_MIN) continue;                    // csr_heavy_rows_kernel (next launch)
        for (int t = li; t < deg; t += SUB) {
            const int vv = tmp_perm[rb + t];
            int rank = 0;
            for (int j = 0; j < deg; ++j) rank += (tmp_perm[rb + j] < vv) ? 1 : 0;
            perm[rb + rank] = vv;
            col[rb + rank] = other ? (int)checked_index(other[vv], num_other, err, PHC_IDX_ERR_EDGE) : vv;
        }
    }
}

struct ScanLevels {
    int levels;
    long long n[6];
    long long blocks[6];
    size_t total_sum_ints;
};

static ScanLevels scan_levels(long long n) {
    ScanLevels L;
    L.levels = 0;
    L.total_sum_ints = 0;
    long long cur = n;
    while (true) {
        const long long b = (cur + SCAN_ELEMS - 1) / SCAN_ELEMS;
        L.n[L.levels] = cur;
        L.blocks[L.levels] = b;
        L.levels++;
        if (b <= 1) break;
        L.total_sum_ints += (size_t)b;
        cur = b;
    }
    return L;
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- both sorted views in five launches (molecule batches) --------------------------------------------------
// (Tried: count + scan + fill of a side by ONE 1024-thread workgroup with the histogram, the scan and the fill cursors in LDS
// and the thread's 29 edge keys in registers — 2 launches instead of 5.  Its serial phases take 13.4 us with no edges at all and
// 22.5 us at 27.6 k edges, against 21.6 us for the four launch-floor-bound launches it replaces: no gain, removed.)
// A training step needs the destination-sorted view (forward) AND the source-sorted one (backward).  Building
// them side by side (blockIdx.y picks the side) with a one-workgroup scan takes 5 launches instead of 22.
constexpr int PAIR_SCAN_THREADS = 1024;
constexpr int PAIR_SCAN_ITEMS_MAX = 32;
constexpr int PAIR_MAX_KEYS = PAIR_SCAN_THREADS * PAIR_SCAN_ITEMS_MAX - 1;      // rowptr has num_keys + 1 entries

struct CsrSide {
    const long long *key, *other;
    int *rowptr, *cursor, *tmp_perm, *perm, *col;
    long long num_keys;
};
struct CsrPair {
    CsrSide s[2];
    // optional: the packed table rows of every edge in the order of side 0 (phc_pack_edge_codes), written by the
    // canonicalising launch as it places each edge — one launch less per batch
    const long long *bond;
    unsigned *codes;
    int F, off[4], vocab[4];
    int *err;                          // registered error word (phc_set_error_word) or nullptr
    // optional: per destination row, how many of its in-edges name each table row — counts[row][CR], CR = 16 or 32 columns
    // (the A operand of the register-resident bond-table gradient, phc_step_tail); R = sum(vocab) <= CR
    float *counts;
    int CR, R;
    int resident;                      // the counter arrays of the workspace are zero on entry and are left zero (no zero launch)
    // optional (round 6): the first four in-edges of every destination row side by side — ell[row][0..3] = source, ell[row][4..7] =
    // packed code, slots past the row's degree name the row itself (a row its reader loads anyway) with code 0 — and the first four
    // out-edges of every source row, ell_t[row][0..3] = destination.  A reader that owns a row gets its edges in the SAME round trip as
    // the row pointers instead of one later (the aggregating fused forward / backward, csrc/mlp_fused.hip).
    int *ell, *ell_t;
    unsigned *codes_t;                 // optional: the packed codes in the order of side 1 (source-sorted: the softmax backward's walk)
};

__global__ void csr_pair_zero_kernel(CsrPair p) {
    const CsrSide &sd = p.s[blockIdx.y];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i <= sd.num_keys; i += (long long)gridDim.x * blockDim.x) {
        sd.rowptr[i] = 0;
        sd.cursor[i] = 0;
    }
}

// resident workspace (phc_csr_build_pair_resident): the counts go into `cursor`, which every build leaves zeroed — no zero launch
__global__ void csr_pair_hist_kernel(CsrPair p, long long E) {
    const CsrSide &sd = p.s[blockIdx.y];
    int *cnt = p.resident ? sd.cursor : sd.rowptr;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (long long)gridDim.x * blockDim.x)
        atomicAdd(&cnt[checked_index(sd.key[e], sd.num_keys, p.err, PHC_IDX_ERR_EDGE)], 1);      // both sides: every src and dst
}

// exclusive scan of the row counts [0 .. num_keys] by ONE workgroup per side.  Global traffic is coalesced (thread t touches
// elements t, t + 1024, ...) and staged through LDS; each thread then scans its own 32 consecutive LDS entries.  The LDS image is
// padded by one word per 32 (entry e sits at e + e / 32), so that the 64 lanes of a wavefront, 32 entries apart, hit 64 different
// banks.  32 767 rows per side: the 1024-graph batches of the 256-wide configurations (23-27 k nodes) take this path too.
// (ITEMS entries per thread: 16 for up to 16 383 rows — a molecule batch of 512 graphs scans in 7.3 us instead of 9.1 —, 32 beyond)
__device__ __forceinline__ int pair_pad(int e) { return e + (e >> 5); }
template <int PAIR_SCAN_ITEMS>
__global__ __launch_bounds__(PAIR_SCAN_THREADS) void csr_pair_scan_kernel(CsrPair p) {
    __shared__ int buf[PAIR_SCAN_THREADS * PAIR_SCAN_ITEMS + PAIR_SCAN_THREADS * PAIR_SCAN_ITEMS / 32];
    __shared__ int wave_tot[PAIR_SCAN_THREADS / 64];
    const CsrSide &sd = p.s[blockIdx.x];
    const int n = (int)(sd.num_keys + 1);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int *cnt = p.resident ? sd.cursor : sd.rowptr;
#pragma unroll 8
    for (int i = 0; i < PAIR_SCAN_ITEMS; ++i) {
        const int e = i * PAIR_SCAN_THREADS + tid;
        const int v = cnt[min(e, n - 1)];
        buf[pair_pad(e)] = e < n ? v : 0;
    }
    if (p.resident) {                                   // the counts are in LDS now: the fill starts its cursors from zero
#pragma unroll 8
        for (int i = 0; i < PAIR_SCAN_ITEMS; ++i) {
            const int e = i * PAIR_SCAN_THREADS + tid;
            if (e < n) sd.cursor[e] = 0;
        }
    }
    __syncthreads();
    const int base = tid * PAIR_SCAN_ITEMS;             // pair_pad(base + j): one pad word per 32 entries
    int tot = 0;
#pragma unroll 8
    for (int j = 0; j < PAIR_SCAN_ITEMS; ++j) tot += buf[pair_pad(base + j)];
    int incl = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int run = incl - tot;
    for (int w = 0; w < wave; ++w) run += wave_tot[w];
#pragma unroll 8
    for (int j = 0; j < PAIR_SCAN_ITEMS; ++j) {
        const int v = buf[pair_pad(base + j)];
        buf[pair_pad(base + j)] = run;
        run += v;
    }
    __syncthreads();
#pragma unroll 8
    for (int i = 0; i < PAIR_SCAN_ITEMS; ++i) {
        const int e = i * PAIR_SCAN_THREADS + tid;
        if (e < n) sd.rowptr[e] = buf[pair_pad(e)];
    }
}

__global__ void csr_pair_fill_kernel(CsrPair p, long long E) {
    const CsrSide &sd = p.s[blockIdx.y];
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (long long)gridDim.x * blockDim.x) {
        const long long k = safe_index(sd.key[e], sd.num_keys);
        const int pos = sd.rowptr[k] + atomicAdd(&sd.cursor[k], 1);
        sd.tmp_perm[pos] = (int)e;
    }
}

// rank sort of every row, 8 lanes per row, any degree
__global__ __launch_bounds__(256) void csr_pair_canon_kernel(CsrPair p) {
    const CsrSide &sd = p.s[blockIdx.y];
    const long long num_other = p.s[1 - blockIdx.y].num_keys;
    constexpr int SUB = 8;
    const int li = threadIdx.x & (SUB - 1);
    const long long grp = ((long long)blockIdx.x * blockDim.x + threadIdx.x) / SUB;
    const long long n_grp = ((long long)gridDim.x * blockDim.x) / SUB;
    for (long long k = grp; k < sd.num_keys; k += n_grp) {
        const int b = sd.rowptr[k], deg = sd.rowptr[k + 1] - b;
        if (p.resident && li == 0) sd.cursor[k] = 0;      // leave the workspace as the next build expects it
        for (int t = li; t < deg; t += SUB) {
            const int v = sd.tmp_perm[b + t];
            int rank = 0;
            for (int j = 0; j < deg; ++j) rank += (sd.tmp_perm[b + j] < v) ? 1 : 0;
            sd.perm[b + rank] = v;
            const int oth = (int)safe_index(sd.other[v], num_other);
            sd.col[b + rank] = oth;
            unsigned c = 0;
            if (blockIdx.y == 0 && p.codes) {
                for (int f = 0; f < p.F; ++f)
                    c |= (unsigned)(p.off[f] + (int)checked_index(p.bond[(long long)v * p.F + f], p.vocab[f], p.err, PHC_IDX_ERR_BOND))
                         << (8 * f);
                p.codes[b + rank] = c;
            }
            if (blockIdx.y == 1 && p.codes_t) {              // (side 0 reports a bad index; here it reads as 0 like everywhere else)
                for (int f = 0; f < p.F; ++f)
                    c |= (unsigned)(p.off[f] + (int)safe_index(p.bond[(long long)v * p.F + f], p.vocab[f])) << (8 * f);
                p.codes_t[b + rank] = c;
            }
            if (rank < 4) {
                if (blockIdx.y == 0 && p.ell) {
                    p.ell[k * 8 + rank] = oth;
                    p.ell[k * 8 + 4 + rank] = (int)c;
                }
                if (blockIdx.y == 1 && p.ell_t) p.ell_t[k * 4 + rank] = oth;
            }
        }
        if (li < 4 && li >= deg) {                           // slots past the degree: the row itself
            if (blockIdx.y == 0 && p.ell) {
                p.ell[k * 8 + li] = (int)k;
                p.ell[k * 8 + 4 + li] = 0;
            }
            if (blockIdx.y == 1 && p.ell_t) p.ell_t[k * 4 + li] = (int)k;
        }
        if (blockIdx.y == 0 && p.counts) {
            // the 8 lanes of the row's group share the table rows; each walks the row's edges for its own rows (read from the
            // bond features, not from the codes the other lanes are writing right now)
            for (int tp = li; tp < p.CR; tp += SUB) {
                int f = 0;
                for (int u = 1; u < p.F; ++u) f += tp >= p.off[u];
                int cnt = 0;
                if (tp < p.R)
                    for (int j = 0; j < deg; ++j) {
                        const long long v = sd.tmp_perm[b + j];
                        cnt += p.off[f] + (int)safe_index(p.bond[v * p.F + f], p.vocab[f]) == tp ? 1 : 0;
                    }
                p.counts[(size_t)k * p.CR + tp] = (float)cnt;
            }
        }
    }
}

// ---- remove isolated nodes (benchmarks/utils.py:36-46, applied to every collated batch: train_hiv.py:171-172) ----
// keep[i] = 1 when node i touches an edge; newid = exclusive scan of keep; edges are relabelled, node rows compacted.
__global__ void iso_flag_kernel(const long long *__restrict__ src, const long long *__restrict__ dst, long long E,
                                int *__restrict__ keep, long long N, int *__restrict__ err) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (long long)gridDim.x * blockDim.x) {
        keep[checked_index(src[e], N, err, PHC_IDX_ERR_EDGE)] = 1;      // benign race: every writer stores the same value
        keep[checked_index(dst[e], N, err, PHC_IDX_ERR_EDGE)] = 1;
    }
}

// newid holds the exclusive scan of keep over N + 1 entries (newid[N] = number of kept nodes)
__global__ void iso_relabel_kernel(const long long *__restrict__ src, const long long *__restrict__ dst, long long E,
                                   const int *__restrict__ newid, long long *__restrict__ src_out, long long *__restrict__ dst_out,
                                   long long N) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (long long)gridDim.x * blockDim.x) {
        src_out[e] = newid[safe_index(src[e], N)];
        dst_out[e] = newid[safe_index(dst[e], N)];
    }
}

__global__ void iso_compact_kernel(const int *__restrict__ newid, long long N, const long long *__restrict__ x, int F,
                                   const long long *__restrict__ batch, long long *__restrict__ x_out,
                                   long long *__restrict__ batch_out, long long *__restrict__ kept_index, int *__restrict__ n_kept) {
    const long long total = N * (long long)(F > 0 ? F : 1);
    const int Fe = F > 0 ? F : 1;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const long long i = t / Fe;
        const int f = (int)(t - i * Fe);
        const int id = newid[i];
        if (newid[i + 1] == id) continue;                   // isolated: dropped
        if (x && x_out) x_out[(long long)id * F + f] = x[i * F + f];
        if (f == 0) {
            if (batch && batch_out) batch_out[id] = batch[i];
            if (kept_index) kept_index[id] = i;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && n_kept) *n_kept = newid[N];
}

}  // namespace phc

using namespace phc;

extern "C" {

size_t phc_csr_ws_bytes(int64_t E, int64_t num_keys) {
    if (E < 0 || num_keys < 0) return 0;
    ScanLevels L = scan_levels(num_keys + 1);
    // cursor[num_keys] | scan block sums | tmp_perm[E]
    return align_up(sizeof(int) * (size_t)(num_keys + 1), 256) + align_up(sizeof(int) * (L.total_sum_ints + 1), 256) +
           align_up(sizeof(int) * (size_t)(E + 1), 256);
}

int phc_segment_ptr_sorted(const int64_t *key_i64, int64_t n, int64_t num_keys, int32_t *rowptr, void *stream) {
    PHC_REQUIRE(n >= 0 && num_keys >= 0 && n < (1LL << 31) - 1, PHC_ERR_INVALID_ARG, "phc_segment_ptr_sorted: n=%lld num_keys=%lld",
                (long long)n, (long long)num_keys);
    PHC_REQUIRE(rowptr && (n == 0 || key_i64), PHC_ERR_INVALID_ARG, "phc_segment_ptr_sorted: null pointer");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_CSR_SCAN, s);
    PHC_LAUNCH(sorted_segment_ptr_kernel, dim3((unsigned)((num_keys + 1 + 255) / 256)), dim3(256), 0, s,
                       reinterpret_cast<const long long *>(key_i64), (long long)n, (long long)num_keys, rowptr);
    return check_launch("sorted_segment_ptr");
}

int phc_csr_build(const int64_t *key_i64, const int64_t *other_i64, int64_t E, int64_t num_keys, int64_t num_other,
                  int32_t *rowptr, int32_t *col, int32_t *perm, void *ws, size_t ws_bytes, void *stream) {
    PHC_REQUIRE(E >= 0 && num_keys >= 0, PHC_ERR_INVALID_ARG, "phc_csr_build: E=%lld num_keys=%lld", (long long)E,
                (long long)num_keys);
    PHC_REQUIRE(E == 0 || num_keys >= 1, PHC_ERR_INVALID_ARG, "phc_csr_build: %lld items but no key range", (long long)E);
    PHC_REQUIRE(!other_i64 || E == 0 || (num_other >= 1 && num_other < (1LL << 31) - 1), PHC_ERR_INVALID_ARG,
                "phc_csr_build: num_other=%lld", (long long)num_other);
    int *err = err_word();
    const int err_bit = other_i64 ? PHC_IDX_ERR_EDGE : PHC_IDX_ERR_SEGMENT;
    PHC_REQUIRE(E < (1LL << 31) - 1 && num_keys < (1LL << 31) - 1, PHC_ERR_UNSUPPORTED,
                "phc_csr_build: int32 index space exceeded (E=%lld, keys=%lld)", (long long)E, (long long)num_keys);
    PHC_REQUIRE(rowptr && (E == 0 || (key_i64 && col && perm)), PHC_ERR_INVALID_ARG, "phc_csr_build: null pointer");
    const size_t need = phc_csr_ws_bytes(E, num_keys);
    PHC_REQUIRE(ws && ws_bytes >= need, PHC_ERR_WORKSPACE, "phc_csr_build: workspace %zu B < required %zu B", ws_bytes, need);
    hipStream_t s = (hipStream_t)stream;

    ScanLevels L = scan_levels(num_keys + 1);
    char *p = static_cast<char *>(ws);
    int *cursor = reinterpret_cast<int *>(p);
    p += align_up(sizeof(int) * (size_t)(num_keys + 1), 256);
    int *sums = reinterpret_cast<int *>(p);
    p += align_up(sizeof(int) * (L.total_sum_ints + 1), 256);
    int *tmp_perm = reinterpret_cast<int *>(p);

    hipMemsetAsync(rowptr, 0, sizeof(int) * (size_t)(num_keys + 1), s);
    hipMemsetAsync(cursor, 0, sizeof(int) * (size_t)(num_keys + 1), s);
    const unsigned egrid = (unsigned)((E + 255) / 256 > 8192 ? 8192 : (E + 255) / 256);
    if (E > 0) {
        ProfScope prof(K_CSR_HIST, s);
        PHC_LAUNCH(csr_hist_kernel, dim3(egrid), dim3(256), 0, s, reinterpret_cast<const long long *>(key_i64),
                           (long long)E, rowptr, (long long)num_keys, err, err_bit);
        int rc = check_launch("csr_hist");
        if (rc) return rc;
    }
    {   // exclusive scan of rowptr[0 .. num_keys] (num_keys + 1 entries, the last count is 0)
        ProfScope prof(K_CSR_SCAN, s);
        int *level_data[6];
        int *level_sums[6];
        level_data[0] = rowptr;
        int *sp = sums;
        for (int l = 0; l < L.levels; ++l) {
            level_sums[l] = (L.blocks[l] > 1) ? sp : nullptr;
            if (L.blocks[l] > 1) {
                level_data[l + 1] = sp;
                sp += L.blocks[l];
            }
        }
        for (int l = 0; l < L.levels; ++l)
            PHC_LAUNCH(scan_block_kernel, dim3((unsigned)L.blocks[l]), dim3(SCAN_THREADS), 0, s, level_data[l],
                               level_sums[l], L.n[l]);
        for (int l = L.levels - 2; l >= 0; --l)
            PHC_LAUNCH(scan_add_kernel, dim3((unsigned)L.blocks[l]), dim3(SCAN_THREADS), 0, s, level_data[l],
                               level_sums[l], L.n[l]);
        int rc = check_launch("csr_scan");
        if (rc) return rc;
    }
    if (E == 0) return PHC_OK;
    {
        ProfScope prof(K_CSR_FILL, s);
        PHC_LAUNCH(csr_fill_kernel, dim3(egrid), dim3(256), 0, s, reinterpret_cast<const long long *>(key_i64),
                           (long long)E, rowptr, cursor, tmp_perm, (long long)num_keys);
        int rc = check_launch("csr_fill");
        if (rc) return rc;
    }
    {
        ProfScope prof(K_CSR_CANON, s);
        const long long groups8 = (num_keys + 31) / 32;        // 256 threads = 32 rows of 8 lanes
        unsigned g8 = (unsigned)(groups8 > 16384 ? 16384 : (groups8 < 1 ? 1 : groups8));
        PHC_LAUNCH((csr_canon_kernel<8>), dim3(g8), dim3(256), 0, s, rowptr, tmp_perm,
                           reinterpret_cast<const long long *>(other_i64), (long long)num_keys, 0, 64, perm, col,
                           (long long)num_other, err);
        const long long groups64 = (num_keys + 3) / 4;
        unsigned g64 = (unsigned)(groups64 > 8192 ? 8192 : (groups64 < 1 ? 1 : groups64));
        PHC_LAUNCH((csr_canon_kernel<64>), dim3(g64), dim3(256), 0, s, rowptr, tmp_perm,
                           reinterpret_cast<const long long *>(other_i64), (long long)num_keys, 64, CSR_HEAVY_MIN, perm, col,
                           (long long)num_other, err);
        const long long hg = (num_keys + 255) / 256;
        PHC_LAUNCH(csr_heavy_rows_kernel, dim3((unsigned)(hg > CSR_HEAVY_GRID ? CSR_HEAVY_GRID : hg)), dim3(256), 0, s, rowptr, tmp_perm,
                   reinterpret_cast<const long long *>(other_i64), (long long)num_keys, (long long)num_other, perm, col, err);
        int rc = check_launch("csr_canon");
        if (rc) return rc;
    }
    return PHC_OK;
}

static int bk_shift(int64_t num_keys) {
    int shift = 0;
    while (((num_keys + (1LL << shift) - 1) >> shift) > BK_MAX_BUCKETS) ++shift;
    if (shift < 10) shift = 10;                              // ~1 K keys per bucket at least: the per-bucket pass stays worthwhile
    return shift;
}

size_t phc_csr_bucketed_ws_bytes(int64_t E, int64_t num_keys) {
    if (E < 0 || num_keys < 0) return 0;
    return align_up(sizeof(int) * 2 * (size_t)(BK_MAX_BUCKETS + 2), 256) + align_up(sizeof(uint2) * (size_t)(E + 1), 256) +
           align_up(sizeof(int) * (size_t)(E + 1), 256);
}

int phc_csr_build_bucketed(const int64_t *key_i64, const int64_t *other_i64, int64_t E, int64_t num_keys, int64_t num_other,
                           int32_t *rowptr, int32_t *col, int32_t *perm, void *ws, size_t ws_bytes, void *stream) {
    PHC_REQUIRE(E >= 1 && num_keys >= 1 && E < (1LL << 31) - 1 && num_keys < (1LL << 31) - 1, PHC_ERR_INVALID_ARG,
                "phc_csr_build_bucketed: E=%lld num_keys=%lld", (long long)E, (long long)num_keys);
    const int shift = bk_shift(num_keys);
    PHC_REQUIRE((1 << shift) <= BK_MAX_KEYS_PER_BUCKET, PHC_ERR_UNSUPPORTED, "phc_csr_build_bucketed: %lld keys exceed %d buckets x %d keys",
                (long long)num_keys, BK_MAX_BUCKETS, BK_MAX_KEYS_PER_BUCKET);
    PHC_REQUIRE(key_i64 && rowptr && col && perm, PHC_ERR_INVALID_ARG, "phc_csr_build_bucketed: null pointer");
    PHC_REQUIRE(!other_i64 || (num_other >= 1 && num_other < (1LL << 31) - 1), PHC_ERR_INVALID_ARG, "phc_csr_build_bucketed: num_other=%lld",
                (long long)num_other);
    PHC_REQUIRE(ws && ws_bytes >= phc_csr_bucketed_ws_bytes(E, num_keys), PHC_ERR_WORKSPACE, "phc_csr_build_bucketed: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int nb = (int)((num_keys + (1LL << shift) - 1) >> shift);
    char *q = static_cast<char *>(ws);
    int *bucket_count = reinterpret_cast<int *>(q);
    int *bucket_cursor = bucket_count + BK_MAX_BUCKETS + 2;
    q += align_up(sizeof(int) * 2 * (size_t)(BK_MAX_BUCKETS + 2), 256);
    uint2 *pairs = reinterpret_cast<uint2 *>(q);
    q += align_up(sizeof(uint2) * (size_t)(E + 1), 256);
    int *tmp_perm = reinterpret_cast<int *>(q);
    int *err = err_word();
    const int err_bit = other_i64 ? PHC_IDX_ERR_EDGE : PHC_IDX_ERR_SEGMENT;
    const long long *key = reinterpret_cast<const long long *>(key_i64);
    const unsigned chunks = (unsigned)((E + BK_CHUNK - 1) / BK_CHUNK);
    const size_t lds = sizeof(int) * (size_t)nb;
    hipMemsetAsync(bucket_count, 0, sizeof(int) * (size_t)(nb + 1), s);
    {
        ProfScope prof(K_CSR_HIST, s);
        PHC_LAUNCH(bk_count_kernel, dim3(chunks), dim3(BK_THREADS), lds, s, key, (long long)E, (long long)num_keys, shift, nb, bucket_count,
                   err, err_bit);
    }
    {
        ProfScope prof(K_CSR_SCAN, s);
        PHC_LAUNCH(bk_scan_kernel, dim3(1), dim3(1024), 0, s, bucket_count, bucket_cursor, nb);
    }
    {
        ProfScope prof(K_CSR_FILL, s);
        PHC_LAUNCH(bk_scatter_kernel, dim3(chunks), dim3(BK_THREADS), lds, s, key, (long long)E, (long long)num_keys, shift, nb, bucket_count,
                   bucket_cursor, pairs);
    }
    ProfScope prof(K_CSR_CANON, s);
    PHC_LAUNCH(bk_sort_kernel, dim3(nb), dim3(BK_THREADS), 0, s, bucket_count, pairs, reinterpret_cast<const long long *>(other_i64),
               (long long)num_keys, (long long)num_other, shift, nb, tmp_perm, rowptr, col, perm, err);
    {
        const long long hg = (num_keys + 255) / 256;
        PHC_LAUNCH(csr_heavy_rows_kernel, dim3((unsigned)(hg > CSR_HEAVY_GRID ? CSR_HEAVY_GRID : hg)), dim3(256), 0, s, rowptr, tmp_perm,
                   reinterpret_cast<const long long *>(other_i64), (long long)num_keys, (long long)num_other, perm, col, err);
    }
    return check_launch("csr_build_bucketed");
}

size_t phc_csr_pair_ws_bytes(int64_t E, int64_t n_src, int64_t n_dst) {
    if (E < 0 || n_src < 0 || n_dst < 0) return 0;
    return align_up(sizeof(int) * (size_t)(n_src + 1), 256) + align_up(sizeof(int) * (size_t)(n_dst + 1), 256) +
           2 * align_up(sizeof(int) * (size_t)(E > 0 ? E : 1), 256);
}

int phc_csr_build_pair(const int64_t *edge_src_i64, const int64_t *edge_dst_i64, int64_t E, int64_t n_src, int64_t n_dst,
                       int32_t *rowptr, int32_t *col, int32_t *perm, int32_t *rowptr_t, int32_t *col_t, int32_t *perm_t, void *ws,
                       size_t ws_bytes, void *stream) {
    return phc_csr_build_pair_codes(edge_src_i64, edge_dst_i64, E, n_src, n_dst, rowptr, col, perm, rowptr_t, col_t, perm_t, nullptr,
                                    nullptr, 0, nullptr, nullptr, ws, ws_bytes, stream);
}

static int csr_build_pair_impl(const int64_t *edge_src_i64, const int64_t *edge_dst_i64, int64_t E, int64_t n_src, int64_t n_dst,
                               int32_t *rowptr, int32_t *col, int32_t *perm, int32_t *rowptr_t, int32_t *col_t, int32_t *perm_t,
                               const int64_t *bond_i64, const int32_t *vocab, int F, uint32_t *codes, float *code_counts, void *ws,
                               size_t ws_bytes, void *stream, int resident, int32_t *ell = nullptr, int32_t *ell_t = nullptr,
                               uint32_t *codes_src = nullptr);

int phc_csr_build_pair_codes(const int64_t *edge_src_i64, const int64_t *edge_dst_i64, int64_t E, int64_t n_src, int64_t n_dst,
                             int32_t *rowptr, int32_t *col, int32_t *perm, int32_t *rowptr_t, int32_t *col_t, int32_t *perm_t,
                             const int64_t *bond_i64, const int32_t *vocab, int F, uint32_t *codes, float *code_counts, void *ws,
                             size_t ws_bytes, void *stream) {
    return csr_build_pair_impl(edge_src_i64, edge_dst_i64, E, n_src, n_dst, rowptr, col, perm, rowptr_t, col_t, perm_t, bond_i64, vocab, F,
                               codes, code_counts, ws, ws_bytes, stream, 0);
}

int phc_csr_build_pair_codes_resident(const int64_t *edge_src_i64, const int64_t *edge_dst_i64, int64_t E, int64_t n_src,
                                      int64_t n_dst, int32_t *rowptr, int32_t *col, int32_t *perm, int32_t *rowptr_t, int32_t *col_t,
                                      int32_t *perm_t, const int64_t *bond_i64, const int32_t *vocab, int F, uint32_t *codes,
                                      float *code_counts, void *ws_zeroed_once, size_t ws_bytes, void *stream) {
    return csr_build_pair_impl(edge_src_i64, edge_dst_i64, E, n_src, n_dst, rowptr, col, perm, rowptr_t, col_t, perm_t, bond_i64, vocab, F,
                               codes, code_counts, ws_zeroed_once, ws_bytes, stream, 1);
}

/* ABI 8: the same build that also leaves the first four in-edges (sources | codes) of every destination row in ell [n_dst][8] and the
 * first four out-edges (destinations) of every source row in ell_t [n_src][4] (either may be NULL); slots past a row's degree name the
 * row itself.  Readers: phc_mlp_norm_fwd_fused / phc_mlp_norm_bwd_fused (a row's edges in the round trip of its row pointers). */
int phc_csr_build_pair_codes_resident_ell(const int64_t *edge_src_i64, const int64_t *edge_dst_i64, int64_t E, int64_t n_src,
                                          int64_t n_dst, int32_t *rowptr, int32_t *col, int32_t *perm, int32_t *rowptr_t, int32_t *col_t,
                                          int32_t *perm_t, const int64_t *bond_i64, const int32_t *vocab, int F, uint32_t *codes,
                                          float *code_counts, int32_t *ell, int32_t *ell_t, void *ws_zeroed_once, size_t ws_bytes,
                                          void *stream) {
    PHC_REQUIRE(!ell || codes, PHC_ERR_INVALID_ARG, "phc_csr_build_pair_codes_resident_ell: ell carries the packed codes: codes must be built too");
    return csr_build_pair_impl(edge_src_i64, edge_dst_i64, E, n_src, n_dst, rowptr, col, perm, rowptr_t, col_t, perm_t, bond_i64, vocab, F,
                               codes, code_counts, ws_zeroed_once, ws_bytes, stream, 1, ell, ell_t);
}

/* ... and the packed codes in SOURCE-sorted order as well (codes_src [E], may be NULL): what phc_pack_edge_codes(perm_t) computes in a
 * launch of its own for the softmax aggregation's backward walk */
int phc_csr_build_pair_codes_resident_ell_src(const int64_t *edge_src_i64, const int64_t *edge_dst_i64, int64_t E, int64_t n_src,
                                              int64_t n_dst, int32_t *rowptr, int32_t *col, int32_t *perm, int32_t *rowptr_t, int32_t *col_t,
                                              int32_t *perm_t, const int64_t *bond_i64, const int32_t *vocab, int F, uint32_t *codes,
                                              float *code_counts, int32_t *ell, int32_t *ell_t, uint32_t *codes_src, void *ws_zeroed_once,
                                              size_t ws_bytes, void *stream) {
    PHC_REQUIRE((!ell && !codes_src) || codes, PHC_ERR_INVALID_ARG, "phc_csr_build_pair_codes_resident_ell_src: ell / codes_src need codes");
    return csr_build_pair_impl(edge_src_i64, edge_dst_i64, E, n_src, n_dst, rowptr, col, perm, rowptr_t, col_t, perm_t, bond_i64, vocab, F,
                               codes, code_counts, ws_zeroed_once, ws_bytes, stream, 1, ell, ell_t, codes_src);
}

static int csr_build_pair_impl(const int64_t *edge_src_i64, const int64_t *edge_dst_i64, int64_t E, int64_t n_src, int64_t n_dst,
                               int32_t *rowptr, int32_t *col, int32_t *perm, int32_t *rowptr_t, int32_t *col_t, int32_t *perm_t,
                               const int64_t *bond_i64, const int32_t *vocab, int F, uint32_t *codes, float *code_counts, void *ws,
                               size_t ws_bytes, void *stream, int resident, int32_t *ell, int32_t *ell_t, uint32_t *codes_src) {
    int code_off[4] = {0, 0, 0, 0};
    int code_rows = 0;
    if (codes) {
        PHC_REQUIRE(bond_i64 && vocab && F >= 1 && F <= 4, PHC_ERR_INVALID_ARG, "phc_csr_build_pair_codes: codes need bond, vocab, 1 <= F <= 4");
        int acc = 0;
        for (int f = 0; f < F; ++f) {
            PHC_REQUIRE(vocab[f] >= 1, PHC_ERR_INVALID_ARG, "phc_csr_build_pair_codes: vocab[%d] = %d", f, vocab[f]);
            code_off[f] = acc;
            acc += vocab[f];
        }
        PHC_REQUIRE(acc <= 256, PHC_ERR_UNSUPPORTED, "phc_csr_build_pair_codes: packed codes need sum(vocab) <= 256");
        code_rows = acc;
    }
    PHC_REQUIRE(!code_counts || (codes && code_rows <= 32), PHC_ERR_INVALID_ARG,
                "phc_csr_build_pair_codes: code_counts need codes and sum(vocab) <= 32");
    PHC_REQUIRE(E >= 1 && n_src >= 1 && n_dst >= 1, PHC_ERR_INVALID_ARG, "phc_csr_build_pair: E=%lld n_src=%lld n_dst=%lld",
                (long long)E, (long long)n_src, (long long)n_dst);
    PHC_REQUIRE(n_src <= PAIR_MAX_KEYS && n_dst <= PAIR_MAX_KEYS && E < (1LL << 31) - 1, PHC_ERR_UNSUPPORTED,
                "phc_csr_build_pair: at most %d rows per side (use phc_csr_build)", PAIR_MAX_KEYS);
    PHC_REQUIRE(edge_src_i64 && edge_dst_i64 && rowptr && col && perm && rowptr_t && col_t && perm_t, PHC_ERR_INVALID_ARG,
                "phc_csr_build_pair: null pointer");
    PHC_REQUIRE(ws && ws_bytes >= phc_csr_pair_ws_bytes(E, n_src, n_dst), PHC_ERR_WORKSPACE, "phc_csr_build_pair: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    char *q = static_cast<char *>(ws);
    CsrPair p;
    // side 0: grouped by destination (forward);  side 1: grouped by source (backward)
    p.s[0].key = reinterpret_cast<const long long *>(edge_dst_i64);
    p.s[0].other = reinterpret_cast<const long long *>(edge_src_i64);
    p.s[0].rowptr = rowptr; p.s[0].col = col; p.s[0].perm = perm; p.s[0].num_keys = n_dst;
    p.s[1].key = reinterpret_cast<const long long *>(edge_src_i64);
    p.s[1].other = reinterpret_cast<const long long *>(edge_dst_i64);
    p.s[1].rowptr = rowptr_t; p.s[1].col = col_t; p.s[1].perm = perm_t; p.s[1].num_keys = n_src;
    p.s[0].cursor = reinterpret_cast<int *>(q); q += align_up(sizeof(int) * (size_t)(n_dst + 1), 256);
    p.s[1].cursor = reinterpret_cast<int *>(q); q += align_up(sizeof(int) * (size_t)(n_src + 1), 256);
    p.s[0].tmp_perm = reinterpret_cast<int *>(q); q += align_up(sizeof(int) * (size_t)E, 256);
    p.s[1].tmp_perm = reinterpret_cast<int *>(q);
    p.bond = reinterpret_cast<const long long *>(bond_i64);
    p.codes = codes;
    p.counts = code_counts;
    p.R = code_rows;
    p.CR = code_rows <= 16 ? 16 : 32;
    p.F = codes ? F : 0;
    for (int f = 0; f < 4; ++f) {
        p.off[f] = code_off[f];
        p.vocab[f] = (codes && f < F) ? vocab[f] : 1;
    }
    p.err = err_word();
    p.resident = resident;
    p.ell = ell;
    p.ell_t = ell_t;
    p.codes_t = codes ? codes_src : nullptr;
    const long long nmax = (n_src > n_dst ? n_src : n_dst) + 1;
    const unsigned ngrid = (unsigned)((nmax + 255) / 256);
    const unsigned egrid = (unsigned)((E + 255) / 256 > 4096 ? 4096 : (E + 255) / 256);
    {
        ProfScope prof(K_CSR_HIST, s);
        if (!resident) PHC_LAUNCH(csr_pair_zero_kernel, dim3(ngrid, 2), dim3(256), 0, s, p);
        PHC_LAUNCH(csr_pair_hist_kernel, dim3(egrid, 2), dim3(256), 0, s, p, (long long)E);
    }
    {
        ProfScope prof(K_CSR_SCAN, s);
        if (nmax <= PAIR_SCAN_THREADS * 16)
            PHC_LAUNCH(csr_pair_scan_kernel<16>, dim3(2), dim3(PAIR_SCAN_THREADS), 0, s, p);
        else
            PHC_LAUNCH(csr_pair_scan_kernel<32>, dim3(2), dim3(PAIR_SCAN_THREADS), 0, s, p);
    }
    {
        ProfScope prof(K_CSR_FILL, s);
        PHC_LAUNCH(csr_pair_fill_kernel, dim3(egrid, 2), dim3(256), 0, s, p, (long long)E);
    }
    ProfScope prof(K_CSR_CANON, s);
    const long long groups = (nmax + 31) / 32;
    PHC_LAUNCH(csr_pair_canon_kernel, dim3((unsigned)(groups > 8192 ? 8192 : groups), 2), dim3(256), 0, s, p);
    return check_launch("csr_build_pair");
}

size_t phc_remove_isolated_ws_bytes(int64_t N) {
    if (N < 0) return 0;
    ScanLevels L = scan_levels(N + 1);
    return align_up(sizeof(int) * (size_t)(N + 1), 256) + align_up(sizeof(int) * (L.total_sum_ints + 1), 256);
}

int phc_remove_isolated_nodes(const int64_t *edge_src_i64, const int64_t *edge_dst_i64, int64_t E, int64_t N, const int64_t *x_i64,
                              int F, const int64_t *batch_i64, int64_t *edge_src_out, int64_t *edge_dst_out, int64_t *x_out,
                              int64_t *batch_out, int64_t *kept_index, int32_t *n_kept_dev, void *ws, size_t ws_bytes,
                              void *stream) {
    PHC_REQUIRE(E >= 0 && N >= 0 && F >= 0, PHC_ERR_INVALID_ARG, "phc_remove_isolated_nodes: E=%lld N=%lld F=%d", (long long)E,
                (long long)N, F);
    PHC_REQUIRE(N < (1LL << 31) - 1, PHC_ERR_UNSUPPORTED, "phc_remove_isolated_nodes: int32 index space exceeded");
    PHC_REQUIRE(E == 0 || (edge_src_i64 && edge_dst_i64 && edge_src_out && edge_dst_out), PHC_ERR_INVALID_ARG,
                "phc_remove_isolated_nodes: null edge pointer");
    PHC_REQUIRE(ws && ws_bytes >= phc_remove_isolated_ws_bytes(N), PHC_ERR_WORKSPACE, "phc_remove_isolated_nodes: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    ScanLevels L = scan_levels(N + 1);
    char *p = static_cast<char *>(ws);
    int *newid = reinterpret_cast<int *>(p);
    p += align_up(sizeof(int) * (size_t)(N + 1), 256);
    int *sums = reinterpret_cast<int *>(p);
    hipMemsetAsync(newid, 0, sizeof(int) * (size_t)(N + 1), s);
    const long long *src = reinterpret_cast<const long long *>(edge_src_i64);
    const long long *dst = reinterpret_cast<const long long *>(edge_dst_i64);
    const unsigned egrid = (unsigned)((E + 255) / 256 > 8192 ? 8192 : ((E + 255) / 256 < 1 ? 1 : (E + 255) / 256));
    if (E > 0) {
        ProfScope prof(K_CSR_HIST, s);
        PHC_LAUNCH(iso_flag_kernel, dim3(egrid), dim3(256), 0, s, src, dst, (long long)E, newid, (long long)N, err_word());
    }
    {
        ProfScope prof(K_CSR_SCAN, s);
        int *level_data[6];
        int *level_sums[6];
        level_data[0] = newid;
        int *sp = sums;
        for (int l = 0; l < L.levels; ++l) {
            level_sums[l] = (L.blocks[l] > 1) ? sp : nullptr;
            if (L.blocks[l] > 1) {
                level_data[l + 1] = sp;
                sp += L.blocks[l];
            }
        }
        for (int l = 0; l < L.levels; ++l)
            PHC_LAUNCH(scan_block_kernel, dim3((unsigned)L.blocks[l]), dim3(SCAN_THREADS), 0, s, level_data[l], level_sums[l], L.n[l]);
        for (int l = L.levels - 2; l >= 0; --l)
            PHC_LAUNCH(scan_add_kernel, dim3((unsigned)L.blocks[l]), dim3(SCAN_THREADS), 0, s, level_data[l], level_sums[l], L.n[l]);
    }
    ProfScope prof(K_CSR_FILL, s);
    if (E > 0)
        PHC_LAUNCH(iso_relabel_kernel, dim3(egrid), dim3(256), 0, s, src, dst, (long long)E, newid,
                   reinterpret_cast<long long *>(edge_src_out), reinterpret_cast<long long *>(edge_dst_out), (long long)N);
    const long long total = N * (long long)(F > 0 ? F : 1);
    const unsigned ngrid = (unsigned)((total + 255) / 256 > 8192 ? 8192 : ((total + 255) / 256 < 1 ? 1 : (total + 255) / 256));
    PHC_LAUNCH(iso_compact_kernel, dim3(ngrid), dim3(256), 0, s, newid, (long long)N, reinterpret_cast<const long long *>(x_i64), F,
               reinterpret_cast<const long long *>(batch_i64), reinterpret_cast<long long *>(x_out),
               reinterpret_cast<long long *>(batch_out), reinterpret_cast<long long *>(kept_index), n_kept_dev);
    return check_launch("remove_isolated_nodes");
}

}  // extern "C"
