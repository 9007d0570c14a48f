// Internal helpers shared by the gfx950 kernels of libphc_hip.so.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <tuple>
#include <type_traits>
#include <utility>
#include <stdio.h>

#include "phc_hip.h"

// experiment switch (python -m phc_gnn_amd.build --variant noxcd -DPHC_XCD_MAP=0): XCD-aware block permutations
#ifndef PHC_XCD_MAP
#define PHC_XCD_MAP 1
#endif

// A/B switch for the row walkers' contiguous-rows-per-XCD order alone (-DPHC_XCD_ROWS=0)
#ifndef PHC_XCD_ROWS
#define PHC_XCD_ROWS PHC_XCD_MAP
#endif

namespace phc {

// Workgroups go to the 8 XCDs round-robin by block id.  A row-walking kernel whose neighbouring rows gather the same rows
// (a molecule's atoms are consecutive rows and its bonds stay inside it) gives every XCD one CONTIGUOUS range of rows: a row
// that two neighbours gather comes from the Infinity Cache once and from that XCD's L2 afterwards.  The grid must be a
// multiple of 8 (xcd_grid); the permuted id stays below it, rows are bound-checked by the callers as before.
__device__ __forceinline__ unsigned xcd_contiguous_block(unsigned b, unsigned nb) {
    return PHC_XCD_ROWS ? (b & 7u) * (nb >> 3) + (b >> 3) : b;
}
static inline long long xcd_grid(long long blocks) { return (blocks + 7) / 8 * 8; }

// ---- error plumbing --------------------------------------------------------
void set_last_error(const char *fmt, ...);

#define PHC_REQUIRE(cond, code, ...)         \
    do {                                     \
        if (!(cond)) {                       \
            phc::set_last_error(__VA_ARGS__); \
            return (code);                   \
        }                                    \
    } while (0)

// ---- index validation (phc_set_error_word) --------------------------------------
int *err_word();                       // device word registered for the current device, or nullptr

// caller-supplied index -> a value that is safe to dereference; the first reader of an array also reports
__device__ __forceinline__ long long checked_index(long long v, long long n, int *err, int bit) {
    if ((unsigned long long)v >= (unsigned long long)n) {
        if (err) atomicOr(err, bit);
        return 0;
    }
    return v;
}
__device__ __forceinline__ long long safe_index(long long v, long long n) { return (unsigned long long)v < (unsigned long long)n ? v : 0; }

// ---- launch profiler ---------------------------------------------------------
// Kernel ids are stable indices into the name table in api.hip.
enum KernelId : int {
    K_PHM_BUILD_H = 0,
    K_PHM_GEMM_FWD,
    K_PHM_GEMM_DX,
    K_PHM_DH_PARTIAL,
    K_PHM_DH_REDUCE,
    K_PHM_CONTRACT,
    K_AGG_FWD,
    K_AGG_BWD,
    K_GATHER_ROWS,
    K_CSR_HIST,
    K_CSR_SCAN,
    K_CSR_FILL,
    K_CSR_CANON,
    K_ENCODE_FWD,
    K_AGG_ENC_FWD,
    K_AGG_ENC_DX,
    K_TABLE_GRAD,
    K_TABLE_GRAD_REDUCE,
    K_BN_STATS,
    K_BN_FINALIZE,
    K_BN_APPLY,
    K_BN_BWD_PARTIAL,
    K_BN_BWD_FINALIZE,
    K_BN_BWD_APPLY,
    K_PACK_CODES,
    K_SOFTMAX_FWD,
    K_SOFTMAX_BWD,
    K_BCE_FWD,
    K_BCE_BWD,
    K_GATED_POOL_FWD,
    K_GATED_POOL_BWD,
    K_CLIP_ADAM,
    K_SMALL_LINEAR,
    K_SUM_N,
    K_DOWNSTREAM_FWD,
    K_DOWNSTREAM_BWD,
    K_PHM_GEMM_AG,
    K_NUM_KERNELS
};

extern bool g_prof_on;
void prof_begin(int kid, hipStream_t s, int *slot);
void prof_end(int slot, hipStream_t s);

// Timing experiments only (results are WRONG): PHC_ABLATE=name[,name...] drops every launch of the named
// kernels (names as in phc_prof_read), so "step time with - without" gives a kernel's real cost inside a
// replayed hipGraph, where neither HIP events nor rocprofv3's serialised durations can be trusted.
extern bool g_ablate[];
extern thread_local bool tl_ablate;

// Launch profiler: the launch inside a profiled ProfScope goes through hipExtLaunchKernelGGL with the scope's event
// pair, which the runtime stamps with the dispatch's own begin / end times (what rocprofv3's kernel trace reports) —
// an event recorded before and after the launch would add the dispatch latency of two extra barrier packets
// (measured: 16 us instead of 11-13 us for the molecule-scale GEMM).  PHC_PROF_EVENT_PAIR=1 selects that older form.
extern thread_local hipEvent_t tl_prof_start, tl_prof_stop;

struct ProfScope {
    int slot = -1;
    hipStream_t s;
    ProfScope(int kid, hipStream_t stream) : s(stream) {
        tl_ablate = g_ablate[kid];
        if (g_prof_on) prof_begin(kid, s, &slot);
    }
    ~ProfScope() {
        if (slot >= 0) prof_end(slot, s);
        tl_ablate = false;
    }
};

// hipExtLaunchKernelGGL wants every kernel parameter spelled out; this form value-initialises the trailing ones a call
// leaves to their defaults (every default parameter of this library's kernels is nullptr / 0 — keep it that way).
template <typename... Formals, typename... Actuals, size_t... I>
inline void ext_launch_impl(void (*kernel)(Formals...), dim3 grid, dim3 block, unsigned shmem, hipStream_t s, hipEvent_t e0,
                            hipEvent_t e1, std::index_sequence<I...>, Actuals... actuals) {
    static_assert(sizeof...(Actuals) <= sizeof...(Formals), "too many kernel arguments");
    std::tuple<std::remove_cv_t<Formals>...> formals{};
    std::tuple<Actuals...> given{actuals...};
    ((std::get<I>(formals) = std::get<I>(given)), ...);
    void *args[sizeof...(Formals)];
    size_t n = 0;
    std::apply([&](auto &...f) { ((args[n++] = (void *)&f), ...); }, formals);
    hipExtLaunchKernel(reinterpret_cast<void *>(kernel), grid, block, args, shmem, s, e0, e1, 0);
}
template <typename... Formals, typename... Actuals>
inline void ext_launch(void (*kernel)(Formals...), dim3 grid, dim3 block, unsigned shmem, hipStream_t s, hipEvent_t e0,
                       hipEvent_t e1, Actuals... actuals) {
    ext_launch_impl(kernel, grid, block, shmem, s, e0, e1, std::index_sequence_for<Actuals...>{}, actuals...);
}

#define PHC_LAUNCH(kernel, grid, block, shmem, stream, ...)                                                               \
    do {                                                                                                                  \
        if (!phc::tl_ablate) {                                                                                            \
            if (phc::tl_prof_start)                                                                                       \
                phc::ext_launch(kernel, grid, block, shmem, stream, phc::tl_prof_start, phc::tl_prof_stop, __VA_ARGS__);  \
            else                                                                                                          \
                hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                                      \
        }                                                                                                                 \
    } while (0)

// ---- batch-norm parameter tables (one BatchNorm1d per hypercomplex component, norm.py:20), by value in the kernarg segment ----
constexpr int BN_MAX_COMP = PHC_MAX_COMPONENTS;
struct BnParams {
    const float *weight[BN_MAX_COMP];
    const float *bias[BN_MAX_COMP];
    float *running_mean[BN_MAX_COMP];
    float *running_var[BN_MAX_COMP];
    long long *num_batches_tracked[BN_MAX_COMP];
    int n_comp;
    int k;              // columns per component
    int has_weight, has_bias, has_running, has_nbt;
    int flat;           // every table present lays its components out back to back (tbl[c] == tbl[0] + c * k): column `col` of a
                        // table is then tbl[0][col] — one load, no pointer hop (phc/hypercomplex/norm.py keeps them that way)
};
inline int fill_params(const char *who, BnParams *p, const float *const *weight, const float *const *bias,
                       float *const *running_mean, float *const *running_var, int64_t *const *nbt, int n_comp, int h) {
    PHC_REQUIRE(n_comp >= 1 && n_comp <= BN_MAX_COMP, PHC_ERR_UNSUPPORTED, "%s: %d components (max %d)", who, n_comp, BN_MAX_COMP);
    PHC_REQUIRE(h >= 1 && h % n_comp == 0, PHC_ERR_INVALID_ARG, "%s: h=%d not divisible by n_comp=%d", who, h, n_comp);
    p->n_comp = n_comp;
    p->k = h / n_comp;
    p->has_weight = weight != nullptr;
    p->has_bias = bias != nullptr;
    p->has_running = running_mean != nullptr && running_var != nullptr;
    p->has_nbt = nbt != nullptr;
    for (int c = 0; c < n_comp; ++c) {
        p->weight[c] = weight ? weight[c] : nullptr;
        p->bias[c] = bias ? bias[c] : nullptr;
        p->running_mean[c] = running_mean ? running_mean[c] : nullptr;
        p->running_var[c] = running_var ? running_var[c] : nullptr;
        p->num_batches_tracked[c] = nbt ? reinterpret_cast<long long *>(nbt[c]) : nullptr;
        PHC_REQUIRE((!weight || weight[c]) && (!bias || bias[c]) && (!p->has_running || (running_mean[c] && running_var[c])),
                    PHC_ERR_INVALID_ARG, "%s: null parameter pointer for component %d", who, c);
    }
    p->flat = 1;
    for (int c = 1; c < n_comp; ++c) {
        const size_t o = (size_t)c * p->k;
        if ((weight && weight[c] != weight[0] + o) || (bias && bias[c] != bias[0] + o) ||
            (p->has_running && (running_mean[c] != running_mean[0] + o || running_var[c] != running_var[0] + o)))
            p->flat = 0;
    }
    return PHC_OK;
}


// ---- statistics accumulators --------------------------------------------------------------------------------------
// Column sums that many workgroups of one launch contribute to, WITHOUT a merge launch behind it: every workgroup adds its
// partial with 64-bit INTEGER atomics (integer adds commute: the total does not depend on the arrival order, results stay
// bit-reproducible) and the next launch reads the totals in its prologue.  A value d is split into two fixed-point words,
// hi = rint(d * 2^12) and lo = rint((d - hi * 2^-12) * 2^62): resolution 2^-62 up to |total| < 2^51 — partials beyond that
// saturate, by then the fp32 activations themselves have long overflowed.  Layout (caller-zeroed, 8-byte words):
// [STAT_ACC_REP replicas][2 quantities][hi | lo][h columns]; a workgroup adds into replica (its row-group index % REP),
// which spreads the same-address traffic (108 adders per address cost +2 us per launch, 27 per address +1).
constexpr int STAT_ACC_REP = 4;
__device__ __forceinline__ void stat_acc_add(unsigned long long *__restrict__ acc, int h, int rep, int qty, int col, double d) {
    const long long hi = __double2ll_rn(d * 4096.0);
    const long long lo = __double2ll_rn((d - (double)hi * (1.0 / 4096.0)) * 4611686018427387904.0);
    unsigned long long *p = acc + ((size_t)(rep * 2 + qty) * 2) * h + col;
    atomicAdd(p, (unsigned long long)hi);
    atomicAdd(p + h, (unsigned long long)lo);
}
// one replica's words of (qty, col) as a double (the caller adds the replicas in a fixed order)
__device__ __forceinline__ double stat_acc_part(const unsigned long long *__restrict__ acc, int h, int rep, int qty, int col) {
    const unsigned long long *p = acc + ((size_t)(rep * 2 + qty) * 2) * h + col;
    const long long hi = (long long)p[0], lo = (long long)p[h];
    return (double)hi * (1.0 / 4096.0) + (double)lo * (1.0 / 4611686018427387904.0);
}
__device__ __forceinline__ double stat_acc_total(const unsigned long long *__restrict__ acc, int h, int qty, int col) {
    long long hi = 0, lo = 0;
#pragma unroll
    for (int rep = 0; rep < STAT_ACC_REP; ++rep) {
        const unsigned long long *p = acc + ((size_t)(rep * 2 + qty) * 2) * h + col;
        hi += (long long)p[0];
        lo += (long long)p[h];
    }
    return (double)hi * (1.0 / 4096.0) + (double)lo * (1.0 / 4611686018427387904.0);
}

// The pointer of component `comp` out of one of BnParams' by-value tables, by a select chain over CONSTANT indices.  As
// `for (q < n_comp) if (q == comp) v = tbl[q][kk]` the table is indexed at run time: hipcc reads the pointer with a vector load from
// the kernel-argument segment and waits for it (vmcnt(0)) before it can load the value — weight, bias, running mean and running
// variance were four serial pairs of round trips in every kernel's prologue (3.3 us of the 6 us in front of the first barrier of
// gemm_hs_ag_kernel, scripts/dbg/ag_phases.py).  With constant indices the pointers are scalar loads issued at kernel entry and the
// values are independent loads that travel together.
template <typename T>
__device__ __forceinline__ T *bn_pick(T *const (&tbl)[BN_MAX_COMP], int comp) {
    T *p = tbl[0];
#pragma unroll
    for (int q = 1; q < BN_MAX_COMP; ++q) p = comp == q ? tbl[q] : p;
    return p;
}

// &table(comp)[kk] = the entry of column col = comp * k + kk: from the first table when the components are contiguous.  (hipcc
// evaluates both sides of the select — the table walk's pointer load still stands in front of the value load; a run-time BRANCH
// around the two loads is worse: every join costs an s_waitcnt vmcnt(0).  Kernels where that round trip is on the critical path take
// the layout as a template parameter instead: gemm_hs_ag_kernel.)
template <typename T>
__device__ __forceinline__ T *bn_col_ptr(T *const (&tbl)[BN_MAX_COMP], int flat, int comp, int kk, int col) {
    return flat ? tbl[0] + col : bn_pick(tbl, comp) + kk;
}

// ---- online softmax of the softmax neighbour aggregation (softmax_agg.hip; mlp_fused.hip's aggregating forward) ----
// torch-scatter 2.0.5's scatter_softmax: max-shifted, eps = 1e-12 added to the segment sum (messagepassing.py:211-214, 297-300).
// One column's running (max, sum, weighted sum) takes the next message m of its segment, in edge order.
constexpr float SM_EPS = 1e-12f;
__device__ __forceinline__ void sm_update(float &mx, float &s, float &w, float m, float beta) {
    // Both cases of the recurrence — a new maximum (rescale the sums by r = exp(mx - sc), add 1 | m) and an ordinary message (add
    // p = exp(sc - mx) | p m) — take ONE exponential, of -|sc - mx| (mx = -inf on the first edge: 0).  Written without a branch: the
    // lanes of a wavefront disagree about the case, and as two branches each lane paid both exponentials.
    const float sc = beta * m;
    const bool gt = sc > mx;
    const float e = __expf(-fabsf(sc - mx));
    s = gt ? fmaf(s, e, 1.f) : s + e;
    w = gt ? fmaf(w, e, m) : fmaf(e, m, w);
    mx = gt ? sc : mx;
}

// Norm constants from a column's accumulator totals (S = sum x, Q = sum x^2 over `count` rows; norm.py:30-35 forward, training mode).
// ONE definition for every kernel that derives them in its prologue (bn_apply_acc_kernel, the normalising-operand GEMMs, the
// aggregations that apply a pending norm): several tests and the step's A/B switches rely on those kernels producing the SAME bits,
// and with -ffp-contract=fast the compiler is free to fuse  b - mean * invstd * w  or  (1 - m) * old + m * new  differently in
// different kernels (seen: the running variance one ulp apart).  Contraction is off inside.
struct BnFromTotals {
    float mean, invstd, scale, shift, unbiased;
};
__device__ __forceinline__ BnFromTotals bn_from_totals(double S, double Q, long long count, float w, float b, float eps) {
#pragma clang fp contract(off)
    BnFromTotals r;
    const double cntd = (double)(count > 1 ? count : 1);
    const double meand = S / cntd;
    const double vard = fmax(Q / cntd - meand * meand, 0.0);
    const float var = (float)vard;
    r.mean = (float)meand;
    r.invstd = 1.f / sqrtf(var + eps);
    r.scale = r.invstd * w;
    r.shift = b - r.mean * r.invstd * w;
    r.unbiased = count > 1 ? (float)(vard * cntd / (cntd - 1.0)) : var;
    return r;
}
__device__ __forceinline__ float bn_running_update(float old, float momentum, float value) {
#pragma clang fp contract(off)
    return (1.f - momentum) * old + momentum * value;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE, PER-KERNEL attribute.  The cache is keyed by the kernel's ADDRESS and
// the device ordinal (api.hip: a mutex-guarded set) — every instantiation of a kernel template has the same function-pointer TYPE,
// so a static inside a template over that type would be shared by the whole family and only the first instantiation launched on a
// device would get the attribute.  false = the runtime refused (the caller takes its fallback path or reports PHC_ERR_LAUNCH).
bool ensure_dynamic_lds_addr(const void *kernel, size_t bytes);
template <typename K>
inline bool ensure_dynamic_lds(K kernel, size_t bytes) {
    return ensure_dynamic_lds_addr(reinterpret_cast<const void *>(kernel), bytes);
}

// ---- dropout fused behind the activation (models.py:211-213, downstream.py:108-111: act -> phm_dropout -> (+ skip)) ----
// phm_dropout (layers.py:31-55): p = drop rate, kept entries scaled by 1 / (1 - p); same = one mask per (row, channel)
// shared by the n hypercomplex components.  The mask is a pure function of (seed, step, site, row, channel): Philox4x32-10
// with key = seed, counter = (element group, site, step), so the backward kernels regenerate it instead of reading a
// stored mask, and a replayed hipGraph draws a fresh mask every step (step is a device counter, phc_rng_advance).
struct BnDrop {
    float p;                        // 0: off
    int same, k, site;              // k = columns per component
    const unsigned long long *state;   // device: {seed, step}
};

__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 key) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = make_uint4(hi1 ^ c.y ^ key.x, lo1, hi0 ^ c.w ^ key.y, lo0);
        key.x += 0x9E3779B9u;
        key.y += 0xBB67AE85u;
    }
    return c;
}

// keep-and-scale factors of the 4 consecutive columns starting at col (col % 4 == 0, k % 4 == 0) of row r — the mask's parameters by
// VALUE: a kernel that passes a member of its by-value argument struct by reference makes hipcc copy the whole struct into scratch
// (seen with GemmHsArgs::ba_drop: 432 - 720 bytes of private segment and an 88 instead of 30 us launch)
__device__ __forceinline__ float4 drop_factors_v(float p, int same, int k, int site, unsigned long long seed, unsigned long long step,
                                                 long long r, int col, int h) {
    const int chan = same ? col % k : col;                           // same: the mask ignores the component
    const int width = same ? k : h;
    const unsigned long long g = (unsigned long long)r * (unsigned long long)(width >> 2) + (unsigned long long)(chan >> 2);
    const uint4 rnd = philox4x32_10(make_uint4((unsigned)g, (unsigned)(g >> 32), (unsigned)site, (unsigned)step),
                                    make_uint2((unsigned)seed, (unsigned)(seed >> 32) ^ (unsigned)(step >> 32)));
    const float inv = 1.f / (1.f - p);
    const float u0 = rnd.x * 2.3283064365386963e-10f, u1 = rnd.y * 2.3283064365386963e-10f, u2 = rnd.z * 2.3283064365386963e-10f,
                u3 = rnd.w * 2.3283064365386963e-10f;
    return make_float4(u0 >= p ? inv : 0.f, u1 >= p ? inv : 0.f, u2 >= p ? inv : 0.f, u3 >= p ? inv : 0.f);
}
__device__ __forceinline__ float4 drop_factors(const BnDrop &d, unsigned long long seed, unsigned long long step, long long r, int col,
                                               int h) {
    return drop_factors_v(d.p, d.same, d.k, d.site, seed, step, r, col, h);
}

inline int make_drop(const char *who, BnDrop *d, float p, int same_k, int site, const uint64_t *state, int h) {
    d->p = 0.f; d->same = 0; d->k = h; d->site = site; d->state = nullptr;
    if (p <= 0.f) return PHC_OK;
    PHC_REQUIRE(p < 1.f && state, PHC_ERR_INVALID_ARG, "%s: dropout p=%g needs 0 <= p < 1 and an rng state", who, (double)p);
    PHC_REQUIRE(h % 4 == 0 && (same_k == 0 || (same_k % 4 == 0 && h % same_k == 0)), PHC_ERR_UNSUPPORTED,
                "%s: fused dropout needs h %% 4 == 0 (and k %% 4 == 0 for the shared mask)", who);
    d->p = p;
    d->same = same_k > 0;
    d->k = same_k > 0 ? same_k : h;
    d->state = reinterpret_cast<const unsigned long long *>(state);
    return PHC_OK;
}

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_last_error("%s: %s", what, hipGetErrorString(e));
        return PHC_ERR_LAUNCH;
    }
    return PHC_OK;
}

// ---- device helpers ------------------------------------------------------------
constexpr int WAVE = 64;

template <int ACT>
__device__ __forceinline__ float act_fwd(float v) {
    if constexpr (ACT == PHC_ACT_RELU) return v > 0.f ? v : 0.f;
    if constexpr (ACT == PHC_ACT_LRELU) return v > 0.f ? v : 0.01f * v;
    if constexpr (ACT == PHC_ACT_ELU) return v > 0.f ? v : expm1f(v);
    if constexpr (ACT == PHC_ACT_SWISH) return v / (1.f + __expf(-v));
    if constexpr (ACT == PHC_ACT_SELU)
        return 1.0507009873554804934193349852946f * (v > 0.f ? v : 1.6732632423543772848170429916717f * expm1f(v));
    return v;
}

// d phi / d v evaluated at the pre-activation v
// activation chosen at run time (wave-uniform switch): used where a template parameter per activation would
// multiply the instantiations of a large kernel
__device__ __forceinline__ float act_fwd_rt(int act, float v) {
    switch (act) {
        case PHC_ACT_RELU: return act_fwd<PHC_ACT_RELU>(v);
        case PHC_ACT_LRELU: return act_fwd<PHC_ACT_LRELU>(v);
        case PHC_ACT_ELU: return act_fwd<PHC_ACT_ELU>(v);
        case PHC_ACT_SWISH: return act_fwd<PHC_ACT_SWISH>(v);
        case PHC_ACT_SELU: return act_fwd<PHC_ACT_SELU>(v);
        default: return v;
    }
}

template <int ACT>
__device__ __forceinline__ float act_grad(float v) {
    if constexpr (ACT == PHC_ACT_RELU) return v > 0.f ? 1.f : 0.f;
    if constexpr (ACT == PHC_ACT_LRELU) return v > 0.f ? 1.f : 0.01f;
    if constexpr (ACT == PHC_ACT_ELU) return v > 0.f ? 1.f : __expf(v);
    if constexpr (ACT == PHC_ACT_SWISH) {
        float s = 1.f / (1.f + __expf(-v));
        return s * (1.f + v * (1.f - s));
    }
    if constexpr (ACT == PHC_ACT_SELU)
        return 1.0507009873554804934193349852946f * (v > 0.f ? 1.f : 1.6732632423543772848170429916717f * __expf(v));
    return 1.f;
}

// derivative of the activation at pre-activation z, chosen at run time (see act_fwd_rt)
__device__ __forceinline__ float act_grad_rt(int act, float z) {
    switch (act) {
        case PHC_ACT_RELU: return act_grad<PHC_ACT_RELU>(z);
        case PHC_ACT_LRELU: return act_grad<PHC_ACT_LRELU>(z);
        case PHC_ACT_ELU: return act_grad<PHC_ACT_ELU>(z);
        case PHC_ACT_SWISH: return act_grad<PHC_ACT_SWISH>(z);
        case PHC_ACT_SELU: return act_grad<PHC_ACT_SELU>(z);
        default: return 1.f;
    }
}

}  // namespace phc
