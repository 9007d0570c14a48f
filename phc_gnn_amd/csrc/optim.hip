// Gradient clipping + Adam for a whole model in two launches.
//
// Reference: benchmarks/train_hiv.py:198-201 — torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm) followed by
// torch.optim.Adam(...).step() (Adam with weight_decay = 0, train_hiv.py:285).  PyTorch runs that as ~21 multi-tensor
// launches over the ~230 small parameter tensors of a PHC model; at molecule scale each launch is pure latency.
// Here: (1) per-chunk sums of squares (+ the per-tensor step counters), (2) one workgroup: total norm (fixed order), clip coefficient,
// (3) the Adam update with the clip coefficient applied on the fly.  Same arithmetic as torch's fused Adam:
//   m = m + (g - m)(1 - b1);  v = b2 v + (1 - b2) g g;  p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// with g = clip_coef * grad, clip_coef = min(1, max_norm / (||grads||_2 + 1e-6)).
#include <stdlib.h>

#include "phc_common.h"

namespace phc {

constexpr int OPT_CHUNK = 4096;                 // elements per workgroup
constexpr int OPT_MAX_TENSORS = 448;            // gradient pointers travel by value (4 KB kernarg limit)

struct OptGrads {
    const float *g[OPT_MAX_TENSORS];
};

// table[t] = {param, exp_avg, exp_avg_sq, numel} (u64 each); chunks[c] = {tensor, first element}
__global__ __launch_bounds__(256) void opt_sumsq_kernel(OptGrads grads, const unsigned long long *__restrict__ table,
                                                      const int *__restrict__ chunks, float *__restrict__ partial,
                                                      float *__restrict__ steps) {
    __shared__ float red[256];
    const int t = chunks[2 * blockIdx.x], off = chunks[2 * blockIdx.x + 1];
    const float *g = grads.g[t];
    const long long numel = (long long)table[4 * t + 3];
    float s = 0.f;
    if (g) {
        // the chunk's OPT_CHUNK / 256 elements of this thread in one round trip (clamped addresses, selected afterwards):
        // a runtime-bounded loop costs a full memory latency per iteration
        const long long end = min((long long)off + OPT_CHUNK, numel);
        float gv[OPT_CHUNK / 256];
#pragma unroll
        for (int q = 0; q < OPT_CHUNK / 256; ++q) gv[q] = g[min((long long)off + threadIdx.x + 256 * q, end - 1)];
#pragma unroll
        for (int q = 0; q < OPT_CHUNK / 256; ++q) {
            const float v = (long long)off + threadIdx.x + 256 * q < end ? gv[q] : 0.f;
            s = fmaf(v, v, s);
        }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {                  // fixed tree
        if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = red[0];
        if (g && off == 0) steps[t] += 1.f;                 // torch.optim.Adam counts steps per parameter tensor
    }
}

// scal[0] = total norm, scal[1] = clip coefficient
__global__ __launch_bounds__(256) void opt_finalize_kernel(const float *__restrict__ partial, int num_chunks, float max_norm,
                                                         float *__restrict__ scal) {
    __shared__ float red[256];
    float s = 0.f;
    for (int c = threadIdx.x; c < num_chunks; c += 256) s += partial[c];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float norm = sqrtf(red[0]);
        scal[0] = norm;
        scal[1] = max_norm > 0.f ? fminf(1.f, max_norm / (norm + 1e-6f)) : 1.f;
    }
}

__global__ __launch_bounds__(256) void opt_adam_kernel(OptGrads grads, const unsigned long long *__restrict__ table,
                                                     const int *__restrict__ chunks, float *__restrict__ scal,
                                                     const float *__restrict__ steps, const float *__restrict__ partial,
                                                     int total_chunks, float max_norm, float lr, float beta1, float beta2,
                                                     float eps, int write_back) {
    const int t = chunks[2 * blockIdx.x], off = chunks[2 * blockIdx.x + 1];
    const float *g = grads.g[t];
    if (!g && blockIdx.x != 0) return;                       // parameter without a gradient this step: untouched
    float *p = reinterpret_cast<float *>(table[4 * t]);
    float *m = reinterpret_cast<float *>(table[4 * t + 1]);
    float *v = reinterpret_cast<float *>(table[4 * t + 2]);
    const long long numel = (long long)table[4 * t + 3];
    const float tstep = steps[t];                            // already incremented by the sums-of-squares launch
    // total norm from the per-chunk partials: every workgroup adds them in the same fixed order (a few hundred values),
    // which is cheaper than a separate single-workgroup launch between the two passes
    __shared__ float red[256];
    __shared__ float s_coef;
    {
        float sacc = 0.f;
        for (int c = threadIdx.x; c < total_chunks; c += 256) sacc += partial[c];
        red[threadIdx.x] = sacc;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            const float norm = sqrtf(red[0]);
            s_coef = max_norm > 0.f ? fminf(1.f, max_norm / (norm + 1e-6f)) : 1.f;
            if (blockIdx.x == 0 && scal) {
                scal[0] = norm;
                scal[1] = s_coef;
            }
        }
        __syncthreads();
    }
    const float coef = s_coef, bc1 = 1.f - powf(beta1, tstep), bc2_sqrt = sqrtf(1.f - powf(beta2, tstep));
    if (!g) return;                                          // (workgroup 0 stayed to publish the norm)
    const float step_size = lr / bc1;
    const long long end = min((long long)off + OPT_CHUNK, numel);
    constexpr int PER = OPT_CHUNK / 256;
    float g_[PER], m_[PER], v_[PER], p_[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {                          // all 4 * PER loads in flight together (clamped addresses)
        const long long e = min((long long)off + threadIdx.x + 256 * q, end - 1);
        g_[q] = g[e]; m_[q] = m[e]; v_[q] = v[e]; p_[q] = p[e];
    }
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const long long e = (long long)off + threadIdx.x + 256 * q;
        if (e < end) {
            const float gv = g_[q] * coef;
            float mv = m_[q], vv = v_[q];
            mv = mv + (gv - mv) * (1.f - beta1);             // lerp, as torch's fused kernel
            vv = beta2 * vv + (1.f - beta2) * gv * gv;
            const float denom = sqrtf(vv) / bc2_sqrt + eps;
            p[e] = p_[q] - step_size * (mv / denom);
            m[e] = mv;
            v[e] = vv;
            if (write_back) const_cast<float *>(g)[e] = gv;  // clip_grad_norm_ scales the gradients in place
        }
    }
}


// ---- clip + Adam in ONE launch (round 6) ------------------------------------------------------------------------------
// The two launches above are one dependent pair: every workgroup of the update needs ALL per-chunk sums of squares.  Inside
// a replayed step the pair costs ~20 us for ~3 us of work (scripts/dbg/r06_ablate.sh), so the hand-over moves inside the
// launch: a workgroup loads its chunk (gradient, both moments, parameter: all four round trips in flight together), reduces
// its sum of squares with the SAME fixed tree as opt_sumsq_kernel, publishes it as a {value, tag} granule (write-through
// store, csrc/downstream.hip) and collects every chunk's granule, repeating the loads until each tag equals this launch's.
// The total norm is then summed in the same fixed order as opt_adam_kernel sums `partial` — the update is bit-identical to
// the two launches'.  All workgroups must be co-resident (they wait for each other): the host launches this form only when
// the chunk count fits the device's real capacity for this kernel, and the poll loop is bounded (OPT_SPIN_MAX ->
// PHC_ERR_RENDEZVOUS in the device error word, a NaN coefficient) so that an unforeseen stall ends the launch.
// The per-tensor step counter is read by every workgroup BEFORE it publishes (so before anybody can be past the collect) and
// written back by the tensor's first chunk AFTER the collect: no workgroup can see the incremented value.
constexpr int OPT_SPIN_MAX = 1 << 21;

__global__ __launch_bounds__(256) void opt_clip_adam_fused_kernel(OptGrads grads, const unsigned long long *__restrict__ table,
                                                                const int *__restrict__ chunks, float *__restrict__ scal,
                                                                float *__restrict__ steps, unsigned long long *__restrict__ gran,
                                                                unsigned *__restrict__ gen, int *__restrict__ err, int total_chunks,
                                                                float max_norm, float lr, float beta1, float beta2, float eps,
                                                                int write_back) {
    __shared__ float red[256];
    __shared__ float s_coef;
    const unsigned tag = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    const int t = chunks[2 * blockIdx.x], off = chunks[2 * blockIdx.x + 1];
    const float *g = grads.g[t];
    float *p = reinterpret_cast<float *>(table[4 * t]);
    float *m = reinterpret_cast<float *>(table[4 * t + 1]);
    float *v = reinterpret_cast<float *>(table[4 * t + 2]);
    const long long numel = (long long)table[4 * t + 3];
    const float tstep = steps[t] + (g ? 1.f : 0.f);           // (in flight in front of the chunk's loads: it has landed before the publish)
    const long long end = min((long long)off + OPT_CHUNK, numel);
    constexpr int PER = OPT_CHUNK / 256;
    float g_[PER], m_[PER], v_[PER], p_[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {                          // all 4 * PER loads in flight together (clamped addresses)
        const long long e = min((long long)off + threadIdx.x + 256 * q, end - 1);
        g_[q] = g ? g[e] : 0.f;
        m_[q] = m[e]; v_[q] = v[e]; p_[q] = p[e];
    }
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const float gv = (long long)off + threadIdx.x + 256 * q < end ? g_[q] : 0.f;
        s = fmaf(gv, gv, s);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {                  // fixed tree (opt_sumsq_kernel's)
        if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0)
        __hip_atomic_store(gran + blockIdx.x, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(red[0]),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();                                         // (red[] is reused below)
    // ---- collect: chunk c by thread c % 256, in opt_adam_kernel's order ----
    float sacc = 0.f;
    bool dead = false;
    for (int c = threadIdx.x; c < total_chunks; c += 256) {
        unsigned long long q;
        int spins = 0;
        do {
            q = __hip_atomic_load(gran + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(q >> 32) == tag) break;
            __builtin_amdgcn_s_sleep(1);
        } while (++spins <= OPT_SPIN_MAX);
        if ((unsigned)(q >> 32) != tag) dead = true;
        sacc += __uint_as_float((unsigned)q);
    }
    if (dead) {
        if (err) atomicOr(err, PHC_ERR_RENDEZVOUS);
        sacc = __int_as_float(0x7fc00000);                   // a timed-out exchange must not look like a result
    }
    red[threadIdx.x] = sacc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float norm = sqrtf(red[0]);
        s_coef = max_norm > 0.f ? fminf(1.f, max_norm / (norm + 1e-6f)) : 1.f;
        if (blockIdx.x == 0) {
            if (scal) { scal[0] = norm; scal[1] = s_coef; }
            // (every other workgroup has published, i.e. has read the generation word: the bump cannot reach one of this launch)
            __hip_atomic_fetch_add(gen, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (g && off == 0) steps[t] = tstep;                 // torch.optim.Adam counts steps per parameter tensor
    }
    __syncthreads();
    if (!g) return;                                          // parameter without a gradient this step: untouched
    const float coef = s_coef, bc1 = 1.f - powf(beta1, tstep), bc2_sqrt = sqrtf(1.f - powf(beta2, tstep));
    const float step_size = lr / bc1;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const long long e = (long long)off + threadIdx.x + 256 * q;
        if (e < end) {
            const float gv = g_[q] * coef;
            float mv = m_[q], vv = v_[q];
            mv = mv + (gv - mv) * (1.f - beta1);             // lerp, as torch's fused kernel
            vv = beta2 * vv + (1.f - beta2) * gv * gv;
            const float denom = sqrtf(vv) / bc2_sqrt + eps;
            p[e] = p_[q] - step_size * (mv / denom);
            m[e] = mv;
            v[e] = vv;
            if (write_back) const_cast<float *>(g)[e] = gv;  // clip_grad_norm_ scales the gradients in place
        }
    }
}


// All gradient tensors of a model -> ONE flat buffer (laid out like the optimizer's flat moment buffers), scaled on the way:
// the send buffer of the data-parallel all-reduce in one launch, mean folded in (sum_r g_r / W = sum_r (g_r / W)).  A parameter
// without a gradient contributes zeros.  The flat offset of tensor t is the offset of its first-moment view in exp_avg_flat.
__global__ __launch_bounds__(256) void opt_pack_kernel(OptGrads grads, const unsigned long long *__restrict__ table,
                                                     const int *__restrict__ chunks, const float *__restrict__ m_base,
                                                     float *__restrict__ out, float scale) {
    const int t = chunks[2 * blockIdx.x], off = chunks[2 * blockIdx.x + 1];
    const float *g = grads.g[t];
    const float *m = reinterpret_cast<const float *>(table[4 * t + 1]);
    const long long numel = (long long)table[4 * t + 3];
    float *dst = out + (m - m_base);
    const long long end = min((long long)off + OPT_CHUNK, numel);
    constexpr int PER = OPT_CHUNK / 256;
    float gv[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) gv[q] = g ? g[min((long long)off + threadIdx.x + 256 * q, end - 1)] : 0.f;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const long long e = (long long)off + threadIdx.x + 256 * q;
        if (e < end) dst[e] = gv[q] * scale;
    }
}


// ---- explicit weight regulariser of the training scripts (phm_weight_regularization) --------------------------------
// reg = coef * sum over PHM layers of mean_{kappa,pi} || W[:, kappa, pi] ||_p   (regularization.py:15-23: weight.norm(p, dim=0)
// .mean(); coef = lr * weight_decay, train_hiv.py:181-196).  ONE launch: a workgroup owns 256 (kappa, pi) cells of one layer,
// adds d reg / d W into that layer's gradient tensor and publishes its share of the value as a {value, tag} granule
// (write-through store, csrc/downstream.hip); workgroup 0 collects all shares in block order (fixed summation order) and
// writes the value / adds it to the loss.  The tag is a generation word bumped by workgroup 0 at the end of every launch.
constexpr int REG_MAX_JOBS = 48;
constexpr int REG_MAX_BLOCKS = 1024;
constexpr int REG_MAX_N = 16;
struct RegJobs {
    const float *W[REG_MAX_JOBS];
    float *dW[REG_MAX_JOBS];
    int n[REG_MAX_JOBS], cells[REG_MAX_JOBS], first_block[REG_MAX_JOBS + 1];      // W is [n][cells], cells = kin * kout
    int num;
};

__global__ __launch_bounds__(256) void phm_weight_reg_kernel(RegJobs jobs, float coef, int p, float *__restrict__ reg_out,
                                                             float *__restrict__ loss_inout, unsigned long long *__restrict__ gran,
                                                             unsigned *__restrict__ gen_word) {
    __shared__ float red[4];
    const int tid = threadIdx.x, b = blockIdx.x;
    const unsigned tag = __hip_atomic_load(gen_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    int j = 0;
    for (int q = 1; q < jobs.num; ++q) j = b >= jobs.first_block[q] ? q : j;       // uniform scan of the by-value table
    const float *W = nullptr;
    float *dW = nullptr;
    int n = 1, cells = 1;
    for (int q = 0; q < jobs.num; ++q)
        if (q == j) { W = jobs.W[q]; dW = jobs.dW[q]; n = jobs.n[q]; cells = jobs.cells[q]; }
    const int c = (b - jobs.first_block[j]) * 256 + tid;
    const bool live = c < cells;
    const int cc = live ? c : 0;
    float w[REG_MAX_N];
#pragma unroll
    for (int i = 0; i < REG_MAX_N; ++i) w[i] = W[(size_t)min(i, n - 1) * cells + cc];     // one round of unconditional loads
    float nrm = 0.f;
#pragma unroll
    for (int i = 0; i < REG_MAX_N; ++i)
        if (i < n) nrm += p == 1 ? fabsf(w[i]) : w[i] * w[i];
    if (p != 1) nrm = sqrtf(nrm);
    if (dW && live) {
        const float scale = coef / (float)cells;
#pragma unroll
        for (int i = 0; i < REG_MAX_N; ++i)
            if (i < n) {
                // d ||v||_1 / dv = sign(v);  d ||v||_2 / dv = v / ||v||  (0 at the origin, as torch's norm backward)
                const float gq = p == 1 ? (w[i] > 0.f ? 1.f : (w[i] < 0.f ? -1.f : 0.f)) : (nrm > 0.f ? w[i] / nrm : 0.f);
                dW[(size_t)i * cells + c] += scale * gq;
            }
    }
    float v = live ? nrm * (coef / (float)cells) : 0.f;            // this cell's share of coef * mean
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    if (tid == 0) {
        const float part = (red[0] + red[1]) + (red[2] + red[3]);
        __hip_atomic_store(gran + b, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(part), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    if (b != 0) return;
    // workgroup 0: all shares, block order
    const int nb = gridDim.x;
    float total = 0.f;
    __shared__ float shares[REG_MAX_BLOCKS];
    for (int q = tid; q < nb; q += 256) {
        unsigned long long g;
        do {
            g = __hip_atomic_load(gran + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } while ((unsigned)(g >> 32) != tag);
        shares[q] = __uint_as_float((unsigned)g);
    }
    __syncthreads();
    if (tid == 0) {
        for (int q = 0; q < nb; ++q) total += shares[q];
        *reg_out = total;
        if (loss_inout) *loss_inout += total;
        __hip_atomic_fetch_add(gen_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace phc

using namespace phc;

extern "C" {

int phc_clip_adam_max_tensors(void) { return OPT_MAX_TENSORS; }
int phc_clip_adam_chunk(void) { return OPT_CHUNK; }

int phc_clip_adam_step(const uint64_t *table_dev, const float *const *grads, int num_tensors, const int32_t *chunks_dev,
                       int num_chunks, int chunk_base, int total_chunks, int phases, float max_norm, float lr, float beta1,
                       float beta2, float eps, float *step_dev, float *scal_dev, float *partial_dev, int write_back_clipped,
                       void *stream) {
    PHC_REQUIRE(num_tensors >= 1 && num_tensors <= OPT_MAX_TENSORS, PHC_ERR_UNSUPPORTED,
                "phc_clip_adam_step: %d tensors (1..%d per call)", num_tensors, OPT_MAX_TENSORS);
    PHC_REQUIRE(num_chunks >= 1 && chunk_base >= 0 && total_chunks >= chunk_base + num_chunks && table_dev && grads && chunks_dev &&
                    step_dev && scal_dev && partial_dev,
                PHC_ERR_INVALID_ARG, "phc_clip_adam_step: bad arguments");
    PHC_REQUIRE(lr > 0.f && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps > 0.f, PHC_ERR_INVALID_ARG,
                "phc_clip_adam_step: bad hyper-parameters");
    hipStream_t s = (hipStream_t)stream;
    OptGrads g;
    for (int t = 0; t < OPT_MAX_TENSORS; ++t) g.g[t] = t < num_tensors ? grads[t] : nullptr;
    ProfScope prof(K_CLIP_ADAM, s);
    if (phases & 1)
        PHC_LAUNCH(opt_sumsq_kernel, dim3(num_chunks), dim3(256), 0, s, g, reinterpret_cast<const unsigned long long *>(table_dev),
                   chunks_dev, partial_dev + chunk_base, step_dev);
    // phase 2 (separate norm launch) is only needed when somebody wants the norm without an update; the update itself
    // derives the norm from the partials in every workgroup
    if ((phases & 2) && !(phases & 4))
        PHC_LAUNCH(opt_finalize_kernel, dim3(1), dim3(256), 0, s, partial_dev, total_chunks, max_norm, scal_dev);
    if (phases & 4)
        PHC_LAUNCH(opt_adam_kernel, dim3(num_chunks), dim3(256), 0, s, g, reinterpret_cast<const unsigned long long *>(table_dev),
                   chunks_dev, chunk_base == 0 ? scal_dev : nullptr, step_dev, partial_dev, total_chunks, max_norm, lr, beta1, beta2,
                   eps, write_back_clipped);
    return check_launch("clip_adam_step");
}

// resident capacity of the one-launch form on the current device (its workgroups wait for each other)
static int opt_fused_capacity() {
    static int cap[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (cap[dev] == 0) {
        int per_cu = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, opt_clip_adam_fused_kernel, 256, 0) != hipSuccess ||
            hipGetDeviceProperties(&prop, dev) != hipSuccess || getenv("HSA_CU_MASK") || getenv("ROC_GLOBAL_CU_MASK"))
            cap[dev] = -1;
        else
            // (a margin: MI355X_MICROARCH.md, the occupancy API can be one block per CU high; two per CU is all this launch asks for)
            cap[dev] = (per_cu > 2 ? 2 : (per_cu > 1 ? per_cu - 1 : 0)) * prop.multiProcessorCount;
        if (cap[dev] == 0) cap[dev] = -1;
    }
    return cap[dev] > 0 ? cap[dev] : 0;
}

size_t phc_clip_adam_fused_ws_bytes(int total_chunks) {
    return total_chunks < 1 ? 0 : sizeof(unsigned long long) * (size_t)total_chunks + 16;
}

int phc_clip_adam_fused_supported(int num_tensors, int total_chunks) {
    return num_tensors >= 1 && num_tensors <= OPT_MAX_TENSORS && total_chunks >= 1 && total_chunks <= opt_fused_capacity();
}

int phc_clip_adam_fused(const uint64_t *table_dev, const float *const *grads, int num_tensors, const int32_t *chunks_dev, int total_chunks,
                        float max_norm, float lr, float beta1, float beta2, float eps, float *step_dev, float *scal_dev, void *ws,
                        size_t ws_bytes, int write_back_clipped, void *stream) {
    PHC_REQUIRE(phc_clip_adam_fused_supported(num_tensors, total_chunks), PHC_ERR_UNSUPPORTED,
                "phc_clip_adam_fused: %d tensors / %d chunks do not fit one co-resident launch (capacity %d chunks)", num_tensors,
                total_chunks, opt_fused_capacity());
    PHC_REQUIRE(table_dev && grads && chunks_dev && step_dev && scal_dev, PHC_ERR_INVALID_ARG, "phc_clip_adam_fused: null pointer");
    PHC_REQUIRE(ws && ws_bytes >= phc_clip_adam_fused_ws_bytes(total_chunks) && (reinterpret_cast<uintptr_t>(ws) & 7) == 0, PHC_ERR_WORKSPACE,
                "phc_clip_adam_fused: workspace missing (phc_clip_adam_fused_ws_bytes, zero before the first use)");
    PHC_REQUIRE(lr > 0.f && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps > 0.f, PHC_ERR_INVALID_ARG,
                "phc_clip_adam_fused: bad hyper-parameters");
    hipStream_t s = (hipStream_t)stream;
    OptGrads g;
    for (int t = 0; t < OPT_MAX_TENSORS; ++t) g.g[t] = t < num_tensors ? grads[t] : nullptr;
    unsigned long long *gran = static_cast<unsigned long long *>(ws);
    unsigned *gen = reinterpret_cast<unsigned *>(gran + total_chunks);
    ProfScope prof(K_CLIP_ADAM, s);
    PHC_LAUNCH(opt_clip_adam_fused_kernel, dim3(total_chunks), dim3(256), 0, s, g, reinterpret_cast<const unsigned long long *>(table_dev),
               chunks_dev, scal_dev, step_dev, gran, gen, err_word(), total_chunks, max_norm, lr, beta1, beta2, eps, write_back_clipped);
    return check_launch("clip_adam_fused");
}

int phc_pack_grads(const uint64_t *table_dev, const float *const *grads, int num_tensors, const int32_t *chunks_dev, int num_chunks,
                   const float *moments_base, float *out_flat, float scale, void *stream) {
    PHC_REQUIRE(num_tensors >= 1 && num_tensors <= OPT_MAX_TENSORS, PHC_ERR_INVALID_ARG, "phc_pack_grads: %d tensors (max %d per call)",
                num_tensors, OPT_MAX_TENSORS);
    PHC_REQUIRE(table_dev && grads && chunks_dev && num_chunks >= 1 && moments_base && out_flat, PHC_ERR_INVALID_ARG,
                "phc_pack_grads: null pointer");
    OptGrads g;
    for (int t = 0; t < num_tensors; ++t) g.g[t] = grads[t];
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_CLIP_ADAM, s);
    PHC_LAUNCH(opt_pack_kernel, dim3(num_chunks), dim3(256), 0, s, g, reinterpret_cast<const unsigned long long *>(table_dev), chunks_dev,
               moments_base, out_flat, scale);
    return check_launch("opt_pack");
}

int phc_phm_weight_reg_max_jobs(void) { return REG_MAX_JOBS; }
int phc_phm_weight_reg_max_blocks(void) { return REG_MAX_BLOCKS; }
size_t phc_phm_weight_reg_ws_bytes(void) { return sizeof(unsigned long long) * REG_MAX_BLOCKS + 16; }

int phc_phm_weight_reg(const float *const *W, float *const *dW, const int32_t *n, const int32_t *cells, int num_jobs, float coef, int p,
                       float *reg_out, float *loss_inout, void *ws, size_t ws_bytes, void *stream) {
    PHC_REQUIRE(num_jobs >= 1 && num_jobs <= REG_MAX_JOBS && W && n && cells && reg_out, PHC_ERR_INVALID_ARG,
                "phc_phm_weight_reg: %d jobs (1..%d)", num_jobs, REG_MAX_JOBS);
    PHC_REQUIRE(p == 1 || p == 2, PHC_ERR_UNSUPPORTED, "phc_phm_weight_reg: p = %d (1 or 2)", p);
    PHC_REQUIRE(ws && ws_bytes >= phc_phm_weight_reg_ws_bytes() && (reinterpret_cast<uintptr_t>(ws) & 7) == 0, PHC_ERR_WORKSPACE,
                "phc_phm_weight_reg: workspace missing (phc_phm_weight_reg_ws_bytes, zero before the first use)");
    RegJobs jobs;
    jobs.num = num_jobs;
    int blocks = 0;
    for (int j = 0; j < num_jobs; ++j) {
        PHC_REQUIRE(W[j] && n[j] >= 1 && n[j] <= REG_MAX_N && cells[j] >= 1, PHC_ERR_INVALID_ARG, "phc_phm_weight_reg: job %d (n <= %d)", j, REG_MAX_N);
        jobs.W[j] = W[j];
        jobs.dW[j] = dW ? dW[j] : nullptr;
        jobs.n[j] = n[j];
        jobs.cells[j] = cells[j];
        jobs.first_block[j] = blocks;
        blocks += (cells[j] + 255) / 256;
    }
    jobs.first_block[num_jobs] = blocks;
    PHC_REQUIRE(blocks <= REG_MAX_BLOCKS, PHC_ERR_UNSUPPORTED, "phc_phm_weight_reg: %d workgroups (max %d)", blocks, REG_MAX_BLOCKS);
    hipStream_t s = (hipStream_t)stream;
    unsigned long long *gran = static_cast<unsigned long long *>(ws);
    unsigned *gen = reinterpret_cast<unsigned *>(gran + REG_MAX_BLOCKS);
    ProfScope prof(K_CLIP_ADAM, s);
    PHC_LAUNCH(phm_weight_reg_kernel, dim3(blocks), dim3(256), 0, s, jobs, coef, p, reg_out, loss_inout, gran, gen);
    return check_launch("phm_weight_reg");
}

}  // extern "C"
