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

__global__ __launch_bounds__(RO_THREADS) void bce_masked_fwd_kernel(const float *__restrict__ logits, const float *__restrict__ y,
                                                                  long long total, float *__restrict__ partial,
                                                                  float *__restrict__ loss, float *__restrict__ stat) {
    __shared__ float red[2 * RO_THREADS];
    float s = 0.f, c = 0.f;
    for (long long e = (long long)blockIdx.x * RO_THREADS + threadIdx.x; e < total; e += (long long)gridDim.x * RO_THREADS) {
        const float yv = y[e];
        if (yv == yv) {                                       // labeled (not NaN)
            s += bce_entry(logits[e], yv);
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

// dlogits = dloss * (sigmoid(z) - y) / count on labeled entries, 0 elsewhere
__global__ __launch_bounds__(256) void bce_masked_bwd_kernel(const float *__restrict__ logits, const float *__restrict__ y,
                                                           const float *__restrict__ dloss, const float *__restrict__ stat,
                                                           long long total, float *__restrict__ dlogits) {
    const float scale = *dloss / stat[1];
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const float yv = y[e];
        float g = 0.f;
        if (yv == yv) {
            const float z = logits[e];
            g = (1.f / (1.f + expf(-z)) - yv) * scale;
        }
        dlogits[e] = g;
    }
}

// ---- gated per-graph sum --------------------------------------------------------------------------
// out[g, c*k + j] = sum_{r in segment g} x[r, c*k + j] * sigmoid(gate[r, j])      (rows grouped by graph: seg_ptr)
// one workgroup per graph; thread -> column (h <= 1024), rows walked in order: fixed summation order.
__global__ __launch_bounds__(256) void gated_pool_fwd_kernel(const float *__restrict__ x, const float *__restrict__ gate,
                                                           const int *__restrict__ seg_ptr, int h, int k,
                                                           float *__restrict__ out) {
    const int g = blockIdx.x;
    const int r0 = seg_ptr[g], r1 = seg_ptr[g + 1];
    for (int col = threadIdx.x; col < h; col += 256) {
        const int j = col % k;
        float acc = 0.f;
        int r = r0;
        for (; r + 4 <= r1; r += 4) {                        // four rows in flight
            float xv[4], gv[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                xv[t] = x[(size_t)(r + t) * h + col];
                gv[t] = gate[(size_t)(r + t) * k + j];
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) acc = fmaf(xv[t], 1.f / (1.f + expf(-gv[t])), acc);
        }
        for (; r < r1; ++r) acc = fmaf(x[(size_t)r * h + col], 1.f / (1.f + expf(-gate[(size_t)r * k + j])), acc);
        out[(size_t)g * h + col] = acc;
    }
}

// dx[r, c*k+j] = dout[g(r), c*k+j] * s_j ;  dgate[r, j] = s_j (1 - s_j) * sum_c dout[g(r), c*k+j] * x[r, c*k+j]
// one workgroup per graph, one wavefront per row (strided), lane -> gate column j.
__global__ __launch_bounds__(256) void gated_pool_bwd_kernel(const float *__restrict__ dout, const float *__restrict__ x,
                                                           const float *__restrict__ gate, const int *__restrict__ seg_ptr,
                                                           int h, int k, int n_comp, float *__restrict__ dx,
                                                           float *__restrict__ dgate) {
    const int g = blockIdx.x;
    const int r0 = seg_ptr[g], r1 = seg_ptr[g + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = r0 + wave; r < r1; r += 4) {
        for (int j = lane; j < k; j += 64) {
            const float s = 1.f / (1.f + expf(-gate[(size_t)r * k + j]));
            float dg = 0.f;
            for (int c = 0; c < n_comp; ++c) {
                const int col = c * k + j;
                const float d = dout[(size_t)g * h + col];
                dg = fmaf(d, x[(size_t)r * h + col], dg);
                if (dx) dx[(size_t)r * h + col] = d * s;
            }
            if (dgate) dgate[(size_t)r * k + j] = dg * s * (1.f - s);
        }
    }
}

}  // namespace phc

using namespace phc;

extern "C" {

size_t phc_bce_masked_ws_bytes(int64_t total) {
    if (total < 0) return 0;
    return sizeof(float) * 2 * 256;
}

int phc_bce_masked_fwd(const float *logits, const float *y, int64_t total, float *loss, float *stat, void *ws, size_t ws_bytes,
                       void *stream) {
    PHC_REQUIRE(total > 0 && logits && y && loss && stat, PHC_ERR_INVALID_ARG, "phc_bce_masked_fwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    int blocks = (int)((total + (1 << 16) - 1) >> 16);          // 64k entries per workgroup
    if (blocks > 256) blocks = 256;
    PHC_REQUIRE(blocks == 1 || (ws && ws_bytes >= phc_bce_masked_ws_bytes(total)), PHC_ERR_WORKSPACE,
                "phc_bce_masked_fwd: workspace too small");
    ProfScope prof(K_BCE_FWD, s);
    PHC_LAUNCH(bce_masked_fwd_kernel, dim3(blocks), dim3(RO_THREADS), 0, s, logits, y, (long long)total, static_cast<float *>(ws),
               loss, stat);
    if (blocks > 1)
        PHC_LAUNCH(bce_masked_finalize_kernel, dim3(1), dim3(RO_THREADS), 0, s, static_cast<const float *>(ws), blocks, loss, stat);
    return check_launch("bce_masked_fwd");
}

int phc_bce_masked_bwd(const float *logits, const float *y, const float *dloss, const float *stat, int64_t total, float *dlogits,
                       void *stream) {
    PHC_REQUIRE(total > 0 && logits && y && dloss && stat && dlogits, PHC_ERR_INVALID_ARG, "phc_bce_masked_bwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    long long blocks = (total + 1023) / 1024;
    if (blocks > 1024) blocks = 1024;
    ProfScope prof(K_BCE_BWD, s);
    PHC_LAUNCH(bce_masked_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, logits, y, dloss, stat, (long long)total, dlogits);
    return check_launch("bce_masked_bwd");
}

int phc_gated_pool_fwd(const float *x, const float *gate, const int32_t *seg_ptr, int64_t num_graphs, int h, int n_comp, float *out,
                       void *stream) {
    PHC_REQUIRE(num_graphs >= 0 && h >= 1 && n_comp >= 1 && h % n_comp == 0, PHC_ERR_INVALID_ARG, "phc_gated_pool_fwd: bad shape");
    if (num_graphs == 0) return PHC_OK;
    PHC_REQUIRE(x && gate && seg_ptr && out, PHC_ERR_INVALID_ARG, "phc_gated_pool_fwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_GATED_POOL_FWD, s);
    PHC_LAUNCH(gated_pool_fwd_kernel, dim3((unsigned)num_graphs), dim3(256), 0, s, x, gate, seg_ptr, h, h / n_comp, out);
    return check_launch("gated_pool_fwd");
}

int phc_gated_pool_bwd(const float *dout, const float *x, const float *gate, const int32_t *seg_ptr, int64_t num_graphs, int h,
                       int n_comp, float *dx, float *dgate, void *stream) {
    PHC_REQUIRE(num_graphs >= 0 && h >= 1 && n_comp >= 1 && h % n_comp == 0, PHC_ERR_INVALID_ARG, "phc_gated_pool_bwd: bad shape");
    if (num_graphs == 0 || (!dx && !dgate)) return PHC_OK;
    PHC_REQUIRE(dout && x && gate && seg_ptr, PHC_ERR_INVALID_ARG, "phc_gated_pool_bwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_GATED_POOL_BWD, s);
    PHC_LAUNCH(gated_pool_bwd_kernel, dim3((unsigned)num_graphs), dim3(256), 0, s, dout, x, gate, seg_ptr, h, h / n_comp, n_comp, dx,
               dgate);
    return check_launch("gated_pool_bwd");
}

}  // extern "C"
