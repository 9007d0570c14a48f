// PHM linear transform on gfx950:  y = x . H(A, W) + b,  H = sum_i A_i (x) W_i
//
// Replaces the reference's einsum-Kronecker + sum + matmul + bias-add chain
// (phc/hypercomplex/kronecker.py:35-48, layers.py:198-219) and its autograd.
//
// Forward / dX kernel (phm_gemm_kernel):
//   * the n*n*n contribution tensor A is staged in LDS once per workgroup;
//   * the [K x 64] panel of H (or H^T for dX) a workgroup needs is ASSEMBLED IN
//     LDS from W (n*kin*kout floats, L2 resident) — H never exists in HBM;
//   * x tiles [64 x K-chunk] are staged through LDS with 16-byte loads and fed
//     to v_mfma_f32_32x32x2_f32 (exact fp32, 4 waves = 2x2 tiles of 32x32).
//   When K <= 128 the H panel is built once and the workgroup streams M tiles
//   under it.
// Backward w.r.t. parameters:
//   dH = x^T . dy as fixed-order split-K slabs (phm_dH_partial_kernel), a slab
//   reduction, then the two small contractions dW = <A, dH>, dA = <W, dH>.
//   No atomics anywhere: results are bitwise reproducible.
#include "phc_common.h"

namespace phc {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int GEMM_BM = 64;        // rows of x per tile
constexpr int GEMM_BN = 64;        // output columns per workgroup
constexpr int GEMM_BK = 128;       // K chunk held in LDS
constexpr int XS_STRIDE = GEMM_BK + 4;   // floats; +4 keeps ds_read_b128 conflict-free (16-lane groups hit 16 distinct 4-bank slots)
constexpr int HS_STRIDE = GEMM_BN + 1;   // floats; odd stride: conflict-free for both fill orders and for the b32 operand reads
constexpr int GEMM_THREADS = 256;
constexpr int MAX_PHM_DIM = 16;

__host__ __device__ inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// Element (k, j) of the GEMM's right operand.
//   TRANS = 0: operand is H    : k = a*kin + kap (input feature), j = c*kout + pi (output feature)
//   TRANS = 1: operand is H^T  : k = c*kout + pi,                 j = a*kin + kap
template <int TRANS>
__device__ __forceinline__ float h_element(const float *__restrict__ As, const float *__restrict__ W, int n, int kin,
                                           int kout, int k, int j) {
    int a, kap, c, pi;
    if (TRANS == 0) {
        a = k / kin; kap = k - a * kin;
        c = j / kout; pi = j - c * kout;
    } else {
        c = k / kout; pi = k - c * kout;
        a = j / kin; kap = j - a * kin;
    }
    const float *w = W + (size_t)kap * kout + pi;
    const float *arow = As + a * n + c;
    const int wstride = kin * kout;
    const int astride = n * n;
    float h = 0.f;
#pragma unroll 4
    for (int i = 0; i < n; ++i) h = fmaf(arow[i * astride], w[(size_t)i * wstride], h);
    return h;
}

// out[M, Nout] = X[M, K] . Op(K, Nout) (+ bias[Nout]) (+ residual[M, Nout])
// grid.x = number of 64-wide output panels, grid.y = workgroups striding over M tiles.
template <int TRANS>
__global__ __launch_bounds__(GEMM_THREADS) void phm_gemm_kernel(
    const float *__restrict__ X, const float *__restrict__ A, const float *__restrict__ W,
    const float *__restrict__ bias, const float *__restrict__ residual, float *__restrict__ out,
    long long M, int K, int Nout, int n, int kin, int kout, int vec4) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *xs = smem;                                   // [GEMM_BM][XS_STRIDE]
    float *hs = xs + GEMM_BM * XS_STRIDE;               // [GEMM_BK][HS_STRIDE]
    float *As = hs + GEMM_BK * HS_STRIDE;               // [n*n*n]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;            // 2 x 2 waves of 32 x 32
    const int r = lane & 31, hh = lane >> 5;
    const int j0 = blockIdx.x * GEMM_BN;

    for (int i = tid; i < n * n * n; i += GEMM_THREADS) As[i] = A[i];

    const int num_chunks = (K + GEMM_BK - 1) / GEMM_BK;
    const long long num_mtiles = (M + GEMM_BM - 1) / GEMM_BM;
    bool h_ready = false;                               // panel resident across M tiles when K fits one chunk

    for (long long mt = blockIdx.y; mt < num_mtiles; mt += gridDim.y) {
        const long long m0 = mt * GEMM_BM;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;

        for (int ch = 0; ch < num_chunks; ++ch) {
            const int k0 = ch * GEMM_BK;
            const int kc = min(GEMM_BK, K - k0);        // valid k in this chunk
            const int kc8 = round_up(kc, 8);            // MFMA consumes k in groups of 8
            __syncthreads();                            // previous chunk's readers are done (also orders the As fill)

            // ---- assemble the H panel chunk in LDS -----------------------------------
            if (!h_ready) {
                if (TRANS == 0) {                       // lanes run along j: W reads are contiguous in pi
                    const int jj = tid & (GEMM_BN - 1);
                    const int j = j0 + jj;
                    for (int kk = tid / GEMM_BN; kk < kc8; kk += GEMM_THREADS / GEMM_BN) {
                        float v = 0.f;
                        if (kk < kc && j < Nout) v = h_element<0>(As, W, n, kin, kout, k0 + kk, j);
                        hs[kk * HS_STRIDE + jj] = v;
                    }
                } else {                                // lanes run along k: W reads are contiguous in pi
                    for (int e = tid; e < kc8 * GEMM_BN; e += GEMM_THREADS) {
                        const int jj = e / kc8;
                        const int kk = e - jj * kc8;
                        const int j = j0 + jj;
                        float v = 0.f;
                        if (kk < kc && j < Nout) v = h_element<1>(As, W, n, kin, kout, k0 + kk, j);
                        hs[kk * HS_STRIDE + jj] = v;
                    }
                }
                if (num_chunks == 1) h_ready = true;
            }

            // ---- stage the x tile chunk ----------------------------------------------
            if (vec4) {
                const int c4n = kc8 >> 2;               // float4 per row (kc is a multiple of 4 here)
                for (int e = tid; e < GEMM_BM * c4n; e += GEMM_THREADS) {
                    const int row = e / c4n;
                    const int c4 = e - row * c4n;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (m0 + row < M && c4 * 4 < kc)
                        v = *reinterpret_cast<const float4 *>(X + (size_t)(m0 + row) * K + k0 + c4 * 4);
                    *reinterpret_cast<float4 *>(xs + row * XS_STRIDE + c4 * 4) = v;
                }
            } else {
                for (int e = tid; e < GEMM_BM * kc8; e += GEMM_THREADS) {
                    const int row = e / kc8;
                    const int kk = e - row * kc8;
                    float v = 0.f;
                    if (m0 + row < M && kk < kc) v = X[(size_t)(m0 + row) * K + k0 + kk];
                    xs[row * XS_STRIDE + kk] = v;
                }
            }
            __syncthreads();

            // ---- MFMA over the chunk ----------------------------------------------------
            // lane (r, hh) supplies A[row r][k] and B[k][col r] for k = kg*8 + 4*hh + q, q = 0..3
            const float *xrow = xs + (wm * 32 + r) * XS_STRIDE + 4 * hh;
            const float *hcol = hs + (4 * hh) * HS_STRIDE + wn * 32 + r;
            for (int kg = 0; kg < kc8; kg += 8) {
                const float4 av = *reinterpret_cast<const float4 *>(xrow + kg);
                const float b0 = hcol[(kg + 0) * HS_STRIDE];
                const float b1 = hcol[(kg + 1) * HS_STRIDE];
                const float b2 = hcol[(kg + 2) * HS_STRIDE];
                const float b3 = hcol[(kg + 3) * HS_STRIDE];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b2, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b3, acc, 0, 0, 0);
            }
        }

        // ---- epilogue: bias, residual, store ---------------------------------------------
        const int col = j0 + wn * 32 + r;
        if (col < Nout) {
            const float bv = bias ? bias[col] : 0.f;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const long long row = m0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
                if (row < M) {
                    float v = acc[reg] + bv;
                    if (residual) v += residual[(size_t)row * Nout + col];
                    out[(size_t)row * Nout + col] = v;
                }
            }
        }
    }
}

// H[k, j] for the whole matrix (exported helper; the hot kernels do not use it).
__global__ void phm_build_H_kernel(const float *__restrict__ A, const float *__restrict__ W, float *__restrict__ H, int n,
                                   int kin, int kout) {
    extern __shared__ float As[];
    for (int i = threadIdx.x; i < n * n * n; i += blockDim.x) As[i] = A[i];
    __syncthreads();
    const int in = n * kin, out = n * kout;
    const long long total = (long long)in * out;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(e / out), j = (int)(e - (long long)k * out);
        H[e] = h_element<0>(As, W, n, kin, kout, k, j);
    }
}

// ------------------------------------------------------------------------------------------
// dH = x^T . dy  as split-K slabs.  Workgroup (tile, split) accumulates the 64 x 64 tile
// (i0.., j0..) of dH over its row range and writes it to slab[split].  Workgroups with
// tile_i == 0 also produce the column sums of dy (db) for their rows.
// ------------------------------------------------------------------------------------------
constexpr int DH_T = 64;            // tile edge
constexpr int DH_RB = 64;           // rows staged per step
constexpr int DH_STRIDE = DH_T;     // lanes read consecutive columns: conflict-free without padding

__global__ __launch_bounds__(256) void phm_dH_partial_kernel(const float *__restrict__ X, const float *__restrict__ DY,
                                                            float *__restrict__ slabs, long long N, int in, int out,
                                                            int tiles_j, long long rows_per_split, int vec4) {
    __shared__ __attribute__((aligned(16))) float xs[DH_RB * DH_STRIDE];
    __shared__ __attribute__((aligned(16))) float ds[DH_RB * DH_STRIDE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int r = lane & 31, hh = lane >> 5;
    const int ti = blockIdx.x / tiles_j, tj = blockIdx.x - ti * tiles_j;
    const int i0 = ti * DH_T, j0 = tj * DH_T;
    const long long split = blockIdx.y;
    const long long r_begin = split * rows_per_split;
    const long long r_end = min(N, r_begin + rows_per_split);
    const size_t slab_elems = (size_t)in * out + out;
    float *slab = slabs + (size_t)split * slab_elems;

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float colsum = 0.f;                                  // threads 0..63 of tile_i == 0 workgroups

    for (long long rb = r_begin; rb < r_end; rb += DH_RB) {
        __syncthreads();
        if (vec4) {
            for (int e = tid; e < DH_RB * (DH_T / 4); e += 256) {
                const int row = e / (DH_T / 4), c4 = e - row * (DH_T / 4);
                float4 xv = make_float4(0.f, 0.f, 0.f, 0.f), dv = xv;
                if (rb + row < r_end) {
                    if (i0 + c4 * 4 < in) xv = *reinterpret_cast<const float4 *>(X + (size_t)(rb + row) * in + i0 + c4 * 4);
                    if (j0 + c4 * 4 < out) dv = *reinterpret_cast<const float4 *>(DY + (size_t)(rb + row) * out + j0 + c4 * 4);
                }
                *reinterpret_cast<float4 *>(xs + row * DH_STRIDE + c4 * 4) = xv;
                *reinterpret_cast<float4 *>(ds + row * DH_STRIDE + c4 * 4) = dv;
            }
        } else {
            for (int e = tid; e < DH_RB * DH_T; e += 256) {
                const int row = e / DH_T, c = e - row * DH_T;
                float xv = 0.f, dv = 0.f;
                if (rb + row < r_end) {
                    if (i0 + c < in) xv = X[(size_t)(rb + row) * in + i0 + c];
                    if (j0 + c < out) dv = DY[(size_t)(rb + row) * out + j0 + c];
                }
                xs[row * DH_STRIDE + c] = xv;
                ds[row * DH_STRIDE + c] = dv;
            }
        }
        __syncthreads();
        // A operand = x^T: lane (r, hh) supplies x[row rr+hh][i0 + wi*32 + r]; B = dy[row rr+hh][j0 + wj*32 + r]
        const float *xa = xs + hh * DH_STRIDE + wi * 32 + r;
        const float *db_ = ds + hh * DH_STRIDE + wj * 32 + r;
#pragma unroll 8
        for (int rr = 0; rr < DH_RB; rr += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[rr * DH_STRIDE], db_[rr * DH_STRIDE], acc, 0, 0, 0);
        if (ti == 0 && tid < DH_T) {
#pragma unroll 8
            for (int rr = 0; rr < DH_RB; ++rr) colsum += ds[rr * DH_STRIDE + tid];
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
    if (ti == 0 && tid < DH_T && j0 + tid < out) slab[(size_t)in * out + j0 + tid] = colsum;
}

// dH[idx] (and the db tail) = sum over splits, in split order.
__global__ void phm_dH_reduce_kernel(const float *__restrict__ slabs, float *__restrict__ dH, size_t slab_elems, int splits) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= slab_elems) return;
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += slabs[(size_t)k * slab_elems + idx];
    dH[idx] = s;
}

// blocks [0, n^3): dA[i,a,c] = sum_{kap,pi} W[i,kap,pi] * dH[a*kin+kap, c*kout+pi]
// blocks [n^3, ..): dW[i,kap,pi] = sum_{a,c} A[i,a,c] * dH[a*kin+kap, c*kout+pi]; the last block also copies db.
__global__ __launch_bounds__(256) void phm_contract_kernel(const float *__restrict__ A, const float *__restrict__ W,
                                                          const float *__restrict__ dH, float *__restrict__ dW,
                                                          float *__restrict__ dA, float *__restrict__ db, int n, int kin,
                                                          int kout, int dA_blocks) {
    const int out = n * kout;
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < dA_blocks) {
        if (!dA) return;
        const int iac = blockIdx.x;
        const int i = iac / (n * n), a = (iac / n) % n, c = iac % n;
        float s = 0.f;
        for (int e = tid; e < kin * kout; e += 256) {
            const int kap = e / kout, pi = e - kap * kout;
            s = fmaf(W[(size_t)i * kin * kout + e], dH[(size_t)(a * kin + kap) * out + c * kout + pi], s);
        }
        __shared__ float red[256];
        red[tid] = s;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {           // fixed tree: reproducible
            if (tid < st) red[tid] += red[tid + st];
            __syncthreads();
        }
        if (tid == 0) dA[iac] = red[0];
        return;
    }
    const int blk = blockIdx.x - dA_blocks;
    const int e = blk * 256 + tid;                       // index into dW [n, kin, kout]
    if (dW && e < n * kin * kout) {
        const int i = e / (kin * kout);
        const int rem = e - i * kin * kout;
        const int kap = rem / kout, pi = rem - kap * kout;
        float s = 0.f;
        for (int a = 0; a < n; ++a)
            for (int c = 0; c < n; ++c)
                s = fmaf(A[(i * n + a) * n + c], dH[(size_t)(a * kin + kap) * out + c * kout + pi], s);
        dW[e] = s;
    }
    if (db && blk == 0)
        for (int j = tid; j < out; j += 256) db[j] = dH[(size_t)n * kin * out + j];
}

// ---- host side ---------------------------------------------------------------------------
static int g_num_cu = 0;
static int num_cu() {
    if (g_num_cu == 0) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) g_num_cu = p.multiProcessorCount;
        if (g_num_cu <= 0) g_num_cu = 256;
    }
    return g_num_cu;
}

static size_t gemm_lds_bytes(int n) {
    return sizeof(float) * ((size_t)GEMM_BM * XS_STRIDE + (size_t)GEMM_BK * HS_STRIDE + (size_t)n * n * n);
}

static int check_phm_dims(const char *who, int64_t N, int n, int kin, int kout) {
    PHC_REQUIRE(N >= 0, PHC_ERR_INVALID_ARG, "%s: N = %lld < 0", who, (long long)N);
    PHC_REQUIRE(n >= 1 && kin >= 1 && kout >= 1, PHC_ERR_INVALID_ARG, "%s: n, kin, kout must be >= 1 (got %d, %d, %d)", who,
                n, kin, kout);
    PHC_REQUIRE(n <= MAX_PHM_DIM, PHC_ERR_UNSUPPORTED, "%s: phm_dim %d > %d", who, n, MAX_PHM_DIM);
    PHC_REQUIRE((int64_t)n * kin < (1 << 20) && (int64_t)n * kout < (1 << 20), PHC_ERR_UNSUPPORTED,
                "%s: feature width too large", who);
    return PHC_OK;
}

template <int TRANS>
static int launch_gemm(const float *X, const float *A, const float *W, const float *bias, const float *residual,
                       float *out, int64_t M, int K, int Nout, int n, int kin, int kout, hipStream_t s) {
    if (M == 0) return PHC_OK;
    const int panels = (Nout + GEMM_BN - 1) / GEMM_BN;
    const int64_t mtiles = (M + GEMM_BM - 1) / GEMM_BM;
    // enough workgroups to fill the chip twice over; the rest is strided inside the kernel
    int64_t gy = (2 * (int64_t)num_cu() + panels - 1) / panels;
    if (gy > mtiles) gy = mtiles;
    if (gy < 1) gy = 1;
    const int vec4 = (K % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
    const size_t lds = gemm_lds_bytes(n);
    static bool attr_set[2] = {false, false};
    if (!attr_set[TRANS]) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(&phm_gemm_kernel<TRANS>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        attr_set[TRANS] = true;
    }
    ProfScope prof(TRANS ? K_PHM_GEMM_DX : K_PHM_GEMM_FWD, s);
    hipLaunchKernelGGL(phm_gemm_kernel<TRANS>, dim3(panels, (unsigned)gy), dim3(GEMM_THREADS), lds, s, X, A, W, bias,
                       residual, out, (long long)M, K, Nout, n, kin, kout, vec4);
    return check_launch(TRANS ? "phm_gemm_dx" : "phm_gemm_fwd");
}

struct DhPlan {
    int tiles_i, tiles_j, splits;
    int64_t rows_per_split;
    size_t slab_elems;
};

static DhPlan dh_plan(int64_t N, int in, int out) {
    DhPlan p;
    p.tiles_i = (in + DH_T - 1) / DH_T;
    p.tiles_j = (out + DH_T - 1) / DH_T;
    const int tiles = p.tiles_i * p.tiles_j;
    int64_t splits = (2 * 256 + tiles - 1) / tiles;             // ~2 workgroups per CU (fixed: plan must not depend on the device)
    const int64_t max_splits = (N + DH_RB - 1) / DH_RB;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    int64_t rps = (N + splits - 1) / splits;
    rps = (rps + DH_RB - 1) / DH_RB * DH_RB;
    if (rps < DH_RB) rps = DH_RB;
    splits = N > 0 ? (N + rps - 1) / rps : 1;
    p.splits = (int)splits;
    p.rows_per_split = rps;
    p.slab_elems = (size_t)in * out + out;
    return p;
}

}  // namespace phc

using namespace phc;

extern "C" {

int phc_phm_build_H(const float *A, const float *W, int n, int kin, int kout, float *H, void *stream) {
    int rc = check_phm_dims("phc_phm_build_H", 0, n, kin, kout);
    if (rc) return rc;
    PHC_REQUIRE(A && W && H, PHC_ERR_INVALID_ARG, "phc_phm_build_H: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const long long total = (long long)n * kin * n * kout;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    ProfScope prof(K_PHM_BUILD_H, s);
    hipLaunchKernelGGL(phm_build_H_kernel, dim3(blocks), dim3(256), sizeof(float) * n * n * n, s, A, W, H, n, kin, kout);
    return check_launch("phm_build_H");
}

int phc_phm_linear_fwd(const float *x, const float *A, const float *W, const float *b, const float *residual, int64_t N,
                       int n, int kin, int kout, float *y, void *stream) {
    int rc = check_phm_dims("phc_phm_linear_fwd", N, n, kin, kout);
    if (rc) return rc;
    PHC_REQUIRE(A && W && (N == 0 || (x && y)), PHC_ERR_INVALID_ARG, "phc_phm_linear_fwd: null pointer");
    PHC_REQUIRE(x != y || N == 0, PHC_ERR_INVALID_ARG, "phc_phm_linear_fwd: y must not alias x");
    return launch_gemm<0>(x, A, W, b, residual, y, N, n * kin, n * kout, n, kin, kout, (hipStream_t)stream);
}

size_t phc_phm_linear_bwd_ws_bytes(int64_t N, int n, int kin, int kout) {
    if (N < 0 || n < 1 || kin < 1 || kout < 1) return 0;
    DhPlan p = dh_plan(N, n * kin, n * kout);
    return sizeof(float) * p.slab_elems * ((size_t)p.splits + 1);
}

int phc_phm_linear_bwd(const float *x, const float *A, const float *W, const float *dy, int64_t N, int n, int kin, int kout,
                       float *dx, float *dW, float *dA, float *db, void *ws, size_t ws_bytes, void *stream) {
    int rc = check_phm_dims("phc_phm_linear_bwd", N, n, kin, kout);
    if (rc) return rc;
    PHC_REQUIRE(A && W && (N == 0 || (x && dy)), PHC_ERR_INVALID_ARG, "phc_phm_linear_bwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int in = n * kin, out = n * kout;
    if (dx) {
        rc = launch_gemm<1>(dy, A, W, nullptr, nullptr, dx, N, out, in, n, kin, kout, s);
        if (rc) return rc;
    }
    if (!dW && !dA && !db) return PHC_OK;
    const size_t need = phc_phm_linear_bwd_ws_bytes(N, n, kin, kout);
    PHC_REQUIRE(ws && ws_bytes >= need, PHC_ERR_WORKSPACE, "phc_phm_linear_bwd: workspace %zu B < required %zu B", ws_bytes,
                need);
    PHC_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 15) == 0, PHC_ERR_WORKSPACE, "phc_phm_linear_bwd: workspace not 16-byte aligned");
    DhPlan p = dh_plan(N, in, out);
    float *slabs = static_cast<float *>(ws);
    float *dH = slabs + p.slab_elems * (size_t)p.splits;
    if (N == 0) {
        hipMemsetAsync(dH, 0, sizeof(float) * p.slab_elems, s);
    } else {
        const int vec4 = (in % 4 == 0) && (out % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) &&
                         ((reinterpret_cast<uintptr_t>(dy) & 15) == 0);
        {
            ProfScope prof(K_PHM_DH_PARTIAL, s);
            hipLaunchKernelGGL(phm_dH_partial_kernel, dim3(p.tiles_i * p.tiles_j, p.splits), dim3(256), 0, s, x, dy, slabs,
                               (long long)N, in, out, p.tiles_j, (long long)p.rows_per_split, vec4);
            rc = check_launch("phm_dH_partial");
            if (rc) return rc;
        }
        {
            ProfScope prof(K_PHM_DH_REDUCE, s);
            hipLaunchKernelGGL(phm_dH_reduce_kernel, dim3((unsigned)((p.slab_elems + 255) / 256)), dim3(256), 0, s, slabs,
                               dH, p.slab_elems, p.splits);
            rc = check_launch("phm_dH_reduce");
            if (rc) return rc;
        }
    }
    const int dA_blocks = n * n * n;
    const int dW_blocks = (n * kin * kout + 255) / 256;
    ProfScope prof(K_PHM_CONTRACT, s);
    hipLaunchKernelGGL(phm_contract_kernel, dim3(dA_blocks + dW_blocks), dim3(256), 0, s, A, W, dH, dW, dA, db, n, kin, kout,
                       dA_blocks);
    return check_launch("phm_contract");
}

}  // extern "C"
