// Probe: what does it cost a 432-workgroup launch to publish its per-column statistics with 64-bit integer atomics into ONE
// [4 x 128] accumulator (no finalize launch afterwards), compared with plain stores of per-workgroup partials?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/atomic_probe scripts/dbg/atomic_stats_probe.hip && /tmp/atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// MODE 0: plain partial stores; 1: int64 atomics (4 per column); 2: fp64 atomics (2 per column); 3: nothing
template <int MODE, int REP = 1>
__global__ __launch_bounds__(256) void probe(const float *__restrict__ x, float *__restrict__ y, float *__restrict__ partial,
                                             unsigned long long *__restrict__ acc, double *__restrict__ accd, int spin) {
    const int panel = blockIdx.x & 3, rg = blockIdx.x >> 2;
    const int tid = threadIdx.x;
    // some streaming work so that workgroups finish staggered like the GEMM's (read 16 KB, write 16 KB)
    const size_t base = ((size_t)rg * 128) * 128 + panel * 32;
    float s = 0.f;
    for (int i = 0; i < 16; ++i) {
        const int r = i * 8 + (tid >> 5), c = tid & 31;
        float v = x[base + (size_t)r * 128 + c];
        for (int k = 0; k < spin; ++k) v = fmaf(v, 1.0001f, 0.5f);
        y[base + (size_t)r * 128 + c] = v;
        s += v;
    }
    __shared__ float red[256];
    red[tid] = s;
    __syncthreads();
    if (tid < 32) {
        float a = 0.f;
        for (int l = 0; l < 8; ++l) a += red[l * 32 + tid];
        const int col = panel * 32 + tid;
        if (MODE == 0) {
            partial[(size_t)rg * 256 + col] = a;
            partial[(size_t)rg * 256 + 128 + col] = a * a;
        } else if (MODE == 1) {
            acc += (size_t)(blockIdx.x % REP) * 512;
            const double d = a, q = (double)a * a;
            const long long h1 = __double2ll_rn(d * 4096.0), h2 = __double2ll_rn(q * 4096.0);
            const long long l1 = __double2ll_rn((d - (double)h1 * (1.0 / 4096.0)) * 4611686018427387904.0);
            const long long l2 = __double2ll_rn((q - (double)h2 * (1.0 / 4096.0)) * 4611686018427387904.0);
            atomicAdd(acc + col, (unsigned long long)h1);
            atomicAdd(acc + 128 + col, (unsigned long long)l1);
            atomicAdd(acc + 256 + col, (unsigned long long)h2);
            atomicAdd(acc + 384 + col, (unsigned long long)l2);
        } else if (MODE == 2) {
            unsafeAtomicAdd(accd + col, (double)a);
            unsafeAtomicAdd(accd + 128 + col, (double)a * a);
        }
    }
}

__global__ void consume(const float *__restrict__ y, float *__restrict__ z, const unsigned long long *acc) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    z[i] = y[i] + (float)(long long)acc[threadIdx.x & 127];
}

template <int MODE, int REP = 1>
int run(const char *name, float *x, float *y, float *z, float *partial, unsigned long long *acc, double *accd, int spin) {
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    // graph of 40 x (probe -> consume) so that launches are back to back like the replayed step
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 40; ++i) {
        probe<MODE, REP><<<432, 256, 0, st>>>(x, y, partial, acc, accd, spin);
        consume<<<6912, 256, 0, st>>>(y, z, acc);
    }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    std::vector<float> ms;
    for (int rep = 0; rep < 12; ++rep) {
        CK(hipEventRecord(e0, st));
        CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    printf("%-28s spin=%3d  %7.2f us per (probe+consume) pair (median of 12 x 40)\n", name, spin, 1e3 * ms[6] / 40);
    return 0;
}

int main() {
    float *x, *y, *z, *partial;
    unsigned long long *acc;
    double *accd;
    const size_t n = (size_t)108 * 128 * 128;
    CK(hipMalloc(&x, n * 4));
    CK(hipMalloc(&y, n * 4));
    CK(hipMalloc(&z, n * 4));
    CK(hipMalloc(&partial, 108 * 256 * 4));
    CK(hipMalloc(&acc, 32 * 512 * 8));
    CK(hipMalloc(&accd, 256 * 8));
    CK(hipMemset(x, 0, n * 4));
    CK(hipMemset(acc, 0, 32 * 512 * 8));
    CK(hipMemset(accd, 0, 256 * 8));
    for (int spin : {0, 64}) {
        if (run<3>("no statistics", x, y, z, partial, acc, accd, spin)) return 1;
        if (run<0>("partial stores", x, y, z, partial, acc, accd, spin)) return 1;
        if (run<1>("4 x int64 atomics / column", x, y, z, partial, acc, accd, spin)) return 1;
        if (run<1, 8>("4 x int64 atomics, 8 replicas", x, y, z, partial, acc, accd, spin)) return 1;
        if (run<1, 32>("4 x int64 atomics, 32 replicas", x, y, z, partial, acc, accd, spin)) return 1;
        if (run<2>("2 x fp64 atomics / column", x, y, z, partial, acc, accd, spin)) return 1;
    }
    return 0;
}
