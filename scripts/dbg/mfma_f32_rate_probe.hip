// Probe: issue rate of v_mfma_f32_32x32x2_f32 from ONE wavefront per SIMD with NACC independent accumulators
// (the register-resident dH kernel holds 16 = 256 accumulator registers).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_probe scripts/dbg/mfma_f32_rate_probe.hip && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NACC>
__global__ __launch_bounds__(256) void probe(float *out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 64 / NACC; ++rep)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
int run(float *out, int wgs) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 64;                  // 64 x 64 MFMAs per wavefront
    probe<NACC><<<wgs, 256>>>(out, iters, 1.f, 2.f);
    CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0));
        probe<NACC><<<wgs, 256>>>(out, iters, 1.f, 2.f);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    const double mf = 64.0 * iters;
    printf("NACC=%2d wgs=%4d: %8.2f us for %d MFMAs per wavefront -> %6.1f ns per MFMA, %6.1f TFLOP/s\n", NACC, wgs, best * 1e3, (int)mf,
           best * 1e6 / mf, wgs * 4.0 * mf * 4096.0 / (best * 1e-3) / 1e12);
    return 0;
}

int main() {
    float *out;
    CK(hipMalloc(&out, 1024 * 256 * 4));
    for (int wgs : {256, 512}) {
        if (run<1>(out, wgs)) return 1;
        if (run<4>(out, wgs)) return 1;
        if (run<8>(out, wgs)) return 1;
        if (run<16>(out, wgs)) return 1;
    }
    return 0;
}
