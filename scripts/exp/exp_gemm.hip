// Kernel-level experiments for the PHM GEMM (build here with hipcc, run on the GPU box).
//   E1: raw fp32 MFMA rate (clock check)      E2: phm_gemm variants with phases removed
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "phc_hip.h"

using f32x16 = __attribute__((ext_vector_type(16))) float;

__global__ __launch_bounds__(256) void mfma_only(float *out, int iters) {
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.f;
    for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    if (s == 12345.678f) out[0] = s;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

static float time_it(hipStream_t s, int iters, auto fn) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) fn();
    CK(hipEventRecord(a, s));
    for (int i = 0; i < iters; ++i) fn();
    CK(hipEventRecord(b, s));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3f / iters;
}

int main(int argc, char **argv) {
    hipStream_t s; CK(hipStreamCreate(&s));
    float *dummy; CK(hipMalloc(&dummy, 1024));
    {   // E1
        int iters = 4096;
        for (int blocks : {256, 512, 1024}) {
            float us = time_it(s, 20, [&] { hipLaunchKernelGGL(mfma_only, dim3(blocks), dim3(256), 0, s, dummy, iters); });
            double flops = (double)blocks * 4 * iters * 32 * 32 * 2 * 2;
            printf("E1 mfma_only blocks=%d: %.1f us  %.1f TFLOP/s\n", blocks, us, flops / us / 1e6);
        }
    }
    // E2: library GEMM at several sizes
    for (long long N : {12874LL, 51496LL, 205984LL, 1250000LL}) {
        int n = 4, kin = 32, kout = 32, h = 128;
        float *x, *y, *A, *W, *b;
        CK(hipMalloc(&x, N * h * 4)); CK(hipMalloc(&y, N * h * 4)); CK(hipMalloc(&A, 64 * 4)); CK(hipMalloc(&W, n * kin * kout * 4)); CK(hipMalloc(&b, h * 4));
        std::vector<float> hx(N * h);
        for (auto &v : hx) v = (rand() % 2001 - 1000) * 1e-3f;
        CK(hipMemcpy(x, hx.data(), N * h * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(A, hx.data(), 64 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(W, hx.data(), n * kin * kout * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hx.data(), h * 4, hipMemcpyHostToDevice));
        float us = time_it(s, 30, [&] { phc_phm_linear_fwd(x, A, W, b, nullptr, N, n, kin, kout, y, s); });
        printf("E2 phm_linear_fwd N=%lld: %.1f us  %.1f TFLOP/s\n", N, us, 2.0 * N * h * h / us / 1e6);
        us = time_it(s, 30, [&] { phc_phm_linear_bwd(x, A, W, y, N, n, kin, kout, y, nullptr, nullptr, nullptr, nullptr, 0, s); });
        printf("E2 phm_linear_bwd(dx) N=%lld: %.1f us  %.1f TFLOP/s\n", N, us, 2.0 * N * h * h / us / 1e6);
        CK(hipFree(x)); CK(hipFree(y)); CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(b));
    }
    return 0;
}
