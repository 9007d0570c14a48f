// Probe for the register-stationary streaming GEMM (csrc/gemm_rs_device.h) against the library's LDS-staged one.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Iphc_gnn_amd/csrc -Iscripts/exp scripts/exp/gemm_rs_probe.hip \
//         -Lphc_gnn_amd -lphc_hip -Wl,-rpath,/root/repo/phc_gnn_amd -o scripts/exp/variants/gemm_rs_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#include "gemm_rs_device.h"
#include "phc_hip.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <typename F>
static float time_us(hipStream_t s, int iters, F fn) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 5; ++i) fn();
    CK(hipEventRecord(a, s));
    for (int i = 0; i < iters; ++i) fn();
    CK(hipEventRecord(b, s));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3f / iters;
}

static unsigned long long *g_stamps = nullptr;      // device buffer; when set, the next launches stamp into it
static int g_last_waves = 0, g_last_blocks = 0;

template <int K, int BN, int WAVES, int MODE = 0>
static void launch_rs(const float *x, const float *hk4, const float *bias, float *y, long long M, int Nout, int groups_override, hipStream_t s) {
    using namespace phc;
    GemmRsArgs a;
    a.stamps = g_stamps;
    a.x = x; a.hk4 = hk4; a.bias = bias; a.residual = nullptr; a.y = y; a.M = M; a.Nout = Nout;
    a.panels = (Nout + BN - 1) / BN;
    const long long slabs = (M + 15) / 16;
    int groups = groups_override > 0 ? groups_override : 256 / a.panels;
    const long long max_groups = (slabs + 3) / 4;                   // one slab per SIMD at least
    if (groups > max_groups) groups = (int)max_groups;
    a.groups = groups;
    const size_t lds = (size_t)K * BN * 4;
    static bool attr = false;
    if (!attr) { CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_rs16_kernel<K, BN, WAVES, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr = true; }
    g_last_waves = WAVES; g_last_blocks = a.panels * a.groups;
    hipLaunchKernelGGL((gemm_rs16_kernel<K, BN, WAVES, MODE>), dim3(a.panels * a.groups), dim3(64 * WAVES), lds, s, a);
}

template <int K, int TPW, int WAVES, int RB, int R>
static void launch_hs(const float *x, const float *hk4, const float *bias, float *y, long long M, int Nout, int groups_override, hipStream_t s) {
    using namespace phc;
    GemmHsArgs a;
    a.stamps = g_stamps;
    a.x = x; a.hk4 = hk4; a.bias = bias; a.residual = nullptr; a.y = y; a.M = M; a.Nout = Nout;
    constexpr int PW = 16 * TPW * WAVES;
    a.panels = (Nout + PW - 1) / PW;
    const long long steps = (M + 16 * RB - 1) / (16 * RB);
    int groups = groups_override > 0 ? groups_override : 256 / a.panels;
    if (groups > steps) groups = (int)steps;
    a.groups = groups;
    constexpr int RPI = 256 / K;
    const size_t lds = (size_t)R * (16 * RB / RPI) * 66 * 16;
    static bool attr = false;
    if (!attr) { CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_hs16_kernel<K, TPW, WAVES, RB, R>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr = true; }
    g_last_waves = WAVES; g_last_blocks = a.panels * a.groups;
    hipLaunchKernelGGL((gemm_hs16_kernel<K, TPW, WAVES, RB, R>), dim3(a.panels * a.groups), dim3(64 * WAVES), lds, s, a);
}

// one stamped launch (after the timed ones): in-kernel clock and phase times
template <typename F>
static void stamped(hipStream_t s, F fn) {
    const size_t cap = 4096 * 16 * 6;
    CK(hipMalloc(&g_stamps, cap * 8));
    CK(hipMemsetAsync(g_stamps, 0, cap * 8, s));
    fn();
    CK(hipStreamSynchronize(s));
    std::vector<unsigned long long> h(cap);
    CK(hipMemcpy(h.data(), g_stamps, cap * 8, hipMemcpyDeviceToHost));
    CK(hipFree(g_stamps));
    g_stamps = nullptr;
    const int nw = g_last_waves * g_last_blocks;
    std::vector<double> clk, fill, loop;
    unsigned long long r_first = ~0ull, r_last = 0, r_loop_first = ~0ull;
    for (int w = 0; w < nw; ++w) {
        const unsigned long long *o = h.data() + (size_t)w * 6;
        if (o[3] == 0) continue;
        r_first = std::min(r_first, o[3]); r_last = std::max(r_last, o[5]); r_loop_first = std::min(r_loop_first, o[4]);
        fill.push_back((o[4] - o[3]) * 0.01);
        loop.push_back((o[5] - o[4]) * 0.01);
        if (o[5] > o[4] + 50) clk.push_back((double)(o[2] - o[1]) / (double)(o[5] - o[4]) * 0.1);
    }
    auto med = [](std::vector<double> &v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    auto mx = [](std::vector<double> &v) { double m = 0; for (double x : v) m = std::max(m, x); return m; };
    printf("      stamped: waves %zu  in-loop clock median %.2f GHz | entry->loop (fill) median %.2f us max %.2f | loop median %.2f us max %.2f | first entry -> last end %.2f us\n",
           fill.size(), med(clk), med(fill), mx(fill), med(loop), mx(loop), (r_last - r_first) * 0.01);
}

static double check(const std::vector<float> &hx, const std::vector<float> &hH, const std::vector<float> &hb, const float *dy, long long M, int K, int N) {
    std::vector<float> y((size_t)M * N);
    CK(hipMemcpy(y.data(), dy, (size_t)M * N * 4, hipMemcpyDeviceToHost));
    double worst = 0, scale = 0;
    const long long rows[] = {0, 1, 15, 16, 17, 31, 32, 1000, 4095, 4096, M / 2, M - 33, M - 17, M - 16, M - 2, M - 1};
    for (long long r : rows) {
        if (r < 0 || r >= M) continue;
        for (int j = 0; j < N; ++j) {
            double acc = hb[j];
            for (int k = 0; k < K; ++k) acc += (double)hx[(size_t)r * K + k] * hH[(size_t)k * N + j];
            worst = fmax(worst, fabs(acc - y[(size_t)r * N + j]));
            scale = fmax(scale, fabs(acc));
        }
    }
    return worst / scale;
}

template <int K>
static void run_case(long long M, int N, hipStream_t s) {
    std::vector<float> hx((size_t)M * K), hH((size_t)K * N), hk4((size_t)K * N), hb(N);
    for (auto &v : hx) v = (rand() % 2001 - 1000) * 1e-3f;
    for (auto &v : hH) v = (rand() % 2001 - 1000) * 1e-3f / sqrtf((float)K);
    for (auto &v : hb) v = (rand() % 2001 - 1000) * 1e-3f;
    for (int k = 0; k < K; ++k)
        for (int j = 0; j < N; ++j) hk4[((size_t)(k >> 2) * N + j) * 4 + (k & 3)] = hH[(size_t)k * N + j];
    float *x, *H, *Hk4, *b, *y;
    CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&H, hH.size() * 4)); CK(hipMalloc(&Hk4, hk4.size() * 4)); CK(hipMalloc(&b, N * 4));
    CK(hipMalloc(&y, (size_t)M * N * 4));
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(H, hH.data(), hH.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(Hk4, hk4.data(), hk4.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb.data(), N * 4, hipMemcpyHostToDevice));
    const double gflop = 2.0 * M * K * N * 1e-9;
    auto report = [&](const char *name, float us) {
        CK(hipStreamSynchronize(s));
        const double err = check(hx, hH, hb, y, M, K, N);
        printf("M=%lld K=%d N=%d  %-34s %8.2f us  %6.1f TFLOP/s  (%.3f of 157.3)  err %.1e\n", M, K, N, name, us, gflop / us * 1e3 * 1e-3, gflop / us * 1e3 * 1e-3 / 157.3, err);
        CK(hipMemsetAsync(y, 0, (size_t)M * N * 4, s));
    };
    report("library phc_gemm_prebuilt", time_us(s, 50, [&] { phc_gemm_prebuilt(x, H, b, nullptr, M, K, N, y, 0, s); }));
    report("rs16 BN=128 W=8", time_us(s, 50, [&] { launch_rs<K, 128, 8>(x, Hk4, b, y, M, N, 0, s); }));
    stamped(s, [&] { launch_rs<K, 128, 8>(x, Hk4, b, y, M, N, 0, s); });
    if (N == 256) {
        report("hs16 TPW=2 W=8 RB=1 R=6", time_us(s, 50, [&] { launch_hs<K, 2, 8, 1, 6>(x, Hk4, b, y, M, N, 0, s); }));
        stamped(s, [&] { launch_hs<K, 2, 8, 1, 6>(x, Hk4, b, y, M, N, 0, s); });
        report("hs16 TPW=2 W=8 RB=1 R=4", time_us(s, 50, [&] { launch_hs<K, 2, 8, 1, 4>(x, Hk4, b, y, M, N, 0, s); }));
        stamped(s, [&] { launch_hs<K, 2, 8, 1, 4>(x, Hk4, b, y, M, N, 0, s); });
        report("hs16 TPW=2 W=8 RB=2 R=4", time_us(s, 50, [&] { launch_hs<K, 2, 8, 2, 4>(x, Hk4, b, y, M, N, 0, s); }));
        stamped(s, [&] { launch_hs<K, 2, 8, 2, 4>(x, Hk4, b, y, M, N, 0, s); });
        report("hs16 TPW=1 W=16 RB=2 R=4", time_us(s, 50, [&] { launch_hs<K, 1, 16, 2, 4>(x, Hk4, b, y, M, N, 0, s); }));
        stamped(s, [&] { launch_hs<K, 1, 16, 2, 4>(x, Hk4, b, y, M, N, 0, s); });
    } else {
        report("hs16 TPW=1 W=8 RB=2 R=4", time_us(s, 50, [&] { launch_hs<K, 1, 8, 2, 4>(x, Hk4, b, y, M, N, 0, s); }));
        stamped(s, [&] { launch_hs<K, 1, 8, 2, 4>(x, Hk4, b, y, M, N, 0, s); });
        report("hs16 TPW=2 W=4 RB=1 R=4", time_us(s, 50, [&] { launch_hs<K, 2, 4, 1, 4>(x, Hk4, b, y, M, N, 0, s); }));
        stamped(s, [&] { launch_hs<K, 2, 4, 1, 4>(x, Hk4, b, y, M, N, 0, s); });
        report("hs16 TPW=2 W=4 RB=2 R=4", time_us(s, 50, [&] { launch_hs<K, 2, 4, 2, 4>(x, Hk4, b, y, M, N, 0, s); }));
        stamped(s, [&] { launch_hs<K, 2, 4, 2, 4>(x, Hk4, b, y, M, N, 0, s); });
        report("hs16 TPW=2 W=4 RB=2 R=4 x2 groups", time_us(s, 50, [&] { launch_hs<K, 2, 4, 2, 4>(x, Hk4, b, y, M, N, 512, s); }));
        stamped(s, [&] { launch_hs<K, 2, 4, 2, 4>(x, Hk4, b, y, M, N, 512, s); });
    }
    CK(hipFree(x)); CK(hipFree(H)); CK(hipFree(Hk4)); CK(hipFree(b)); CK(hipFree(y));
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    run_case<256>(23600, 256, s);
    run_case<256>(27216, 256, s);
    run_case<256>(1000000, 256, s);
    run_case<128>(12874, 128, s);
    run_case<128>(1000000, 128, s);
    return 0;
}
