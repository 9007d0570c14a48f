// libphc_hip.so — the aggregation-inside-the-first-GEMM launch for the wide layers (hidden 256, and the published 180 / 200 on the
// 192 / 208 register geometry): gemm_hs_agw_device.h.  Its own translation unit: nine heavily unrolled instantiations that compile
// beside phm_linear.hip instead of behind it.  Entry point: phc_gemm_aggregate_stats (phm_linear.hip) calls launch_gemm_agw.
#include <stdlib.h>
#include "phc_common.h"
#include "gemm_hs_agw_device.h"

namespace phc {

bool gemm_agw_shape_ok(int h, int Nout, int vocab_rows) {
    return h == Nout && (h == 256 || h == 180 || h == 200) && vocab_rows <= HS_AGW_MAX_ROWS;
}

// `a.groups` is set by the caller (the count gemm_hs16_kernel<K, 2, 8> would run on the same rows: identical statistics totals)
int launch_gemm_agw(const GemmHsArgs &a, const HsAgArgs &ag, bool normed, hipStream_t s) {
    const int tblR = ag.off[ag.F];
    const size_t dyn = (size_t)tblR * a.K * sizeof(float);
    constexpr size_t DYN_MAX = (size_t)HS_AGW_MAX_ROWS * 256 * sizeof(float);
#define PHC_AGW1(KC, NC, RC, NEC, FC)                                                                                          \
    do {                                                                                                                       \
        PHC_REQUIRE(ensure_dynamic_lds(&gemm_hs_agw_kernel<KC, NC, RC, NEC, FC>, DYN_MAX), PHC_ERR_LAUNCH,                      \
                    "phc_gemm_aggregate_stats: the device refused the dynamic LDS");                                          \
        PHC_LAUNCH((gemm_hs_agw_kernel<KC, NC, RC, NEC, FC>), dim3((unsigned)a.groups), dim3(HS_AGW_THREADS), dyn, s, a, ag);  \
    } while (0)
    // edges of a row fetched ahead: 4 (plain: both rows a whole step ahead) / 3 (a norm pending: every gathered row is two loads,
    // y and the skip, and one row's pieces are live at a time)
#define PHC_AGW(KC)                                   \
    if (normed && ag.norm_act == PHC_ACT_RELU)        \
        PHC_AGW1(KC, true, true, 3, false);           \
    else if (normed)                                  \
        PHC_AGW1(KC, true, false, 3, false);          \
    else                                              \
        PHC_AGW1(KC, false, true, 4, true)
    if (a.K == 256) {
        PHC_AGW(256);
    } else if (a.K == 200) {
        PHC_AGW(208);
    } else {
        PHC_AGW(192);
    }
#undef PHC_AGW
#undef PHC_AGW1
    return PHC_OK;
}

}  // namespace phc
