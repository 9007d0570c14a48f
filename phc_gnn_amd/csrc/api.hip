// libphc_hip.so — library-level entry points: version, errors, device info and
// the opt-in launch profiler (include/phc_hip.h).
#include <stdarg.h>
#include <string.h>

#include <mutex>
#include <unordered_map>
#include <vector>

#include "phc_common.h"
#include <string>

namespace phc {

static thread_local char t_last_error[512] = "";

void set_last_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t_last_error, sizeof(t_last_error), fmt, ap);
    va_end(ap);
}

static int *g_err_words[64] = {};
int *err_word() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    return g_err_words[dev];
}

static const char *const kKernelNames[K_NUM_KERNELS] = {
    "phm_build_H",      "phm_gemm_fwd", "phm_gemm_dx", "phm_dH_partial", "phm_dH_reduce", "phm_contract_dW_dA",
    "aggregate_fwd",    "aggregate_bwd", "gather_rows", "csr_histogram",  "csr_scan",      "csr_fill",
    "csr_canonicalize", "encode_fwd",   "aggregate_enc_fwd", "aggregate_enc_dx", "table_grad", "table_grad_reduce",
    "bn_stats_partial", "bn_finalize", "bn_apply", "bn_bwd_partial", "bn_bwd_finalize", "bn_bwd_apply",
    "pack_edge_codes", "aggregate_softmax_fwd", "aggregate_softmax_bwd",
    "bce_masked_fwd", "bce_masked_bwd", "gated_pool_fwd", "gated_pool_bwd", "clip_adam", "small_linear", "sum_n",
    "downstream_fwd", "downstream_bwd", "phm_gemm_ag_fwd",
};

// ---- profiler ------------------------------------------------------------------
bool g_prof_on = false;
bool g_ablate[K_NUM_KERNELS] = {};
thread_local bool tl_ablate = false;
thread_local hipEvent_t tl_prof_start = nullptr, tl_prof_stop = nullptr;
static const bool g_prof_event_pair = getenv("PHC_PROF_EVENT_PAIR") != nullptr;

// PHC_ABLATE is read once, when the library is loaded
static const bool g_ablate_init = [] {
    const char *env = getenv("PHC_ABLATE");
    if (!env) return false;
    std::string list(env);
    size_t pos = 0;
    while (pos <= list.size()) {
        size_t comma = list.find(',', pos);
        if (comma == std::string::npos) comma = list.size();
        const std::string name = list.substr(pos, comma - pos);
        for (int k = 0; k < K_NUM_KERNELS; ++k)
            if (name == kKernelNames[k]) {
                g_ablate[k] = true;
                fprintf(stderr, "[libphc_hip] PHC_ABLATE: launches of '%s' are DROPPED (timing experiment, results invalid)\n",
                        kKernelNames[k]);
            }
        pos = comma + 1;
    }
    return true;
}();

struct ProfSlot {
    hipEvent_t start, stop;
    int kid;
};
static std::mutex g_prof_mu;
static std::vector<ProfSlot> g_slots;   // event pairs recorded since the last read
static std::vector<ProfSlot> g_free;    // recycled pairs
static uint64_t g_launches[K_NUM_KERNELS];
static double g_total_ms[K_NUM_KERNELS];

void prof_begin(int kid, hipStream_t s, int *slot) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfSlot p;
    if (!g_free.empty()) {
        p = g_free.back();
        g_free.pop_back();
    } else {
        if (hipEventCreate(&p.start) != hipSuccess || hipEventCreate(&p.stop) != hipSuccess) {
            *slot = -1;
            return;
        }
    }
    p.kid = kid;
    if (g_prof_event_pair) {
        hipEventRecord(p.start, s);
    } else {
        tl_prof_start = p.start;                 // PHC_LAUNCH hands them to hipExtLaunchKernelGGL
        tl_prof_stop = p.stop;
    }
    g_slots.push_back(p);
    *slot = (int)g_slots.size() - 1;
}

void prof_end(int slot, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (g_prof_event_pair) {
        if (slot >= 0 && slot < (int)g_slots.size()) hipEventRecord(g_slots[slot].stop, s);
    } else {
        tl_prof_start = tl_prof_stop = nullptr;
    }
}

static void prof_drain_locked() {
    for (ProfSlot &p : g_slots) {
        float ms = 0.f;
        if (hipEventSynchronize(p.stop) == hipSuccess && hipEventElapsedTime(&ms, p.start, p.stop) == hipSuccess) {
            g_launches[p.kid] += 1;
            g_total_ms[p.kid] += (double)ms;
        }
        g_free.push_back(p);
    }
    g_slots.clear();
}

// (kernel address, device ordinal) -> largest dynamic-LDS size the attribute was set to (phc_common.h: ensure_dynamic_lds)
static std::mutex g_lds_mu;
static std::unordered_map<unsigned long long, size_t> g_lds_done;
bool ensure_dynamic_lds_addr(const void *kernel, size_t bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 256) {
        (void)hipGetLastError();
        return false;
    }
    // code addresses fit 56 bits: the device ordinal rides in the top byte of the key
    const unsigned long long key = ((unsigned long long)(uintptr_t)kernel & 0x00ffffffffffffffull) | ((unsigned long long)dev << 56);
    std::lock_guard<std::mutex> lk(g_lds_mu);
    auto it = g_lds_done.find(key);
    if (it != g_lds_done.end() && it->second >= bytes) return true;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    g_lds_done[key] = bytes;
    return true;
}

}  // namespace phc

using namespace phc;

extern "C" {

int phc_version(void) { return PHC_ABI_VERSION; }

const char *phc_error_string(int code) {
    switch (code) {
        case PHC_OK: return "ok";
        case PHC_ERR_INVALID_ARG: return "invalid argument";
        case PHC_ERR_UNSUPPORTED: return "unsupported shape";
        case PHC_ERR_WORKSPACE: return "workspace missing or too small";
        case PHC_ERR_LAUNCH: return "HIP launch error";
        case PHC_ERR_NO_DEVICE: return "no gfx950 device";
        default: return "unknown error";
    }
}

const char *phc_last_error(void) { return t_last_error; }

int phc_set_error_word(int device, int32_t *dev_word) {
    PHC_REQUIRE(device >= 0 && device < 64, PHC_ERR_INVALID_ARG, "phc_set_error_word: device %d", device);
    g_err_words[device] = reinterpret_cast<int *>(dev_word);
    return PHC_OK;
}

int phc_device_info(int *num_cu, int *lds_bytes_per_cu, char *arch, int arch_len) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        set_last_error("hipGetDevice failed: no HIP device visible");
        return PHC_ERR_NO_DEVICE;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        set_last_error("hipGetDeviceProperties failed");
        return PHC_ERR_NO_DEVICE;
    }
    if (num_cu) *num_cu = prop.multiProcessorCount;
    if (lds_bytes_per_cu) *lds_bytes_per_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
    if (arch && arch_len > 0) {
        strncpy(arch, prop.gcnArchName, (size_t)arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    return PHC_OK;
}

int phc_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
    return PHC_OK;
}

int phc_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    prof_drain_locked();
    memset(g_launches, 0, sizeof(g_launches));
    memset(g_total_ms, 0, sizeof(g_total_ms));
    return PHC_OK;
}

int phc_prof_num_kernels(void) { return K_NUM_KERNELS; }

int phc_prof_read(int kernel_id, const char **name, uint64_t *launches, double *total_ms) {
    PHC_REQUIRE(kernel_id >= 0 && kernel_id < K_NUM_KERNELS, PHC_ERR_INVALID_ARG, "phc_prof_read: bad kernel id %d",
                kernel_id);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    prof_drain_locked();
    if (name) *name = kKernelNames[kernel_id];
    if (launches) *launches = g_launches[kernel_id];
    if (total_ms) *total_ms = g_total_ms[kernel_id];
    return PHC_OK;
}

}  // extern "C"
