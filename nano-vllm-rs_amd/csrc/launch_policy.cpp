// launch_policy.cpp -- launch-policy overrides and the per-device CU count (see nvllm_kernels.hpp).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "nvllm_kernels.hpp"

namespace nvllm {
namespace kernels {

namespace {

LaunchPolicy g_policy;
std::once_flag g_once;
std::mutex g_mu;

struct Field {
    const char *name;
    int LaunchPolicy::*member;
    int dflt;
};
const Field kFields[] = {
    {"NVLLM_PAGED_MAX_ROWS", &LaunchPolicy::paged_max_rows, 0},
    {"NVLLM_PAGED_SPLITS", &LaunchPolicy::paged_splits, 0},
    {"NVLLM_PAGED_WAVE_TARGET", &LaunchPolicy::paged_wave_target, 2048},
    {"NVLLM_PAGED_NT", &LaunchPolicy::paged_nt, -1},
    {"NVLLM_PAGED_HPB", &LaunchPolicy::paged_hpb, 0},
    {"NVLLM_PAGED_FLASH", &LaunchPolicy::paged_flash, 1},
    {"NVLLM_K1_PAIRING", &LaunchPolicy::k1_pairing, -1},
    {"NVLLM_K1_PACK", &LaunchPolicy::k1_pack, -1},
    {"NVLLM_K1_KSPLIT", &LaunchPolicy::k1_ksplit, 0},
};

int parse(const Field &f, const char *s) {
    if (!s || !*s) return f.dflt;
    if (f.member == &LaunchPolicy::k1_pairing) {
        if (s[0] == 'a') return 1;
        if (s[0] == 'b') return 0;
    }
    return std::atoi(s);
}

void load_env() {
    for (const Field &f : kFields) g_policy.*(f.member) = parse(f, std::getenv(f.name));
}

}  // namespace

const LaunchPolicy &launch_policy() {
    std::call_once(g_once, load_env);
    return g_policy;
}

bool set_launch_policy(const char *name, int value) {
    if (!name) return false;
    std::call_once(g_once, load_env);
    std::lock_guard<std::mutex> lock(g_mu);
    for (const Field &f : kFields)
        if (std::strcmp(name, f.name) == 0) {
            g_policy.*(f.member) = value;
            return true;
        }
    return false;
}

void reset_launch_policy() {
    std::call_once(g_once, load_env);
    std::lock_guard<std::mutex> lock(g_mu);
    load_env();
}

int device_cus() {
    constexpr int kMaxDevices = 64;
    static std::atomic<int> cus[kMaxDevices];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 256;
    int n = cus[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        cus[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

}  // namespace kernels
}  // namespace nvllm
