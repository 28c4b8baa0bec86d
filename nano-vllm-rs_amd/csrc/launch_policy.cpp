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

// one relaxed atomic per field: launches and workspace queries read them without a lock and never
// see a torn value; set / reset are serialised by g_mu
constexpr int kNumFields = 11;
std::atomic<int> g_values[kNumFields];
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
    {"NVLLM_K1_CHAIN", &LaunchPolicy::k1_chain, -1},
    {"NVLLM_PAGED_MID", &LaunchPolicy::paged_mid, -1},
};

static_assert(sizeof(kFields) / sizeof(kFields[0]) == kNumFields, "one atomic per field");

// values a launcher could not work with are brought into range here, once, not on a launch path
int sanitise(const Field &f, int v) {
    if (f.member == &LaunchPolicy::paged_wave_target && v < 1) return 1;
    return v;
}

int parse(const Field &f, const char *s) {
    if (!s || !*s) return f.dflt;
    if (f.member == &LaunchPolicy::k1_pairing) {
        if (s[0] == 'a') return 1;
        if (s[0] == 'b') return 0;
    }
    return sanitise(f, std::atoi(s));
}

void load_env() {
    for (int i = 0; i < kNumFields; ++i)
        g_values[i].store(parse(kFields[i], std::getenv(kFields[i].name)), std::memory_order_relaxed);
}

}  // namespace

LaunchPolicy launch_policy() {
    std::call_once(g_once, load_env);
    LaunchPolicy p{};
    for (int i = 0; i < kNumFields; ++i) p.*(kFields[i].member) = g_values[i].load(std::memory_order_relaxed);
    return p;
}

bool set_launch_policy(const char *name, int value) {
    if (!name) return false;
    std::call_once(g_once, load_env);
    std::lock_guard<std::mutex> lock(g_mu);
    for (int i = 0; i < kNumFields; ++i)
        if (std::strcmp(name, kFields[i].name) == 0) {
            g_values[i].store(sanitise(kFields[i], value), std::memory_order_relaxed);
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
