// Host-only sanitizer build (tests/test_host_sanitizers.py): capi.cpp, nvllm_host.cpp,
// nvllm_sched.cpp and launch_policy.cpp are compiled by g++ with -fsanitize=address,undefined and
// no device code, so the launchers the .hip files define are replaced by these stand-ins for OUR
// OWN kernels: every launch reports "no device", the shape/policy helpers answer as conservatively
// as possible.  Nothing here is product code.
#include "nvllm_kernels.hpp"

namespace nvllm {
namespace kernels {

hipError_t launch_reshape_and_cache(const void *, const void *, void *, void *, const int64_t *, int64_t,
                                    int64_t, int64_t, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_gather_blocks(const void *, const int32_t *, void *, int64_t, int64_t, int64_t,
                                hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_qk_norm_rope_cache(DType, const QkNormRopeCacheArgs &, hipStream_t) { return hipErrorNoDevice; }
bool paged_attn_mid_supported(DType, const PagedAttnArgs &) { return false; }
hipError_t launch_paged_attn_mid(const PagedAttnArgs &, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_validate_tables(const int32_t *, const int32_t *, int, int, int, int64_t, int32_t *,
                                  hipStream_t) { return hipErrorNoDevice; }
bool paged_attn_fast_path(DType, int, int, int, int, int, int, int) { return false; }
int paged_attn_num_splits(int, int, int) { return 1; }
size_t paged_attn_workspace_bytes(int num_seqs, int, int, int, int, int, int) { return num_seqs == 7 ? 4096 : 0; }
size_t paged_attn_launch_workspace_bytes(DType dt, const PagedAttnArgs &a) {
    return dt == DType::BF16 ? paged_attn_workspace_bytes(a.num_seqs, 0, 0, 0, 0, 0, 0) : 0;
}
PagedPath paged_attn_path(DType, const PagedAttnArgs &) { return PagedPath::Generic; }
hipError_t noalloc_prepare_device() { return hipSuccess; }
hipError_t launch_paged_attn(DType, const PagedAttnArgs &, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_paged_attn_noalloc(DType, const PagedAttnArgs &, hipStream_t) { return hipErrorNoDevice; }
bool flash_attn_paged_supported(DType, const PagedAttnArgs &) { return false; }
hipError_t launch_flash_attn_paged(const PagedAttnArgs &, hipStream_t) { return hipErrorNoDevice; }
size_t flash_attn_paged_workspace_bytes(const PagedAttnArgs &) { return 0; }
bool flash_attn_fast_path(DType, const FlashAttnArgs &) { return false; }
hipError_t launch_flash_attn(DType, const FlashAttnArgs &, hipStream_t) { return hipErrorNoDevice; }

}  // namespace kernels
}  // namespace nvllm
