// flash_attn_mfma.hip -- K1 fast path (bf16 MFMA, head_dim 128).  Placeholder
// until the MFMA kernel lands: reports "unsupported" so every shape runs the
// generic kernel of flash_attn.hip.
#include <hip/hip_runtime.h>

#include "nvllm_kernels.hpp"

namespace nvllm {
namespace kernels {

bool flash_attn_mfma_supported(const FlashAttnArgs &) { return false; }

hipError_t launch_flash_attn_mfma(const FlashAttnArgs &, hipStream_t) {
    return hipErrorNotSupported;
}

}  // namespace kernels
}  // namespace nvllm
