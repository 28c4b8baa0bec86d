#!/bin/bash
# usage: asm.sh file.hip out.s [extra flags]  -> device ISA + resource usage
f=$1; o=$2; shift 2
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I/root/repo/nano-vllm-rs_amd/csrc -I/root/repo/include "$@" -x hip --cuda-device-only -S $f -o $o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|Function Name|VGPRs:|Spill|LDS Size" | sed 's/.*remark: //'
