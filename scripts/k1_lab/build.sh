#!/bin/bash
# usage: build.sh name1=path1.hip[:extra -D flags] name2=path2.hip ...   -> scripts/k1_lab/k1_lab
set -e
HERE=$(cd $(dirname $0) && pwd)
ROOT=$(cd $HERE/../.. && pwd)
CSRC=$ROOT/nano-vllm-rs_amd/csrc
OUT=$HERE/_build
mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -DNVLLM_K1_LAB -I$CSRC -I$ROOT/include -Wall -Wno-unused-function -fno-gpu-rdc"
: > $OUT/variants.inc
OBJS=""
pids=""
for spec in "$@"; do
  name=${spec%%=*}; rest=${spec#*=}
  path=${rest%%:*}; extra=""
  if [[ "$rest" == *:* ]]; then extra=${rest#*:}; fi
  echo "LAB_VARIANT($name)" >> $OUT/variants.inc
  /opt/rocm/bin/hipcc $FLAGS $extra -Dlaunch_flash_attn_pp=launch_$name -Dflash_attn_pp_supported=supported_$name -Dlaunch_flash_attn_pp_paged_stub=launch_paged_$name -Dlaunch_flash_attn_mfma=launch_$name -Dflash_attn_mfma_supported=supported_$name -Dlaunch_flash_attn_paged=launch_paged_$name -Dflash_attn_paged_supported=paged_supported_$name -Dflash_attn_paged_workspace_bytes=paged_ws_$name \
      -x hip -c $path -o $OUT/$name.o &
  pids="$pids $!"
  OBJS="$OBJS $OUT/$name.o"
done
/opt/rocm/bin/hipcc $FLAGS -Dlaunch_flash_attn_mfma=lab_unused_launch -Dflash_attn_mfma_supported=lab_unused_supported \
    -x hip -c $CSRC/flash_attn.hip -o $OUT/generic.o &
pids="$pids $!"
/opt/rocm/bin/hipcc $FLAGS -x hip -c $CSRC/launch_policy.cpp -o $OUT/launch_policy.o &
pids="$pids $!"
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc $FLAGS -I$OUT -x hip -c $HERE/lab.cpp -o $OUT/lab.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -o $HERE/k1_lab $OUT/lab.o $OUT/generic.o $OUT/launch_policy.o $OBJS
echo built $HERE/k1_lab
