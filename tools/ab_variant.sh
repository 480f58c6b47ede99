#!/bin/bash
# Build a variant of the HIP library for a same-box A/B timing run: recompiles cc_attn_decode.hip with extra flags and links it
# with the other objects of the current build.
#   tools/ab_variant.sh NAME "-DSOME_EXPERIMENT"   ->  .ab/libNAME.so      (.ab/ is git-ignored; it travels with gpurun)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/.ab"
cd "$root/cold_compress_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1 -mllvm -amdgpu-kernarg-preload-count=14 $2 \
  -c cc_attn_decode.hip -o "/tmp/ab_$1.o"
objs=$(ls *.o | grep -v '^cc_attn_decode.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/.ab/lib$1.so" "/tmp/ab_$1.o" $objs
echo "built .ab/lib$1.so"
