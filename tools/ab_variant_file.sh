#!/bin/bash
# Build a variant of the HIP library for a same-box A/B run: recompiles ONE translation unit with extra flags and links it with the
# other objects of the current build.
#   tools/ab_variant_file.sh NAME cc_attn_prefill_mfma.hip "-DCC_KSTAT_SGB=0"   ->  .ab/libNAME.so   (.ab/ is git-ignored; it travels with gpurun)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/.ab"
cd "$root/cold_compress_amd/csrc"
base=$(basename "$2" .hip)
extra=""
case "$base" in cc_attn_decode|cc_attn_decode_qkv) extra="-mllvm -amdgpu-mfma-vgpr-form=1 -mllvm -amdgpu-kernarg-preload-count=14";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function $extra $3 -c "$2" -o "/tmp/abf_$1.o"
objs=$(ls *.o | grep -v "^$base.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/.ab/lib$1.so" "/tmp/abf_$1.o" $objs
echo "built .ab/lib$1.so"
