#!/bin/bash
# Build a variant of the HIP library that differs in cc_attn_decode_qkv.hip only (the QKV form of the layer step: 12 s to compile):
#   tools/ab_variant_qkv.sh NAME "-DCC_QKV_TRACE=1"   ->  .ab/libNAME.so   (load it with CC_LIB=.ab/libNAME.so tools/trace_qkv.py)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/.ab"
cd "$root/cold_compress_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1 -mllvm -amdgpu-kernarg-preload-count=14 $2 \
  -c cc_attn_decode_qkv.hip -o "/tmp/abq_$1.o"
objs=$(ls *.o | grep -v '^cc_attn_decode_qkv.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/.ab/lib$1.so" "/tmp/abq_$1.o" $objs
echo "built .ab/lib$1.so"
