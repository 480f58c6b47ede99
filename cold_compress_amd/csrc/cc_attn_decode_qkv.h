// cc_attn_decode_qkv.h — the QKV instantiations of the single-launch layer step live in their own translation unit
// (cc_attn_decode_qkv.hip); cc_attn_decode.hip, which builds the step's arguments, reaches them through these two functions.
// The argument block is the kernels header's SplitArgs, passed as bytes: that struct sits in an anonymous namespace (one type per
// translation unit, the same layout in both — they include the same header with the same flags).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

// workgroups of the instantiation the device keeps resident at once (0: no such instantiation, or the runtime could not say)
int cc_qkv_step_capacity(int dtype, int rt, int nw, int xl2);
// launch it: grid (grid_x, grid_y, 1), nw * 64 threads; CC_OK / CC_ERR_*
int cc_qkv_step_launch(const void* split_args, size_t split_args_bytes, int dtype, int rt, int nw, int xl2, int grid_x, int grid_y,
                       hipStream_t stream);
