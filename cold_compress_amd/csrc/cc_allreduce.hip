// cc_allreduce.hip — one-shot sum all-reduce over the GPUs of ONE node for the decode-size messages of tensor parallelism.
//
// ref: tp.py:134-138, 156-160 — two `all_reduce(sum)` per layer on a [1, 1, dim] activation: 8-16 KiB at decode time,
// 160 of them per token on Llama-3-70B at TP = 8.  At that size a ring collective is pure latency (2 (N - 1) hops of a
// per-link protocol); xGMI is a point-to-point mesh, so every rank can instead WRITE its vector straight into a slot of
// every peer's buffer, raise a flag, wait for the N - 1 flags raised for it, and add the N slots itself:
//
//   all-gather by remote stores (one hop, all links busy at once)  ->  flag  ->  local reduction in RANK ORDER
//
// one launch of one workgroup per rank, no intermediate kernel boundaries, capturable in a hipGraph (the epoch lives in
// device memory), and — the additions being performed in the same order everywhere — bitwise-identical results on every
// rank, which keeps replicated state (sampled tokens, head-constant eviction decisions) in lock step.
//
// Memory: each rank owns ONE symmetric buffer (uncached fine-grained device memory, so remote stores and polls need no cache
// maintenance), exported as a hipIpcMemHandle_t; the host exchanges the handles (torch.distributed all_gather) and every rank
// maps its peers' buffers.  Layout: [header: epoch word, status word][flags: 2 sets x world x 64 B][slots: 2 sets x world x
// max_bytes].  Two sets alternate with the epoch's parity: a rank may run at most one all-reduce ahead of its slowest peer
// (it cannot finish e + 1 before every peer has arrived in e + 1, i.e. left e), so set e & 1 is never written while a peer
// still reads it.  Every wait is bounded; a timeout sets the status word instead of hanging the device.
#include <stdlib.h>
#include <string.h>

#include "cc_common.h"

namespace {

constexpr int kArMaxWorld = 16;
constexpr int kArThreads = 1024;
constexpr size_t kArHeader = 256;
constexpr size_t kArFlagStride = 64;  // one flag per 64-byte line: a peer's flag store never shares a line with another's

struct cc_comm_impl {
  int rank, world;
  size_t max_bytes, total_bytes;
  char* local;                 // this rank's buffer
  char* peer[kArMaxWorld];     // peer[r] = rank r's buffer as mapped here (peer[rank] == local)
  bool mapped[kArMaxWorld];
  bool connected;
};

struct ArArgs {
  char* peer[kArMaxWorld];
  int rank, world;
  size_t max_bytes;
  void* data;
  long long n;
};

__device__ __forceinline__ size_t ar_flag_off(int world, int set, int r) { return kArHeader + ((size_t)set * world + r) * kArFlagStride; }
__device__ __forceinline__ size_t ar_slot_off(int world, size_t max_bytes, int set, int r) {
  return kArHeader + 2 * (size_t)world * kArFlagStride + ((size_t)set * world + r) * max_bytes;
}

template <typename T>
__global__ __launch_bounds__(kArThreads) void allreduce_oneshot_kernel(ArArgs a) {
  __shared__ unsigned sm_epoch;
  __shared__ int sm_bad;
  char* mine = a.peer[a.rank];
  if (threadIdx.x == 0) {
    unsigned* ep = reinterpret_cast<unsigned*>(mine);
    sm_epoch = *ep + 1u;  // only this rank's kernels touch its epoch word, one launch at a time (stream order)
    *ep = sm_epoch;
    sm_bad = 0;
  }
  __syncthreads();
  const unsigned epoch = sm_epoch;
  const int set = (int)(epoch & 1u);
  constexpr int VEC = 16 / (int)sizeof(T);
  const long long nvec = (a.n + VEC - 1) / VEC;  // the caller's buffer is padded to 16 bytes (checked on the host)
  // ---- all-gather: my vector into slot `rank` of every rank's buffer (my own included)
  for (long long i = threadIdx.x; i < nvec; i += kArThreads) {
    const uint4 v = reinterpret_cast<const uint4*>(a.data)[i];
    for (int r = 0; r < a.world; r++) {
      uint4* dst = reinterpret_cast<uint4*>(a.peer[r] + ar_slot_off(a.world, a.max_bytes, set, a.rank)) + i;
      __builtin_nontemporal_store(v.x, &dst->x);
      __builtin_nontemporal_store(v.y, &dst->y);
      __builtin_nontemporal_store(v.z, &dst->z);
      __builtin_nontemporal_store(v.w, &dst->w);
    }
  }
  __threadfence_system();  // my stores are visible system-wide before any of my flags
  __syncthreads();
  if ((int)threadIdx.x < a.world) {
    unsigned* f = reinterpret_cast<unsigned*>(a.peer[threadIdx.x] + ar_flag_off(a.world, set, a.rank));
    __hip_atomic_store(f, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // ---- wait for every rank's flag in MY buffer (bounded: ~2 s of polling)
  if ((int)threadIdx.x < a.world) {
    const unsigned* f = reinterpret_cast<const unsigned*>(mine + ar_flag_off(a.world, set, threadIdx.x));
    bool ok = false;
    for (unsigned spins = 0; spins < (1u << 24); spins++) {
      if (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == epoch) {
        ok = true;
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
    if (!ok) {
      sm_bad = 1;
      reinterpret_cast<unsigned*>(mine)[1] = 1u;  // status word: this all-reduce did not complete
    }
  }
  __syncthreads();
  if (sm_bad) {
    // a peer's flag did not arrive in time: the status word is set (above) AND the output is poisoned, so that a caller that
    // never reads the word cannot mistake this rank's partial sum for the result (NaN propagates into the logits)
    for (long long i = threadIdx.x; i < a.n; i += kArThreads) ElemTraits<T>::store(reinterpret_cast<T*>(a.data), i, NAN);
    return;
  }
  // ---- local reduction, slots in rank order 0 .. world - 1 (fp32 accumulate, one rounding): identical on every rank
  for (long long i = threadIdx.x; i < nvec; i += kArThreads) {
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; e++) acc[e] = 0.f;
    for (int r = 0; r < a.world; r++) {
      Vec16<T> v;
      v.raw = reinterpret_cast<const uint4*>(mine + ar_slot_off(a.world, a.max_bytes, set, r))[i];
      float f[VEC];
      v.unpack(f);
#pragma unroll
      for (int e = 0; e < VEC; e++) acc[e] = r == 0 ? f[e] : acc[e] + f[e];
    }
    T* out = reinterpret_cast<T*>(a.data) + i * VEC;
#pragma unroll
    for (int e = 0; e < VEC; e++)
      if (i * VEC + e < a.n) ElemTraits<T>::store(out, e, acc[e]);
  }
}

}  // namespace

extern "C" {

size_t cc_allreduce_handle_bytes(void) { return sizeof(hipIpcMemHandle_t); }

int cc_allreduce_create(int32_t rank, int32_t world, size_t max_bytes, cc_comm** out) {
  CC_ENTRY();
  if (!out || world < 1 || world > kArMaxWorld || rank < 0 || rank >= world || max_bytes == 0) return CC_ERR_BAD_ARG;
  cc_comm_impl* c = reinterpret_cast<cc_comm_impl*>(calloc(1, sizeof(cc_comm_impl)));
  if (!c) return CC_ERR_HIP;
  c->rank = rank;
  c->world = world;
  c->max_bytes = (max_bytes + 255) & ~(size_t)255;
  c->total_bytes = kArHeader + 2 * (size_t)world * kArFlagStride + 2 * (size_t)world * c->max_bytes;
  void* p = nullptr;
  // uncached fine-grained device memory: stores from peers over xGMI and polls by the owner are coherent without fences on
  // the cache hierarchy of either side
  // (NO fallback to plain hipMalloc: on coarse-grained, L2-cacheable memory the owner's plain loads of a slot may be served a
  //  stale line after its flag has been seen — wrong sums that a short self-test can miss.  The caller keeps RCCL instead.)
  if (hipExtMallocWithFlags(&p, c->total_bytes, hipDeviceMallocUncached) != hipSuccess) {
    (void)hipGetLastError();
    free(c);
    return CC_ERR_HIP;
  }
  if (hipMemset(p, 0, c->total_bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    (void)hipFree(p);
    free(c);
    return CC_ERR_HIP;
  }
  c->local = reinterpret_cast<char*>(p);
  c->peer[rank] = c->local;
  c->connected = world == 1;
  *out = reinterpret_cast<cc_comm*>(c);
  return CC_OK;
}

int cc_allreduce_export(cc_comm* comm, void* handle_out) {
  CC_ENTRY();
  cc_comm_impl* c = reinterpret_cast<cc_comm_impl*>(comm);
  if (!c || !handle_out) return CC_ERR_BAD_ARG;
  hipIpcMemHandle_t h;
  if (hipIpcGetMemHandle(&h, c->local) != hipSuccess) return CC_ERR_HIP;
  memcpy(handle_out, &h, sizeof(h));
  return CC_OK;
}

int cc_allreduce_connect(cc_comm* comm, const void* handles) {
  CC_ENTRY();
  cc_comm_impl* c = reinterpret_cast<cc_comm_impl*>(comm);
  if (!c || !handles) return CC_ERR_BAD_ARG;
  for (int r = 0; r < c->world; r++) {
    if (r == c->rank || c->mapped[r]) continue;
    hipIpcMemHandle_t h;
    memcpy(&h, reinterpret_cast<const char*>(handles) + (size_t)r * sizeof(h), sizeof(h));
    void* p = nullptr;
    if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) return CC_ERR_HIP;
    c->peer[r] = reinterpret_cast<char*>(p);
    c->mapped[r] = true;
  }
  c->connected = true;
  return CC_OK;
}

int cc_allreduce_sum(cc_comm* comm, void* data, int64_t n, int32_t dtype, cc_stream_t stream) {
  CC_ENTRY();
  cc_comm_impl* c = reinterpret_cast<cc_comm_impl*>(comm);
  if (!c || !data || n <= 0 || !cc_dt_ok(dtype) || !c->connected) return CC_ERR_BAD_ARG;
  const size_t bytes = (((size_t)n * cc_dt_size(dtype)) + 15) & ~(size_t)15;
  if (bytes > c->max_bytes || (reinterpret_cast<uintptr_t>(data) & 15)) return CC_ERR_UNSUPPORTED;
  ArArgs a{};
  for (int r = 0; r < c->world; r++) a.peer[r] = c->peer[r];
  a.rank = c->rank; a.world = c->world; a.max_bytes = c->max_bytes; a.data = data; a.n = n;
  hipStream_t st = (hipStream_t)stream;
  switch (dtype) {
    case CC_DT_F32: hipLaunchKernelGGL(allreduce_oneshot_kernel<float>, dim3(1), dim3(kArThreads), 0, st, a); break;
    case CC_DT_BF16: hipLaunchKernelGGL(allreduce_oneshot_kernel<bf16_t>, dim3(1), dim3(kArThreads), 0, st, a); break;
    default: hipLaunchKernelGGL(allreduce_oneshot_kernel<f16_t>, dim3(1), dim3(kArThreads), 0, st, a); break;
  }
  CC_LAUNCH_CHECK();
  return CC_OK;
}

int32_t cc_allreduce_status(cc_comm* comm) {
  cc_comm_impl* c = reinterpret_cast<cc_comm_impl*>(comm);
  if (!c) return -1;
  unsigned st = 0;
  if (hipMemcpy(&st, c->local + sizeof(unsigned), sizeof(st), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return (int32_t)st;
}

int cc_allreduce_destroy(cc_comm* comm) {
  cc_comm_impl* c = reinterpret_cast<cc_comm_impl*>(comm);
  if (!c) return CC_OK;
  for (int r = 0; r < c->world; r++)
    if (c->mapped[r]) (void)hipIpcCloseMemHandle(c->peer[r]);
  (void)hipFree(c->local);
  free(c);
  return CC_OK;
}

}  // extern "C"
