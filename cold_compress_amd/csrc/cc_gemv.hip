// cc_gemv.hip — the five dense matrix-vector products of a decode layer, with the caller glue fused in:
//   wqkv : RMSNorm(x + delta) prologue (the pending residual add included) + RoPE epilogue on the q / k rows
//   wo   : plain
//   w1/w3: RMSNorm prologue + SwiGLU epilogue  silu(w1·n) * (w3·n)  (one pass over both matrices)
//   w2   : plain
// ref (caller side): model.py:317-327 (pre-norm block), :375-387 (split + apply_rotary_emb), :442-443 (FFN),
// :452-457 (RMSNorm), :507-519 (RoPE).  Eleven launches per layer become six; every rounding point of the eager
// bf16 chain is kept (each tensor op rounds to the model dtype).
//
// HBM-bound streaming of W (batch 1: 2 flops per 2-byte weight).  One WAVE owns a row at a time: its 64 lanes
// stride over K with 16-byte loads (one fully coalesced 1 KiB segment per load instruction), RB rows and CU
// column steps are in flight together (RB*CU 16-byte loads per lane), products go through v_dot2c_f32_bf16 /
// v_dot2_f32_f16 (fp32 accumulate), the row total through 4 DPP steps + 4 v_readlane.  The input vector is
// normalised ONCE per workgroup into LDS as packed 16-bit (every workgroup recomputes the 8 KiB norm rather than
// paying a launch for it) and kept in the registers of the lanes that multiply it.  No atomics, fixed orders.
#include <cstdio>
#include <cstdlib>

#include "cc_common.h"
#include "cc_gemv_core.h"

namespace {

struct GemvArgs {
  const void* W;
  const void* W3;      // second matrix of the SwiGLU pair, or null
  const void* x;       // [K]
  const void* delta;   // [K] pending residual, or null
  const void* norm_w;  // [K] RMSNorm weight, or null (no norm prologue)
  const void* bias;    // [N] or null
  const void* freqs;   // [head_dim/2, 2] (cos, sin) of this position, or null
  void* h_out;         // [K] x + delta, or null
  void* y;             // [N]
  float eps;
  int N, K, rope_rows, head_dim;
};

constexpr int kGvThreads = 256;
constexpr int kGvWaves = kGvThreads / 64;

template <typename T, bool SWIGLU, int RB, int CU, int XS>
__global__ __launch_bounds__(kGvThreads) void gemv_kernel(GemvArgs a) {
  constexpr int VEC = 16 / (int)sizeof(T);
  __shared__ float sm_red[kGvWaves];
  __shared__ float sm_part[kGvWaves][2][RB];
  const int K = a.K, N = a.N;
  const int nch = K / VEC;
  const T* xg = reinterpret_cast<const T*>(a.x);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

  // ---- Work decomposition: a WORKGROUP owns RB consecutive rows at a time; its four waves split K (wave w takes
  //      the 1 KiB segments w, w+4, w+8, ... of every row), CU segments per row in flight per wave, and the four
  //      partial sums meet in LDS.  For the decode shapes this puts a whole matrix in flight in one phase
  //      (wo: 2 rows x 2 segments per wave; w2: 4 rows x 4 of 7 segments) instead of looping inside a wave.
  //      The FIRST tile's weight loads are issued before the input vector is touched: W does not depend on x.
  const int nseg = (nch + 63) / 64;                      // 1 KiB segments per row
  const int nstep = (nseg + kGvWaves - 1) / kGvWaves;    // segments per wave (<= XS, checked by the launcher)
  const uint4* Wv = reinterpret_cast<const uint4*>(a.W);
  const uint4* W3v = reinterpret_cast<const uint4*>(a.W3);
  uint4 w[RB][CU], w3[RB][CU];
  auto issue = [&](int r0, int s0) {
#pragma unroll
    for (int u = 0; u < CU; u++) {
      const int c = ((s0 + u) * kGvWaves + wave) * 64 + lane;
      const bool cin = (s0 + u < nstep) && c < nch;
#pragma unroll
      for (int r = 0; r < RB; r++) {
        const bool in = cin && (r0 + r < N);
        const size_t off = (size_t)(r0 + r) * nch + c;
        // non-temporal: 15 GB of weights pass once per token, nothing is gained by keeping them in L2 / the 256 MB
        // Infinity Cache (measured: w1+w3 43.1 -> 40.1 us, w2 24.8 -> 22.5 us)
        w[r][u] = in ? nt_load(Wv + off) : make_uint4(0, 0, 0, 0);
        if (SWIGLU) w3[r][u] = in ? nt_load(W3v + off) : make_uint4(0, 0, 0, 0);
      }
    }
  };
  const int row_first = blockIdx.x * RB;
  if (row_first < N) issue(row_first, 0);

  // ---- the input vector: every lane keeps exactly the 16-byte chunks it will multiply (no LDS staging).  With a
  //      norm prologue, x / delta / norm weight of those chunks are requested together (one memory round trip), the
  //      four waves exchange their partial sums of squares through LDS (one barrier), and the normalised chunks
  //      replace h in the same registers.
  uint4 xr[XS];
  if (a.norm_w != nullptr) {
    const T* dg = reinterpret_cast<const T*>(a.delta);
    const T* wg = reinterpret_cast<const T*>(a.norm_w);
    Vec16<T> xv[XS], dv[XS], nv[XS];
#pragma unroll
    for (int j = 0; j < XS; j++) {
      const int c = (j * kGvWaves + wave) * 64 + lane;
      if (j < nstep && c < nch) {
        xv[j].load(xg + (size_t)c * VEC);
        if (dg != nullptr) dv[j].load(dg + (size_t)c * VEC);
        nv[j].load(wg + (size_t)c * VEC);
      }
    }
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < XS; j++) {
      const int c = (j * kGvWaves + wave) * 64 + lane;
      xr[j] = make_uint4(0, 0, 0, 0);
      if (j < nstep && c < nch) {
        float h[VEC];
        xv[j].unpack(h);
        if (dg != nullptr) {
          float d[VEC];
          dv[j].unpack(d);
#pragma unroll
          for (int e = 0; e < VEC; e++) h[e] = ElemTraits<T>::rnd(__fadd_rn(h[e], d[e]));  // model-dtype residual add
        }
#pragma unroll
        for (int e = 0; e < VEC; e++) ss = fmaf(h[e], h[e], ss);
        xr[j] = pack16<T>(h);  // h is already rounded to T: packing is exact
        if (a.h_out != nullptr && blockIdx.x == 0) reinterpret_cast<uint4*>(a.h_out)[c] = xr[j];
      }
    }
    ss = gv_wave_sum(ss);
    if (lane == 0) sm_red[wave] = ss;
    __syncthreads();
    const float tot = (sm_red[0] + sm_red[1]) + (sm_red[2] + sm_red[3]);
    const float rs = rsqrtf(tot / (float)K + a.eps);  // ref: model.py:452-457 (fp32 inside)
#pragma unroll
    for (int j = 0; j < XS; j++) {
      const int c = (j * kGvWaves + wave) * 64 + lane;
      if (j < nstep && c < nch) {
        Vec16<T> hv;
        float h[VEC], wf[VEC], o[VEC];
        hv.raw = xr[j];
        hv.unpack(h);
        nv[j].unpack(wf);
#pragma unroll
        for (int e = 0; e < VEC; e++) o[e] = ElemTraits<T>::rnd(__fmul_rn(ElemTraits<T>::rnd(cc_opaque_f32(__fmul_rn(h[e], rs))), wf[e]));  // (opaque: see the epilogue)
        xr[j] = pack16<T>(o);
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < XS; j++) {
      const int c = (j * kGvWaves + wave) * 64 + lane;
      xr[j] = (j < nstep && c < nch) ? reinterpret_cast<const uint4*>(xg)[c] : make_uint4(0, 0, 0, 0);
    }
  }

  T* yo = reinterpret_cast<T*>(a.y);
  bool first = true;
  for (int r0 = row_first; r0 < N; r0 += gridDim.x * RB) {
    float acc[RB], acc3[RB];
#pragma unroll
    for (int r = 0; r < RB; r++) acc[r] = acc3[r] = 0.f;
#pragma unroll
    for (int s0 = 0; s0 < XS; s0 += CU) {
      if (s0 < nstep) {
        if (!first) issue(r0, s0);
        first = false;
#pragma unroll
        for (int u = 0; u < CU; u++) {
          const uint4 xv = xr[s0 + u < XS ? s0 + u : XS - 1];  // beyond nstep the weights are zero-filled
#pragma unroll
          for (int r = 0; r < RB; r++) {
            acc[r] = Dot16<T>::run(w[r][u], xv, acc[r]);
            if (SWIGLU) acc3[r] = Dot16<T>::run(w3[r][u], xv, acc3[r]);
          }
        }
      }
    }
    // ---- the four K-quarters of every row meet in LDS (fixed order: deterministic)
#pragma unroll
    for (int r = 0; r < RB; r++) {
      const float s = gv_wave_sum(acc[r]);
      float s3 = 0.f;
      if (SWIGLU) s3 = gv_wave_sum(acc3[r]);
      if (lane == 0) {
        sm_part[wave][0][r] = s;
        if (SWIGLU) sm_part[wave][1][r] = s3;
      }
    }
    __syncthreads();
    if (wave == 0) {
      float out = 0.f;  // lane r finishes row r0 + r
      const int r = lane < RB ? lane : 0;
      float s = (sm_part[0][0][r] + sm_part[1][0][r]) + (sm_part[2][0][r] + sm_part[3][0][r]);
      const int row = r0 + r;
      if (a.bias != nullptr && row < N) s += ElemTraits<T>::load(reinterpret_cast<const T*>(a.bias), (size_t)row);
      // (cc_opaque_f32: the fp32 value first, THEN its rounding — otherwise the f16 instantiations may fuse the bias add / the
      //  rotation with the conversion into one v_fma_mix rounding, differently in different kernels: the QKV form of the layer step,
      //  cc_attn_decode_qkv.hip, runs these operations too and must give the same bits)
      s = ElemTraits<T>::rnd(cc_opaque_f32(s));  // the Linear's output in the model dtype
      if (SWIGLU) {
        const float s3 = ElemTraits<T>::rnd((sm_part[0][1][r] + sm_part[1][1][r]) + (sm_part[2][1][r] + sm_part[3][1][r]));
        const float sl = ElemTraits<T>::rnd(__fdiv_rn(s, 1.0f + expf(-s)));  // F.silu -> dtype (model.py:443)
        s = __fmul_rn(sl, s3);
      }
      out = s;
      if (a.freqs != nullptr) {  // RoPE on the (even, odd) row pairs of the q / k heads (model.py:507-519)
        const float other = gv_dpp<0xB1>(out);  // the pair partner lives in lane ^ 1 (r0 is even)
        if (row < a.rope_rows) {
          const int pr = (row % a.head_dim) >> 1;
          const float c = ElemTraits<T>::load(reinterpret_cast<const T*>(a.freqs), (size_t)pr * 2);
          const float sn = ElemTraits<T>::load(reinterpret_cast<const T*>(a.freqs), (size_t)pr * 2 + 1);
          out = (row & 1) ? __fadd_rn(__fmul_rn(out, c), __fmul_rn(other, sn)) : __fsub_rn(__fmul_rn(out, c), __fmul_rn(other, sn));
        }
      }
      if (lane < RB && row < N) ElemTraits<T>::store(yo, (size_t)row, cc_opaque_f32(out));
    }
    __syncthreads();  // sm_part is reused by the next row group
  }
}

struct GvCfg {
  int rb, cu, cap;
};

static GvCfg pick_cfg(const GemvArgs& a, int vec) {
  static int env_rb = -1, env_cu = -1, env_cap = 2048;
  if (env_rb < 0) {  // tuning hook: CC_GEMV_CFG="RB,CU[,max workgroups]"
    env_rb = env_cu = 0;
    if (const char* e = getenv("CC_GEMV_CFG")) sscanf(e, "%d,%d,%d", &env_rb, &env_cu, &env_cap);
  }
  if (env_rb > 0 && env_cu > 0) return {env_rb, env_cu, env_cap};
  const int nseg = (a.K / vec + 63) / 64, nstep = (nseg + kGvWaves - 1) / kGvWaves;
  // measured on MI355X (tools/bench_gemv.py): 2 rows x 2 segments per wave for the SwiGLU pair at K = 4096 (w1 + w3 39.5 us =
  // 5.95 TB/s), 4 x 4 for K = 14336 (w2 20.9 us)
  GvCfg c;
  c.cap = 2048;
  if (nstep <= 2 && a.W3 == nullptr) {
    // r5 (same-box sweeps, tools/bench_gemv.py with CC_GEMV_CFG, two repetitions each): 4 rows per workgroup and 1024 workgroups — 8
    // loads in flight per lane, four workgroups per CU — instead of 2 rows and 2048: wqkv with its norm prologue and RoPE epilogue
    // 12.25 -> 11.0 us, wo 8.14 -> 7.86 (2048 / 1536 / 1280 / 896 / 768 / 640 / 512 workgroups: 12.4 / 12.4 / 12.1 / 11.5 / 11.6 /
    // 12.0 / 12.7; 8 rows: 11.8-11.9); the LM head (128256 rows: 125 rounds per workgroup) prefers 2 rows x 1024: 163.4 -> 157.7.
    // The arithmetic of a row does not depend on either number: bit-identical results.  (Also tried, r5: the SwiGLU pair as one wave
    // per row pair — 256 workgroups x 8 waves, no LDS, no barrier, the geometry at which a stream-only launch moves the pair's 235 MB
    // in 37.1 us: 41.1-41.4 us against this kernel's 40.0; removed.)
    if (a.N >= 32768) {
      c.rb = 2; c.cu = 2;
    } else {
      c.rb = 4; c.cu = 2;
    }
    c.cap = 1024;
  } else if (nstep <= 2) {
    c.rb = 2; c.cu = 2;
  } else if (a.W3 != nullptr) {
    c.rb = 2; c.cu = (nstep <= 4 || nstep > 8) ? 4 : 8;
  } else {
    c.rb = 4; c.cu = 4;
    c.cap = 512;  // long rows: two workgroups per CU looping over row groups (w2: 21.1 vs 21.8 us at 1024+)
  }
  return c;
}

template <typename T, bool SWIGLU, int RB, int CU, int XS>
static void launch_cfg(const GemvArgs& a, hipStream_t st, int cap) {
  int blocks = (a.N + RB - 1) / RB;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL((gemv_kernel<T, SWIGLU, RB, CU, XS>), dim3(blocks), dim3(kGvThreads), 0, st, a);
}

template <typename T>
static int launch_gemv(const GemvArgs& a, hipStream_t st) {
  const int vec = 16 / (int)sizeof(T);
  const GvCfg c = pick_cfg(a, vec);
  const int nseg = (a.K / vec + 63) / 64, nstep = (nseg + kGvWaves - 1) / kGvWaves;
  const int xs = nstep <= 2 ? 2 : nstep <= 8 ? 8 : 16;  // input chunks kept per lane
  if (nstep > 16 || c.cu > xs) return CC_ERR_UNSUPPORTED;
  const int key = (a.W3 ? 100000 : 0) + xs * 1000 + c.rb * 10 + c.cu;
  switch (key) {
    case 2022: launch_cfg<T, false, 2, 2, 2>(a, st, c.cap); break;
    case 2042: launch_cfg<T, false, 4, 2, 2>(a, st, c.cap); break;
    case 2082: launch_cfg<T, false, 8, 2, 2>(a, st, c.cap); break;
    case 8022: launch_cfg<T, false, 2, 2, 8>(a, st, c.cap); break;
    case 8024: launch_cfg<T, false, 2, 4, 8>(a, st, c.cap); break;
    case 8044: launch_cfg<T, false, 4, 4, 8>(a, st, c.cap); break;
    case 8028: launch_cfg<T, false, 2, 8, 8>(a, st, c.cap); break;
    case 8048: launch_cfg<T, false, 4, 8, 8>(a, st, c.cap); break;
    case 16044: launch_cfg<T, false, 4, 4, 16>(a, st, c.cap); break;
    case 16028: launch_cfg<T, false, 2, 8, 16>(a, st, c.cap); break;
    case 102022: launch_cfg<T, true, 2, 2, 2>(a, st, c.cap); break;
    case 102042: launch_cfg<T, true, 4, 2, 2>(a, st, c.cap); break;
    case 108022: launch_cfg<T, true, 2, 2, 8>(a, st, c.cap); break;
    case 108024: launch_cfg<T, true, 2, 4, 8>(a, st, c.cap); break;
    case 108028: launch_cfg<T, true, 2, 8, 8>(a, st, c.cap); break;
    case 116024: launch_cfg<T, true, 2, 4, 16>(a, st, c.cap); break;
    default: return CC_ERR_UNSUPPORTED;
  }
  CC_LAUNCH_CHECK();
  return CC_OK;
}

}  // namespace

extern "C" int cc_gemv_fused(const void* W, const void* W3, const void* x, const void* delta, const void* norm_w, float eps,
                             void* h_out, const void* bias, const void* freqs, int32_t rope_rows, int32_t head_dim, void* y,
                             int32_t N, int32_t K, int32_t dtype, cc_stream_t stream) {
  CC_ENTRY();
  if (!W || !x || !y || N <= 0 || K <= 0 || !cc_dt_ok(dtype)) return CC_ERR_BAD_ARG;
  if ((delta || h_out) && !norm_w) return CC_ERR_BAD_ARG;
  if (freqs && (W3 || rope_rows < 0 || rope_rows > N || head_dim <= 0 || (head_dim & 1) || (rope_rows % head_dim))) return CC_ERR_BAD_ARG;
  if (W3 && bias) return CC_ERR_BAD_ARG;
  const int vec = 16 / (int)cc_dt_size(dtype);
  if (K % vec) return CC_ERR_UNSUPPORTED;
  if ((size_t)K * cc_dt_size(dtype) > 64 * 1024) return CC_ERR_UNSUPPORTED;  // 16 input chunks per lane at most
  GemvArgs a{W, W3, x, delta, norm_w, bias, freqs, h_out, y, eps, N, K, freqs ? rope_rows : 0, freqs ? head_dim : 2};
  hipStream_t st = (hipStream_t)stream;
  switch (dtype) {
    case CC_DT_F32: return launch_gemv<float>(a, st);
    case CC_DT_BF16: return launch_gemv<bf16_t>(a, st);
    default: return launch_gemv<f16_t>(a, st);
  }
}
