// Fused single-head spatial self-attention (modules.py:129-164 of the reference AttnBlock) for the
// N = 256-token level (16x16) and the N = 64-token middle block (8x8), head dim D = C in {64, 128}, bf16 activations:
//   forward   O = softmax(Q K^T * scale) V            one launch  (was bmm, softmax, bmm)
//   backward  dQ, dK, dV                              two launches (was 4 bmm + softmax backward)
// qkv is the [B, N, 3D] output of the fused q/k/v 1x1 conv (token-major, q | k | v along channels).
//
// (below for N = 256; N = 64 is the same kernel with a quarter of the score tiles and one block per image)
// One block = 64 "row" tokens (16 per wave) against ALL 256 "column" tokens, whose two [256][D]
// operands sit in LDS for the whole kernel.  The score products take the LDS operand K-contiguous
// (ds_read_b128) and the row operand from registers; the MFMA result leaves each lane holding, for
// ONE row token (lane & 15), 64 of its 256 scores -- the other 192 are in the three lanes 16 / 32 /
// 48 apart, so a row softmax is a register reduction plus two shuffles, and the probabilities are
// ALREADY laid out as the B operand of the second product (k slot (g, j) = column 32s + 4g + j /
// 32s + 16 + 4g + j - 4), whose A operand comes from the same LDS tile through the gfx950
// transposed read ds_read_b64_tr_b16.  Scores, probabilities and dS never touch memory.
//
//   fwd   : LDS = K, V       rows = queries   P = softmax(S)            O  = P V      (+ row logsumexp)
//   bwd-A : LDS = K, V       rows = queries   dS = P o (dP - sum P dP)  dQ = dS K     (+ row sum P dP)
//   bwd-B : LDS = Q, dO      rows = keys      P^T, dS^T from the saved row statistics
//                                             dV = P^T dO, dK = dS^T Q
#include "idf_common.h"
#include <atomic>
#include <stdlib.h>

namespace {

constexpr int ANT = 256;     // threads: 4 waves x 16 rows

template <int D, int AN = 256> struct ACfg {
  static constexpr int PITCH = D + 16;          // elements; (2D + 32) bytes: tr reads conflict-free
  static constexpr int KS = D / 32;             // k-steps of a score product
  static constexpr int CT = D / 16;             // channel tiles of an output product
  static constexpr int NT16 = AN / 16;           // 16-column score tiles per row
  static constexpr int NS32 = AN / 32;           // 32-row k-steps of an output product
  static constexpr size_t LDS = (size_t)2 * AN * PITCH * sizeof(bf16_t) + 2 * AN * sizeof(float);
};

__device__ __forceinline__ s16x4_t atr_read(const bf16_t* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p));
}
__device__ __forceinline__ bf16x8_t amk(s16x4_t lo, s16x4_t hi) {
  union { struct { s16x4_t a, b; } s; bf16x8_t v; } u;
  u.s.a = lo; u.s.b = hi;
  return u.v;
}

// [256][D] rows of `src` (row pitch ld elements) -> LDS, pitch ACfg<D>::PITCH
template <int D, int AN>
__device__ __forceinline__ void stage_rows(const bf16_t* __restrict__ src, int ld, bf16_t* lds, int tid) {
  constexpr int VPR = D / 8, NV = AN * VPR / ANT;
  uint4 r[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    int idx = tid + k * ANT, row = idx / VPR, v = idx - row * VPR;
    r[k] = *reinterpret_cast<const uint4*>(src + (size_t)row * ld + v * 8);
  }
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    int idx = tid + k * ANT, row = idx / VPR, v = idx - row * VPR;
    *reinterpret_cast<uint4*>(lds + row * ACfg<D>::PITCH + v * 8) = r[k];
  }
}

// this lane's row operand: row (lane & 15) of `rows`, 8 channels (lane >> 4) * 8 of every k-step
template <int D>
__device__ __forceinline__ void load_rowfrag(const bf16_t* __restrict__ rows, int ld, int lane, bf16x8_t (&f)[ACfg<D>::KS]) {
  const bf16_t* p = rows + (size_t)(lane & 15) * ld + (lane >> 4) * 8;
#pragma unroll
  for (int s = 0; s < ACfg<D>::KS; ++s) f[s] = *reinterpret_cast<const bf16x8_t*>(p + s * 32);
}

// acc[t][r] = <X[16 t + 4 (lane >> 4) + r], row (lane & 15)>
template <int D, int AN>
__device__ __forceinline__ void scores(f32x4_t (&acc)[AN / 16], const bf16_t* X, const bf16x8_t (&f)[ACfg<D>::KS], int lane) {
  const bf16_t* xb = X + (lane & 15) * ACfg<D>::PITCH + (lane >> 4) * 8;
#pragma unroll
  for (int t = 0; t < AN / 16; ++t) {
    acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < ACfg<D>::KS; ++s) {
      bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(xb + t * 16 * ACfg<D>::PITCH + s * 32);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, f[s], acc[t], 0, 0, 0);
    }
  }
}

// 64 fp32 weights per lane (as `scores` lays them out) -> the 8 B-operand fragments of `outprod`
template <int AN>
__device__ __forceinline__ void pack_w(const f32x4_t (&w)[AN / 16], bf16x8_t (&pk)[AN / 32]) {
#pragma unroll
  for (int s = 0; s < AN / 32; ++s) {
    bf16x8_t v;
#pragma unroll
    for (int r = 0; r < 4; ++r) { v[r] = (__bf16)w[2 * s][r]; v[4 + r] = (__bf16)w[2 * s + 1][r]; }
    pk[s] = v;
  }
}

// out[c][r] = sum over the 256 LDS rows j of w[j] * X[j][16 c + 4 (lane >> 4) + r]   (for row lane & 15)
template <int D, int AN>
__device__ __forceinline__ void outprod(f32x4_t (&out)[ACfg<D>::CT], const bf16_t* X, const bf16x8_t (&pk)[AN / 32], int lane) {
  const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
#pragma unroll
  for (int c = 0; c < ACfg<D>::CT; ++c) out[c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < AN / 32; ++s) {
    const bf16_t* x0 = X + (32 * s + 4 * g + q) * ACfg<D>::PITCH + 4 * pp;
    const bf16_t* x1 = x0 + 16 * ACfg<D>::PITCH;
#pragma unroll
    for (int c = 0; c < ACfg<D>::CT; ++c) {
      bf16x8_t a = amk(atr_read(x0 + c * 16), atr_read(x1 + c * 16));
      out[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pk[s], out[c], 0, 0, 0);
    }
  }
}

// row (lane & 15) of dst gets channels 16 c + 4 (lane >> 4) + r
template <int D>
__device__ __forceinline__ void store_out(bf16_t* __restrict__ dst, int ld, const f32x4_t (&out)[ACfg<D>::CT], int lane,
                                          float alpha) {
  bf16_t* p = dst + (size_t)(lane & 15) * ld + (lane >> 4) * 4;
#pragma unroll
  for (int c = 0; c < ACfg<D>::CT; ++c) {
    uint32_t lo = (uint32_t)f32_to_bf16(out[c][0] * alpha) | ((uint32_t)f32_to_bf16(out[c][1] * alpha) << 16);
    uint32_t hi = (uint32_t)f32_to_bf16(out[c][2] * alpha) | ((uint32_t)f32_to_bf16(out[c][3] * alpha) << 16);
    *reinterpret_cast<uint2*>(p + c * 16) = make_uint2(lo, hi);
  }
}

__device__ __forceinline__ float quad_max(float v) {   // over the 4 lanes that share lane & 15
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float quad_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

// ------------------------------------------------------------------ forward
// rb_per_block: 64-row blocks one workgroup walks with K and V staged ONCE (big batches: B = 256 sampling launches 1024
// workgroups of which every four re-staged the same 128 KB; at B = 32 one row block per workgroup keeps 128 CUs busy)
template <int D, int AN>
__global__ __launch_bounds__(ANT) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o,
                                                       float* __restrict__ lse, float scale, int rb_per_block) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);
  bf16_t* Vs = Ks + AN * ACfg<D>::PITCH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const bf16_t* base = qkv + (size_t)b * AN * 3 * D;
  stage_rows<D, AN>(base + D, 3 * D, Ks, tid);
  stage_rows<D, AN>(base + 2 * D, 3 * D, Vs, tid);
  __syncthreads();
  for (int rb = blockIdx.x * rb_per_block; rb < (int)(blockIdx.x + 1) * rb_per_block; ++rb) {
  const int r0 = rb * 64 + wave * 16;
  bf16x8_t qf[ACfg<D>::KS];
  load_rowfrag<D>(base + (size_t)r0 * 3 * D, 3 * D, lane, qf);
  f32x4_t s[AN / 16];
  scores<D, AN>(s, Ks, qf, lane);
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < AN / 16; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s[t][r] *= scale; mx = fmaxf(mx, s[t][r]); }
  mx = quad_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < AN / 16; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s[t][r] = __expf(s[t][r] - mx); sum += s[t][r]; }
  sum = quad_sum(sum);
  const float inv = 1.0f / sum;
#pragma unroll
  for (int t = 0; t < AN / 16; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) s[t][r] *= inv;
  bf16x8_t pk[AN / 32];
  pack_w<AN>(s, pk);
  f32x4_t out[ACfg<D>::CT];
  outprod<D, AN>(out, Vs, pk, lane);
  store_out<D>(o + ((size_t)b * AN + r0) * D, D, out, lane, 1.0f);
  if (lse && lane < 16) lse[(size_t)b * AN + r0 + lane] = mx + __logf(sum);
  }
}

// ------------------------------------------------- backward A: dQ and the row sums  sum_j P dP
template <int D, int AN>
__device__ __forceinline__ void attn_bwd_q_body(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dO,
                                                bf16_t* __restrict__ dqkv, float* __restrict__ dsum, float scale,
                                                unsigned char* smem) {
  bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);
  bf16_t* Vs = Ks + AN * ACfg<D>::PITCH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y, r0 = blockIdx.x * 64 + wave * 16;
  const bf16_t* base = qkv + (size_t)b * AN * 3 * D;
  stage_rows<D, AN>(base + D, 3 * D, Ks, tid);
  stage_rows<D, AN>(base + 2 * D, 3 * D, Vs, tid);
  bf16x8_t qf[ACfg<D>::KS], gf[ACfg<D>::KS];
  load_rowfrag<D>(base + (size_t)r0 * 3 * D, 3 * D, lane, qf);
  load_rowfrag<D>(dO + ((size_t)b * AN + r0) * D, D, lane, gf);
  __syncthreads();
  f32x4_t p[AN / 16], dp[AN / 16];
  scores<D, AN>(p, Ks, qf, lane);
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < AN / 16; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) { p[t][r] *= scale; mx = fmaxf(mx, p[t][r]); }
  mx = quad_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < AN / 16; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) { p[t][r] = __expf(p[t][r] - mx); sum += p[t][r]; }
  sum = quad_sum(sum);
  const float inv = 1.0f / sum;
  scores<D, AN>(dp, Vs, gf, lane);
  float dot = 0.f;
#pragma unroll
  for (int t = 0; t < AN / 16; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) { p[t][r] *= inv; dot += p[t][r] * dp[t][r]; }
  dot = quad_sum(dot);
#pragma unroll
  for (int t = 0; t < AN / 16; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) p[t][r] = p[t][r] * (dp[t][r] - dot);
  bf16x8_t pk[AN / 32];
  pack_w<AN>(p, pk);
  f32x4_t out[ACfg<D>::CT];
  outprod<D, AN>(out, Ks, pk, lane);
  store_out<D>(dqkv + ((size_t)b * AN + r0) * 3 * D, 3 * D, out, lane, scale);
  if (dsum && lane < 16) dsum[(size_t)b * AN + r0 + lane] = dot;
}

template <int D, int AN>
__global__ __launch_bounds__(ANT) void attn_bwd_q_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dO,
                                                         bf16_t* __restrict__ dqkv, float* __restrict__ dsum, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  attn_bwd_q_body<D, AN>(qkv, dO, dqkv, dsum, scale, smem);
}

// ------------------------------------------------- backward B: dK and dV (rows = keys)
// o != null: the row sums D_i = sum_j P_ij dP_ij are formed here as dO_i . O_i (O = P V, so the two are the same number up
// to the bf16 rounding of O) instead of being read from the query launch -- the two halves of the backward then depend on
// nothing of each other and run as ONE launch (attn_bwd_kernel)
template <int D, int AN>
__device__ __forceinline__ void attn_bwd_kv_body(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dO,
                                                 const float* __restrict__ lse, const float* __restrict__ dsum,
                                                 const bf16_t* __restrict__ o, bf16_t* __restrict__ dqkv, float scale,
                                                 unsigned char* smem) {
  bf16_t* Qs = reinterpret_cast<bf16_t*>(smem);
  bf16_t* Gs = Qs + AN * ACfg<D>::PITCH;
  float* Ls = reinterpret_cast<float*>(Gs + AN * ACfg<D>::PITCH);   // [256] row logsumexp
  float* Ds = Ls + AN;                                               // [256] row sum P dP
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y, r0 = blockIdx.x * 64 + wave * 16;
  const bf16_t* base = qkv + (size_t)b * AN * 3 * D;
  stage_rows<D, AN>(base, 3 * D, Qs, tid);
  stage_rows<D, AN>(dO + (size_t)b * AN * D, D, Gs, tid);
  if (tid < AN) {
    Ls[tid] = lse[(size_t)b * AN + tid];
    if (o) {
      const uint4* orow = reinterpret_cast<const uint4*>(o + ((size_t)b * AN + tid) * D);
      const uint4* grow = reinterpret_cast<const uint4*>(dO + ((size_t)b * AN + tid) * D);
      float acc = 0.f;
#pragma unroll 4
      for (int v = 0; v < D / 8; ++v) {
        const uint4 a = orow[v], g = grow[v];
        const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, gw[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc += __uint_as_float(aw[i] << 16) * __uint_as_float(gw[i] << 16) +
                 __uint_as_float(aw[i] & 0xffff0000u) * __uint_as_float(gw[i] & 0xffff0000u);
      }
      Ds[tid] = acc;
    } else Ds[tid] = dsum[(size_t)b * AN + tid];
  }
  bf16x8_t kf[ACfg<D>::KS], vf[ACfg<D>::KS];
  load_rowfrag<D>(base + (size_t)r0 * 3 * D + D, 3 * D, lane, kf);
  load_rowfrag<D>(base + (size_t)r0 * 3 * D + 2 * D, 3 * D, lane, vf);
  __syncthreads();
  const int g4 = (lane >> 4) * 4;
  f32x4_t p[AN / 16], dp[AN / 16];
  scores<D, AN>(p, Qs, kf, lane);            // p[t][r]: query 16 t + g4 + r  x  key (lane & 15)
#pragma unroll
  for (int t = 0; t < AN / 16; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) p[t][r] = __expf(p[t][r] * scale - Ls[16 * t + g4 + r]);
  bf16x8_t pk[AN / 32];
  f32x4_t out[ACfg<D>::CT];
  pack_w<AN>(p, pk);
  outprod<D, AN>(out, Gs, pk, lane);         // dV = P^T dO
  store_out<D>(dqkv + ((size_t)b * AN + r0) * 3 * D + 2 * D, 3 * D, out, lane, 1.0f);
  scores<D, AN>(dp, Gs, vf, lane);           // dP^T
#pragma unroll
  for (int t = 0; t < AN / 16; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) p[t][r] = p[t][r] * (dp[t][r] - Ds[16 * t + g4 + r]);
  pack_w<AN>(p, pk);
  outprod<D, AN>(out, Qs, pk, lane);         // dK = dS^T Q
  store_out<D>(dqkv + ((size_t)b * AN + r0) * 3 * D + D, 3 * D, out, lane, scale);
}

template <int D, int AN>
__global__ __launch_bounds__(ANT) void attn_bwd_kv_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dO,
                                                          const float* __restrict__ lse, const float* __restrict__ dsum,
                                                          bf16_t* __restrict__ dqkv, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  attn_bwd_kv_body<D, AN>(qkv, dO, lse, dsum, nullptr, dqkv, scale, smem);
}

// both halves in one launch (grid.z = 2: query blocks | key-value blocks): at B = 32 each half is 128 workgroups on 256 CUs
template <int D, int AN>
__global__ __launch_bounds__(ANT) void attn_bwd_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dO,
                                                       const float* __restrict__ lse, const bf16_t* __restrict__ o,
                                                       bf16_t* __restrict__ dqkv, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (blockIdx.z == 0) attn_bwd_q_body<D, AN>(qkv, dO, dqkv, nullptr, scale, smem);
  else attn_bwd_kv_body<D, AN>(qkv, dO, lse, nullptr, o, dqkv, scale, smem);
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute: raised once per (kernel set, device), not per
// launch; a failure is reported to the caller (the launch would otherwise run without the LDS opt-in).
template <int D, int AN>
hipError_t raise_lds_once() {
  static std::atomic<unsigned> done{0};        // bit d: device d has the attribute (devices >= 32: set every time)
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 32 && (done.load(std::memory_order_relaxed) >> dev) & 1u) return hipSuccess;
  const int lds = (int)ACfg<D, AN>::LDS;
  e = hipFuncSetAttribute((const void*)attn_fwd_kernel<D, AN>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attn_bwd_q_kernel<D, AN>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attn_bwd_kv_kernel<D, AN>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attn_bwd_kernel<D, AN>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e == hipSuccess && dev < 32) done.fetch_or(1u << dev, std::memory_order_relaxed);
  return e;
}

template <int D, int AN>
hipError_t launch_fwd(const void* qkv, void* o, float* lse, int B, float scale, hipStream_t st) {
  if (hipError_t e = raise_lds_once<D, AN>(); e != hipSuccess) return e;
  const size_t lds = ACfg<D, AN>::LDS;
  static const int rb_min_b = getenv("IDF_ATTN_RB_MINB") ? atoi(getenv("IDF_ATTN_RB_MINB")) : 128;
  const int rb = (B >= rb_min_b) ? AN / 64 : 1;       // whole image per workgroup once the batch alone fills the chip
  hipLaunchKernelGGL((attn_fwd_kernel<D, AN>), dim3(AN / 64 / rb, B), dim3(ANT), lds, st, (const bf16_t*)qkv,
                     (bf16_t*)o, lse, scale, rb);
  return hipGetLastError();
}

template <int D, int AN>
hipError_t launch_bwd(const void* qkv, const void* dO, const float* lse, float* dsum, const void* o, void* dqkv, int B,
                      float scale, hipStream_t st) {
  if (hipError_t e = raise_lds_once<D, AN>(); e != hipSuccess) return e;
  const dim3 g(AN / 64, B);
  const size_t lds = ACfg<D, AN>::LDS;
  if (o) {
    hipLaunchKernelGGL((attn_bwd_kernel<D, AN>), dim3(AN / 64, B, 2), dim3(ANT), lds, st, (const bf16_t*)qkv, (const bf16_t*)dO,
                       lse, (const bf16_t*)o, (bf16_t*)dqkv, scale);
    return hipGetLastError();
  }
  hipLaunchKernelGGL((attn_bwd_q_kernel<D, AN>), g, dim3(ANT), lds, st, (const bf16_t*)qkv, (const bf16_t*)dO,
                     (bf16_t*)dqkv, dsum, scale);
  if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;       // a failure here is the q launch's, not the kv launch's
  hipLaunchKernelGGL((attn_bwd_kv_kernel<D, AN>), g, dim3(ANT), lds, st, (const bf16_t*)qkv, (const bf16_t*)dO,
                     lse, dsum, (bf16_t*)dqkv, scale);
  return hipGetLastError();
}

}  // namespace

extern "C" int idf_attn_fused_ok(int N, int D, int dtype) {
  return ((N == 256 || N == 64) && (D == 64 || D == 128) && dtype == IDF_BF16) ? 1 : 0;
}

extern "C" int idf_attn_fwd(const void* qkv, void* o, float* lse, int B, int N, int D, float scale, int dtype,
                            void* stream) {
  if (!idf_attn_fused_ok(N, D, dtype)) IDF_FAIL(IDF_ERR_UNSUPPORTED, "attn_fwd: N=%d D=%d dtype=%d not covered", N, D, dtype);
  if (B == 0) return IDF_OK;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e;
  if (N == 256) e = D == 128 ? launch_fwd<128, 256>(qkv, o, lse, B, scale, st) : launch_fwd<64, 256>(qkv, o, lse, B, scale, st);
  else e = D == 128 ? launch_fwd<128, 64>(qkv, o, lse, B, scale, st) : launch_fwd<64, 64>(qkv, o, lse, B, scale, st);
  if (e != hipSuccess) IDF_FAIL(IDF_ERR_HIP, "attn_fwd: %s", hipGetErrorString(e));
  return IDF_OK;
}

static int attn_bwd_impl(const void* qkv, const void* dO, const float* lse, float* dsum, const void* o, void* dqkv, int B, int N,
                         int D, float scale, int dtype, void* stream) {
  if (!idf_attn_fused_ok(N, D, dtype)) IDF_FAIL(IDF_ERR_UNSUPPORTED, "attn_bwd: N=%d D=%d dtype=%d not covered", N, D, dtype);
  if (!o && !dsum) IDF_FAIL(IDF_ERR_BADARG, "attn_bwd: the row-sum scratch is missing");
  if (B == 0) return IDF_OK;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e;
  if (N == 256) e = D == 128 ? launch_bwd<128, 256>(qkv, dO, lse, dsum, o, dqkv, B, scale, st) : launch_bwd<64, 256>(qkv, dO, lse, dsum, o, dqkv, B, scale, st);
  else e = D == 128 ? launch_bwd<128, 64>(qkv, dO, lse, dsum, o, dqkv, B, scale, st) : launch_bwd<64, 64>(qkv, dO, lse, dsum, o, dqkv, B, scale, st);
  if (e != hipSuccess) IDF_FAIL(IDF_ERR_HIP, "attn_bwd: %s (query / key-value launch)", hipGetErrorString(e));
  return IDF_OK;
}

extern "C" int idf_attn_bwd(const void* qkv, const void* dO, const float* lse, float* dsum, void* dqkv, int B, int N,
                            int D, float scale, int dtype, void* stream) {
  return attn_bwd_impl(qkv, dO, lse, dsum, nullptr, dqkv, B, N, D, scale, dtype, stream);
}

// the backward as ONE launch: with the forward's output o [B, N, D] the key-value half forms the row sums itself
extern "C" int idf_attn_bwd_o(const void* qkv, const void* dO, const float* lse, const void* o, void* dqkv, int B, int N,
                              int D, float scale, int dtype, void* stream) {
  if (!o) IDF_FAIL(IDF_ERR_BADARG, "attn_bwd_o: the forward output is missing");
  return attn_bwd_impl(qkv, dO, lse, nullptr, o, dqkv, B, N, D, scale, dtype, stream);
}
