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
#include "idf_gnfold.h"
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
    uint32_t lo = idf_pack_bf16(out[c][0] * alpha, out[c][1] * alpha);
    uint32_t hi = idf_pack_bf16(out[c][2] * alpha, out[c][3] * alpha);
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


// sum over the 16 lanes of a DPP row (every lane of the row ends with the total)
__device__ __forceinline__ float row16_sum_ab(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
  return v;
}

// ---------------------------------------------------------------- the whole attention block in ONE launch
// modules.py:145-164 for the N = 256-token level at C = 128:  y = x + softmax(q k^T C^-1/2) v',  q | k | v' = conv1x1(GroupNorm(x)) with
// the proj conv folded into V (Wv' = Wp Wv, b' = Wp bv + bp: idf_attn_fold_batched).
// One 8-wave workgroup per image, K and V of the image in LDS (as attn_fwd_kernel keeps them), nothing but x read:
//   fold   GroupNorm coefficients (sc, sh) per channel from the statistics partials x's producer left behind
//   K | V  eight slabs of 32 pixels: h = x * sc + sh staged as bf16 (four 32-channel chunk images, pixel pitch 96 B: conflict-free
//          ds_read_b128), each wave owns 64 of the 256 K | V couts with its 16 weight fragments in registers (fragment-major
//          shadow: 1 KB per wave instruction), results + bias rounded to bf16 straight into the K / V tiles
//   Q      per 16-row block of a wave, from x's rows (registers) and the 32 q fragments in registers; the MFMA result leaves a
//          lane holding 4 consecutive channels of ITS row per 16-channel tile -- which IS a B operand of the score product once
//          the contraction's channel order is permuted the same way on the K side (two ds_read_b64 instead of one b128)
//   P V    scores, softmax and the output product exactly as attn_fwd_kernel (same functions)
//   y      = x + O in the output product's registers (O rounded to bf16 first: the tensor the backward pass reads), per-channel
//          statistics of y for the next GroupNorm
// Training additionally stores what the backward pass reads (q | k | v, h for the weight gradient, O, the row logsumexp, the
// GroupNorm's mean / rstd / sc / sh): the data-gradient side stays the existing launches.
struct AbP {
  const bf16_t* x; const float* st; int T;
  const float* gamma; const float* beta; float eps;
  const bf16_t* wqkv; const float* bqkv;
  bf16_t* y; float* st_out;
  bf16_t* qkv; bf16_t* h; bf16_t* o; float* lse; float* mean; float* rstd; float* sc; float* sh;
  float scale;
};
constexpr int AB_SLAB = 32, AB_PPB = 96;
constexpr size_t AB_LDS = (size_t)2 * 256 * ACfg<128>::PITCH * sizeof(bf16_t) + (size_t)4 * AB_SLAB * AB_PPB + 128 * 2 * sizeof(float);

__device__ __forceinline__ void ab_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ s16x4_t ab_read8(const bf16_t* p) {
  return *reinterpret_cast<const __attribute__((address_space(3))) s16x4_t*>(
      (const __attribute__((address_space(3))) unsigned char*)(p));
}
__device__ __forceinline__ void ab_unpack(const uint4& r, float* o) {
  const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) { o[2 * i] = __uint_as_float(w[i] << 16); o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
}
__device__ __forceinline__ uint32_t ab_pack2(float lo, float hi) {
  return idf_pack_bf16(lo, hi);
}
// the B fragments of a product contracting over channels, from accumulator tiles that hold channels 16 c + 4 (lane >> 4) + r:
// k slot (g, e) of step s = channel 32 s + 4 g + e (e < 4) / 32 s + 16 + 4 g + e - 4
__device__ __forceinline__ void ab_frags(const f32x4_t (&t)[8], const float4* bias, bf16x8_t (&f)[4]) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    bf16x8_t v;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float b0 = bias ? (&bias[2 * s].x)[r] : 0.f, b1 = bias ? (&bias[2 * s + 1].x)[r] : 0.f;
      v[r] = (__bf16)(t[2 * s][r] + b0); v[4 + r] = (__bf16)(t[2 * s + 1][r] + b1);
    }
    f[s] = v;
  }
}

// NW waves per workgroup (4 or 8): each owns 16 / NW of the 16 K | V cout fragments and 16 / NW of the 16 query-row blocks
template <int NW>
__global__ __launch_bounds__(64 * NW) void attnblock_fwd_kernel(const AbP p) {
  constexpr int D = 128, AN = 256, PITCH = ACfg<D>::PITCH;
  constexpr int NT = 64 * NW, KVF = 16 / NW, RBW = 16 / NW, SVT = 512 / NT;     // threads; K | V fragments, row blocks per wave; staged vectors per thread
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);
  bf16_t* Vs = Ks + AN * PITCH;
  unsigned char* slab = reinterpret_cast<unsigned char*>(Vs + AN * PITCH);
  float* cof = reinterpret_cast<float*>(slab + 4 * AB_SLAB * AB_PPB);      // [128][2] (sc, sh)
  float* scr = reinterpret_cast<float*>(slab);                             // fold scratch [128][2]; at the end statistics [NW][128][2]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r16 = lane & 15;
  const int b = blockIdx.x;
  const bf16_t* xb = p.x + (size_t)b * AN * D;
  const bool train = p.qkv != nullptr;

  // ---- GroupNorm coefficients (the arithmetic of idf_conv3x3.hip's pro_coefficients: same partials, same order, same bits)
  if (tid < D) {
    const float2 S = idf_sum_partials(reinterpret_cast<const float2*>(p.st) + (size_t)b * p.T * D + tid, p.T, (size_t)D);
    scr[2 * tid] = S.x; scr[2 * tid + 1] = S.y;
  }
  __syncthreads();
  if (tid < D) {
    const int c = tid, g0 = c & ~3;
    double a = 0.0, d = 0.0;
    for (int k = g0; k < g0 + 4; ++k) { a += scr[2 * k]; d += scr[2 * k + 1]; }
    float r, mf;
    idf_group_stats(a, d, 1.0 / ((double)AN * 4.0), p.eps, &mf, &r);
    const float ga = p.gamma ? p.gamma[c] : 1.f, be = p.beta ? p.beta[c] : 0.f;
    const float sc = r * ga, sh = be - mf * sc;
    cof[2 * c] = sc; cof[2 * c + 1] = sh;
    if (p.sc) {
      p.sc[(size_t)b * D + c] = sc; p.sh[(size_t)b * D + c] = sh;
      if (c == g0) { p.mean[b * 32 + (c >> 2)] = mf; p.rstd[b * 32 + (c >> 2)] = r; }
    }
  }
  __syncthreads();

  // ---- K | V: slabs of 32 pixels
  {
    const int sv = tid & 15, spx = tid >> 4;       // this thread stages channel vector sv of pixels spx (+ 16 with four waves) of a slab
    float scv[8], shv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { scv[e] = cof[2 * (sv * 8 + e)]; shv[e] = cof[2 * (sv * 8 + e) + 1]; }
    const int soff = (sv >> 2) * (AB_SLAB * AB_PPB) + (sv & 3) * 16;
    uint4 xr[SVT];
#pragma unroll
    for (int j = 0; j < SVT; ++j) xr[j] = *reinterpret_cast<const uint4*>(xb + (size_t)(spx + 16 * j) * D + sv * 8);
    bf16x8_t wkv[KVF][4];
    float4 bkv[KVF];
#pragma unroll
    for (int j = 0; j < KVF; ++j) {
      const int nf = 8 + KVF * wave + j;
#pragma unroll
      for (int s = 0; s < 4; ++s)
        wkv[j][s] = *reinterpret_cast<const bf16x8_t*>(p.wqkv + ((size_t)((((s >> 1) * 24 + nf) * 2 + (s & 1)) * 64 + lane)) * 8);
      bkv[j] = *reinterpret_cast<const float4*>(p.bqkv + 16 * nf + 4 * g);
    }
    bf16_t* dstT = wave < NW / 2 ? Ks : Vs;
    const int part = 1 + wave / (NW / 2);          // 1: k, 2: v (the q | k | v tensor's channel third)
#pragma unroll 1
    for (int i = 0; i < AN / AB_SLAB; ++i) {
#pragma unroll
      for (int j = 0; j < SVT; ++j) {
        float f[8];
        ab_unpack(xr[j], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = f[e] * scv[e] + shv[e];
        const uint4 hv = make_uint4(ab_pack2(f[0], f[1]), ab_pack2(f[2], f[3]), ab_pack2(f[4], f[5]), ab_pack2(f[6], f[7]));
        *reinterpret_cast<uint4*>(slab + soff + (spx + 16 * j) * AB_PPB) = hv;
        if (train) *reinterpret_cast<uint4*>(p.h + ((size_t)b * AN + i * AB_SLAB + spx + 16 * j) * D + sv * 8) = hv;
      }
      if (i + 1 < AN / AB_SLAB) {
#pragma unroll
        for (int j = 0; j < SVT; ++j)
          xr[j] = *reinterpret_cast<const uint4*>(xb + (size_t)((i + 1) * AB_SLAB + spx + 16 * j) * D + sv * 8);
      }
      ab_barrier();
      f32x4_t acc[KVF][2];
#pragma unroll
      for (int j = 0; j < KVF; ++j)
#pragma unroll
        for (int pf = 0; pf < 2; ++pf) acc[j][pf] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int pf = 0; pf < 2; ++pf)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const bf16x8_t hf = *reinterpret_cast<const bf16x8_t*>(slab + s * (AB_SLAB * AB_PPB) + (pf * 16 + r16) * AB_PPB + g * 16);
#pragma unroll
          for (int j = 0; j < KVF; ++j) acc[j][pf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wkv[j][s], hf, acc[j][pf], 0, 0, 0);
        }
#pragma unroll
      for (int j = 0; j < KVF; ++j)
#pragma unroll
        for (int pf = 0; pf < 2; ++pf) {
          const int row = i * AB_SLAB + pf * 16 + r16, col = 16 * (KVF * (wave % (NW / 2)) + j) + 4 * g;
          const uint2 u = make_uint2(ab_pack2(acc[j][pf][0] + bkv[j].x, acc[j][pf][1] + bkv[j].y),
                                     ab_pack2(acc[j][pf][2] + bkv[j].z, acc[j][pf][3] + bkv[j].w));
          *reinterpret_cast<uint2*>(dstT + row * PITCH + col) = u;
          if (train) *reinterpret_cast<uint2*>(p.qkv + ((size_t)b * AN + row) * 3 * D + part * D + col) = u;
        }
      ab_barrier();
    }
  }

  // ---- Q of this wave's 16-row blocks (row block wave + NW k)
  bf16x8_t qp[RBW][4];
  {
    bf16x8_t wq[8][4];
    float4 bq[8];
#pragma unroll
    for (int f = 0; f < 8; ++f) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
        wq[f][s] = *reinterpret_cast<const bf16x8_t*>(p.wqkv + ((size_t)((((s >> 1) * 24 + f) * 2 + (s & 1)) * 64 + lane)) * 8);
      bq[f] = *reinterpret_cast<const float4*>(p.bqkv + 16 * f + 4 * g);
    }
#pragma unroll
    for (int k = 0; k < RBW; ++k) {
      const int r0 = 16 * (wave + NW * k);
      bf16x8_t hf[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const uint4 raw = *reinterpret_cast<const uint4*>(xb + (size_t)(r0 + r16) * D + s * 32 + g * 8);
        float f[8];
        ab_unpack(raw, f);
        bf16x8_t v;
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          const float4 c4 = *reinterpret_cast<const float4*>(cof + 2 * (s * 32 + g * 8 + e));
          v[e] = (__bf16)(f[e] * c4.x + c4.y); v[e + 1] = (__bf16)(f[e + 1] * c4.z + c4.w);
        }
        hf[s] = v;
      }
      f32x4_t acc[8];
#pragma unroll
      for (int f = 0; f < 8; ++f) {
        acc[f] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[f][s], hf[s], acc[f], 0, 0, 0);
      }
      ab_frags(acc, bq, qp[k]);
      if (train) {
        bf16_t* qrow = p.qkv + ((size_t)b * AN + r0 + r16) * 3 * D + 4 * g;
#pragma unroll
        for (int f = 0; f < 8; ++f)
          *reinterpret_cast<uint2*>(qrow + 16 * f) = make_uint2(ab_pack2(acc[f][0] + bq[f].x, acc[f][1] + bq[f].y),
                                                                ab_pack2(acc[f][2] + bq[f].z, acc[f][3] + bq[f].w));
      }
    }
  }

  // ---- softmax(Q K^T) V' per row block, the residual and the statistics of y straight from the output product
  float s1[8][4], s2[8][4];
#pragma unroll
  for (int f = 0; f < 8; ++f)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[f][r] = 0.f; s2[f][r] = 0.f; }
#pragma unroll
  for (int k = 0; k < RBW; ++k) {
    const int r0 = 16 * (wave + NW * k);
    f32x4_t sc_[AN / 16];
    {
      const bf16_t* kb = Ks + r16 * PITCH + 4 * g;
#pragma unroll
      for (int t = 0; t < AN / 16; ++t) {
        sc_[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const bf16x8_t a = amk(ab_read8(kb + t * 16 * PITCH + s * 32), ab_read8(kb + t * 16 * PITCH + s * 32 + 16));
          sc_[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qp[k][s], sc_[t], 0, 0, 0);
        }
      }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < AN / 16; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) { sc_[t][r] *= p.scale; mx = fmaxf(mx, sc_[t][r]); }
    mx = quad_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < AN / 16; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) { sc_[t][r] = __expf(sc_[t][r] - mx); sum += sc_[t][r]; }
    sum = quad_sum(sum);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int t = 0; t < AN / 16; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) sc_[t][r] *= inv;
    bf16x8_t pk[AN / 32];
    pack_w<AN>(sc_, pk);
    f32x4_t out[ACfg<D>::CT];
    outprod<D, AN>(out, Vs, pk, lane);
    if (train) {
      store_out<D>(p.o + ((size_t)b * AN + r0) * D, D, out, lane, 1.0f);
      if (lane < 16) p.lse[(size_t)b * AN + r0 + lane] = mx + __logf(sum);
    }
    {
      // the proj conv is inside V (Wv' = Wp Wv, bias b' on V'): y = x + O, O rounded to bf16 first (the stored o)
      const size_t row = (size_t)b * AN + r0 + r16;
#pragma unroll
      for (int f = 0; f < 8; ++f) {
        const uint2 rs = *reinterpret_cast<const uint2*>(p.x + row * D + 16 * f + 4 * g);
        const float o0 = bf16_to_f32(f32_to_bf16(out[f][0])), o1 = bf16_to_f32(f32_to_bf16(out[f][1]));
        const float o2 = bf16_to_f32(f32_to_bf16(out[f][2])), o3 = bf16_to_f32(f32_to_bf16(out[f][3]));
        const uint2 u = make_uint2(ab_pack2(o0 + __uint_as_float(rs.x << 16), o1 + __uint_as_float(rs.x & 0xffff0000u)),
                                   ab_pack2(o2 + __uint_as_float(rs.y << 16), o3 + __uint_as_float(rs.y & 0xffff0000u)));
        *reinterpret_cast<uint2*>(p.y + row * D + 16 * f + 4 * g) = u;
        const float h0 = __uint_as_float(u.x << 16), h1 = __uint_as_float(u.x & 0xffff0000u);
        const float h2 = __uint_as_float(u.y << 16), h3 = __uint_as_float(u.y & 0xffff0000u);
        s1[f][0] += h0; s1[f][1] += h1; s1[f][2] += h2; s1[f][3] += h3;
        s2[f][0] += h0 * h0; s2[f][1] += h1 * h1; s2[f][2] += h2 * h2; s2[f][3] += h3 * h3;
      }
    }
  }
  if (p.st_out) {
#pragma unroll
    for (int f = 0; f < 8; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a = row16_sum_ab(s1[f][r]), q = row16_sum_ab(s2[f][r]);
        if (r16 == 0) { scr[(wave * D + 16 * f + 4 * g + r) * 2] = a; scr[(wave * D + 16 * f + 4 * g + r) * 2 + 1] = q; }
      }
    __syncthreads();
    if (tid < D) {
      float a = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) { a += scr[(w * D + tid) * 2]; q += scr[(w * D + tid) * 2 + 1]; }
      p.st_out[((size_t)b * D + tid) * 2] = a; p.st_out[((size_t)b * D + tid) * 2 + 1] = q;
    }
  }
}


// ---------------------------------------------------------------- the proj conv folded into V
// (P V) Wp^T = P (V Wp^T): with  Wv' = Wp Wv  and  b' = Wp bv + bp  (the rows of P sum to one, so a bias on V' passes through the
// softmax product unchanged) the block is  y = x + P V'  -- no proj launch, no proj data gradient, no proj weight gradient.
//   attn_fold_kernel      Wv', (bq | bk | b') of every attention block of a network, once per optimizer step
//   attn_fwd_res_kernel   attn_fwd_kernel + the residual and the statistics partials of y in its epilogue
//   attn_fold_bwd_a / _b  the chain rule back to the parameters, in place in the gradient arena, after the weight gradients ran:
//                         G = dL/dWv' and gb = dL/db' arrive in proj_v's slots;  dWp = G Wv^T + gb bv^T, dbp = gb,
//                         dWv = Wp^T G, dbv = Wp^T gb
struct FoldRow {
  const float* wp; const float* bp; const float* wv; const float* bv; const float* bq; const float* bk;
  float* wvf; float* bf;       // [C][C], [3 C]
  int C, pad;
};
// 16 x 16 output tiles, the contraction staged through LDS 16 columns at a time (128^3 products: latency, not throughput)
__global__ __launch_bounds__(256) void attn_fold_kernel(const FoldRow* __restrict__ tab) {
  __shared__ float As[16][17], Bs[16][17];
  const FoldRow d = tab[blockIdx.z];
  const int C = d.C, tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  if ((int)blockIdx.y * 16 >= C) {
    // the bias vector (bq | bk | Wp bv + bp): the extra tile row, block x = 0 only
    if (blockIdx.x != 0) return;
    for (int j = threadIdx.x; j < 3 * C; j += 256) {
      float v;
      if (j < C) v = d.bq[j];
      else if (j < 2 * C) v = d.bk[j - C];
      else {
        const int o = j - 2 * C;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int k = 0; k < C; k += 4) {
          a0 += d.wp[o * C + k] * d.bv[k]; a1 += d.wp[o * C + k + 1] * d.bv[k + 1];
          a2 += d.wp[o * C + k + 2] * d.bv[k + 2]; a3 += d.wp[o * C + k + 3] * d.bv[k + 3];
        }
        v = d.bp[o] + ((a0 + a1) + (a2 + a3));
      }
      d.bf[j] = v;
    }
    return;
  }
  const int o0 = blockIdx.y * 16, i0 = blockIdx.x * 16;
  if (o0 >= C || i0 >= C) return;
  float acc = 0.f;
  for (int k0 = 0; k0 < C; k0 += 16) {                 // Wv'[o][i] = sum_k Wp[o][k] Wv[k][i]
    As[ty][tx] = d.wp[(o0 + ty) * C + k0 + tx];
    Bs[ty][tx] = d.wv[(k0 + ty) * C + i0 + tx];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) acc += As[ty][k] * Bs[k][tx];
    __syncthreads();
  }
  d.wvf[(o0 + ty) * C + i0 + tx] = acc;
}

struct FoldBwdRow {
  float* g; float* gb;         // in: dL/dWv' [C][C], dL/db' [C] (proj_v's gradient slots);  out: dWv, dbv
  const float* wp; const float* wv; const float* bv;
  float* dwp; float* dbp;      // proj's gradient slots (written)
  float* gs;                   // scratch [C][C] + [C]: pass A leaves a copy of G and gb there, pass B reads it and writes g / gb
  int C, pad;
};
// pass A (reads G, gb; writes proj's slots and the scratch copy):  dWp[o][k] = sum_i G[o][i] Wv[k][i] + gb[o] bv[k],  dbp = gb
__global__ __launch_bounds__(256) void attn_fold_bwd_a_kernel(const FoldBwdRow* __restrict__ tab) {
  __shared__ float As[16][17], Bs[16][17];
  const FoldBwdRow d = tab[blockIdx.z];
  const int C = d.C, tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int o0 = blockIdx.y * 16, k0 = blockIdx.x * 16;
  if (o0 >= C || k0 >= C) return;
  float acc = 0.f;
  for (int i0 = 0; i0 < C; i0 += 16) {
    const float gv = d.g[(o0 + ty) * C + i0 + tx];
    As[ty][tx] = gv;
    if (blockIdx.x == 0) d.gs[(o0 + ty) * C + i0 + tx] = gv;      // the column-tile-0 blocks of a row tile cover G once
    Bs[ty][tx] = d.wv[(k0 + ty) * C + i0 + tx];       // Bs[k][i]
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += As[ty][i] * Bs[tx][i];
    __syncthreads();
  }
  const float gbo = d.gb[o0 + ty];
  d.dwp[(o0 + ty) * C + k0 + tx] = acc + gbo * d.bv[k0 + tx];
  if (blockIdx.x == 0 && tx == 0) { d.dbp[o0 + ty] = gbo; d.gs[C * C + o0 + ty] = gbo; }
}
// pass B (after pass A, from the scratch copy):  dWv[k][i] = sum_o Wp[o][k] G[o][i] into g,  dbv[k] = sum_o Wp[o][k] gb[o] into gb
__global__ __launch_bounds__(256) void attn_fold_bwd_b_kernel(const FoldBwdRow* __restrict__ tab) {
  __shared__ float As[16][17], Bs[16][17];
  const FoldBwdRow d = tab[blockIdx.z];
  const int C = d.C, tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int k0 = blockIdx.y * 16, i0 = blockIdx.x * 16;
  if (k0 >= C || i0 >= C) return;
  float acc = 0.f, accb = 0.f;
  for (int o0 = 0; o0 < C; o0 += 16) {
    As[ty][tx] = d.wp[(o0 + ty) * C + k0 + tx];       // As[o][k]
    Bs[ty][tx] = d.gs[(o0 + ty) * C + i0 + tx];       // Bs[o][i]
    __syncthreads();
#pragma unroll
    for (int o = 0; o < 16; ++o) acc += As[o][ty] * Bs[o][tx];
    if (blockIdx.x == 0 && tx == 0) {
#pragma unroll
      for (int o = 0; o < 16; ++o) accb += As[o][ty] * d.gs[C * C + o0 + o];
    }
    __syncthreads();
  }
  d.g[(k0 + ty) * C + i0 + tx] = acc;
  if (blockIdx.x == 0 && tx == 0) d.gb[k0 + ty] = accb;
}

template <int D, int AN>
__global__ __launch_bounds__(ANT) void attn_fwd_res_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ xres,
                                                           bf16_t* __restrict__ o, float* __restrict__ lse,
                                                           bf16_t* __restrict__ y, float* __restrict__ st_out, float scale,
                                                           int rb_per_block) {
  constexpr int PITCH = ACfg<D>::PITCH, CT = ACfg<D>::CT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);
  bf16_t* Vs = Ks + AN * PITCH;
  float* scr = reinterpret_cast<float*>(Vs + AN * PITCH);           // [4 waves][D][2]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, r16 = lane & 15;
  const int b = blockIdx.y;
  const bf16_t* base = qkv + (size_t)b * AN * 3 * D;
  stage_rows<D, AN>(base + D, 3 * D, Ks, tid);
  stage_rows<D, AN>(base + 2 * D, 3 * D, Vs, tid);
  __syncthreads();
  float s1[CT][4], s2[CT][4];
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[c][r] = 0.f; s2[c][r] = 0.f; }
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
    f32x4_t out[CT];
    outprod<D, AN>(out, Vs, pk, lane);
    if (o) store_out<D>(o + ((size_t)b * AN + r0) * D, D, out, lane, 1.0f);
    if (lse && lane < 16) lse[(size_t)b * AN + r0 + lane] = mx + __logf(sum);
    // y = x + O (O rounded to bf16 first, as the stored o is: what the backward pass differentiates); row lane & 15,
    // channels 16 c + 4 (lane >> 4) + r
    const size_t row = (size_t)b * AN + r0 + r16;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      const uint2 rs = *reinterpret_cast<const uint2*>(xres + row * D + 16 * c + 4 * g);
      const float o0 = bf16_to_f32(f32_to_bf16(out[c][0])), o1 = bf16_to_f32(f32_to_bf16(out[c][1]));
      const float o2 = bf16_to_f32(f32_to_bf16(out[c][2])), o3 = bf16_to_f32(f32_to_bf16(out[c][3]));
      const float v0 = o0 + __uint_as_float(rs.x << 16), v1 = o1 + __uint_as_float(rs.x & 0xffff0000u);
      const float v2 = o2 + __uint_as_float(rs.y << 16), v3 = o3 + __uint_as_float(rs.y & 0xffff0000u);
      const uint2 u = make_uint2(ab_pack2(v0, v1), ab_pack2(v2, v3));
      *reinterpret_cast<uint2*>(y + row * D + 16 * c + 4 * g) = u;
      const float h0 = __uint_as_float(u.x << 16), h1 = __uint_as_float(u.x & 0xffff0000u);
      const float h2 = __uint_as_float(u.y << 16), h3 = __uint_as_float(u.y & 0xffff0000u);
      s1[c][0] += h0; s1[c][1] += h1; s1[c][2] += h2; s1[c][3] += h3;
      s2[c][0] += h0 * h0; s2[c][1] += h1 * h1; s2[c][2] += h2 * h2; s2[c][3] += h3 * h3;
    }
  }
  if (st_out) {
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a = row16_sum_ab(s1[c][r]), q = row16_sum_ab(s2[c][r]);
        if (r16 == 0) { scr[(wave * D + 16 * c + 4 * g + r) * 2] = a; scr[(wave * D + 16 * c + 4 * g + r) * 2 + 1] = q; }
      }
    __syncthreads();
    if (tid < D) {
      float a = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { a += scr[(w * D + tid) * 2]; q += scr[(w * D + tid) * 2 + 1]; }
      float* dst = st_out + (((size_t)b * gridDim.x + blockIdx.x) * D + tid) * 2;
      dst[0] = a; dst[1] = q;
    }
  }
}

template <int D, int AN>
hipError_t launch_fwd_res(const void* qkv, const void* xres, void* o, float* lse, void* y, float* st_out, int B, float scale,
                          int tiles, hipStream_t st) {
  constexpr size_t lds = (size_t)2 * AN * ACfg<D>::PITCH * sizeof(bf16_t) + (size_t)4 * D * 2 * sizeof(float);
  static IdfLdsGrant grant;
  if (hipError_t e = idf_ensure_lds((const void*)attn_fwd_res_kernel<D, AN>, lds, grant); e != hipSuccess) return e;
  const int rb = (AN / 64) / tiles;
  hipLaunchKernelGGL((attn_fwd_res_kernel<D, AN>), dim3(tiles, B), dim3(ANT), lds, st, (const bf16_t*)qkv, (const bf16_t*)xres,
                     (bf16_t*)o, lse, (bf16_t*)y, st_out, scale, rb);
  return hipGetLastError();
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
  static const int rb_min_b = 128;
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

// ---- the attention block of the N = 256, C = 128 level in one launch (attnblock_fwd_kernel)
extern "C" int idf_attnblock_ok(int N, int C, int dtype) { return (N == 256 && C == 128 && dtype == IDF_BF16) ? 1 : 0; }

extern "C" int idf_attnblock_fwd(const void* x, const float* st, int T, const float* gamma, const float* beta, float eps,
                                 const void* wqkv_frag, const float* bqkv, void* y, float* st_out, void* qkv, void* h, void* o,
                                 float* lse, float* mean, float* rstd, float* sc, float* sh, float scale, int B, int N, int C,
                                 void* stream) {
  if (!idf_attnblock_ok(N, C, IDF_BF16)) IDF_FAIL(IDF_ERR_UNSUPPORTED, "attnblock_fwd: N=%d C=%d not covered", N, C);
  if (!x || !st || T < 1 || !wqkv_frag || !bqkv || !y) IDF_FAIL(IDF_ERR_BADARG, "attnblock_fwd: null argument");
  const bool any = qkv || h || o || lse || mean || rstd || sc || sh, all = qkv && h && o && lse && mean && rstd && sc && sh;
  if (any != all) IDF_FAIL(IDF_ERR_BADARG, "attnblock_fwd: the training outputs (qkv, h, o, lse, mean, rstd, sc, sh) go together");
  if (B == 0) return IDF_OK;
  // eight waves (two per SIMD): the four-wave form of the same kernel, one wave per SIMD with every LDS / memory latency exposed,
  // measured 302 - 307 against 307 - 312 img/s on DDIM-100 at B = 256 (profiles/r04_attn_block.txt)
  static IdfLdsGrant grant;
  if (hipError_t e = idf_ensure_lds((const void*)attnblock_fwd_kernel<8>, AB_LDS, grant); e != hipSuccess)
    IDF_FAIL(IDF_ERR_HIP, "attnblock_fwd: %d bytes of LDS refused: %s", (int)AB_LDS, hipGetErrorString(e));
  AbP p;
  p.x = (const bf16_t*)x; p.st = st; p.T = T; p.gamma = gamma; p.beta = beta; p.eps = eps;
  p.wqkv = (const bf16_t*)wqkv_frag; p.bqkv = bqkv;
  p.y = (bf16_t*)y; p.st_out = st_out; p.qkv = (bf16_t*)qkv; p.h = (bf16_t*)h; p.o = (bf16_t*)o; p.lse = lse;
  p.mean = mean; p.rstd = rstd; p.sc = sc; p.sh = sh; p.scale = scale;
  hipLaunchKernelGGL(attnblock_fwd_kernel<8>, dim3(B), dim3(512), AB_LDS, (hipStream_t)stream, p);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}

// ---- attention with the residual and the statistics of y in its epilogue (the proj conv folded into V: idf_attn_fold_batched)
// tiles: statistics tiles per image (st_out [B][tiles][D][2]); 0: shape not covered (the shapes of idf_attn_fused_ok)
extern "C" int idf_attn_res_tiles(int B, int N, int D, int dtype) {
  if (!idf_attn_fused_ok(N, D, dtype)) return 0;
  static const int rb_min_b = 128;
  return (N == 256 && B < rb_min_b) ? 4 : 1;
}

extern "C" int idf_attn_fwd_res(const void* qkv, const void* xres, void* o, float* lse, void* y, float* st_out, int B, int N,
                                int D, float scale, void* stream) {
  const int tiles = idf_attn_res_tiles(B, N, D, IDF_BF16);
  if (!tiles) IDF_FAIL(IDF_ERR_UNSUPPORTED, "attn_fwd_res: N=%d D=%d not covered", N, D);
  if (!qkv || !xres || !y) IDF_FAIL(IDF_ERR_BADARG, "attn_fwd_res: null argument");
  if ((o != nullptr) != (lse != nullptr)) IDF_FAIL(IDF_ERR_BADARG, "attn_fwd_res: o and lse go together");
  if (B == 0) return IDF_OK;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e;
  if (N == 256) e = D == 128 ? launch_fwd_res<128, 256>(qkv, xres, o, lse, y, st_out, B, scale, tiles, st)
                             : launch_fwd_res<64, 256>(qkv, xres, o, lse, y, st_out, B, scale, tiles, st);
  else e = D == 128 ? launch_fwd_res<128, 64>(qkv, xres, o, lse, y, st_out, B, scale, tiles, st)
                    : launch_fwd_res<64, 64>(qkv, xres, o, lse, y, st_out, B, scale, tiles, st);
  if (e != hipSuccess) IDF_FAIL(IDF_ERR_HIP, "attn_fwd_res: %s", hipGetErrorString(e));
  return IDF_OK;
}

// table (device): nrows x {wp, bp, wv, bv, bq, bk (const float*), wvf, bf (float*), int C, pad} (72 bytes): the folded V weights
// Wv' = Wp Wv [C][C] and the bias vector (bq | bk | Wp bv + bp) [3 C] of every attention block of a network
extern "C" int idf_attn_fold_batched(const void* table, int nrows, int max_C, void* stream) {
  if (nrows <= 0) return IDF_OK;
  if (!table || max_C <= 0) IDF_FAIL(IDF_ERR_BADARG, "attn_fold_batched: bad arguments");
  static_assert(sizeof(FoldRow) == 72, "the host builds 72-byte rows");
  if (max_C % 16) IDF_FAIL(IDF_ERR_UNSUPPORTED, "attn_fold_batched: C %% 16 != 0");
  hipLaunchKernelGGL(attn_fold_kernel, dim3((unsigned)(max_C / 16), (unsigned)(max_C / 16 + 1), (unsigned)nrows), dim3(256), 0,
                     (hipStream_t)stream, (const FoldRow*)table);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}

// table (device): nrows x {g, gb (float*, in place), wp, wv, bv (const float*), dwp, dbp, gs (float*: scratch [C*C + C]), int C, pad}
// (72 bytes)
extern "C" int idf_attn_fold_bwd_batched(const void* table, int nrows, int max_C, void* stream) {
  if (nrows <= 0) return IDF_OK;
  if (!table || max_C <= 0 || max_C > 512) IDF_FAIL(IDF_ERR_BADARG, "attn_fold_bwd_batched: bad arguments");
  static_assert(sizeof(FoldBwdRow) == 72, "the host builds 72-byte rows");
  if (max_C % 16) IDF_FAIL(IDF_ERR_UNSUPPORTED, "attn_fold_bwd_batched: C %% 16 != 0");
  hipLaunchKernelGGL(attn_fold_bwd_a_kernel, dim3((unsigned)(max_C / 16), (unsigned)(max_C / 16), (unsigned)nrows), dim3(256), 0,
                     (hipStream_t)stream, (const FoldBwdRow*)table);
  hipLaunchKernelGGL(attn_fold_bwd_b_kernel, dim3((unsigned)(max_C / 16), (unsigned)(max_C / 16), (unsigned)nrows), dim3(256), 0,
                     (hipStream_t)stream, (const FoldBwdRow*)table);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}
