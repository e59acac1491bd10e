// Image-resident ResBlock forward for the 8x8 maps (bf16, gfx950): ONE launch computes a whole
//   AuxResBlock / ResBlock    (modules.py:261-328, 206-258):  h1 = conv1(SiLU(GN1(x)));  h2 = conv2(drop(SiLU(FiLM(GN2(h1)))));
//                                                            y  = conv3(drop(SiLU(GN3(h2)))) + shortcut(x)
//   ResBlock_encoder          (modules.py:331-366):          h1 = conv1(SiLU(GN1(x)));  y = conv2(drop(SiLU(GN2(h1)))) + shortcut(x)
// with one 512-thread workgroup per IMAGE.  At 8x8 a whole image is 64 pixels: the workgroup owns every pixel of every
// channel, so each GroupNorm's statistics close inside the workgroup and the activated tensor of the next stage is written
// straight into the LDS halo image its conv reads -- no launch boundary, no statistics round trip through memory, no
// re-staging between the stages.  The per-op path needs one launch per stage (17 us each at B = 32: launch floor + one
// memory round trip + coefficient fold + epilogue) on 64 of 256 CUs; this kernel runs the stages back to back on 32 CUs
// at the pace of its MFMA / weight stream.
//
// LDS:  Abuf [Cin/32][100 halo pixels][64 B]   the activated conv input, 32-channel chunk major, halo-tile layout of
//                                              idf_conv3x3.hip (64-byte pixel rows, 16-byte slots XOR-swizzled); the border
//                                              pixels are zeroed once and stay zero (the reference pads the ACTIVATED tensor)
//       Wbuf [9][128][64 B]                    one 32-channel weight slab of all 128 couts; the next slab travels in
//                                              registers while the MFMAs run (register double buffering), across stage
//                                              boundaries too; the fp32 output tile Hbuf [64][132] aliases it between stages
//       cof / chs / part                       coefficients, channel sums, wave partials
// Arithmetic per element is the per-op path's (pro_vec: fold, SiLU, dropout keyed by the element's index in the dense
// activated tensor; statistics of the bf16-ROUNDED outputs; the shortcut rounded to bf16 before it joins): results agree
// with the per-op launches up to the summation order of the statistics.
#include "idf_common.h"
#include "idf_gnfold.h"
#include "../../include/infodiff_hip.h"
#include <type_traits>

namespace {

constexpr int CK = 32, NPX = 64, WH = 10, NPH = 100, BN = 128, NT = 512, PF = BN + 4;
constexpr int WSLAB = 9 * BN * 64;              // bytes of a weight slab
constexpr int WV = 9 * BN * 4 / NT;             // 16-byte weight vectors per thread per slab (9)
constexpr int MAXC = 256;

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

__device__ __forceinline__ int swz(int row, int q) { return q ^ (((row >> 2) & 1) << 1); }

// a = act(x * sc + sh) on one 16-byte vector (8 bf16 channels), dropout keyed by the vector's index in the dense activated
// tensor: idf_conv3x3.hip's pro_vec
__device__ __forceinline__ uint4 act_vec(const uint4 raw, const float (&scv)[8], const float (&shv)[8], bool drop, uint64_t seedv,
                                         uint32_t salt, uint32_t thr, float dscale, uint32_t vec) {
  const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
  float v[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
  const uint32_t h = drop ? idf_vec_hash(seedv, salt, vec) : 0u;
  idf_act_vec<8>(v, scv, shv, 2, drop, h, 0, thr, dscale);
  uint32_t o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = (uint32_t)f32_to_bf16(v[2 * i]) | ((uint32_t)f32_to_bf16(v[2 * i + 1]) << 16);
  return make_uint4(o[0], o[1], o[2], o[3]);
}

// (sc, sh) of channel c from the per-channel sums chs [C][2] (sum, sum of squares over the image's 64 pixels): the fold of
// idf_conv3x3.hip's pro_coefficients, same expressions in the same order
__device__ __forceinline__ void fold_channel(const float* chs, int c, int C, int b, const IdfResblockStage& s, float eps,
                                             float* cof) {
  const int cpg = C >> 5, g = c / cpg;
  double a = 0.0, d = 0.0;
  for (int k = g * cpg; k < (g + 1) * cpg; ++k) { a += chs[2 * k]; d += chs[2 * k + 1]; }
  const double n = (double)NPX * cpg;
  double mu = a / n, var = d / n - mu * mu;
  if (var < 0.0) var = 0.0;
  const float r = (float)(1.0 / sqrt(var + (double)eps)), mf = (float)mu;
  const float ga = s.gamma ? s.gamma[c] : 1.f, be = s.beta ? s.beta[c] : 0.f;
  float sc = r * ga, sh = be - mf * sc;
  if (s.film_t) { const float f = 1.f + s.film_t[(size_t)b * s.ld_t + c]; sc *= f; sh = sh * f + s.film_t[(size_t)b * s.ld_t + C + c]; }
  if (s.film_a) { const float f = 1.f + s.film_a[(size_t)b * s.ld_a + c]; sc *= f; sh = sh * f + s.film_a[(size_t)b * s.ld_a + C + c]; }
  cof[2 * c] = sc; cof[2 * c + 1] = sh;
  if (s.sc) {
    s.sc[(size_t)b * C + c] = sc; s.sh[(size_t)b * C + c] = sh;
    if (c == g * cpg) { s.mean[b * 32 + g] = mf; s.rstd[b * 32 + g] = r; }
  }
}

struct RbK { IdfResblockArgs a; uint32_t thr; float dscale; };

__global__ __launch_bounds__(NT) void resblock8_fwd_kernel(const RbK k_in) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const IdfResblockArgs& p = k_in.a;
  const bf16_t* const px = reinterpret_cast<const bf16_t*>(p.x);
  const bf16_t* const px2 = reinterpret_cast<const bf16_t*>(p.x2);
  const uint32_t thr = k_in.thr;
  const float dscale = k_in.dscale;
  const int Cin = p.Cin, nck1 = Cin / CK;
  unsigned char* const Abuf = smem;                                   // [nck1][NPH][64]
  unsigned char* const Wbuf = smem + (size_t)nck1 * NPH * 64;         // [9][BN][64]   (Hbuf aliases it)
  float* const Hbuf = reinterpret_cast<float*>(Wbuf);                 // [NPX][PF]
  float* const cof = reinterpret_cast<float*>(Wbuf + WSLAB);          // [MAXC][2]
  float* const chs = cof + 2 * MAXC;                                  // [MAXC][2]
  float* const part = chs + 2 * MAXC;                                 // [8][BN][2]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
  const int fr = lane & 15, fq = lane >> 4;
  const int wm0 = (wave & 1) * 32, wn0 = (wave >> 1) * 32;            // this wave's 32 pixels x 32 couts
  const bool drop_any = p.seed != nullptr;
  const uint64_t seedv = drop_any ? *p.seed : 0;

  int hbase[2], wbase[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pl = wm0 + i * 16 + fr;
    hbase[i] = (pl >> 3) * WH + (pl & 7);
    const int n = wn0 + i * 16 + fr;
    wbase[i] = n * 64 + swz(n, fq) * 16;
  }
  // weight slab plan: idx over [BN][9][4 slots]
  int wsrc[WV], wlds[WV];
#pragma unroll
  for (int k = 0; k < WV; ++k) {
    const int idx = tid + k * NT, ch = idx & 3, r = idx >> 2, tap = r % 9, n = r / 9;
    wsrc[k] = (n * 9 + tap) * 4 + ch;                      // (row, slot): element offset = row * Cin_stage + chunk * 32 + slot * 8
    wlds[k] = (tap * BN + n) * 64 + swz(n, ch) * 16;
  }
  // (builtin vectors, not HIP's uint4 struct: its assignment from memory is a memcpy the optimizer does not split, which
  // leaves the array in scratch)
  u32x4_t wreg[WV];
#define load_w(wptr, cin_, ck_)                                                                                          \
  do {                                                                                                                   \
    _Pragma("unroll") for (int k_ = 0; k_ < WV; ++k_)                                                                    \
      wreg[k_] = *reinterpret_cast<const u32x4_t*>((wptr) + (size_t)(wsrc[k_] >> 2) * (cin_) + (ck_) * CK + (wsrc[k_] & 3) * 8); \
  } while (0)
#define store_w()                                                                                                        \
  do {                                                                                                                   \
    _Pragma("unroll") for (int k_ = 0; k_ < WV; ++k_) *reinterpret_cast<u32x4_t*>(Wbuf + wlds[k_]) = wreg[k_];             \
  } while (0)

  // ---- stage 0: the block input.  Its vectors and the statistics partials are fetched together (one round trip).
  // element-wise thread map over a 64-pixel x C-channel tensor: vector v = tid + k * NT -> pixel v / (C / 8), slot v % (C / 8)
  const int vpp1 = Cin / 8;                                  // vectors per pixel of the input
  const int nv1 = NPX * vpp1 / NT;                           // vectors per thread: 2 (128 channels) or 4 (256)
  uint4 xraw[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    xraw[k] = make_uint4(0, 0, 0, 0);
    if (k < nv1) {
      const int v = tid + k * NT, pl = v / vpp1, c = (v - pl * vpp1) * 8;
      const bf16_t* src = (px2 && c >= p.C1) ? px2 + ((size_t)(b * NPX + pl) * (Cin - p.C1) + (c - p.C1))
                                             : px + ((size_t)(b * NPX + pl) * (px2 ? p.C1 : Cin) + c);
      xraw[k] = *reinterpret_cast<const uint4*>(src);
    }
  }
  for (int c = tid; c < Cin; c += NT) {
    const float* st = p.st1;
    int T = p.T1, Cs = p.x2 ? p.C1 : Cin, cl = c;
    if (p.x2 && c >= p.C1) { st = p.st2; T = p.T2; Cs = Cin - p.C1; cl = c - p.C1; }
    const float2 S = idf_sum_partials(reinterpret_cast<const float2*>(st) + (size_t)b * T * Cs + cl, T, (size_t)Cs);
    chs[2 * c] = S.x; chs[2 * c + 1] = S.y;
  }
  // zero the halo border of every chunk image (36 border pixels x 4 slots per chunk)
  for (int i = tid; i < nck1 * 36 * 4; i += NT) {
    const int ck = i / 144, r = i - ck * 144, bp = r >> 2, q = r & 3;
    // border pixel bp: top row 0..9, bottom row 10..19, left column rows 1..8 (20..27), right column (28..35)
    int hy, hx;
    if (bp < 10) { hy = 0; hx = bp; } else if (bp < 20) { hy = 9; hx = bp - 10; }
    else if (bp < 28) { hy = bp - 19; hx = 0; } else { hy = bp - 27; hx = 9; }
    const int h = hy * WH + hx;
    *reinterpret_cast<uint4*>(Abuf + (size_t)ck * NPH * 64 + h * 64 + swz(h, q) * 16) = make_uint4(0, 0, 0, 0);
  }
  __syncthreads();
  for (int c = tid; c < Cin; c += NT) fold_channel(chs, c, Cin, b, p.s[0], p.eps, cof);

  f32x4_t sacc[2][2];        // the 1x1 shortcut, MFMA layout
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 2; ++i) sacc[a][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  auto a_slot = [&](int pl, int c) -> unsigned char* {      // LDS home of the 8 channels c.. of pixel pl
    const int h = ((pl >> 3) + 1) * WH + (pl & 7) + 1;
    return Abuf + (size_t)(c >> 5) * NPH * 64 + h * 64 + swz(h, (c & 31) >> 3) * 16;
  };
  if (p.w_sc) {
    // raw input -> Abuf, the whole [128][Cin] shortcut weight -> Wbuf ([chunk][n][64 B]), centre-tap MFMAs
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < nv1) {
        const int v = tid + k * NT, pl = v / vpp1, c = (v - pl * vpp1) * 8;
        *reinterpret_cast<uint4*>(a_slot(pl, c)) = xraw[k];
      }
    for (int i = tid; i < BN * nck1 * 4; i += NT) {
      const int q = i & 3, r = i >> 2, ck = r % nck1, n = r / nck1;
      *reinterpret_cast<uint4*>(Wbuf + ((size_t)ck * BN + n) * 64 + swz(n, q) * 16) =
          *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(p.w_sc) + (size_t)n * Cin + ck * CK + q * 8);
    }
    __syncthreads();
    for (int ck = 0; ck < nck1; ++ck) {
      bf16x8_t wf[2], xf[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) wf[a] = *reinterpret_cast<const bf16x8_t*>(Wbuf + (size_t)ck * BN * 64 + wbase[a]);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int h = hbase[i] + WH + 1;
        xf[i] = *reinterpret_cast<const bf16x8_t*>(Abuf + (size_t)ck * NPH * 64 + h * 64 + swz(h, fq) * 16);
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 2; ++i) sacc[a][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[a], xf[i], sacc[a][i], 0, 0, 0);
    }
    __syncthreads();         // Abuf / Wbuf free again (and cof complete)
  } else {
    __syncthreads();         // cof complete
  }
  load_w(reinterpret_cast<const bf16_t*>(p.s[0].w), Cin, 0);  // first slab of conv1 flies during the activation
  {
    const IdfResblockStage& s = p.s[0];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < nv1) {
        const int v = tid + k * NT, pl = v / vpp1, c = (v - pl * vpp1) * 8;
        float scv[8], shv[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 t4 = *reinterpret_cast<const float4*>(cof + 2 * c + 4 * q);
          scv[2 * q] = t4.x; shv[2 * q] = t4.y; scv[2 * q + 1] = t4.z; shv[2 * q + 1] = t4.w;
        }
        const unsigned e0 = (unsigned)((b * NPX + pl) * Cin + c);
        const uint4 o = act_vec(xraw[k], scv, shv, s.drop && drop_any, seedv, s.salt, thr, dscale, e0 >> 3);
        *reinterpret_cast<uint4*>(a_slot(pl, c)) = o;
        if (s.a_out) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(s.a_out) + e0) = o;
      }
  }

  // ---- the conv stages
  const int cc = (tid & 15) * 8;                             // this thread's 8 couts in the element-wise passes (fixed)
  // (the stage index is a compile-time constant in each copy of the body: p.s[] is then read from the kernel arguments,
  // not from a scratch copy of the struct)
  auto run_stage = [&](auto ST) __attribute__((always_inline)) {
    constexpr int st = decltype(ST)::value;
    const IdfResblockStage& s = p.s[st];
    const IdfResblockStage& nx = p.s[st + 1 < 3 ? st + 1 : 2];
    const int cin = st == 0 ? Cin : BN, nck = cin / CK;
    const bool last = st + 1 == p.nstage;
    f32x4_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[a][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int ck = 0; ck < nck; ++ck) {
      store_w();
      __syncthreads();       // slab in place; the activated image (written before the loop / by the previous epilogue) too
      if (ck + 1 < nck) load_w(reinterpret_cast<const bf16_t*>(s.w), cin, ck + 1);
      else if (!last) load_w(reinterpret_cast<const bf16_t*>(nx.w), BN, 0);          // the next stage's first slab travels through its epilogue
      const unsigned char* Xs = Abuf + (size_t)ck * NPH * 64;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int toff = (tap / 3) * WH + (tap % 3);
        bf16x8_t wf[2], xf[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) wf[a] = *reinterpret_cast<const bf16x8_t*>(Wbuf + tap * BN * 64 + wbase[a]);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int h = hbase[i] + toff;
          xf[i] = *reinterpret_cast<const bf16x8_t*>(Xs + h * 64 + swz(h, fq) * 16);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[a][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[a], xf[i], acc[a][i], 0, 0, 0);
      }
      __syncthreads();
    }
    // ---- epilogue: fp32 tile (+ bias, + the shortcut) -> Hbuf (the weight slab is dead); the order of the additions is the
    // per-op path's: (acc + bias) + residual, the shortcut rounded to bf16 first (it was a tensor of its own there)
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int nl = wn0 + a * 16 + fq * 4;
      const float4 bb = *reinterpret_cast<const float4*>(s.bias + nl);
      const float bbv[4] = {bb.x, bb.y, bb.z, bb.w};
      float bsv[4] = {0.f, 0.f, 0.f, 0.f};
      if (last && p.w_sc && p.b_sc) {
        const float4 bs = *reinterpret_cast<const float4*>(p.b_sc + nl);
        bsv[0] = bs.x; bsv[1] = bs.y; bsv[2] = bs.z; bsv[3] = bs.w;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          acc[a][i][r] += bbv[r];
          if (last && p.w_sc) acc[a][i][r] += bf16_to_f32(f32_to_bf16(sacc[a][i][r] + bsv[r]));
        }
        const int pl = wm0 + i * 16 + fr;
        *reinterpret_cast<float4*>(Hbuf + pl * PF + nl) = make_float4(acc[a][i][0], acc[a][i][1], acc[a][i][2], acc[a][i][3]);
      }
    }
    __syncthreads();
    float ssum[8], ssq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) ssum[e] = ssq[e] = 0.f;
    uint4 hv[2];             // this thread's two output vectors (pixels tid / 16 and tid / 16 + 32), bf16
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int pl = (tid >> 4) + kk * 32;
      float o[8];
      const float4 v0 = *reinterpret_cast<const float4*>(Hbuf + pl * PF + cc), v1 = *reinterpret_cast<const float4*>(Hbuf + pl * PF + cc + 4);
      o[0] = v0.x; o[1] = v0.y; o[2] = v0.z; o[3] = v0.w; o[4] = v1.x; o[5] = v1.y; o[6] = v1.z; o[7] = v1.w;
      const size_t e0 = (size_t)(b * NPX + pl) * BN + cc;
      if (last && !p.w_sc) {                                 // identity residual: the raw block input
        float r[8];
        Vec16<bf16_t>::load(px + e0, r);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += r[e];
      }
      uint32_t w4[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) w4[i] = (uint32_t)f32_to_bf16(o[2 * i]) | ((uint32_t)f32_to_bf16(o[2 * i + 1]) << 16);
      hv[kk] = make_uint4(w4[0], w4[1], w4[2], w4[3]);
#pragma unroll
      for (int i = 0; i < 4; ++i) {                          // statistics of what a reader of the tensor sees (rounded)
        const float lo = __uint_as_float(w4[i] << 16), hi = __uint_as_float(w4[i] & 0xffff0000u);
        ssum[2 * i] += lo; ssq[2 * i] += lo * lo; ssum[2 * i + 1] += hi; ssq[2 * i + 1] += hi * hi;
      }
      if (s.h_out) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(s.h_out) + e0) = hv[kk];
    }
    // lanes 16 apart hold the same couts: fold them, then the waves through LDS
#pragma unroll
    for (int off = 32; off >= 16; off >>= 1)
#pragma unroll
      for (int e = 0; e < 8; ++e) { ssum[e] += __shfl_xor(ssum[e], off, 64); ssq[e] += __shfl_xor(ssq[e], off, 64); }
    if (lane < 16) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { part[(wave * BN + cc + e) * 2] = ssum[e]; part[(wave * BN + cc + e) * 2 + 1] = ssq[e]; }
    }
    __syncthreads();
    if (tid < BN) {
      float a = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < NT / 64; ++w) { a += part[(w * BN + tid) * 2]; q += part[(w * BN + tid) * 2 + 1]; }
      chs[2 * tid] = a; chs[2 * tid + 1] = q;
      if (last && p.st_out) reinterpret_cast<float2*>(p.st_out)[(size_t)b * BN + tid] = make_float2(a, q);
    }
    if (last) return;
    __syncthreads();
    if (tid < BN) fold_channel(chs, tid, BN, b, nx, p.eps, cof);
    __syncthreads();
    {
      float scv[8], shv[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 t4 = *reinterpret_cast<const float4*>(cof + 2 * cc + 4 * q);
        scv[2 * q] = t4.x; shv[2 * q] = t4.y; scv[2 * q + 1] = t4.z; shv[2 * q + 1] = t4.w;
      }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int pl = (tid >> 4) + kk * 32;
        const unsigned e0 = (unsigned)((b * NPX + pl) * BN + cc);
        const uint4 o = act_vec(hv[kk], scv, shv, nx.drop && drop_any, seedv, nx.salt, thr, dscale, e0 >> 3);
        *reinterpret_cast<uint4*>(a_slot(pl, cc)) = o;
        if (nx.a_out) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(nx.a_out) + e0) = o;
      }
    }
    // (the next stage's loop starts with store_w + barrier: Hbuf's readers are past it by then, Abuf's writes before it)
    __syncthreads();
  };
  run_stage(std::integral_constant<int, 0>{});
  run_stage(std::integral_constant<int, 1>{});
  if (p.nstage == 3) run_stage(std::integral_constant<int, 2>{});
#undef load_w
#undef store_w
}

}  // namespace

extern "C" int idf_resblock_small_ok(int B, int H, int W, int Cin, int C1, int Cout, int nstage) {
  if (B <= 0 || H != 8 || W != 8 || Cout != BN || (Cin != 128 && Cin != 256) || (nstage != 2 && nstage != 3)) return 0;
  if (C1 && (C1 <= 0 || C1 >= Cin || (C1 % CK))) return 0;
  if ((long)B * NPX * Cin >= (1L << 31)) return 0;
  return 1;
}

extern "C" int idf_resblock_small_fwd(const IdfResblockArgs* args, void* stream) {
  if (!args) IDF_FAIL(IDF_ERR_BADARG, "resblock_small_fwd: null arguments");
  const IdfResblockArgs& p = *args;
  if (!idf_resblock_small_ok(p.B, 8, 8, p.Cin, p.x2 ? p.C1 : 0, BN, p.nstage))
    IDF_FAIL(IDF_ERR_UNSUPPORTED, "resblock_small_fwd: B%d Cin%d (C1 %d) nstage%d not covered (8x8 maps, 128 couts)", p.B, p.Cin,
             p.C1, p.nstage);
  if (!p.x || !p.st1 || p.T1 < 1 || (p.x2 && (!p.st2 || p.T2 < 1)) || !p.y) IDF_FAIL(IDF_ERR_BADARG, "resblock_small_fwd: null tensor");
  if (!p.w_sc && p.Cin != BN) IDF_FAIL(IDF_ERR_BADARG, "resblock_small_fwd: an identity residual needs Cin == 128");
  if (!p.w_sc && p.x2) IDF_FAIL(IDF_ERR_BADARG, "resblock_small_fwd: a two-source input needs the 1x1 shortcut");
  for (int i = 0; i < p.nstage; ++i) {
    const IdfResblockStage& s = p.s[i];
    if (!s.w || !s.bias) IDF_FAIL(IDF_ERR_BADARG, "resblock_small_fwd: stage %d weights missing", i);
    if ((s.sc != nullptr) != (s.sh != nullptr) || (s.sc != nullptr) != (s.mean != nullptr) || (s.sc != nullptr) != (s.rstd != nullptr))
      IDF_FAIL(IDF_ERR_BADARG, "resblock_small_fwd: mean / rstd / sc / sh go together (stage %d)", i);
  }
  if (p.s[p.nstage - 1].h_out != p.y) IDF_FAIL(IDF_ERR_BADARG, "resblock_small_fwd: the last stage's output is y");
  const size_t lds = (size_t)(p.Cin / CK) * NPH * 64 + WSLAB + (size_t)(4 * MAXC + 8 * BN * 2) * sizeof(float);
  static IdfLdsGrant grant;
  if (hipError_t e = idf_ensure_lds((const void*)resblock8_fwd_kernel, lds, grant); e != hipSuccess)
    IDF_FAIL(IDF_ERR_HIP, "resblock_small_fwd: %zu bytes of LDS refused: %s", lds, hipGetErrorString(e));
  RbK k;
  k.a = p;
  k.thr = idf_drop_thresh(p.p_drop);
  k.dscale = 1.0f / (1.0f - (float)k.thr / 65536.0f);
  if (!(p.p_drop > 0.f)) k.a.seed = nullptr;
  hipLaunchKernelGGL(resblock8_fwd_kernel, dim3(p.B), dim3(NT), lds, (hipStream_t)stream, k);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}
