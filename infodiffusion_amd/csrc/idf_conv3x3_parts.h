// Parts of the 3x3 / 1x1 bf16 conv kernels shared by idf_conv3x3.hip (halo-tile, direct-to-LDS, persistent forms) and
// idf_conv_rs.hip (weights-in-registers row-stream form): the launch descriptor, the GroupNorm coefficient fold, the staged-vector
// transform and the epilogues' layout-independent halves.  Everything lives in an anonymous namespace per translation unit.
#pragma once
#include "idf_common.h"
#include "idf_gnfold.h"

namespace {


struct C3P {
  const bf16_t* x;      // source activations [B, Hs, Ws, Cin]
  const bf16_t* x2;     // DUAL: channels C1.. of the input live here ([B, Hs, Ws, Cin - C1]); x holds [.., C1]
  int C1;
  const bf16_t* w;      // [Cout][9][Cin]
  const float* bias;    // [Cout] or null
  const bf16_t* res;    // [B, H, W, Cout] or null
  bf16_t* y;            // [B, H, W, Cout]
  int B, H, W, Hs, Ws, Cin, Cout;
  int R, tiles_per_img, n_tiles, wshift;
  unsigned wh_magic;    // (pix * wh_magic) >> 16 == pix / (W + 2*halo) for every halo pixel index (checked on the host)
  // ---- statistics epilogue: per-(image, pixel tile, cout) partial (sum y, sum y^2) of the bf16-rounded output,
  // st_out [B][tiles_per_img][Cout][2].  Each block owns its entries: plain stores, no zeroing, fixed order.
  float* st_out;
  int aux_off;          // LDS byte offset of the auxiliary region (coefficients / statistics scratch)
  // ---- GroupNorm prologue (PRO): the conv input is act(x * sc[b,c] + sh[b,c]) with sc / sh folded in-block from
  // the producers' statistics partials st1 [B][T1][C1][2] (st2 [B][T2][Cin-C1][2] for the second source)
  const float* st1; const float* st2;
  int T1, T2;
  const float* gamma; const float* beta; const float* film_t; const float* film_a;
  int ld_t, ld_a;
  float eps;
  int act;              // 1: affine only (AttnBlock GroupNorm), 2: SiLU (+ dropout when seed != null)
  const uint64_t* seed;
  uint32_t salt, thr;
  float dscale;
  const float* cof_in;  // [B][Cin][2] (sc, sh) folded by pro_coef_kernel beforehand (big launches: many blocks per image)
  bf16_t* a_out;        // training: the activated tensor [B, H, W, Cin] (dense), kept for the weight gradient
  float* mean_out; float* rstd_out; float* sc_out; float* sh_out;   // training: saved for the GroupNorm backward
  // ---- GroupNorm backward in the epilogue of a data-gradient conv (GNB; the tile is a whole image, so the block owns
  // every pixel of its 64 channels and the (sample, group) sums need no other block): the accumulator tile is dA, the
  // gradient w.r.t. the activated tensor a = act(x * sc + sh); the block turns it into dx as idf_gn_fused_bwd does.
  // res / gnb_res2 = gradients arriving over the residual / skip branches, y = dx, gamma .. ld_a / act / seed .. as above.
  const bf16_t* gnb_x;          // the GroupNorm's input [B, H, W, Cout]
  const bf16_t* gnb_res2;
  const float* gnb_sc; const float* gnb_sh; const float* gnb_mean; const float* gnb_rstd;
  float* gnb_dfilm_t; float* gnb_dfilm_a; float* gnb_dgb; float* gnb_dgam; float* gnb_dbet;
  // ---- backward chain at the big maps (BWD & 2, "du epilogue"): the accumulator tile is dA, the gradient w.r.t. the activated
  // tensor a = dropout(act(x * sc + sh)) of a GroupNorm stage; the epilogue writes du = dA * act'(x*sc+sh) * mask (bf16) into y
  // and the per-(image, pixel tile, channel) partial sums (sum du, sum du * x) into st_out.  x may be the pair x | x2.
  const bf16_t* due_x; const bf16_t* due_x2; int due_C1;
  const float* due_sc; const float* due_sh;
  // (BWD & 1, "dy prologue"): the conv input is itself the gradient of a GroupNorm stage that exists only as (du, partials):
  // dy = A * du + K1 * xg + K0 with x = du, dyp_x = xg (that GroupNorm's input), coefficients folded in-block from dyp_f
  // (gn_bwd_fold; block (row tile 0, cout tile 0) of an image stores that GroupNorm's parameter / FiLM gradients);
  // dyp_out: dy written once (interior vectors of cout tile 0), kept for the weight gradient of the conv that produced xg
  const bf16_t* dyp_x; bf16_t* dyp_out; GnFoldP dyp_f;
  // ---- an auxiliary 1x1 job riding in the same launch (conv3x3_halo_bf16, MODE 0, KS 3, BN 64): blocks >= main_blocks run
  // y_aux = conv1x1(x_aux | x2_aux) + bias_aux over the same pixel tiles -- the ResBlock shortcut (modules.py:228, 248) beside
  // the block's first conv (forward: both read the block input) or its data gradient beside the first conv's (backward):
  // the input is staged raw (no prologue), only the centre tap is contracted, the epilogue is the plain one.
  int main_blocks, aux_blocks;
  int hw_main;                  // conv3x3_halo_bf16: > 0 -- blocks >= hw_main are helper workgroups (idf_warm_lines over the launch's weights)
  const bf16_t* aux_x; const bf16_t* aux_x2; int aux_C1, aux_Cin;
  const bf16_t* aux_w;          // [aux_Cout][aux_Cin]
  const float* aux_bias; bf16_t* aux_y; int aux_Cout, aux_n_tiles;
  // ---- persistent wave-specialised form (conv_ps_bf16): a pixel tile = NI images x R rows x W columns = 256 pixels
  int ps_NI, ps_rwshift;        // images per tile, log2(R * W)
  int ps_npi;                   // halo pixels per image, (R + 2 halo)(W + 2 halo)
  unsigned ps_magic_img;        // (pix * magic) >> 16 == pix / ps_npi over the tile's halo pixels
  int ps_nptiles, ps_work;      // pixel tiles, work items (= pixel tiles x cout tiles)
  int ps_hbytes;                // bytes of one halo ring slot (whole 1-KB groups)
  // ---- weights-in-registers row-reuse form (idf_conv_rs.hip): w = the fragment-major shadow; a workgroup walks rs_per consecutive
  // (cout tile, pixel tile) items of the rs_total; LDS: chunk images | fp32 epilogue tile at rs_os_off | statistics scratch at
  // aux_off | coefficients at rs_cof_off
  int rs_per, rs_total, rs_os_off, rs_cof_off;
  // group-synchronised du epilogue (EPI 3): the GroupNorm backward applied in the launch -- y = dx (x | x2 sources: y | rs_dx2),
  // res / gnb_res2 = the residual-branch gradients (dense, Cout channels), dyp_f = the fold's parameters (part = st_out)
  bf16_t* rs_dx2; unsigned* rs_sync; unsigned* rs_sync_err; unsigned rs_spin_max; int rs_stash_off;
  int rs_x0, rs_tidx, rs_halves;   // half-width tiles (two 256-thread workgroups per CU): first column and statistics-tile index of the current tile; tiles per row strip
};

// Workgroups go round-robin over the 8 XCDs (block b on XCD b % 8): with the map below every XCD owns one CONTIGUOUS eighth of the
// tile order, so row tiles that share halo rows -- and the cout tiles of one pixel tile -- meet in one L2 (speed only).
// (same-box A/B, tools/ab_libs.sh default xcd: train step 9.174 -> 9.152 ms, DDIM-100 328.6 -> 331.9 img/s; profiles/r04_conv_wr.txt)
#ifndef IDF_TILE_XCD
#define IDF_TILE_XCD 1
#endif
__device__ __forceinline__ int xcd_tile_id(int b, int G) {
  return (IDF_TILE_XCD && !(G & 7)) ? (b & 7) * (G >> 3) + (b >> 3) : b;
}
constexpr int HALO_VEC_MAX_256 = 1280, HALO_VEC_MAX_512 = 2048, HALO_VEC_MAX_S2 = 1536;   // (R+2)*(W+2)*4 budget per block size
constexpr int CK = 32;

__device__ __forceinline__ int swz(int row, int q) { return q ^ (((row >> 2) & 1) << 1); }

// pixel index in [B * H * W] of pixel pl of a tile that starts at row oy0: whole rows (TWS = 0: the tile's pixels are contiguous), or
// 2^TWS columns from column p.rs_x0 on (idf_conv_rs.hip's half-width tiles)
template <int TWS>
__device__ __forceinline__ int tile_pix(const C3P& p, int b, int oy0, int pl) {
  if constexpr (TWS == 0) return (b * p.H + oy0) * p.W + pl;
  else return (b * p.H + oy0 + (pl >> TWS)) * p.W + p.rs_x0 + (pl & ((1 << TWS) - 1));
}

// In-block fold of the GroupNorm statistics + gamma / beta + FiLM pairs into cof[c] = (sc, sh) (the fold of
// idf_groupnorm.hip's gn_finalize, from per-channel partial sums).  One image per block; the partials are summed
// in a fixed order, so the result does not depend on which block computes it.
// PF: this thread's first channel's gamma / beta / FiLM values are fetched together with the partials (they do not depend on
// them): one memory round trip in front of the block's first MFMA instead of two.
#ifndef IDF_PRO_PF
#define IDF_PRO_PF true
#endif
template <int NT, bool PF = IDF_PRO_PF>
__device__ __forceinline__ void pro_coefficients(const C3P& p, int b, bool writer, float* cof, float* chs, int tid) {
  const int C = p.Cin, cpg = C >> 5;
  if (p.cof_in) {                 // folded once per image by pro_coef_kernel: just fetch
    for (int c = tid; c < 2 * C; c += NT) cof[c] = p.cof_in[(size_t)b * 2 * C + c];
    __syncthreads();
    return;
  }
  float pf[6] = {1.f, 0.f, 0.f, 0.f, 0.f, 0.f};     // gamma, beta, FiLM_t scale / shift, FiLM_a scale / shift of channel `tid`
  if (PF && tid < C) {
    if (p.gamma) pf[0] = p.gamma[tid];
    if (p.beta) pf[1] = p.beta[tid];
    if (p.film_t) { pf[2] = p.film_t[(size_t)b * p.ld_t + tid]; pf[3] = p.film_t[(size_t)b * p.ld_t + C + tid]; }
    if (p.film_a) { pf[4] = p.film_a[(size_t)b * p.ld_a + tid]; pf[5] = p.film_a[(size_t)b * p.ld_a + C + tid]; }
  }
  for (int c = tid; c < C; c += NT) {
    const float* st = p.st1;
    int T = p.T1, Cs = p.C1, cl = c;
    if (c >= p.C1) { st = p.st2; T = p.T2; Cs = C - p.C1; cl = c - p.C1; }
    const float2 S = idf_sum_partials(reinterpret_cast<const float2*>(st) + (size_t)b * T * Cs + cl, T, (size_t)Cs);
    chs[2 * c] = S.x; chs[2 * c + 1] = S.y;
  }
  __syncthreads();
  const double inv_n = 1.0 / ((double)p.H * p.W * cpg);       // (block-uniform: one scalar-path division)
  for (int c = tid; c < C; c += NT) {
    const int g = c / cpg;
    double a = 0.0, d = 0.0;
    for (int k = g * cpg; k < (g + 1) * cpg; ++k) { a += chs[2 * k]; d += chs[2 * k + 1]; }
    float r, mf;
    idf_group_stats(a, d, inv_n, p.eps, &mf, &r);
    float ga, be, ft0 = 0.f, ft1 = 0.f, fa0 = 0.f, fa1 = 0.f;
    if (PF && c == tid) {
      ga = pf[0]; be = pf[1]; ft0 = pf[2]; ft1 = pf[3]; fa0 = pf[4]; fa1 = pf[5];
    } else {
      ga = p.gamma ? p.gamma[c] : 1.f; be = p.beta ? p.beta[c] : 0.f;
      if (p.film_t) { ft0 = p.film_t[(size_t)b * p.ld_t + c]; ft1 = p.film_t[(size_t)b * p.ld_t + C + c]; }
      if (p.film_a) { fa0 = p.film_a[(size_t)b * p.ld_a + c]; fa1 = p.film_a[(size_t)b * p.ld_a + C + c]; }
    }
    float sc = r * ga, sh = be - mf * sc;
    if (p.film_t) { float f = 1.f + ft0; sc *= f; sh = sh * f + ft1; }
    if (p.film_a) { float f = 1.f + fa0; sc *= f; sh = sh * f + fa1; }
    cof[2 * c] = sc; cof[2 * c + 1] = sh;
    if (writer && p.sc_out) {
      p.sc_out[(size_t)b * C + c] = sc; p.sh_out[(size_t)b * C + c] = sh;
      if (c == g * cpg) { p.mean_out[b * 32 + g] = mf; p.rstd_out[b * 32 + g] = r; }
    }
  }
  __syncthreads();
}

// grid B: the fold once per image, for launches whose images are cut into many tiles (every block of the conv would
// otherwise repeat it: two dependent memory round trips and a double-precision rsqrt in front of its first MFMA)
__global__ __launch_bounds__(256) void pro_coef_kernel(const C3P p, float* __restrict__ cof_out) {
  extern __shared__ __attribute__((aligned(16))) float cs[];        // cof [Cin][2] | chs [Cin][2]
  C3P q = p;
  q.cof_in = nullptr;
  const int b = blockIdx.x;
  pro_coefficients<256>(q, b, true, cs, cs + 2 * p.Cin, threadIdx.x);
  for (int c = threadIdx.x; c < 2 * p.Cin; c += 256) cof_out[(size_t)b * 2 * p.Cin + c] = cs[c];
}

// a = act(x * sc + sh) on one 16-byte vector (8 bf16 channels); vec = index of the vector in the dense activated
// tensor (the dropout key, as idf_groupnorm.hip's gn_apply_kernel / du_vec use it)
// G = elements activated together (their exp / rcp chains interleave; 8 costs ~10 more live registers than 4)
#ifndef IDF_DLDS_PRO_G
#define IDF_DLDS_PRO_G 1
#endif
#ifndef IDF_HALO_PRO_G
#define IDF_HALO_PRO_G 8
#endif
#ifndef IDF_RES_PF
#define IDF_RES_PF 0          // plain epilogue: residual vectors issued under the last chunk's MFMA phase -- built, measured, OFF:
                              // +25 registers in the plain kernels, 9.88 vs 9.83 ms per step (the du epilogue's x prefetch below pays: -0.05 ms)
#endif
#ifndef IDF_DUE_PF
#define IDF_DUE_PF 1          // du epilogue: x vectors issued under the last chunk's MFMA phase
#endif
#ifndef IDF_SMALL_PFD
#define IDF_SMALL_PFD 1        // chunks in flight in the 64-pixel-tile launches: 2 was built and measured -- 128->128 @8x8 8.5 -> 8.9 us,
                               // DDIM-100 at B = 256 297 -> 287 img/s (registers: 143 -> 214): the per-chunk 1.2 us is not load latency
#endif

// the same with the activation / dropout switches as template arguments (resolved once per tile by the caller: straight-line code)
template <bool SILU, bool DROP, int G = 8>
__device__ __forceinline__ uint4 pro_vec_t(const uint4 raw, const float (&scv)[8], const float (&shv)[8], uint64_t seedv, uint32_t salt,
                                           uint32_t thr, float dscale, uint32_t vec) {
  const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
  float v[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
  const uint32_t h = DROP ? idf_vec_hash(seedv, salt, vec) : 0u;
#pragma unroll
  for (int g0 = 0; g0 < 8; g0 += G) idf_act_vec_t<G, SILU, DROP>(v + g0, scv + g0, shv + g0, h, g0, thr, dscale);
  return make_uint4(idf_pack_bf16(v[0], v[1]), idf_pack_bf16(v[2], v[3]), idf_pack_bf16(v[4], v[5]), idf_pack_bf16(v[6], v[7]));
}

template <int G = 8>
__device__ __forceinline__ uint4 pro_vec(const uint4 raw, const float (&scv)[8], const float (&shv)[8], int act, bool drop,
                                         uint64_t seedv, uint32_t salt, uint32_t thr, float dscale, uint32_t vec) {
  const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
  float v[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
  const uint32_t h = drop ? idf_vec_hash(seedv, salt, vec) : 0u;
#pragma unroll
  for (int g0 = 0; g0 < 8; g0 += G) idf_act_vec<G>(v + g0, scv + g0, shv + g0, act, drop, h, g0, thr, dscale);
  uint32_t o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = idf_pack_bf16(v[2 * i], v[2 * i + 1]);
  return make_uint4(o[0], o[1], o[2], o[3]);
}

// Epilogue through LDS.  A lane holds couts n..n+3 of one pixel, i.e. 8-byte pieces scattered over 16 pixel rows
// per store instruction; the fp32 tile goes through LDS (the staging buffers are free now) and is written back as
// whole 16-byte chunks, consecutive lanes covering one pixel's contiguous couts: full-line HBM writes, coalesced
// bias / residual reads.  With p.st_out the per-cout (sum, sum of squares) of the block's bf16-rounded outputs are
// stored too: the GroupNorm that reads y needs no pass of its own over it.
// the residual vectors of this thread's outputs (same (pixel, 8-cout) slots as lds_epilogue's loop): issued by the caller
// under its last MFMA phase
template <int BM, int BN, int NT>
__device__ __forceinline__ void res_fetch(const C3P& p, uint4 (&rr)[(BM * (BN / 8) + NT - 1) / NT], int b, int oy0, int n0, int KT,
                                          int tid) {
  constexpr int CPR = BN / 8, NI = (BM * CPR + NT - 1) / NT;
  const int ncols = min(BN, p.Cout - n0);
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    const int idx = tid + k * NT, pl = idx / CPR, cc = (idx - pl * CPR) * 8;
    rr[k] = make_uint4(0, 0, 0, 0);
    if (idx < BM * CPR && pl < KT && cc < ncols)
      rr[k] = *reinterpret_cast<const uint4*>(p.res + ((size_t)(b * p.H + oy0) * p.W + pl) * p.Cout + n0 + cc);
  }
}

// second half of lds_epilogue: the fp32 tile Os [BM][BN + 4] is in LDS (the caller's barrier is behind it); bias, residual,
// rounding, full-line stores and the statistics partials.  LDSONLY: the waits in front of the internal barrier are for LDS only
// (the row-stream kernel keeps global loads in flight across it).
template <int BM, int BN, int NT, bool LDSONLY = false, int TWS = 0>
__device__ __forceinline__ void lds_epilogue_tail(const C3P& p, unsigned char* smem, int b, int oy0, int n0, int KT, int tid,
                                                  const uint4 (&rpre)[(BM * (BN / 8) + NT - 1) / NT], bool have_rpre) {
  const int lane = tid & 63, wave = tid >> 6;
  const int W = p.W, R = p.R;
  const int ncols = min(BN, p.Cout - n0);          // valid couts of this tile
  constexpr int PF = BN + 4;                       // fp32 row pitch (floats)
  float* Os = reinterpret_cast<float*>(smem);      // [BM][PF]
  constexpr int CPR = BN / 8;                      // 16-byte output chunks per pixel row
  float ssum[8], ssq[8];                           // statistics of this thread's 8 couts (cc is fixed per thread)
#pragma unroll
  for (int k = 0; k < 8; ++k) ssum[k] = ssq[k] = 0.f;
#pragma unroll
  for (int kk = 0; kk < (BM * CPR + NT - 1) / NT; ++kk) {
    const int idx = tid + kk * NT;
    if (idx >= BM * CPR) break;
    int pl = idx / CPR, cc = (idx - pl * CPR) * 8;
    if (pl >= KT || cc >= ncols) continue;
    float o[8];
    float4 v0 = *reinterpret_cast<const float4*>(Os + pl * PF + cc);
    float4 v1 = *reinterpret_cast<const float4*>(Os + pl * PF + cc + 4);
    o[0] = v0.x; o[1] = v0.y; o[2] = v0.z; o[3] = v0.w; o[4] = v1.x; o[5] = v1.y; o[6] = v1.z; o[7] = v1.w;
    const unsigned e = (unsigned)(tile_pix<TWS>(p, b, oy0, pl) * p.Cout + n0 + cc);      // tensors < 2^31 elements (checked on the host)
    if (p.bias) {
      float4 b0 = *reinterpret_cast<const float4*>(p.bias + n0 + cc);
      float4 b1 = *reinterpret_cast<const float4*>(p.bias + n0 + cc + 4);
      o[0] += b0.x; o[1] += b0.y; o[2] += b0.z; o[3] += b0.w; o[4] += b1.x; o[5] += b1.y; o[6] += b1.z; o[7] += b1.w;
    }
    if (p.res) {
      float r[8];
      if (have_rpre) {
        const uint32_t w4[4] = {rpre[kk].x, rpre[kk].y, rpre[kk].z, rpre[kk].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { r[2 * i] = __uint_as_float(w4[i] << 16); r[2 * i + 1] = __uint_as_float(w4[i] & 0xffff0000u); }
      } else Vec16<bf16_t>::load(p.res + e, r);
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] += r[k];
    }
    uint32_t ow[4];                                // rounded once, as pairs (v_cvt_pk_bf16_f32)
#pragma unroll
    for (int k = 0; k < 4; ++k) ow[k] = idf_pack_bf16(o[2 * k], o[2 * k + 1]);
    if (p.st_out) {                                // statistics of the values a reader of y will see (bf16-rounded)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float lo = __uint_as_float(ow[k] << 16), hi = __uint_as_float(ow[k] & 0xffff0000u);
        ssum[2 * k] += lo; ssq[2 * k] += lo * lo; ssum[2 * k + 1] += hi; ssq[2 * k + 1] += hi * hi;
      }
    }
    *reinterpret_cast<uint4*>(p.y + e) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
  }
  if (p.st_out) {
    // lanes CPR apart hold the same couts: fold them, then the waves through LDS (outside the fp32 tile)
    if constexpr (CPR == 8) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        ssum[k] = idf_xor8_sum(idf_xor16_sum(idf_xor32_sum(ssum[k])));
        ssq[k] = idf_xor8_sum(idf_xor16_sum(idf_xor32_sum(ssq[k])));
      }
    } else {
#pragma unroll
      for (int off = 32; off >= CPR; off >>= 1)
#pragma unroll
        for (int k = 0; k < 8; ++k) { ssum[k] += __shfl_xor(ssum[k], off, 64); ssq[k] += __shfl_xor(ssq[k], off, 64); }
    }
    float* part = reinterpret_cast<float*>(smem + p.aux_off);     // [waves][BN][2]
    if (lane < CPR) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        part[(wave * BN + lane * 8 + k) * 2] = ssum[k];
        part[(wave * BN + lane * 8 + k) * 2 + 1] = ssq[k];
      }
    }
    if (LDSONLY) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); else __syncthreads();
    for (int c = tid; c < ncols; c += NT) {
      float a = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < NT / 64; ++w) { a += part[(w * BN + c) * 2]; q += part[(w * BN + c) * 2 + 1]; }
      reinterpret_cast<float2*>(p.st_out)[((size_t)b * p.tiles_per_img + (TWS ? p.rs_tidx : oy0 / R)) * p.Cout + n0 + c] = make_float2(a, q);
    }
  }
}

template <int TM, int TN, int BM, int BN, int NT>
__device__ __forceinline__ void lds_epilogue(const C3P& p, const f32x4_t (&acc)[TN][TM], unsigned char* smem, int b, int oy0,
                                             int n0, int KT, int tid, int wm0, int wn0,
                                             const uint4 (&rpre)[(BM * (BN / 8) + NT - 1) / NT], bool have_rpre) {   // res_fetch's registers
  const int lane = tid & 63, fr = lane & 15, fq = lane >> 4;
  constexpr int PF = BN + 4;                       // fp32 row pitch (floats)
  float* Os = reinterpret_cast<float*>(smem);      // [BM][PF]; fits: BM*(BN+4)*4 <= (halo + weights) bytes
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    int pl = wm0 + i * 16 + fr;
#pragma unroll
    for (int a = 0; a < TN; ++a) {
      int nl = wn0 + a * 16 + fq * 4;
      *reinterpret_cast<float4*>(Os + pl * PF + nl) = make_float4(acc[a][i][0], acc[a][i][1], acc[a][i][2], acc[a][i][3]);
    }
  }
  __syncthreads();
  lds_epilogue_tail<BM, BN, NT>(p, smem, b, oy0, n0, KT, tid, rpre, have_rpre);
}

// Epilogue of a data-gradient conv whose tile is one whole image: the GroupNorm / FiLM / SiLU / dropout backward of
// idf_groupnorm.hip's gn_small_bwd on the accumulator tile (same sums, same coefficient algebra, same outputs), so the
// 16x16 / 8x8 levels need no GroupNorm-backward launch.  dA never leaves the chip (and is not rounded to bf16 on the way).
// What the epilogue reads from memory -- x, the branch gradient, the coefficients -- does not depend on the conv: it is
// fetched BEFORE the conv's main loop (GnbPre, ~50 registers of a kernel that runs at two waves per SIMD anyway), so the
// epilogue starts with its operands in registers instead of with a memory round trip.
template <int NI>
struct GnbPre {
  uint4 xr[NI], rr[NI];
  float scv[8], shv[8], pf[7];
  uint64_t seedv;
};

template <int BM, int BN, int NT>
__device__ __forceinline__ void gnb_prefetch(const C3P& p, int b, int n0, int KT, int tid, GnbPre<BM * (BN / 8) / NT>& g) {
  constexpr int CPR = BN / 8, NI = BM * CPR / NT;
  const int C = p.Cout, cpg = C >> 5, HW = KT;
  const int cc = (tid % CPR) * 8;                  // this thread's 8 channels (fixed: NT % CPR == 0)
  const size_t cbase = (size_t)b * C + n0 + cc;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    float4 a4 = *reinterpret_cast<const float4*>(p.gnb_sc + cbase + 4 * q), b4 = *reinterpret_cast<const float4*>(p.gnb_sh + cbase + 4 * q);
    g.scv[4 * q] = a4.x; g.scv[4 * q + 1] = a4.y; g.scv[4 * q + 2] = a4.z; g.scv[4 * q + 3] = a4.w;
    g.shv[4 * q] = b4.x; g.shv[4 * q + 1] = b4.y; g.shv[4 * q + 2] = b4.z; g.shv[4 * q + 3] = b4.w;
  }
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    const int pl = (tid + k * NT) / CPR;
    g.xr[k] = g.rr[k] = make_uint4(0, 0, 0, 0);
    if (pl < KT) {
      const size_t e0 = ((size_t)b * HW + pl) * C + n0 + cc;
      g.xr[k] = *reinterpret_cast<const uint4*>(p.gnb_x + e0);
      if (p.res) g.rr[k] = *reinterpret_cast<const uint4*>(p.res + e0);
    }
  }
  // the coefficient phase's per-channel parameters (thread c < BN owns channel n0 + c)
  g.pf[0] = 0.f; g.pf[1] = 0.f; g.pf[2] = 1.f; g.pf[3] = 0.f; g.pf[4] = 0.f; g.pf[5] = 0.f; g.pf[6] = 0.f;
  if (tid < BN) {
    const int c = n0 + tid, gr = c / cpg;
    g.pf[0] = p.gnb_mean[b * 32 + gr]; g.pf[1] = p.gnb_rstd[b * 32 + gr];
    if (p.gamma) g.pf[2] = p.gamma[c];
    if (p.beta) g.pf[3] = p.beta[c];
    if (p.film_t) { g.pf[4] = p.film_t[(size_t)b * p.ld_t + c]; g.pf[5] = p.film_t[(size_t)b * p.ld_t + C + c]; }
    if (p.film_a) g.pf[6] = p.film_a[(size_t)b * p.ld_a + c];
  }
  g.seedv = (p.act == 2 && p.seed != nullptr) ? *p.seed : 0;
}

template <int TM, int TN, int BM, int BN, int NT>
__device__ __forceinline__ void gnb_epilogue(const C3P& p, const f32x4_t (&acc)[TN][TM], unsigned char* smem, int b, int n0,
                                             int KT, int tid, int wm0, int wn0, const GnbPre<BM * (BN / 8) / NT>& g) {
  const int lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
  constexpr int PF = BN + 4, CPR = BN / 8, NI = BM * CPR / NT, NW = NT / 64;
  const int C = p.Cout, cpg = C >> 5, HW = KT, GS = BN / cpg;
  float* Os = reinterpret_cast<float*>(smem);      // [BM][PF]: dA, then du in place
  float* part = reinterpret_cast<float*>(smem + p.aux_off);     // [NW][BN][2]
  float* pc = part + NW * BN * 2;                  // [BN][2]
  float* kk = pc + BN * 2;                         // [GS][2]
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    int pl = wm0 + i * 16 + fr;
#pragma unroll
    for (int a = 0; a < TN; ++a) {
      int nl = wn0 + a * 16 + fq * 4;
      *reinterpret_cast<float4*>(Os + pl * PF + nl) = make_float4(acc[a][i][0], acc[a][i][1], acc[a][i][2], acc[a][i][3]);
    }
  }
  const int cc = (tid % CPR) * 8;
  const uint4 (&xr)[NI] = g.xr;
  const uint4 (&rr)[NI] = g.rr;
  const float (&scv)[8] = g.scv;
  const float (&shv)[8] = g.shv;
  const float (&pf)[7] = g.pf;
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
  const bool drop = p.act == 2 && p.seed != nullptr;
  const uint64_t seedv = g.seedv;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    const int pl = (tid + k * NT) / CPR;
    if (pl < KT) {
      float dav[8], xv[8], du[8];
      float4 v0 = *reinterpret_cast<const float4*>(Os + pl * PF + cc), v1 = *reinterpret_cast<const float4*>(Os + pl * PF + cc + 4);
      dav[0] = v0.x; dav[1] = v0.y; dav[2] = v0.z; dav[3] = v0.w; dav[4] = v1.x; dav[5] = v1.y; dav[6] = v1.z; dav[7] = v1.w;
      const uint32_t w4[4] = {xr[k].x, xr[k].y, xr[k].z, xr[k].w};
#pragma unroll
      for (int i = 0; i < 4; ++i) { xv[2 * i] = __uint_as_float(w4[i] << 16); xv[2 * i + 1] = __uint_as_float(w4[i] & 0xffff0000u); }
      const size_t e0 = ((size_t)b * HW + pl) * C + n0 + cc;
      const uint32_t h = drop ? idf_vec_hash(seedv, p.salt, e0 >> 3) : 0u;
      if (p.act == 2) {
#pragma unroll
        for (int g0 = 0; g0 < 8; g0 += 4) {
          if (drop) idf_dact_vec_t<4, true, true>(dav + g0, xv + g0, scv + g0, shv + g0, h, g0, p.thr, p.dscale, du + g0);
          else idf_dact_vec_t<4, true, false>(dav + g0, xv + g0, scv + g0, shv + g0, h, g0, p.thr, p.dscale, du + g0);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) du[e] = dav[e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) { s1[e] += du[e]; s2[e] += du[e] * xv[e]; }
      *reinterpret_cast<float4*>(Os + pl * PF + cc) = make_float4(du[0], du[1], du[2], du[3]);
      *reinterpret_cast<float4*>(Os + pl * PF + cc + 4) = make_float4(du[4], du[5], du[6], du[7]);
    }
  }
  // lanes CPR apart hold the same channels: fold them, then the waves through LDS
#pragma unroll
  for (int off = 32; off >= CPR; off >>= 1)
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] += __shfl_xor(s1[e], off, 64); s2[e] += __shfl_xor(s2[e], off, 64); }
  if (lane < CPR) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { part[(wave * BN + lane * 8 + e) * 2] = s1[e]; part[(wave * BN + lane * 8 + e) * 2 + 1] = s2[e]; }
  }
  __syncthreads();
  if (tid < BN) {
    const int c = n0 + tid;
    float S1 = 0.f, S2 = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) { S1 += part[(w * BN + tid) * 2]; S2 += part[(w * BN + tid) * 2 + 1]; }
    const float mu = pf[0], r = pf[1], ga = pf[2], be = pf[3], st = pf[4], bt = pf[5], sa = pf[6];
    const float D1 = S1, D2 = r * (S2 - mu * S1);
    const float f = (1.f + st) * (1.f + sa);
    const float Gf = ga * D2 + be * D1, Ge = D1;
    if (p.gnb_dfilm_t) { p.gnb_dfilm_t[(size_t)b * 2 * C + c] = Gf * (1.f + sa); p.gnb_dfilm_t[(size_t)b * 2 * C + C + c] = Ge * (1.f + sa); }
    if (p.gnb_dfilm_a) { p.gnb_dfilm_a[(size_t)b * 2 * C + c] = Gf * (1.f + st) + Ge * bt; p.gnb_dfilm_a[(size_t)b * 2 * C + C + c] = Ge; }
    if (p.gnb_dgb) { p.gnb_dgb[((size_t)b * 2 + 0) * C + c] = f * D2; p.gnb_dgb[((size_t)b * 2 + 1) * C + c] = f * D1; }
    if (p.gnb_dgam) atomicAdd(p.gnb_dgam + c, f * D2);
    if (p.gnb_dbet) atomicAdd(p.gnb_dbet + c, f * D1);
    pc[tid * 2] = ga * f * D1; pc[tid * 2 + 1] = ga * f * D2;
  }
  __syncthreads();
  if (tid < GS) {
    float P1 = 0.f, P2 = 0.f;
    for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) { P1 += pc[c * 2]; P2 += pc[c * 2 + 1]; }
    const int g = n0 / cpg + tid;
    const float mu = p.gnb_mean[b * 32 + g], r = p.gnb_rstd[b * 32 + g];
    const float invN = 1.f / ((float)HW * cpg);
    kk[tid * 2] = -r * r * P2 * invN;
    kk[tid * 2 + 1] = (-r * P1 + r * r * mu * P2) * invN;
  }
  __syncthreads();
  float k1v[8], k0v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { const int gl = (cc + e) / cpg; k1v[e] = kk[gl * 2]; k0v[e] = kk[gl * 2 + 1]; }
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    const int pl = (tid + k * NT) / CPR;
    if (pl < KT) {
      float du[8], xv[8], o[8];
      float4 v0 = *reinterpret_cast<const float4*>(Os + pl * PF + cc), v1 = *reinterpret_cast<const float4*>(Os + pl * PF + cc + 4);
      du[0] = v0.x; du[1] = v0.y; du[2] = v0.z; du[3] = v0.w; du[4] = v1.x; du[5] = v1.y; du[6] = v1.z; du[7] = v1.w;
      const uint32_t w4[4] = {xr[k].x, xr[k].y, xr[k].z, xr[k].w};
#pragma unroll
      for (int i = 0; i < 4; ++i) { xv[2 * i] = __uint_as_float(w4[i] << 16); xv[2 * i + 1] = __uint_as_float(w4[i] & 0xffff0000u); }
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = scv[e] * du[e] + k1v[e] * xv[e] + k0v[e];
      const size_t e0 = ((size_t)b * HW + pl) * C + n0 + cc;
      if (p.res) {
        const uint32_t r4[4] = {rr[k].x, rr[k].y, rr[k].z, rr[k].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { o[2 * i] += __uint_as_float(r4[i] << 16); o[2 * i + 1] += __uint_as_float(r4[i] & 0xffff0000u); }
      }
      if (p.gnb_res2) {
        float rv[8];
        Vec16<bf16_t>::load(p.gnb_res2 + e0, rv);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += rv[e];
      }
      Vec16<bf16_t>::store(p.y + e0, o);
    }
  }
}

// Epilogue of a data-gradient conv at the big maps (a tile is a slice of an image): du = dA * act'(x*sc+sh) * mask goes out
// in place of dA, together with the per-channel partial sums the GroupNorm backward needs (sum du, sum du * x) -- the
// backward mirror of lds_epilogue's statistics.  x is fetched at the top (its latency hides behind the accumulators' trip
// through LDS); the sums are of the bf16-rounded du, i.e. of what the consumer of du will read.
// this thread's vectors of the GroupNorm input x for the du epilogue (x [.., C1] | x2 [.., C - C1], C1 % BN == 0: a cout
// tile lies in one of them): issued by the caller while its last MFMA phase still runs, consumed by due_epilogue
template <int BM, int BN, int NT, int TWS = 0>
__device__ __forceinline__ void due_fetch_x(const C3P& p, uint4 (&xr)[BM * (BN / 8) / NT], int b, int oy0, int n0, int KT, int tid) {
  constexpr int CPR = BN / 8, NI = BM * CPR / NT;
  const int C = p.Cout, cc = (tid % CPR) * 8;
  const bf16_t* xs = p.due_x;
  int xpitch = C, xc = n0 + cc;
  if (p.due_x2) {
    if (n0 < p.due_C1) xpitch = p.due_C1;
    else { xs = p.due_x2; xpitch = C - p.due_C1; xc -= p.due_C1; }
  }
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    const int pl = (tid + k * NT) / CPR;
    xr[k] = make_uint4(0, 0, 0, 0);
    if (pl < KT) xr[k] = *reinterpret_cast<const uint4*>(xs + (size_t)tile_pix<TWS>(p, b, oy0, pl) * xpitch + xc);
  }
}

// this thread's folded coefficients of the du epilogue (channels n0 + (tid % CPR) * 8 ..) and the step's dropout seed
template <int BN>
__device__ __forceinline__ void due_fetch_coef(const C3P& p, int b, int n0, int tid, float (&scv)[8], float (&shv)[8], uint64_t& seedv) {
  const int C = p.Cout, cc = (tid % (BN / 8)) * 8;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const float4 a4 = *reinterpret_cast<const float4*>(p.due_sc + (size_t)b * C + n0 + cc + 4 * q);
    const float4 b4 = *reinterpret_cast<const float4*>(p.due_sh + (size_t)b * C + n0 + cc + 4 * q);
    scv[4 * q] = a4.x; scv[4 * q + 1] = a4.y; scv[4 * q + 2] = a4.z; scv[4 * q + 3] = a4.w;
    shv[4 * q] = b4.x; shv[4 * q + 1] = b4.y; shv[4 * q + 2] = b4.z; shv[4 * q + 3] = b4.w;
  }
  seedv = (p.act == 2 && p.seed != nullptr) ? *p.seed : 0;
}

// second half of due_epilogue: dA [BM][BN + 4] is in LDS (the caller's barrier is behind it).  LDSONLY as lds_epilogue_tail's.
// KEEP (idf_conv_rs.hip's group-synchronised form): du stays in LDS (packed bf16, in the slot of the dA it came from: a thread's own
// slots) instead of going to memory, and the
// partials are published write-through (one 8-byte `sc1` store per channel, whole 128-byte lines per instruction of wave 0) for
// the other workgroups of the image to read inside this launch.
template <int BM, int BN, int NT, bool LDSONLY, bool SILU, bool DROP, int TWS = 0, bool KEEP = false>
__device__ __forceinline__ void due_epilogue_tail_t(const C3P& p, unsigned char* smem, int b, int oy0, int n0, int KT, int tid,
                                                    const uint4 (&xr)[BM * (BN / 8) / NT], const float (&scv)[8], const float (&shv)[8],
                                                    uint64_t seedv) {
  const int lane = tid & 63, wave = tid >> 6;
  const int W = p.W, R = p.R, C = p.Cout;
  constexpr int PF = BN + 4, CPR = BN / 8, NI = BM * CPR / NT;
  static_assert(NT % CPR == 0 && (BM * CPR) % NT == 0, "a thread keeps one channel slot");
  float* Os = reinterpret_cast<float*>(smem);      // [BM][PF]
  const int cc = (tid % CPR) * 8;                  // this thread's 8 channels of the tile
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    const int pl = (tid + k * NT) / CPR;
    if (pl < KT) {
      float dav[8], xv[8], du[8];
      const float4 v0 = *reinterpret_cast<const float4*>(Os + pl * PF + cc), v1 = *reinterpret_cast<const float4*>(Os + pl * PF + cc + 4);
      dav[0] = v0.x; dav[1] = v0.y; dav[2] = v0.z; dav[3] = v0.w; dav[4] = v1.x; dav[5] = v1.y; dav[6] = v1.z; dav[7] = v1.w;
      const uint32_t w4[4] = {xr[k].x, xr[k].y, xr[k].z, xr[k].w};
#pragma unroll
      for (int i = 0; i < 4; ++i) { xv[2 * i] = __uint_as_float(w4[i] << 16); xv[2 * i + 1] = __uint_as_float(w4[i] & 0xffff0000u); }
      const unsigned e0 = (unsigned)(tile_pix<TWS>(p, b, oy0, pl) * C + n0 + cc);      // index in the dense activated tensor (< 2^31: host check)
      const uint32_t h = DROP ? idf_vec_hash(seedv, p.salt, e0 >> 3) : 0u;
#pragma unroll
      for (int g0 = 0; g0 < 8; g0 += 4)
        idf_dact_vec_t<4, SILU, DROP>(dav + g0, xv + g0, scv + g0, shv + g0, h, g0, p.thr, p.dscale, du + g0);
      uint32_t dw[4];                              // rounded once, as pairs; the sums are of what the consumer of du will read
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        dw[e] = idf_pack_bf16(du[2 * e], du[2 * e + 1]);
        const float lo = __uint_as_float(dw[e] << 16), hi = __uint_as_float(dw[e] & 0xffff0000u);
        s1[2 * e] += lo; s2[2 * e] += lo * xv[2 * e]; s1[2 * e + 1] += hi; s2[2 * e + 1] += hi * xv[2 * e + 1];
      }
      if constexpr (KEEP) *reinterpret_cast<uint4*>(Os + pl * PF + cc) = make_uint4(dw[0], dw[1], dw[2], dw[3]);   // in place of the dA it came from
      else *reinterpret_cast<uint4*>(p.y + e0) = make_uint4(dw[0], dw[1], dw[2], dw[3]);
    }
  }
  // lanes CPR apart hold the same channels: fold them, then the waves through LDS (outside the fp32 tile)
  if constexpr (CPR == 8) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      s1[e] = idf_xor8_sum(idf_xor16_sum(idf_xor32_sum(s1[e])));
      s2[e] = idf_xor8_sum(idf_xor16_sum(idf_xor32_sum(s2[e])));
    }
  } else {
#pragma unroll
    for (int off = 32; off >= CPR; off >>= 1)
#pragma unroll
      for (int e = 0; e < 8; ++e) { s1[e] += __shfl_xor(s1[e], off, 64); s2[e] += __shfl_xor(s2[e], off, 64); }
  }
  float* part = reinterpret_cast<float*>(smem + p.aux_off);     // [waves][BN][2]
  if (lane < CPR) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { part[(wave * BN + lane * 8 + e) * 2] = s1[e]; part[(wave * BN + lane * 8 + e) * 2 + 1] = s2[e]; }
  }
  if (LDSONLY) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); else __syncthreads();
  for (int c = tid; c < BN; c += NT) {
    float a = 0.f, q = 0.f;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) { a += part[(w * BN + c) * 2]; q += part[(w * BN + c) * 2 + 1]; }
    float2* dst = reinterpret_cast<float2*>(p.st_out) + ((size_t)b * p.tiles_per_img + (TWS ? p.rs_tidx : oy0 / R)) * C + n0 + c;
    if constexpr (KEEP) {
      const unsigned long long v = ((unsigned long long)__float_as_uint(q) << 32) | __float_as_uint(a);
      __hip_atomic_store(reinterpret_cast<__attribute__((address_space(1))) unsigned long long*>(reinterpret_cast<uintptr_t>(dst)), v,
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else *dst = make_float2(a, q);
  }
}


// the activation / dropout switches are launch-uniform: resolved ONCE per tile, each case straight-line (with the conditions inside
// the vector loop hipcc emitted a branch, a wait and a partial copy of the body per 4 elements)
template <int BM, int BN, int NT, bool LDSONLY = false, int TWS = 0, bool KEEP = false>
__device__ __forceinline__ void due_epilogue_tail(const C3P& p, unsigned char* smem, int b, int oy0, int n0, int KT, int tid,
                                                  const uint4 (&xr)[BM * (BN / 8) / NT], const float (&scv)[8], const float (&shv)[8],
                                                  uint64_t seedv) {
  if (p.act == 2) {
    if (p.seed != nullptr) due_epilogue_tail_t<BM, BN, NT, LDSONLY, true, true, TWS, KEEP>(p, smem, b, oy0, n0, KT, tid, xr, scv, shv, seedv);
    else due_epilogue_tail_t<BM, BN, NT, LDSONLY, true, false, TWS, KEEP>(p, smem, b, oy0, n0, KT, tid, xr, scv, shv, seedv);
  } else due_epilogue_tail_t<BM, BN, NT, LDSONLY, false, false, TWS, KEEP>(p, smem, b, oy0, n0, KT, tid, xr, scv, shv, seedv);
}

template <int TM, int TN, int BM, int BN, int NT>
__device__ __forceinline__ void due_epilogue(const C3P& p, const f32x4_t (&acc)[TN][TM], unsigned char* smem, int b, int oy0,
                                             int n0, int KT, int tid, int wm0, int wn0, const uint4 (&xr)[BM * (BN / 8) / NT]) {
  const int lane = tid & 63, fr = lane & 15, fq = lane >> 4;
  constexpr int PF = BN + 4;
  float* Os = reinterpret_cast<float*>(smem);      // [BM][PF]
  float scv[8], shv[8];
  uint64_t seedv;
  due_fetch_coef<BN>(p, b, n0, tid, scv, shv, seedv);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    int pl = wm0 + i * 16 + fr;
#pragma unroll
    for (int a = 0; a < TN; ++a) {
      int nl = wn0 + a * 16 + fq * 4;
      *reinterpret_cast<float4*>(Os + pl * PF + nl) = make_float4(acc[a][i][0], acc[a][i][1], acc[a][i][2], acc[a][i][3]);
    }
  }
  __syncthreads();
  due_epilogue_tail<BM, BN, NT>(p, smem, b, oy0, n0, KT, tid, xr, scv, shv, seedv);
}

}  // namespace
