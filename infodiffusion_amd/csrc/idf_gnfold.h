// GroupNorm / FiLM backward, coefficient fold from per-channel partial sums (shared by the streaming apply kernel of
// idf_groupnorm.hip and the data-gradient conv whose prologue applies it, idf_conv3x3.hip).
//
// A data-gradient conv with the `du` epilogue leaves, for the GroupNorm stage whose activated output it differentiated,
//   du = dA * act'(x * sc + sh) * mask        (bf16, the tensor's own shape)
//   part[b][t][c] = (sum du, sum du * x)      over the pixels of tile t (of the bf16-rounded du)
// From the sums S1, S2 per (sample, channel) everything else follows (derivation: idf_groupnorm.hip header;
// modules.py:312-318 backward):
//   dx = A * du + K1 * x + K0,   A = sc[b,c],   K1 = -r^2 P2 / N,   K0 = (-r P1 + r^2 mu P2) / N   per (sample, group)
//   with D1 = S1, D2 = r (S2 - mu S1), f = (1+s_t)(1+s_a), P1 = sum_c gamma f D1, P2 = sum_c gamma f D2 over the group,
//   dgamma += f D2, dbeta += f D1, dFiLM_t = ((gamma D2 + beta D1)(1+s_a), D1 (1+s_a)),
//   dFiLM_a = ((gamma D2 + beta D1)(1+s_t) + D1 b_t, D1).
#pragma once
#include "idf_common.h"

// Sum of T per-tile partials (float2, `stride` float2s apart), all loads of a batch of 16 in flight before the first add.
// (A plain `for t` loop makes hipcc wait for each pair of loads before it issues the next: T / 2 dependent L2 round trips
// -- 4 us in front of every block at T = 16, the 64x64 maps.)  Two interleaved chains, even / odd t, in a fixed order:
// every block that folds the same partials gets the same bits.
__device__ __forceinline__ float2 idf_sum_partials(const float2* __restrict__ src, int T, size_t stride) {
  float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
  for (int t0 = 0; t0 < T; t0 += 16) {
    float2 v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = (t0 + j < T) ? src[(size_t)(t0 + j) * stride] : make_float2(0.f, 0.f);
#pragma unroll
    for (int j = 0; j < 16; j += 2) { s0 += v[j].x; q0 += v[j].y; s1 += v[j + 1].x; q1 += v[j + 1].y; }
  }
  return make_float2(s0 + s1, q0 + q1);
}

struct GnFoldP {
  const float* part; int T;              // [B][T][C][2]
  const float* mean; const float* rstd;  // [B][32]
  const float* sc;                       // [B][C]  the forward's folded scale
  const float* gamma; const float* beta; const float* film_t; const float* film_a;
  int ld_t, ld_a;
  float* dfilm_t; float* dfilm_a;        // [B][2C] or null
  float* dgb;                            // [B][2][C] per-sample (dgamma, dbeta) contributions, or null
  float* dgam; float* dbet;              // [C] accumulated with atomics (pre-zeroed parameter gradients), or null
  int C, HW;
};

// cof[4c + {0,1,2}] = (A, K1, K0) of image b; pc = scratch [C][2].  `writer`: this block stores the side outputs
// (exactly one block per image must).  The partials are summed in a fixed order: every block computes the same values.
template <int NT>
__device__ __forceinline__ void gn_bwd_fold(const GnFoldP& f, int b, bool writer, float* cof, float* pc, int tid) {
  const int C = f.C, cpg = C >> 5;
  for (int c = tid; c < C; c += NT) {
    const float2 S = idf_sum_partials(reinterpret_cast<const float2*>(f.part) + (size_t)b * f.T * C + c, f.T, (size_t)C);
    const float S1 = S.x, S2 = S.y;
    const int g = c / cpg;
    const float mu = f.mean[b * 32 + g], r = f.rstd[b * 32 + g];
    const float ga = f.gamma ? f.gamma[c] : 1.f, be = f.beta ? f.beta[c] : 0.f;
    float st = 0.f, bt = 0.f, sa = 0.f;
    if (f.film_t) { st = f.film_t[(size_t)b * f.ld_t + c]; bt = f.film_t[(size_t)b * f.ld_t + C + c]; }
    if (f.film_a) sa = f.film_a[(size_t)b * f.ld_a + c];
    const float D1 = S1, D2 = r * (S2 - mu * S1);
    const float fm = (1.f + st) * (1.f + sa);
    if (writer) {
      const float Gf = ga * D2 + be * D1, Ge = D1;
      if (f.dfilm_t) { f.dfilm_t[(size_t)b * 2 * C + c] = Gf * (1.f + sa); f.dfilm_t[(size_t)b * 2 * C + C + c] = Ge * (1.f + sa); }
      if (f.dfilm_a) { f.dfilm_a[(size_t)b * 2 * C + c] = Gf * (1.f + st) + Ge * bt; f.dfilm_a[(size_t)b * 2 * C + C + c] = Ge; }
      if (f.dgb) { f.dgb[((size_t)b * 2 + 0) * C + c] = fm * D2; f.dgb[((size_t)b * 2 + 1) * C + c] = fm * D1; }
      if (f.dgam) atomicAdd(f.dgam + c, fm * D2);
      if (f.dbet) atomicAdd(f.dbet + c, fm * D1);
    }
    pc[2 * c] = ga * fm * D1; pc[2 * c + 1] = ga * fm * D2;
    cof[4 * c] = f.sc[(size_t)b * C + c];
  }
  __syncthreads();
  const float invN = 1.f / ((float)f.HW * cpg);
  for (int c = tid; c < C; c += NT) {
    const int g = c / cpg;
    float P1 = 0.f, P2 = 0.f;
    for (int k = g * cpg; k < (g + 1) * cpg; ++k) { P1 += pc[2 * k]; P2 += pc[2 * k + 1]; }
    const float mu = f.mean[b * 32 + g], r = f.rstd[b * 32 + g];
    cof[4 * c + 1] = -r * r * P2 * invN;
    cof[4 * c + 2] = (-r * P1 + r * r * mu * P2) * invN;
  }
  __syncthreads();
}
