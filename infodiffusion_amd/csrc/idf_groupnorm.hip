// GroupNorm(32) statistics and the AdaGN / FiLM coefficient fold, forward and
// backward (HBM-bound kernels).  Replaces nn.GroupNorm(32, C) + the FiLM
// scale-shift of modules.py:312-318 (and 132, 214-228, 335-344; models.py:280-284).
//
// Forward: per (b, group) mean / rstd over (C/32)*H*W elements (eps 1e-5, biased
// variance), folded with gamma/beta and the two FiLM pairs into one affine per
// (b, channel):   u = x * sc[b,c] + sh[b,c]
//   sc = rstd*gamma*(1+s_t)*(1+s_a)
//   sh = ((beta - mean*rstd*gamma)*(1+s_t) + b_t)*(1+s_a) + b_a
// which the conv kernel applies while staging (followed by SiLU / dropout).
//
// Backward (given dA = gradient w.r.t. the activated tensor fed to the conv):
//   du = dA * keep * dsilu(u);  S1[b,c] = sum_p du;  S2[b,c] = sum_p du*x
//   dx = sc*du + k1[b,g]*x + k0[b,g]      (+ dres)
// with k1, k0, dgamma, dbeta and the FiLM gradients derived from S1, S2 only.
#include "idf_common.h"
#include "idf_gnfold.h"
#include <stdlib.h>

namespace {

constexpr int G = 32;

// ---------------------------------------------------------------- statistics
// grid (nchunk, B); each block reduces `chunk` pixels x all C channels to
// per-group (sum, sumsq) partials.
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_partial(const T* __restrict__ x, float2* __restrict__ part,
                                                        int HW, int C, int chunk) {
  constexpr int VE = Elem<T>::VE;
  extern __shared__ __attribute__((aligned(16))) float red[];   // [lanes][C][2] then [C][2]
  const int vpp = C / VE, lanes = 256 / vpp, tid = threadIdx.x;
  const int b = blockIdx.y, ck = blockIdx.x, nchunk = gridDim.x;
  const int v = tid % vpp, pl = tid / vpp;
  float s[VE], ss[VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) s[e] = ss[e] = 0.f;
  const int pend = min(HW, (ck + 1) * chunk);
  if (pl < lanes) {
    for (int p = ck * chunk + pl; p < pend; p += lanes) {
      float xv[VE];
      Vec16<T>::load(x + ((size_t)b * HW + p) * C + v * VE, xv);
#pragma unroll
      for (int e = 0; e < VE; ++e) { s[e] += xv[e]; ss[e] += xv[e] * xv[e]; }
    }
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      red[(pl * C + v * VE + e) * 2] = s[e];
      red[(pl * C + v * VE + e) * 2 + 1] = ss[e];
    }
  }
  __syncthreads();
  float* chs = red + lanes * C * 2;
  for (int c = tid; c < C; c += 256) {
    float a = 0.f, q = 0.f;
    for (int l = 0; l < lanes; ++l) { a += red[(l * C + c) * 2]; q += red[(l * C + c) * 2 + 1]; }
    chs[c * 2] = a; chs[c * 2 + 1] = q;
  }
  __syncthreads();
  if (tid < G) {
    const int cpg = C / G;
    float a = 0.f, q = 0.f;
    for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) { a += chs[c * 2]; q += chs[c * 2 + 1]; }
    part[((size_t)b * nchunk + ck) * G + tid] = make_float2(a, q);
  }
}

// grid (nchunk, B): per-CHANNEL partial (sum, sumsq) of `chunk` pixels -> part[b][ck][c].  The statistics a conv
// launch leaves behind for the GroupNorm that reads its output (idf_conv3x3.hip, st_out), for tensors that did
// not come out of such a launch.
template <typename T>
__global__ __launch_bounds__(256) void gn_partials_kernel(const T* __restrict__ x, float2* __restrict__ part,
                                                          int HW, int C, int chunk) {
  constexpr int VE = Elem<T>::VE;
  extern __shared__ __attribute__((aligned(16))) float red[];   // [lanes][C][2]
  const int vpp = C / VE, lanes = 256 / vpp, tid = threadIdx.x;
  const int b = blockIdx.y, ck = blockIdx.x, nchunk = gridDim.x;
  const int v = tid % vpp, pl = tid / vpp;
  float s[VE], ss[VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) s[e] = ss[e] = 0.f;
  const int pend = min(HW, (ck + 1) * chunk);
  if (pl < lanes) {
    for (int p = ck * chunk + pl; p < pend; p += lanes) {
      float xv[VE];
      Vec16<T>::load(x + ((size_t)b * HW + p) * C + v * VE, xv);
#pragma unroll
      for (int e = 0; e < VE; ++e) { s[e] += xv[e]; ss[e] += xv[e] * xv[e]; }
    }
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      red[(pl * C + v * VE + e) * 2] = s[e];
      red[(pl * C + v * VE + e) * 2 + 1] = ss[e];
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float a = 0.f, q = 0.f;
    for (int l = 0; l < lanes; ++l) { a += red[(l * C + c) * 2]; q += red[(l * C + c) * 2 + 1]; }
    part[((size_t)b * nchunk + ck) * C + c] = make_float2(a, q);
  }
}

// grid B.  Merge partials (in double), write mean/rstd, fold coefficients.
__global__ __launch_bounds__(256) void gn_finalize(const float2* __restrict__ part, int nchunk, int HW, int C,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   const float* __restrict__ film_t, const float* __restrict__ film_a,
                                                   int ld_t, int ld_a, float eps, float* __restrict__ mean, float* __restrict__ rstd,
                                                   float* __restrict__ sc, float* __restrict__ sh) {
  __shared__ float sm[G], sr[G];
  __shared__ double pa[8][G], pq[8][G];
  const int b = blockIdx.x, tid = threadIdx.x, cpg = C / G;
  {   // all 256 threads: 8 interleaved chunk subsets per group, so the partial loads overlap
    const int g = tid & (G - 1), kq = tid >> 5;
    double a = 0.0, q = 0.0;
    for (int k = kq; k < nchunk; k += 8) {
      float2 v = part[((size_t)b * nchunk + k) * G + g];
      a += v.x; q += v.y;
    }
    pa[kq][g] = a; pq[kq][g] = q;
  }
  __syncthreads();
  if (tid < G) {
    double a = 0.0, q = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { a += pa[k][tid]; q += pq[k][tid]; }
    double n = (double)HW * cpg;
    double mu = a / n, var = q / n - mu * mu;
    if (var < 0.0) var = 0.0;
    float r = (float)(1.0 / sqrt(var + (double)eps));
    sm[tid] = (float)mu; sr[tid] = r;
    mean[b * G + tid] = (float)mu; rstd[b * G + tid] = r;
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    int g = c / cpg;
    float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
    float a = sr[g] * ga, d = be - sm[g] * a;
    if (film_t) {
      float f = 1.f + film_t[(size_t)b * ld_t + c];
      a *= f; d = d * f + film_t[(size_t)b * ld_t + C + c];
    }
    if (film_a) {
      float f = 1.f + film_a[(size_t)b * ld_a + c];
      a *= f; d = d * f + film_a[(size_t)b * ld_a + C + c];
    }
    sc[(size_t)b * C + c] = a; sh[(size_t)b * C + c] = d;
  }
}

// ------------------------------------------------------------------ backward
#ifndef IDF_GN_GV
#define IDF_GN_GV 4
#endif
template <typename T>
__device__ __forceinline__ void du_vec(const float* dav, const float* xv, const float* scv, const float* shv,
                                       int act, const uint64_t* seed, uint32_t salt, uint32_t thr, float dscale,
                                       size_t e0, float* du) {
  constexpr int VE = Elem<T>::VE;
  const uint32_t h = (act == 2 && seed) ? idf_vec_hash(*seed, salt, e0 >> 3) : 0u;   // one hash per vector
  const int l0 = (int)(e0 & 7);
  constexpr int GV = IDF_GN_GV;   // elements whose exp / rcp chains interleave (8 at once costs ~10 registers more: spills at 128)
  if (act == 2) {
#pragma unroll
    for (int g0 = 0; g0 < VE; g0 += GV) {
      if (seed) idf_dact_vec_t<GV, true, true>(dav + g0, xv + g0, scv + g0, shv + g0, h, l0 + g0, thr, dscale, du + g0);
      else idf_dact_vec_t<GV, true, false>(dav + g0, xv + g0, scv + g0, shv + g0, h, l0 + g0, thr, dscale, du + g0);
    }
  } else {
    idf_dact_vec_t<VE, false, false>(dav, xv, scv, shv, h, l0, thr, dscale, du);
  }
}

// grid (nchunk, B): per-channel partial S1 = sum du, S2 = sum du*x
template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_partial(const T* __restrict__ dA, const T* __restrict__ x,
                                                      const float* __restrict__ sc, const float* __restrict__ sh,
                                                      float2* __restrict__ part, int HW, int C, int chunk, int act,
                                                      const uint64_t* seed, uint32_t salt, uint32_t thr, float dscale) {
  constexpr int VE = Elem<T>::VE;
  extern __shared__ __attribute__((aligned(16))) float red[];   // [lanes][C][2]
  const int vpp = C / VE, lanes = 256 / vpp, tid = threadIdx.x;
  const int b = blockIdx.y, ck = blockIdx.x, nchunk = gridDim.x;
  const int v = tid % vpp, pl = tid / vpp;
  float s1[VE], s2[VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) s1[e] = s2[e] = 0.f;
  const int pend = min(HW, (ck + 1) * chunk);
  if (pl < lanes) {
    float scv[VE], shv[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      scv[e] = sc[(size_t)b * C + v * VE + e];
      shv[e] = sh[(size_t)b * C + v * VE + e];
    }
    for (int p = ck * chunk + pl; p < pend; p += lanes) {
      size_t e0 = ((size_t)b * HW + p) * C + v * VE;
      float xv[VE], dav[VE], du[VE];
      Vec16<T>::load(x + e0, xv);
      Vec16<T>::load(dA + e0, dav);
      du_vec<T>(dav, xv, scv, shv, act, seed, salt, thr, dscale, e0, du);
#pragma unroll
      for (int e = 0; e < VE; ++e) { s1[e] += du[e]; s2[e] += du[e] * xv[e]; }
    }
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      red[(pl * C + v * VE + e) * 2] = s1[e];
      red[(pl * C + v * VE + e) * 2 + 1] = s2[e];
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float a = 0.f, q = 0.f;
    for (int l = 0; l < lanes; ++l) { a += red[(l * C + c) * 2]; q += red[(l * C + c) * 2 + 1]; }
    part[((size_t)b * nchunk + ck) * C + c] = make_float2(a, q);
  }
}

// grid B.  From S1,S2 derive k1,k0 per group, FiLM grads, and the per-sample
// gamma/beta gradient contributions dgb[b][0][c] (dgamma), dgb[b][1][c] (dbeta).
__global__ __launch_bounds__(256) void gn_bwd_finalize(const float2* __restrict__ part, int nchunk, int HW, int C,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ film_t, const float* __restrict__ film_a,
                                                       int ld_t, int ld_a, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       float* __restrict__ k1, float* __restrict__ k0,
                                                       float* __restrict__ dfilm_t, float* __restrict__ dfilm_a,
                                                       float* __restrict__ dgb, float* __restrict__ dgam_acc,
                                                       float* __restrict__ dbet_acc) {
  extern __shared__ float sm[];   // P1c[C], P2c[C]
  float* P1c = sm;
  float* P2c = sm + C;
  const int b = blockIdx.x, tid = threadIdx.x, cpg = C / G;
  for (int c = tid; c < C; c += 256) {
    float S1 = 0.f, S2 = 0.f, T1 = 0.f, T2 = 0.f, U1 = 0.f, U2 = 0.f, V1 = 0.f, V2 = 0.f;
    int k = 0;
    for (; k + 4 <= nchunk; k += 4) {      // four independent chains keep the partial loads in flight
      float2 v0 = part[((size_t)b * nchunk + k) * C + c], v1 = part[((size_t)b * nchunk + k + 1) * C + c];
      float2 v2 = part[((size_t)b * nchunk + k + 2) * C + c], v3 = part[((size_t)b * nchunk + k + 3) * C + c];
      S1 += v0.x; S2 += v0.y; T1 += v1.x; T2 += v1.y; U1 += v2.x; U2 += v2.y; V1 += v3.x; V2 += v3.y;
    }
    for (; k < nchunk; ++k) {
      float2 v = part[((size_t)b * nchunk + k) * C + c];
      S1 += v.x; S2 += v.y;
    }
    S1 = (S1 + T1) + (U1 + V1); S2 = (S2 + T2) + (U2 + V2);
    int g = c / cpg;
    float mu = mean[b * G + g], r = rstd[b * G + g];
    float D1 = S1, D2 = r * (S2 - mu * S1);
    float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
    float st = 0.f, bt = 0.f, sa = 0.f;
    if (film_t) { st = film_t[(size_t)b * ld_t + c]; bt = film_t[(size_t)b * ld_t + C + c]; }
    if (film_a) { sa = film_a[(size_t)b * ld_a + c]; }
    float f = (1.f + st) * (1.f + sa);
    float Gf = ga * D2 + be * D1, Ge = D1;
    if (dfilm_t) {
      dfilm_t[(size_t)b * 2 * C + c] = Gf * (1.f + sa);
      dfilm_t[(size_t)b * 2 * C + C + c] = Ge * (1.f + sa);
    }
    if (dfilm_a) {
      dfilm_a[(size_t)b * 2 * C + c] = Gf * (1.f + st) + Ge * bt;
      dfilm_a[(size_t)b * 2 * C + C + c] = Ge;
    }
    if (dgb) {
      dgb[((size_t)b * 2 + 0) * C + c] = f * D2;
      dgb[((size_t)b * 2 + 1) * C + c] = f * D1;
    }
    if (dgam_acc) atomicAdd(dgam_acc + c, f * D2);     // B adds per address: accumulate straight into the
    if (dbet_acc) atomicAdd(dbet_acc + c, f * D1);     // (pre-zeroed) parameter gradients, no column-sum pass
    P1c[c] = ga * f * D1;
    P2c[c] = ga * f * D2;
  }
  __syncthreads();
  if (tid < G) {
    float P1 = 0.f, P2 = 0.f;
    for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) { P1 += P1c[c]; P2 += P2c[c]; }
    float mu = mean[b * G + tid], r = rstd[b * G + tid];
    float invN = 1.f / ((float)HW * cpg);
    k1[b * G + tid] = -r * r * P2 * invN;
    k0[b * G + tid] = (-r * P1 + r * r * mu * P2) * invN;
  }
}

// elementwise: dx = sc*du + k1*x + k0 (+ dres).  grid (nchunk, B): a thread keeps one
// 16-byte channel slot for its whole pixel loop, so the per-(b,c) coefficients live in
// registers and the loop has no integer division.
template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_apply(const T* __restrict__ dA, const T* __restrict__ x,
                                                    const T* __restrict__ dres, T* __restrict__ dx,
                                                    const float* __restrict__ sc, const float* __restrict__ sh,
                                                    const float* __restrict__ k1, const float* __restrict__ k0,
                                                    int HW, int C, int chunk, int act, const uint64_t* seed,
                                                    uint32_t salt, uint32_t thr, float dscale) {
  constexpr int VE = Elem<T>::VE;
  const int vpp = C / VE, lanes = 256 / vpp, cpg = C / G, tid = threadIdx.x;
  const int b = blockIdx.y, v = tid % vpp, pl = tid / vpp;
  if (pl >= lanes) return;
  float scv[VE], shv[VE], k1v[VE], k0v[VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) {
    int c = v * VE + e, g = c / cpg;
    scv[e] = sc[(size_t)b * C + c]; shv[e] = sh[(size_t)b * C + c];
    k1v[e] = k1[b * G + g]; k0v[e] = k0[b * G + g];
  }
  const int pend = min(HW, (int)(blockIdx.x + 1) * chunk);
  for (int p = blockIdx.x * chunk + pl; p < pend; p += lanes) {
    size_t e0 = ((size_t)b * HW + p) * C + v * VE;
    float xv[VE], dav[VE], du[VE], o[VE];
    Vec16<T>::load(x + e0, xv);
    Vec16<T>::load(dA + e0, dav);
    du_vec<T>(dav, xv, scv, shv, act, seed, salt, thr, dscale, e0, du);
#pragma unroll
    for (int e = 0; e < VE; ++e) o[e] = scv[e] * du[e] + k1v[e] * xv[e] + k0v[e];
    if (dres) {
      float rv[VE];
      Vec16<T>::load(dres + e0, rv);
#pragma unroll
      for (int e = 0; e < VE; ++e) o[e] += rv[e];
    }
    Vec16<T>::store(dx + e0, o);
  }
}

// elementwise: a = dropout(SiLU(x*sc + sh))  (act 2)  or  x*sc + sh  (act 1); same geometry
// x2 != null: the input is the channel concatenation x [.., C1] | x2 [.., C - C1] (a skip pair read in place, C1 % VE == 0);
// `out` is dense over all C channels
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, const T* __restrict__ x2, int C1,
                                                       T* __restrict__ out,
                                                       const float* __restrict__ sc, const float* __restrict__ sh,
                                                       int HW, int C, int chunk, int act, const uint64_t* seed,
                                                       uint32_t salt, uint32_t thr, float dscale) {
  constexpr int VE = Elem<T>::VE;
  const int vpp = C / VE, lanes = 256 / vpp, tid = threadIdx.x;
  const int b = blockIdx.y, v = tid % vpp, pl = tid / vpp;
  if (pl >= lanes) return;
  const T* src = x;
  int spitch = C, sc0 = v * VE;                    // this thread's vector inside its source tensor
  if (x2) {
    if (v * VE < C1) spitch = C1;
    else { src = x2; spitch = C - C1; sc0 = v * VE - C1; }
  }
  float scv[VE], shv[VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) {
    scv[e] = sc[(size_t)b * C + v * VE + e];
    shv[e] = sh[(size_t)b * C + v * VE + e];
  }
  const int pend = min(HW, (int)(blockIdx.x + 1) * chunk);
  for (int p = blockIdx.x * chunk + pl; p < pend; p += lanes) {
    size_t e0 = ((size_t)b * HW + p) * C + v * VE;
    float xv[VE];
    Vec16<T>::load(src + ((size_t)b * HW + p) * spitch + sc0, xv);
    const uint32_t h = (act == 2 && seed) ? idf_vec_hash(*seed, salt, e0 >> 3) : 0u;
    const int l0 = (int)(e0 & 7);
    idf_act_vec<VE>(xv, scv, shv, act, seed != nullptr, h, l0, thr, dscale);
    Vec16<T>::store(out + e0, xv);
  }
}

// ------------------------------------------------- one-launch forms for small samples
// When a (sample, channel slice) fits one workgroup's registers the whole GroupNorm-FiLM-SiLU-dropout
// pass -- statistics, coefficient fold and apply -- is ONE launch, x read once and held in
// registers; likewise its backward (S1/S2 sums, k1/k0, FiLM / gamma / beta gradients, dx).
// Groups are independent, so a sample is cut along channels into slices of whole groups
// (grid = B x C/CS): one block per sample would leave 7/8 of the CUs idle at B = 32.
constexpr int SNV = 8;        // 16-byte vectors per thread (template NV: 8, or 16 for the largest samples)
constexpr int SNV_MAX = 16;

template <typename T> __device__ __forceinline__ void unpack16(const uint4& r, float* o);
template <> __device__ __forceinline__ void unpack16<float>(const uint4& r, float* o) {
  o[0] = __uint_as_float(r.x); o[1] = __uint_as_float(r.y); o[2] = __uint_as_float(r.z); o[3] = __uint_as_float(r.w);
}
template <> __device__ __forceinline__ void unpack16<bf16_t>(const uint4& r, float* o) {
  uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) { o[2 * i] = __uint_as_float(w[i] << 16); o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
}

// per-wave partial sums of the slice's CS channels -> red[wave][CS][2]
template <int VE>
__device__ __forceinline__ void small_reduce(float (&s)[VE], float (&q)[VE], float* red, int vs, int CS, int v) {
  // lanes of a wave that share the channel slot v sit vs apart
  for (int off = 32; off >= vs; off >>= 1) {
#pragma unroll
    for (int e = 0; e < VE; ++e) { s[e] += __shfl_xor(s[e], off, 64); q[e] += __shfl_xor(q[e], off, 64); }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane < vs && v * VE < CS) {          // vs may exceed CS / VE (padding lanes of a non-power-of-two slice)
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      red[(wave * CS + v * VE + e) * 2] = s[e];
      red[(wave * CS + v * VE + e) * 2 + 1] = q[e];
    }
  }
}

template <typename T, int NV>
__global__ __launch_bounds__(1024) void gn_small_fwd(const T* __restrict__ x, const T* __restrict__ x2, int C1,
                                                     T* __restrict__ out,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const float* __restrict__ film_t, const float* __restrict__ film_a,
                                                     int ld_t, int ld_a, float eps, float* __restrict__ mean,
                                                     float* __restrict__ rstd, float* __restrict__ sc,
                                                     float* __restrict__ sh, int HW, int C, int CS, int vs, int act,
                                                     const uint64_t* seed, uint32_t salt, uint32_t thr, float dscale) {
  constexpr int VE = Elem<T>::VE;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int NT = blockDim.x, nw = NT >> 6, cpg = C / G, GS = CS / cpg;
  float* red = sm;                       // [nw][CS][2]
  float* chs = red + nw * CS * 2;        // [CS][2]
  float* gst = chs + CS * 2;             // [GS][2] mean, rstd
  float* cof = gst + GS * 2;             // [CS][2] sc, sh
  // vs lanes per pixel (a power of two >= CS / VE; the lanes beyond CS / VE idle: 192-channel tensors have
  // 24-channel slices = 3 vectors on 4 lanes)
  const int lanes = NT / vs, tid = threadIdx.x;
  const int b = blockIdx.x, c0 = blockIdx.y * CS, v = tid % vs, pl = tid / vs;
  const bool live = v * VE < CS;
  // the input may be the channel concatenation of two tensors (skip connection): x [.., C1] | x2 [.., C - C1],
  // never materialised -- each 16-byte vector is read from the tensor it lies in (C1 % VE == 0)
  const T* src = x;
  int spitch = C, sc0 = c0 + v * VE;
  if (x2) {
    if (sc0 < C1) spitch = C1;
    else { src = x2; spitch = C - C1; sc0 -= C1; }
  }
  uint4 xr[NV];                 // the sample's slice stays in registers, packed
  float s[VE], q[VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) s[e] = q[e] = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    int p = pl + k * lanes;
    if (p < HW && live) {
      xr[k] = *reinterpret_cast<const uint4*>(src + ((size_t)b * HW + p) * spitch + sc0);
      float xv[VE];
      unpack16<T>(xr[k], xv);
#pragma unroll
      for (int e = 0; e < VE; ++e) { s[e] += xv[e]; q[e] += xv[e] * xv[e]; }
    }
  }
  small_reduce<VE>(s, q, red, vs, CS, v);
  __syncthreads();
  for (int c = tid; c < CS; c += NT) {
    float a = 0.f, d = 0.f;
    for (int w = 0; w < nw; ++w) { a += red[(w * CS + c) * 2]; d += red[(w * CS + c) * 2 + 1]; }
    chs[c * 2] = a; chs[c * 2 + 1] = d;
  }
  __syncthreads();
  for (int gl = tid; gl < GS; gl += NT) {
    double a = 0.0, d = 0.0;
    for (int c = gl * cpg; c < (gl + 1) * cpg; ++c) { a += chs[c * 2]; d += chs[c * 2 + 1]; }
    double n = (double)HW * cpg, mu = a / n, var = d / n - mu * mu;
    if (var < 0.0) var = 0.0;
    float r = (float)(1.0 / sqrt(var + (double)eps));
    gst[gl * 2] = (float)mu; gst[gl * 2 + 1] = r;
    const int g = c0 / cpg + gl;
    mean[b * G + g] = (float)mu; rstd[b * G + g] = r;
  }
  __syncthreads();
  for (int cl = tid; cl < CS; cl += NT) {
    const int gl = cl / cpg, c = c0 + cl;
    float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
    float a = gst[gl * 2 + 1] * ga, d = be - gst[gl * 2] * a;
    if (film_t) { float f = 1.f + film_t[(size_t)b * ld_t + c]; a *= f; d = d * f + film_t[(size_t)b * ld_t + C + c]; }
    if (film_a) { float f = 1.f + film_a[(size_t)b * ld_a + c]; a *= f; d = d * f + film_a[(size_t)b * ld_a + C + c]; }
    cof[cl * 2] = a; cof[cl * 2 + 1] = d;
    sc[(size_t)b * C + c] = a; sh[(size_t)b * C + c] = d;
  }
  __syncthreads();
  float scv[VE], shv[VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) {
    scv[e] = live ? cof[(v * VE + e) * 2] : 0.f;
    shv[e] = live ? cof[(v * VE + e) * 2 + 1] : 0.f;
  }
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    int p = pl + k * lanes;
    if (p < HW && live) {
      size_t e0 = ((size_t)b * HW + p) * C + c0 + v * VE;
      const uint32_t h = (act == 2 && seed) ? idf_vec_hash(*seed, salt, e0 >> 3) : 0u;
      const int l0 = (int)(e0 & 7);
      float xv[VE];
      unpack16<T>(xr[k], xv);
      idf_act_vec<VE>(xv, scv, shv, act, seed != nullptr, h, l0, thr, dscale);
      Vec16<T>::store(out + e0, xv);
    }
  }
}

// KEEP > 0: at most KEEP vectors per thread and du = dA * act'(...) stays in registers between the two
// phases (no second read of dA, no second sigmoid / dropout-hash evaluation); KEEP = 0: up to SNV
// vectors per thread, du recomputed in the apply phase (register budget of a 1024-thread block).
template <typename T, int KEEP>
__global__ __launch_bounds__(1024) void gn_small_bwd(const T* __restrict__ dA, const T* __restrict__ x,
                                                     const T* __restrict__ x2, int C1,
                                                     const T* __restrict__ dres, const T* __restrict__ dres2,
                                                     T* __restrict__ dx, T* __restrict__ dx2,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const float* __restrict__ film_t, const float* __restrict__ film_a,
                                                     int ld_t, int ld_a, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ sc,
                                                     const float* __restrict__ sh, float* __restrict__ dfilm_t,
                                                     float* __restrict__ dfilm_a, float* __restrict__ dgb,
                                                     float* __restrict__ dgam_acc, float* __restrict__ dbet_acc, int HW, int C,
                                                     int CS, int vs, int act, const uint64_t* seed, uint32_t salt, uint32_t thr,
                                                     float dscale, int pre) {
  constexpr int VE = Elem<T>::VE;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int NT = blockDim.x, nw = NT >> 6, cpg = C / G, GS = CS / cpg;
  float* red = sm;                       // [nw][CS][2]
  float* pc = red + nw * CS * 2;         // [CS][2]  ga*f*D1, ga*f*D2
  float* kk = pc + CS * 2;               // [GS][2]  k1, k0
  const int lanes = NT / vs, tid = threadIdx.x;          // vs lanes per pixel, see gn_small_fwd
  const int b = blockIdx.x, c0 = blockIdx.y * CS, v = tid % vs, pl = tid / vs;
  const bool live = v * VE < CS;
  float scv[VE], shv[VE], s1[VE], s2[VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) {
    scv[e] = live ? sc[(size_t)b * C + c0 + v * VE + e] : 0.f;
    shv[e] = live ? sh[(size_t)b * C + c0 + v * VE + e] : 0.f;
    s1[e] = s2[e] = 0.f;
  }
  // the per-channel parameters of the coefficient phase are independent of the statistics: fetch them now, beside the
  // x / dA loads, instead of after the reduction (one dependent memory round trip less on the small maps, where the
  // launch is nothing but such round trips)
  constexpr bool PF = KEEP == 2;              // the small-map variant: registers to spare
  const bool pf_ok = PF && tid < CS;
  float pf[7] = {0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f};
  if (pf_ok) {
    const int c = c0 + tid, g = c / cpg;
    pf[0] = mean[b * G + g]; pf[1] = rstd[b * G + g];
    if (gamma) pf[2] = gamma[c];
    if (beta) pf[3] = beta[c];
    if (film_t) { pf[4] = film_t[(size_t)b * ld_t + c]; pf[5] = film_t[(size_t)b * ld_t + C + c]; }
    if (film_a) pf[6] = film_a[(size_t)b * ld_a + c];
  }
  float pg[2] = {0.f, 0.f};
  if (PF && tid < GS) { pg[0] = mean[b * G + c0 / cpg + tid]; pg[1] = rstd[b * G + c0 / cpg + tid]; }
  const T* src = x;             // two-source input (see gn_small_fwd); dx goes back to the matching tensor
  T* dst = dx;
  int spitch = C, sc0 = c0 + v * VE;
  if (x2) {
    if (sc0 < C1) spitch = C1;
    else { src = x2; dst = dx2; spitch = C - C1; sc0 -= C1; }
  }
  constexpr int NV = KEEP > 0 ? KEEP : (KEEP < 0 ? SNV_MAX : SNV);
  uint4 xr[KEEP < 0 ? 1 : NV];  // packed x stays in registers (KEEP < 0: nothing kept, x and dA are re-read)
  float duk[KEEP > 0 ? KEEP : 1][VE];
  uint4 rr[KEEP > 0 ? KEEP : 1];  // KEEP > 0: the residual-branch gradient is fetched with x and dA (one HBM latency, not two)
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    int p = pl + k * lanes;
    if (p < HW && live) {
      size_t e0 = ((size_t)b * HW + p) * C + c0 + v * VE;
      const uint4 xraw = *reinterpret_cast<const uint4*>(src + ((size_t)b * HW + p) * spitch + sc0);
      if (KEEP >= 0) xr[k] = xraw;
      if (KEEP > 0 && dres && pre) rr[k] = *reinterpret_cast<const uint4*>(dres + e0);
      float xv[VE], dav[VE], du[VE];
      unpack16<T>(xraw, xv);
      Vec16<T>::load(dA + e0, dav);
      du_vec<T>(dav, xv, scv, shv, act, seed, salt, thr, dscale, e0, du);
#pragma unroll
      for (int e = 0; e < VE; ++e) {
        s1[e] += du[e]; s2[e] += du[e] * xv[e];
        if (KEEP > 0) duk[k][e] = du[e];
      }
    }
  }
  small_reduce<VE>(s1, s2, red, vs, CS, v);
  __syncthreads();
  for (int cl = tid; cl < CS; cl += NT) {
    const int c = c0 + cl, g = c / cpg;
    float S1 = 0.f, S2 = 0.f;
    for (int w = 0; w < nw; ++w) { S1 += red[(w * CS + cl) * 2]; S2 += red[(w * CS + cl) * 2 + 1]; }
    float mu, r, ga, be, st = 0.f, bt = 0.f, sa = 0.f;
    if (cl == tid && pf_ok) {       // fetched before the statistics phase (see below): no memory round trip here
      mu = pf[0]; r = pf[1]; ga = pf[2]; be = pf[3]; st = pf[4]; bt = pf[5]; sa = pf[6];
    } else {
      mu = mean[b * G + g]; r = rstd[b * G + g];
      ga = gamma ? gamma[c] : 1.f; be = beta ? beta[c] : 0.f;
      if (film_t) { st = film_t[(size_t)b * ld_t + c]; bt = film_t[(size_t)b * ld_t + C + c]; }
      if (film_a) { sa = film_a[(size_t)b * ld_a + c]; }
    }
    float D1 = S1, D2 = r * (S2 - mu * S1);
    float f = (1.f + st) * (1.f + sa);
    float Gf = ga * D2 + be * D1, Ge = D1;
    if (dfilm_t) { dfilm_t[(size_t)b * 2 * C + c] = Gf * (1.f + sa); dfilm_t[(size_t)b * 2 * C + C + c] = Ge * (1.f + sa); }
    if (dfilm_a) { dfilm_a[(size_t)b * 2 * C + c] = Gf * (1.f + st) + Ge * bt; dfilm_a[(size_t)b * 2 * C + C + c] = Ge; }
    if (dgb) {
      dgb[((size_t)b * 2 + 0) * C + c] = f * D2;
      dgb[((size_t)b * 2 + 1) * C + c] = f * D1;
    }
    if (dgam_acc) atomicAdd(dgam_acc + c, f * D2);
    if (dbet_acc) atomicAdd(dbet_acc + c, f * D1);
    pc[cl * 2] = ga * f * D1; pc[cl * 2 + 1] = ga * f * D2;
  }
  __syncthreads();
  for (int gl = tid; gl < GS; gl += NT) {
    float P1 = 0.f, P2 = 0.f;
    for (int c = gl * cpg; c < (gl + 1) * cpg; ++c) { P1 += pc[c * 2]; P2 += pc[c * 2 + 1]; }
    const int g = c0 / cpg + gl;
    float mu, r;
    if (PF && gl == tid) { mu = pg[0]; r = pg[1]; }
    else { mu = mean[b * G + g]; r = rstd[b * G + g]; }
    float invN = 1.f / ((float)HW * cpg);
    kk[gl * 2] = -r * r * P2 * invN;
    kk[gl * 2 + 1] = (-r * P1 + r * r * mu * P2) * invN;
  }
  __syncthreads();
  float k1v[VE], k0v[VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) {
    int gl = (v * VE + e) / cpg;
    k1v[e] = live ? kk[gl * 2] : 0.f;
    k0v[e] = live ? kk[gl * 2 + 1] : 0.f;
  }
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    int p = pl + k * lanes;
    if (p < HW && live) {
      size_t e0 = ((size_t)b * HW + p) * C + c0 + v * VE;
      float xv[VE], du[VE], o[VE];
      if (KEEP < 0) {
        const uint4 xraw = *reinterpret_cast<const uint4*>(src + ((size_t)b * HW + p) * spitch + sc0);
        unpack16<T>(xraw, xv);
      } else {
        unpack16<T>(xr[KEEP < 0 ? 0 : k], xv);
      }
      if (KEEP > 0) {
#pragma unroll
        for (int e = 0; e < VE; ++e) du[e] = duk[k][e];
      } else {
        float dav[VE];
        Vec16<T>::load(dA + e0, dav);        // L2-hot second read
        du_vec<T>(dav, xv, scv, shv, act, seed, salt, thr, dscale, e0, du);
      }
#pragma unroll
      for (int e = 0; e < VE; ++e) o[e] = scv[e] * du[e] + k1v[e] * xv[e] + k0v[e];
      if (dres) {                 // gradient arriving over the block's residual / shortcut branch
        float rv[VE];
        if (KEEP > 0 && pre) unpack16<T>(rr[k], rv);
        else Vec16<T>::load(dres + e0, rv);
#pragma unroll
        for (int e = 0; e < VE; ++e) o[e] += rv[e];
      }
      if (dres2) {                // ... and over a skip connection that branched off the same input
        float rv[VE];
        Vec16<T>::load(dres2 + e0, rv);
#pragma unroll
        for (int e = 0; e < VE; ++e) o[e] += rv[e];
      }
      Vec16<T>::store(dst + ((size_t)b * HW + p) * spitch + sc0, o);
    }
  }
}

// ---------------------------------------------------------- streaming backward from conv-epilogue partials
// The data-gradient conv that produced dA already formed du = dA * act'(x*sc+sh) * mask and the per-channel sums
// (idf_conv_dgrad_du_bf16): what is left of the GroupNorm backward is a streaming pass with no reduction in it --
//   dx = A*du + K1*x + K0 (+ dres + dres2)
// grid (chunks, B).  Every block folds its image's partials into (A, K1, K0) per channel (gn_bwd_fold; the block of chunk
// 0 also stores dgamma / dbeta / dFiLM), then streams its pixel chunk: a thread keeps one 16-byte channel slot for its
// whole pixel loop.  x may be the never-materialised concatenation x | x2 (dx goes back to the matching tensor).
__global__ __launch_bounds__(256) void gn_bwd_apply_loop(const bf16_t* __restrict__ du, const bf16_t* __restrict__ x,
                                                         const bf16_t* __restrict__ x2, int C1,
                                                         const bf16_t* __restrict__ dres, const bf16_t* __restrict__ dres2,
                                                         bf16_t* __restrict__ dx, bf16_t* __restrict__ dx2, const GnFoldP f,
                                                         int chunk) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // cof [C][4] | pc [C][2]
  const int C = f.C, HW = f.HW, tid = threadIdx.x, b = blockIdx.y;
  float* cof = sm;
  const int vpp = C / 8, lanes = 256 / vpp, v = tid % vpp, pl = tid / vpp;
  const bool live = pl < lanes;
  const bf16_t* src = x;
  bf16_t* dst = dx;
  int spitch = C, sc0 = v * 8;
  if (x2) {
    if (sc0 < C1) spitch = C1;
    else { src = x2; dst = dx2; spitch = C - C1; sc0 -= C1; }
  }
  const int pend = min(HW, (int)(blockIdx.x + 1) * chunk);
  // the first TWO pixels' operands are on their way before the fold starts (they do not depend on it): the stream's first round
  // trips run beside the fold's instead of behind it, and the loop keeps two pixels per thread in flight (round 6: one pixel was
  // 16 KB per block against a ~2 us round trip)
  int p = blockIdx.x * chunk + pl;
  struct Px { uint4 xr, dr, rr, r2; };
  auto fetch = [&](int q_) {
    Px w;
    w.xr = w.dr = w.rr = w.r2 = make_uint4(0, 0, 0, 0);
    if (live && q_ < pend) {
      const size_t e0 = ((size_t)b * HW + q_) * C + v * 8, es = ((size_t)b * HW + q_) * spitch + sc0;
      w.xr = *reinterpret_cast<const uint4*>(src + es);
      w.dr = *reinterpret_cast<const uint4*>(du + e0);
      if (dres) w.rr = *reinterpret_cast<const uint4*>(dres + e0);
      if (dres2) w.r2 = *reinterpret_cast<const uint4*>(dres2 + e0);
    }
    return w;
  };
  Px cur = fetch(p), nxt = fetch(p + lanes);
  gn_bwd_fold<256>(f, b, blockIdx.x == 0, cof, sm + 4 * C, tid);
  if (!live) return;
  float av[8], k1v[8], k0v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float4 t4 = *reinterpret_cast<const float4*>(cof + 4 * (v * 8 + e));
    av[e] = t4.x; k1v[e] = t4.y; k0v[e] = t4.z;
  }
  for (; p < pend; p += lanes) {
    const size_t es = ((size_t)b * HW + p) * spitch + sc0;
    float xv[8], dv[8], o[8], rv[8];
    unpack16<bf16_t>(cur.xr, xv);
    unpack16<bf16_t>(cur.dr, dv);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = av[e] * dv[e] + k1v[e] * xv[e] + k0v[e];
    if (dres) {
      unpack16<bf16_t>(cur.rr, rv);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] += rv[e];
    }
    if (dres2) {
      unpack16<bf16_t>(cur.r2, rv);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] += rv[e];
    }
    cur = nxt;
    nxt = fetch(p + 2 * lanes);               // two pixels ahead, before this pixel's store
    Vec16<bf16_t>::store(dst + es, o);
  }
}

// Small tensors (the launch is a handful of dependent round trips): the streamed operands do not depend on the
// coefficients, so a thread fetches its NV vectors of each FIRST and the fold runs beside the loads instead of in front of
// them (16x16 x 128 channels at B = 32: 6.6 -> 5.9 us; big tensors lose occupancy to the registers: 16.6 -> 18.5 us at
// 64x64 x 64 channels -- they keep the loop form above).
template <int NV>
__global__ __launch_bounds__(256) void gn_bwd_apply_part(const bf16_t* __restrict__ du, const bf16_t* __restrict__ x,
                                                         const bf16_t* __restrict__ x2, int C1,
                                                         const bf16_t* __restrict__ dres, const bf16_t* __restrict__ dres2,
                                                         bf16_t* __restrict__ dx, bf16_t* __restrict__ dx2, const GnFoldP f,
                                                         int chunk) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // cof [C][4] | pc [C][2]
  const int C = f.C, HW = f.HW, tid = threadIdx.x, b = blockIdx.y;
  float* cof = sm;
  const int vpp = C / 8, lanes = 256 / vpp, v = tid % vpp, pl = tid / vpp;
  const bool live = pl < lanes;
  const bf16_t* src = x;
  bf16_t* dst = dx;
  int spitch = C, sc0 = v * 8;
  if (x2) {
    if (sc0 < C1) spitch = C1;
    else { src = x2; dst = dx2; spitch = C - C1; sc0 -= C1; }
  }
  const int p0 = blockIdx.x * chunk + pl, pend = min(HW, (int)(blockIdx.x + 1) * chunk);
  uint4 xr[NV], dr[NV], rr[NV], r2[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int p = p0 + k * lanes;
    xr[k] = dr[k] = rr[k] = r2[k] = make_uint4(0, 0, 0, 0);
    if (live && p < pend) {
      const size_t e0 = ((size_t)b * HW + p) * C + v * 8, es = ((size_t)b * HW + p) * spitch + sc0;
      xr[k] = *reinterpret_cast<const uint4*>(src + es);
      dr[k] = *reinterpret_cast<const uint4*>(du + e0);
      if (dres) rr[k] = *reinterpret_cast<const uint4*>(dres + e0);
      if (dres2) r2[k] = *reinterpret_cast<const uint4*>(dres2 + e0);
    }
  }
  gn_bwd_fold<256>(f, b, blockIdx.x == 0, cof, sm + 4 * C, tid);
  if (!live) return;
  float av[8], k1v[8], k0v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float4 t4 = *reinterpret_cast<const float4*>(cof + 4 * (v * 8 + e));
    av[e] = t4.x; k1v[e] = t4.y; k0v[e] = t4.z;
  }
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int p = p0 + k * lanes;
    if (p < pend) {
      const size_t es = ((size_t)b * HW + p) * spitch + sc0;
      float xv[8], dv[8], o[8];
      unpack16<bf16_t>(xr[k], xv);
      unpack16<bf16_t>(dr[k], dv);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = av[e] * dv[e] + k1v[e] * xv[e] + k0v[e];
      if (dres) {
        float rv[8];
        unpack16<bf16_t>(rr[k], rv);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += rv[e];
      }
      if (dres2) {
        float rv[8];
        unpack16<bf16_t>(r2[k], rv);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += rv[e];
      }
      Vec16<bf16_t>::store(dst + es, o);
    }
  }
}

// Slice plan for the one-launch kernels: CS channels per block (whole groups, whole 16-byte vectors,
// CS/VE a power of two <= 64) and the block size.  Largest slice that still gives >= ~256 blocks.
struct SmallPlan { int CS, NT, VS; };   // VS = lanes per pixel (power of two >= CS / VE)
bool small_plan(int B, int HW, int C, int VE, SmallPlan* plan, int C1 = 0) {
  static const int max_hw = 4096;
  static const int want = 256;
  if (C % G || C % VE || C > 1024 || HW > max_hw || HW < 1) return false;
  const int cpg = C / G;
  int unit = cpg;                       // lcm(cpg, VE)
  while (unit % VE) unit += cpg;
  int best = 0;
  if (C1 > 0 && (C1 % VE)) return false;         // a 16-byte vector must lie inside one of the two sources
  auto lanes_for = [&](int cs) { int v = 1; while (v < cs / VE) v *= 2; return v; };
  for (int cs = unit; cs <= C; cs *= 2) {
    const int vs = lanes_for(cs);
    if (C % cs || vs > 64) continue;
    if ((long)HW * vs > 1024L * SNV_MAX) break;
    if (!best || (long)B * (C / cs) >= want) best = cs;
  }
  if (!best) return false;
  // big batches of big samples (DDIM sampling at B = 256): slices this narrow (<= 32-byte segments) run at
  // ~2.4 TB/s, the fully coalesced three-launch path at ~4 TB/s, and launch costs no longer matter
  // (not for a two-source input: there the alternative is materialising the concatenation first)
  if (C1 == 0 && best * (VE == 8 ? 2 : 4) <= 32 && (long)B * HW * C * (VE == 8 ? 2 : 4) >= (64L << 20)) return false;
  plan->VS = lanes_for(best);
  const int vectors = HW * plan->VS;
  int nt = ((vectors + 1) / 2 + 63) / 64 * 64;
  if (nt < 64) nt = 64;
  if (nt > 1024) nt = 1024;
  plan->CS = best; plan->NT = nt;
  return true;
}

int pick_chunk(int B, int HW) {
  // aim for >= ~1024 blocks, chunks of at least 64 pixels
  int nchunk = idf_cdiv(1024, B);
  int chunk = idf_cdiv(HW, nchunk);
  if (chunk < 64) chunk = 64;
  if (chunk > HW) chunk = HW;
  return chunk;
}

// elementwise passes: more, smaller blocks (no partial buffers to pay for)
int pick_chunk_ew(int B, int HW) {
  int nchunk = idf_cdiv(4096, B);
  int chunk = idf_cdiv(HW, nchunk);
  if (chunk < 32) chunk = 32;
  if (chunk > HW) chunk = HW;
  return chunk;
}

}  // namespace

extern "C" int idf_gn_workspace_floats(int B, int HW, int C) {
  int chunk = pick_chunk(B, HW);
  int nchunk = idf_cdiv(HW, chunk);
  return B * nchunk * (C > G ? C : G) * 2;
}

// Per-channel statistics partials of a tensor: part [B][T][C][2] (sum, sum of squares over T pixel chunks),
// T = idf_gn_partials_chunks(B, HW).  What the conv launches write as st_out, for tensors produced elsewhere.
extern "C" int idf_gn_partials_chunks(int B, int HW) { return idf_cdiv(HW, pick_chunk(B, HW)); }

extern "C" int idf_gn_partials(const void* x, float* part, int B, int HW, int C, int dtype, void* stream) {
  if (B == 0) return IDF_OK;
  int VE = dtype == IDF_F32 ? 4 : 8;
  if (C % VE || C / VE > 256) IDF_FAIL(IDF_ERR_UNSUPPORTED, "gn_partials: C=%d unsupported", C);
  hipStream_t st = (hipStream_t)stream;
  int chunk = pick_chunk(B, HW), nchunk = idf_cdiv(HW, chunk);
  int lanes = 256 / (C / VE);
  size_t lds = (size_t)lanes * C * 2 * sizeof(float);
  dim3 g(nchunk, B);
  if (dtype == IDF_F32)
    hipLaunchKernelGGL(gn_partials_kernel<float>, g, dim3(256), lds, st, (const float*)x, (float2*)part, HW, C, chunk);
  else
    hipLaunchKernelGGL(gn_partials_kernel<bf16_t>, g, dim3(256), lds, st, (const bf16_t*)x, (float2*)part, HW, C, chunk);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}

extern "C" int idf_gn_coef_fwd(const void* x, const float* gamma, const float* beta, const float* film_t,
                               const float* film_a, int ld_t, int ld_a, float eps, float* mean, float* rstd, float* sc, float* sh,
                               float* workspace, int B, int HW, int C, int dtype, void* stream) {
  if (B == 0) return IDF_OK;
  int VE = dtype == IDF_F32 ? 4 : 8;
  if (C % G || C % VE || C / VE > 256) IDF_FAIL(IDF_ERR_UNSUPPORTED, "groupnorm: C=%d unsupported", C);
  hipStream_t st = (hipStream_t)stream;
  int chunk = pick_chunk(B, HW), nchunk = idf_cdiv(HW, chunk);
  int lanes = 256 / (C / VE);
  size_t lds = (size_t)(lanes + 1) * C * 2 * sizeof(float);
  dim3 g(nchunk, B);
  if (dtype == IDF_F32)
    hipLaunchKernelGGL(gn_stats_partial<float>, g, dim3(256), lds, st, (const float*)x, (float2*)workspace, HW, C, chunk);
  else
    hipLaunchKernelGGL(gn_stats_partial<bf16_t>, g, dim3(256), lds, st, (const bf16_t*)x, (float2*)workspace, HW, C, chunk);
  IDF_CHECK_LAUNCH();
  hipLaunchKernelGGL(gn_finalize, dim3(B), dim3(256), 0, st, (const float2*)workspace, nchunk, HW, C, gamma, beta,
                     film_t, film_a, ld_t ? ld_t : 2 * C, ld_a ? ld_a : 2 * C, eps, mean, rstd, sc, sh);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}

extern "C" int idf_gn_coef_bwd(const void* dA, const void* x, const void* dres, void* dx, const float* gamma,
                               const float* beta, const float* film_t, const float* film_a, int ld_t, int ld_a,
                               const float* mean, const float* rstd, const float* sc, const float* sh, float* dfilm_t, float* dfilm_a,
                               float* dgb, float* dgamma_acc, float* dbeta_acc, float* k1, float* k0, float* workspace,
                               const uint64_t* seed, uint32_t salt, float p_drop, int act, int B, int HW, int C,
                               int dtype, void* stream) {
  if (B == 0) return IDF_OK;
  int VE = dtype == IDF_F32 ? 4 : 8;
  if (C % G || C % VE || C / VE > 256) IDF_FAIL(IDF_ERR_UNSUPPORTED, "groupnorm bwd: C=%d unsupported", C);
  hipStream_t st = (hipStream_t)stream;
  int chunk = pick_chunk(B, HW), nchunk = idf_cdiv(HW, chunk);
  int lanes = 256 / (C / VE);
  size_t lds = (size_t)lanes * C * 2 * sizeof(float);
  uint32_t thr = idf_drop_thresh(p_drop);
  float dscale = 1.0f / (1.0f - (float)thr / 65536.0f);
  const uint64_t* sd = (act == 2 && p_drop > 0.f) ? seed : nullptr;
  dim3 g(nchunk, B);
  if (dtype == IDF_F32)
    hipLaunchKernelGGL(gn_bwd_partial<float>, g, dim3(256), lds, st, (const float*)dA, (const float*)x, sc, sh,
                       (float2*)workspace, HW, C, chunk, act, sd, salt, thr, dscale);
  else
    hipLaunchKernelGGL(gn_bwd_partial<bf16_t>, g, dim3(256), lds, st, (const bf16_t*)dA, (const bf16_t*)x, sc, sh,
                       (float2*)workspace, HW, C, chunk, act, sd, salt, thr, dscale);
  IDF_CHECK_LAUNCH();
  hipLaunchKernelGGL(gn_bwd_finalize, dim3(B), dim3(256), 2 * C * sizeof(float), st, (const float2*)workspace, nchunk,
                     HW, C, gamma, beta, film_t, film_a, ld_t ? ld_t : 2 * C, ld_a ? ld_a : 2 * C, mean, rstd, k1, k0,
                     dfilm_t, dfilm_a, dgb, dgamma_acc, dbeta_acc);
  IDF_CHECK_LAUNCH();
  int achunk = pick_chunk_ew(B, HW);
  dim3 ga(idf_cdiv(HW, achunk), B);
  if (dtype == IDF_F32)
    hipLaunchKernelGGL(gn_bwd_apply<float>, ga, dim3(256), 0, st, (const float*)dA, (const float*)x,
                       (const float*)dres, (float*)dx, sc, sh, k1, k0, HW, C, achunk, act, sd, salt, thr, dscale);
  else
    hipLaunchKernelGGL(gn_bwd_apply<bf16_t>, ga, dim3(256), 0, st, (const bf16_t*)dA, (const bf16_t*)x,
                       (const bf16_t*)dres, (bf16_t*)dx, sc, sh, k1, k0, HW, C, achunk, act, sd, salt, thr, dscale);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}

// a = act(x*sc+sh): the GroupNorm-apply + FiLM + SiLU + dropout pass (one read, one write)
static int gn_apply_impl(const void* x, const void* x2, int C1, void* out, const float* sc, const float* sh, const uint64_t* seed,
                         uint32_t salt, float p_drop, int act, int B, int HW, int C, int dtype, void* stream) {
  if (B == 0) return IDF_OK;
  int VE = dtype == IDF_F32 ? 4 : 8;
  if (C % VE) IDF_FAIL(IDF_ERR_UNSUPPORTED, "gn_apply: C=%d unsupported", C);
  if (x2 && (C1 <= 0 || C1 >= C || (C1 % VE))) IDF_FAIL(IDF_ERR_UNSUPPORTED, "gn_apply: C1=%d of %d unsupported", C1, C);
  if (!x2) C1 = C;
  if (act != 1 && act != 2) IDF_FAIL(IDF_ERR_BADARG, "gn_apply: act must be 1 or 2");
  hipStream_t st = (hipStream_t)stream;
  uint32_t thr = idf_drop_thresh(p_drop);
  float dscale = 1.0f / (1.0f - (float)thr / 65536.0f);
  const uint64_t* sd = (act == 2 && p_drop > 0.f) ? seed : nullptr;
  if (C / VE > 256) IDF_FAIL(IDF_ERR_UNSUPPORTED, "gn_apply: C=%d unsupported", C);
  int achunk = pick_chunk_ew(B, HW);
  dim3 ga(idf_cdiv(HW, achunk), B);
  if (dtype == IDF_F32)
    hipLaunchKernelGGL(gn_apply_kernel<float>, ga, dim3(256), 0, st, (const float*)x, (const float*)x2, C1, (float*)out, sc, sh, HW, C,
                       achunk, act, sd, salt, thr, dscale);
  else
    hipLaunchKernelGGL(gn_apply_kernel<bf16_t>, ga, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)x2, C1, (bf16_t*)out, sc, sh, HW, C,
                       achunk, act, sd, salt, thr, dscale);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}

extern "C" int idf_gn_apply(const void* x, void* out, const float* sc, const float* sh, const uint64_t* seed,
                            uint32_t salt, float p_drop, int act, int B, int HW, int C, int dtype, void* stream) {
  return gn_apply_impl(x, nullptr, 0, out, sc, sh, seed, salt, p_drop, act, B, HW, C, dtype, stream);
}

// the same pass over the never-materialised concatenation x [.., C1] | x2 [.., C - C1] of a skip pair (models.py:321)
extern "C" int idf_gn_apply2(const void* x, const void* x2, int C1, void* out, const float* sc, const float* sh,
                             const uint64_t* seed, uint32_t salt, float p_drop, int act, int B, int HW, int C, int dtype,
                             void* stream) {
  if (!x2) IDF_FAIL(IDF_ERR_BADARG, "gn_apply2: second source missing");
  return gn_apply_impl(x, x2, C1, out, sc, sh, seed, salt, p_drop, act, B, HW, C, dtype, stream);
}

// 1 when idf_gn_fused_fwd / idf_gn_fused_bwd cover this shape (the host picks the path with it).
extern "C" int idf_gn_fused_ok(int B, int HW, int C, int C1, int dtype) {
  SmallPlan sp;
  if (C1 < 0 || C1 >= C) return 0;
  return small_plan(B, HW, C, dtype == IDF_F32 ? 4 : 8, &sp, C1) ? 1 : 0;
}

// One-launch GroupNorm + FiLM fold + apply for small samples (statistics, sc/sh, a = act(x*sc+sh)).
// IDF_ERR_UNSUPPORTED when a sample does not fit one workgroup: use idf_gn_coef_fwd + idf_gn_apply.
extern "C" int idf_gn_fused_fwd(const void* x, const void* x2, int C1, void* out, const float* gamma, const float* beta, const float* film_t,
                                const float* film_a, int ld_t, int ld_a, float eps, float* mean, float* rstd,
                                float* sc, float* sh, const uint64_t* seed, uint32_t salt, float p_drop, int act,
                                int B, int HW, int C, int dtype, void* stream) {
  int VE = dtype == IDF_F32 ? 4 : 8;
  SmallPlan sp;
  if (!x2) C1 = 0;
  if (C1 < 0 || C1 >= C || !small_plan(B, HW, C, VE, &sp, C1))
    IDF_FAIL(IDF_ERR_UNSUPPORTED, "gn_fused_fwd: HW=%d C=%d C1=%d does not fit one workgroup", HW, C, C1);
  if (act != 1 && act != 2) IDF_FAIL(IDF_ERR_BADARG, "gn_fused_fwd: act must be 1 or 2");
  if (B == 0) return IDF_OK;
  uint32_t thr = idf_drop_thresh(p_drop);
  float dscale = 1.0f / (1.0f - (float)thr / 65536.0f);
  const uint64_t* sd = (act == 2 && p_drop > 0.f) ? seed : nullptr;
  size_t lds = ((size_t)(sp.NT / 64) * sp.CS * 2 + sp.CS * 2 + G * 2 + sp.CS * 2) * sizeof(float);
  ld_t = ld_t ? ld_t : 2 * C; ld_a = ld_a ? ld_a : 2 * C;
  hipStream_t st = (hipStream_t)stream;
  const int nvt_f = idf_cdiv((long)HW * sp.VS, sp.NT);           // vectors per thread
#define IDF_GN_FWD(T, N)                                                                                          \
  hipLaunchKernelGGL((gn_small_fwd<T, N>), dim3(B, C / sp.CS), dim3(sp.NT), lds, st, (const T*)x, (const T*)x2, C1, \
                     (T*)out, gamma, beta, film_t, film_a, ld_t, ld_a, eps, mean, rstd, sc, sh, HW, C, sp.CS, sp.VS, \
                     act, sd, salt, thr, dscale)
  if (dtype == IDF_F32) { if (nvt_f > SNV) IDF_GN_FWD(float, SNV_MAX); else IDF_GN_FWD(float, SNV); }
  else { if (nvt_f > SNV) IDF_GN_FWD(bf16_t, SNV_MAX); else IDF_GN_FWD(bf16_t, SNV); }
#undef IDF_GN_FWD
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}

extern "C" int idf_gn_fused_bwd(const void* dA, const void* x, const void* x2, int C1, const void* dres, const void* dres2,
                                void* dx, void* dx2,
                                const float* gamma, const float* beta,
                                const float* film_t, const float* film_a, int ld_t, int ld_a, const float* mean,
                                const float* rstd, const float* sc, const float* sh, float* dfilm_t, float* dfilm_a,
                                float* dgb, float* dgamma_acc, float* dbeta_acc, const uint64_t* seed, uint32_t salt,
                                float p_drop, int act, int B, int HW,
                                int C, int dtype, void* stream) {
  int VE = dtype == IDF_F32 ? 4 : 8;
  SmallPlan sp;
  if (!x2) C1 = 0;
  if (C1 < 0 || C1 >= C || (x2 && !dx2) || !small_plan(B, HW, C, VE, &sp, C1))
    IDF_FAIL(IDF_ERR_UNSUPPORTED, "gn_fused_bwd: HW=%d C=%d C1=%d does not fit one workgroup", HW, C, C1);
  if (B == 0) return IDF_OK;
  uint32_t thr = idf_drop_thresh(p_drop);
  float dscale = 1.0f / (1.0f - (float)thr / 65536.0f);
  const uint64_t* sd = (act == 2 && p_drop > 0.f) ? seed : nullptr;
  size_t lds = ((size_t)(sp.NT / 64) * sp.CS * 2 + sp.CS * 2 + G * 2) * sizeof(float);
  ld_t = ld_t ? ld_t : 2 * C; ld_a = ld_a ? ld_a : 2 * C;
  hipStream_t st = (hipStream_t)stream;
  static const int keep_env = 1;
  const int nvt = idf_cdiv((long)HW * sp.VS, sp.NT);             // vectors per thread
  const bool keep = keep_env && nvt <= 4;
  const bool keep2 = keep && nvt <= 2;      // small maps: half the kept vectors, the coefficient phase's parameters prefetched
  static const int pre = 1;
#define IDF_GN_BWD(T, K)                                                                                          \
  hipLaunchKernelGGL((gn_small_bwd<T, K>), dim3(B, C / sp.CS), dim3(sp.NT), lds, st, (const T*)dA, (const T*)x,   \
                     (const T*)x2, C1, (const T*)dres, (const T*)dres2, (T*)dx, (T*)dx2, gamma, beta, film_t, film_a, ld_t, ld_a, mean, rstd, sc, sh, dfilm_t, \
                     dfilm_a, dgb, dgamma_acc, dbeta_acc, HW, C, sp.CS, sp.VS, act, sd, salt, thr, dscale, pre)
  if (dtype == IDF_F32) {
    if (keep2) IDF_GN_BWD(float, 2); else if (keep) IDF_GN_BWD(float, 4); else if (nvt > SNV) IDF_GN_BWD(float, -1); else IDF_GN_BWD(float, 0);
  } else {
    if (keep2) IDF_GN_BWD(bf16_t, 2); else if (keep) IDF_GN_BWD(bf16_t, 4); else if (nvt > SNV) IDF_GN_BWD(bf16_t, -1); else IDF_GN_BWD(bf16_t, 0);
  }
#undef IDF_GN_BWD
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}

// dx (and the GroupNorm's parameter / FiLM gradients) from du and the per-channel partials a data-gradient conv left
// behind (idf_conv_dgrad_du_bf16): bf16, C % 32 == 0, C <= 1024.  part [B][T][C][2]; x may be the pair x [.., C1] |
// x2 [.., C - C1] (then dx2 receives the second part's gradient).  Side outputs as idf_gn_fused_bwd.
extern "C" int idf_gn_bwd_apply(const void* du, const float* part, int T, const void* x, const void* x2, int C1,
                                const void* dres, const void* dres2, void* dx, void* dx2, const float* gamma,
                                const float* beta, const float* film_t, const float* film_a, int ld_t, int ld_a,
                                const float* mean, const float* rstd, const float* sc, float* dfilm_t, float* dfilm_a,
                                float* dgb, float* dgamma_acc, float* dbeta_acc, int B, int HW, int C, void* stream) {
  if (!x2) C1 = 0;
  if (C % G || C % 8 || C > 1024 || C / 8 > 256 || T < 1 || C1 < 0 || C1 >= C || (C1 % 8) || (x2 && !dx2))
    IDF_FAIL(IDF_ERR_UNSUPPORTED, "gn_bwd_apply: C=%d C1=%d T=%d not covered", C, C1, T);
  if (!du || !part || !x || !dx || !mean || !rstd || !sc) IDF_FAIL(IDF_ERR_BADARG, "gn_bwd_apply: null argument");
  if (B == 0) return IDF_OK;
  GnFoldP f;
  f.part = part; f.T = T; f.mean = mean; f.rstd = rstd; f.sc = sc; f.gamma = gamma; f.beta = beta;
  f.film_t = film_t; f.film_a = film_a; f.ld_t = ld_t ? ld_t : 2 * C; f.ld_a = ld_a ? ld_a : 2 * C;
  f.dfilm_t = dfilm_t; f.dfilm_a = dfilm_a; f.dgb = dgb; f.dgam = dgamma_acc; f.dbet = dbeta_acc; f.C = C; f.HW = HW;
  const int lanes = 256 / (C / 8);
  const size_t lds = (size_t)C * 6 * sizeof(float);
  static const long small_max = (1L << 22);
  if ((long)B * HW * C <= small_max) {
    // a thread streams NV (4, or 2 to keep >= 256 blocks) pixels of one 16-byte channel slot, all fetched before the fold
    int nv = (long)B * idf_cdiv(HW, lanes * 4) >= 256 ? 4 : 2;
    int chunk = lanes * nv;
    if (chunk > HW) chunk = HW;
    dim3 g(idf_cdiv(HW, chunk), B);
    if (nv == 4)
      hipLaunchKernelGGL(gn_bwd_apply_part<4>, g, dim3(256), lds, (hipStream_t)stream, (const bf16_t*)du, (const bf16_t*)x,
                         (const bf16_t*)x2, C1, (const bf16_t*)dres, (const bf16_t*)dres2, (bf16_t*)dx, (bf16_t*)dx2, f, chunk);
    else
      hipLaunchKernelGGL(gn_bwd_apply_part<2>, g, dim3(256), lds, (hipStream_t)stream, (const bf16_t*)du, (const bf16_t*)x,
                         (const bf16_t*)x2, C1, (const bf16_t*)dres, (const bf16_t*)dres2, (bf16_t*)dx, (bf16_t*)dx2, f, chunk);
  } else {
    // ~4 blocks per CU, all resident at once: the fold in front of every block's stream is paid once, in parallel
#ifndef IDF_GNAPPLY_WANT
#define IDF_GNAPPLY_WANT 1024
#endif
    static const int want = IDF_GNAPPLY_WANT;
    int chunk = idf_cdiv(HW, idf_cdiv(want, B));
    if (chunk < lanes) chunk = lanes;
    if (chunk > HW) chunk = HW;
    dim3 g(idf_cdiv(HW, chunk), B);
    hipLaunchKernelGGL(gn_bwd_apply_loop, g, dim3(256), lds, (hipStream_t)stream, (const bf16_t*)du, (const bf16_t*)x,
                       (const bf16_t*)x2, C1, (const bf16_t*)dres, (const bf16_t*)dres2, (bf16_t*)dx, (bf16_t*)dx2, f, chunk);
  }
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}

// ------------------------------------------------------------------ deterministic mode: GroupNorm affine gradients, batched
// Every GroupNorm backward of a pass leaves per-image rows dgb [B][2][C] (what the kernels above write when no accumulation slots
// are given); ONE launch at the end of the pass adds each stage's rows, in image order, into its gamma / beta gradient slots
// (round 5: the per-stage column sums were 124 launches / 0.6 ms of a deterministic CelebA step).  table: n entries of
// {rows, dgamma, dbeta, B, C}; grid (n, ceil(2 maxC / 256)).
struct GnRowsDesc { const float* rows; float* dgam; float* dbet; int B, C; };

__global__ __launch_bounds__(256) void gn_param_reduce_kernel(const GnRowsDesc* __restrict__ tab) {
  const GnRowsDesc d = tab[blockIdx.x];
  const int j = blockIdx.y * 256 + threadIdx.x;
  if (j >= 2 * d.C) return;
  float s0 = 0.f, s1 = 0.f;
  int b = 0;
  for (; b + 1 < d.B; b += 2) { s0 += d.rows[(size_t)b * 2 * d.C + j]; s1 += d.rows[(size_t)(b + 1) * 2 * d.C + j]; }
  if (b < d.B) s0 += d.rows[(size_t)b * 2 * d.C + j];
  float* dst = j < d.C ? d.dgam + j : d.dbet + (j - d.C);
  *dst += s0 + s1;
}

extern "C" int idf_gn_rows_desc_bytes(void) { return (int)sizeof(GnRowsDesc); }

extern "C" int idf_gn_param_reduce_batched(const void* table, int n, int max_c, void* stream) {
  if (n <= 0) return IDF_OK;
  if (!table || max_c <= 0) IDF_FAIL(IDF_ERR_BADARG, "gn_param_reduce_batched: null table / max_c");
  hipLaunchKernelGGL(gn_param_reduce_kernel, dim3(n, idf_cdiv(2 * max_c, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const GnRowsDesc*)table);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}
