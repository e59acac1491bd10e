// HBM-bound elementwise / reduction kernels of the path: forward-diffusion
// q_sample (models.py:702-704), time-table gather (modules.py:23), SiLU, the
// epsilon-MSE + reconstruction loss (models.py:640-646), sampler updates
// (sampling.py:29-37, 52-59, 71-72), RBF-kernel MMD (utils.py:74-90), column sums
// (bias / gamma / beta gradients), 2x2 sum-pool (UpSample data gradient) and the
// conv weight shadow pack.
//
// Arithmetic that the reference performs as separate fp32 torch ops is kept as
// separately rounded operations (__fmul_rn / __fadd_rn: no FMA contraction) so the
// fp32 results are bit-identical to the CPU path.
#include "idf_common.h"

// separately rounded mul/add (no FMA contraction) so fp32 results match the CPU path bit for bit
#pragma clang fp contract(off)

namespace {

template <typename T> __device__ __forceinline__ float ldT(const void* p, size_t i) {
  return Elem<T>::ld(reinterpret_cast<const T*>(p) + i);
}

// -------------------------------------------------------------- q_sample
// x, eps fp32 (any layout, elementwise); per-sample gather of sqrt(alpha_bar), sqrt(1-alpha_bar)
// (tables computed by the same torch CPU ops the reference uses, so the gather is bit-exact).
template <typename TO>
__global__ void qsample_kernel(const float* __restrict__ x, const float* __restrict__ eps,
                               const long* __restrict__ idx, const float* __restrict__ sqrt_ab,
                               const float* __restrict__ sqrt_1mab, float* __restrict__ xt32,
                               TO* __restrict__ xt, long per, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    long b = i / per;
    float a = sqrt_ab[idx[b]], s = sqrt_1mab[idx[b]];
    float v = __fadd_rn(__fmul_rn(a, x[i]), __fmul_rn(s, eps[i]));
    if (xt32) xt32[i] = v;
    if (xt) Elem<TO>::st(xt + i, v);
  }
}

// Input pipeline on the device (reference data.py:149-171: ToTensor -> RandomHorizontalFlip -> Normalize(0.5, 0.5)):
// uint8 NHWC image bytes -> fp32 NHWC-dense activations (x / 255 - 0.5) / 0.5, each op rounded separately
// (this file is compiled with -ffp-contract=off), so the result is bit-identical to the torchvision chain;
// flip[b] != 0 mirrors sample b horizontally.
__global__ void prep_u8_kernel(const uint8_t* __restrict__ src, const uint8_t* __restrict__ flip,
                               float* __restrict__ dst, int H, int W, int C, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    long pix = i / C;
    int c = (int)(i - pix * C);
    long row = pix / W;
    int w = (int)(pix - row * W);
    long b = row / H;
    long j = i;
    if (flip && flip[b]) j = (row * W + (W - 1 - w)) * C + c;
    float v = __fdiv_rn((float)src[j], 255.0f);
    dst[i] = __fdiv_rn(__fsub_rn(v, 0.5f), 0.5f);
  }
}

__global__ void gather_rows_kernel(const float* __restrict__ table, const long* __restrict__ idx,
                                   float* __restrict__ out, int D, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    long b = i / D;
    out[i] = table[idx[b] * D + (i - b * D)];
  }
}

__global__ void silu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = silu_f(x[i]);
}
__global__ void silu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dx[i] = dy[i] * dsilu_f(x[i]);
}

// ------------------------------------------------------------------ loss
// partial[blk] = (sum (out-eps)^2, sum (x0-x)^2), x0 = c0*(x - c1*out)
template <typename T>
__global__ __launch_bounds__(256) void loss_partial_kernel(const T* __restrict__ out, const float* __restrict__ eps,
                                                           const float* __restrict__ x, float c0, float c1,
                                                           float2* __restrict__ part, long n) {
  float a = 0.f, r = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float o = Elem<T>::ld(out + i);
    float d = o - eps[i];
    a += d * d;
    float x0 = c0 * (x[i] - c1 * o);
    float e = x0 - x[i];
    r += e * e;
  }
  __shared__ float sa[4], sr[4];
  a = wave_sum(a); r = wave_sum(r);
  if ((threadIdx.x & 63) == 0) { sa[threadIdx.x >> 6] = a; sr[threadIdx.x >> 6] = r; }
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = make_float2(sa[0] + sa[1] + sa[2] + sa[3], sr[0] + sr[1] + sr[2] + sr[3]);
}
__global__ void loss_final_kernel(const float2* __restrict__ part, int nb, float inv_n, float inv_T,
                                  float* __restrict__ res) {
  double a = 0.0, r = 0.0;
  for (int i = threadIdx.x; i < nb; i += 64) { a += part[i].x; r += part[i].y; }
  float fa = wave_sum((float)a), fr = wave_sum((float)r);
  if (threadIdx.x == 0) { res[0] = fa * inv_n; res[1] = fr * inv_n * inv_T; }
}
// dout = g0 * 2(out-eps)/n + g1 * 2(x0-x)*(-c0*c1)/(n*T);  g0 = g[0], g1 = g[gs] (gs = 0: one upstream scalar for both terms)
template <typename T>
__global__ void loss_bwd_kernel(const T* __restrict__ out, const float* __restrict__ eps, const float* __restrict__ x,
                                float c0, float c1, const float* __restrict__ g, int gs, float inv_n, float inv_T,
                                T* __restrict__ dout, long n) {
  float g0 = g[0] * 2.f * inv_n, g1 = g[gs] * 2.f * inv_n * inv_T * (-c0 * c1);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float o = Elem<T>::ld(out + i);
    float x0 = c0 * (x[i] - c1 * o);
    Elem<T>::st(dout + i, g0 * (o - eps[i]) + g1 * (x0 - x[i]));
  }
}

// ------------------------------------------------------------- samplers
// mode 0 DDPM, 1 DDIM, 2 reverse DDIM.  The step's scalars are gathered from a
// [T][8] coefficient table the host builds with the reference's own fp32 torch
// expressions (sampling.py:30,35 / 52,57-58 / 71-72):
//   DDPM: c0 = sqrt(1/alpha_t), c1 = beta_t/sqrt(1-ab_t), c2 = sqrt_tilde_beta
//   DDIM: c0 = sqrt(1-apb_t), c1 = sqrt(apb_t), d0 = sqrt(apb_{t-1}), d1 = sqrt(1-apb_{t-1}-sigma^2), sigma
//   REV : c0, c1 as DDIM, c2 = sqrt(apb_{t+1}), c3 = sqrt(1-apb_{t+1})
template <typename T>
__global__ void sampler_step_kernel(const float* __restrict__ x, const T* __restrict__ eps,
                                    const float* __restrict__ noise, float* __restrict__ xo, T* __restrict__ xo_t,
                                    const long* __restrict__ idx_p, const float* __restrict__ coef, int mode, long n) {
  const long t = *idx_p;
  const float* cf = coef + t * 8;
  const float c0 = cf[0], c1 = cf[1], c2 = cf[2], c3 = cf[3], d0 = cf[4], d1 = cf[5], sigma = cf[6];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float xv = x[i], e = Elem<T>::ld(eps + i), o;
    if (mode == 0) {
      float mu = __fmul_rn(c0, __fsub_rn(xv, __fmul_rn(c1, e)));
      float nz = (t == 0) ? 0.f : noise[i];
      o = __fadd_rn(mu, __fmul_rn(c2, nz));
    } else if (mode == 1) {
      float x0 = __fdiv_rn(__fsub_rn(xv, __fmul_rn(c0, e)), c1);
      if (t == 0) o = x0;
      else {
        o = __fadd_rn(__fmul_rn(d0, x0), __fmul_rn(d1, e));
        o = __fadd_rn(o, __fmul_rn(sigma, noise[i]));
      }
    } else {
      float x0 = __fdiv_rn(__fsub_rn(xv, __fmul_rn(c0, e)), c1);
      o = __fadd_rn(__fmul_rn(c2, x0), __fmul_rn(c3, e));
    }
    xo[i] = o;
    if (xo_t) Elem<T>::st(xo_t + i, o);
  }
}

// -------------------------------------------------------------------- MMD
// one block per (which, i): which 0 = K(x,x), 1 = K(y,y), 2 = K(x,y); row sums
__global__ __launch_bounds__(256) void mmd_rows_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                       int n, int m, int D, float* __restrict__ rows) {
  int which, i = blockIdx.x;
  const float *P, *Q; int nq;
  if (i < n) { which = 0; P = x + (size_t)i * D; Q = x; nq = n; }
  else if (i < n + m) { which = 1; P = y + (size_t)(i - n) * D; Q = y; nq = m; }
  else { which = 2; P = x + (size_t)(i - n - m) * D; Q = y; nq = m; }
  (void)which;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc = 0.f;
  for (int j = wave; j < nq; j += 4) {
    float d2 = 0.f;
    for (int d = lane; d < D; d += 64) { float t = P[d] - Q[(size_t)j * D + d]; d2 += t * t; }
    d2 = wave_sum(d2);
    acc += expf(-(d2 / D) / D);
  }
  __shared__ float s[4];
  if (lane == 0) s[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) rows[i] = s[0] + s[1] + s[2] + s[3];
}
__global__ void mmd_final_kernel(const float* __restrict__ rows, int n, int m, float* __restrict__ out) {
  float a = 0.f, b = 0.f, c = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) { a += rows[i]; c += rows[n + m + i]; }
  for (int i = threadIdx.x; i < m; i += 64) b += rows[n + i];
  a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
  if (threadIdx.x == 0) out[0] = a / ((float)n * n) + b / ((float)m * m) - 2.f * c / ((float)n * m);
}
// The whole objective's tail in one block: the two loss terms from their partials (loss_final_kernel), the MMD from its
// row sums (mmd_final_kernel), and total = denoise + recon + w_mmd * mmd (models.py:640-646, 674-678 as ONE scalar).
// res = {denoise, recon, mmd, total}
__global__ void objective_final_kernel(const float2* __restrict__ part, int nb, float inv_n, float inv_T,
                                       const float* __restrict__ rows, int n, int m, float w_mmd, float* __restrict__ res) {
  double a = 0.0, r = 0.0;
  for (int i = threadIdx.x; i < nb; i += 64) { a += part[i].x; r += part[i].y; }
  const float fa = wave_sum((float)a), fr = wave_sum((float)r);
  float ka = 0.f, kb = 0.f, kc = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) { ka += rows[i]; kc += rows[n + m + i]; }
  for (int i = threadIdx.x; i < m; i += 64) kb += rows[n + i];
  ka = wave_sum(ka); kb = wave_sum(kb); kc = wave_sum(kc);
  if (threadIdx.x == 0) {
    const float d = fa * inv_n, rc = fr * inv_n * inv_T;
    const float mmd = ka / ((float)n * n) + kb / ((float)m * m) - 2.f * kc / ((float)n * m);
    res[0] = d; res[1] = rc; res[2] = mmd;
    res[3] = (d + rc) + w_mmd * mmd;          // the reference's order: (denoise + recon) + alpha * mmd
  }
}
// dy_j = g * [ (2/m^2) sum_i kyy_ij * (-2/D^2)(y_j - y_i) - (2/(nm)) sum_i kxy_ij * (-2/D^2)(y_j - x_i) ]
__global__ __launch_bounds__(256) void mmd_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                      int n, int m, int D, const float* __restrict__ g, float gscale,
                                                      float* __restrict__ dy) {
  extern __shared__ float acc[];   // [4][D]
  const int j = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* yj = y + (size_t)j * D;
  for (int d = lane; d < D; d += 64) acc[wave * D + d] = 0.f;
  const float cyy = 2.f / ((float)m * m) * (-2.f / ((float)D * D));
  const float cxy = -2.f / ((float)n * m) * (-2.f / ((float)D * D));
  for (int i = wave; i < n + m; i += 4) {
    const float* q = (i < m) ? y + (size_t)i * D : x + (size_t)(i - m) * D;
    float d2 = 0.f;
    for (int d = lane; d < D; d += 64) { float t = yj[d] - q[d]; d2 += t * t; }
    d2 = wave_sum(d2);
    float k = expf(-(d2 / D) / D) * ((i < m) ? cyy : cxy);
    for (int d = lane; d < D; d += 64) acc[wave * D + d] += k * (yj[d] - q[d]);
  }
  __syncthreads();
  for (int d = threadIdx.x; d < D; d += 256)
    dy[(size_t)j * D + d] = (g[0] * gscale) * (acc[d] + acc[D + d] + acc[2 * D + d] + acc[3 * D + d]);
}

// ---------------------------------------------------------------- colsum
// in [R][N] (T or float) -> partial[chunk][N]; then final sum.  N % VE == 0 not required.
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ in, float* __restrict__ part,
                                                             long R, int N, long rows_per_blk) {
  const long r0 = (long)blockIdx.x * rows_per_blk, r1 = min(R, r0 + rows_per_blk);
  // thread -> column (tid % ncol), row lane (tid / ncol)
  const int ncol = N < 256 ? N : 256;
  const int lanes = 256 / ncol;
  extern __shared__ float red[];   // [lanes][ncol]
  for (int c0 = 0; c0 < N; c0 += ncol) {
    int c = c0 + threadIdx.x % ncol, l = threadIdx.x / ncol;
    float a = 0.f;
    if (l < lanes && c < N)
      for (long r = r0 + l; r < r1; r += lanes) a += Elem<T>::ld(in + r * N + c);
    if (l < lanes) red[l * ncol + threadIdx.x % ncol] = a;
    __syncthreads();
    if (threadIdx.x < ncol && c0 + threadIdx.x < N) {
      float s = 0.f;
      for (int k = 0; k < lanes; ++k) s += red[k * ncol + threadIdx.x];
      part[(size_t)blockIdx.x * N + c0 + threadIdx.x] = s;
    }
    __syncthreads();
  }
}
// few rows (per-sample partials, linear biases): one thread per column, no second pass
template <typename T>
__global__ void colsum_small_kernel(const T* __restrict__ in, float* __restrict__ out, int R, int N) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= N) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;   // independent chains keep 4+ loads in flight
  int r = 0;
  for (; r + 4 <= R; r += 4) {
    s0 += Elem<T>::ld(in + (size_t)r * N + c);
    s1 += Elem<T>::ld(in + (size_t)(r + 1) * N + c);
    s2 += Elem<T>::ld(in + (size_t)(r + 2) * N + c);
    s3 += Elem<T>::ld(in + (size_t)(r + 3) * N + c);
  }
  for (; r < R; ++r) s0 += Elem<T>::ld(in + (size_t)r * N + c);
  out[c] = (s0 + s1) + (s2 + s3);
}

// one wave per column
__global__ void colsum_final_kernel(const float* __restrict__ part, int nb, int N, float* __restrict__ out) {
  int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= N) return;
  float s = 0.f;
  for (int k = threadIdx.x & 63; k < nb; k += 64) s += part[(size_t)k * N + c];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) out[c] = s;
}

// ------------------------------------------------------------ 2x2 sum pool
template <typename T>
__global__ void pool2_sum_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int Ho, int Wo, int C) {
  long n = (long)B * Ho * Wo * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % C);
    long p = i / C;
    int ox = (int)(p % Wo); p /= Wo;
    int oy = (int)(p % Ho);
    long b = p / Ho;
    size_t base = (((size_t)b * 2 * Ho + 2 * oy) * 2 * Wo + 2 * ox) * C + c;
    float v = Elem<T>::ld(in + base) + Elem<T>::ld(in + base + C) + Elem<T>::ld(in + base + (size_t)2 * Wo * C) +
              Elem<T>::ld(in + base + (size_t)2 * Wo * C + C);
    Elem<T>::st(out + i, v);
  }
}

// ---------------------------------------------------------- weight shadows
// src fp32 with logical index (o, i, tap) at o*so + i*si + tap*st.
// fwd shadow  [O][taps][I];  dgrad shadow [I][taps(flipped)][O].
template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ src, long so, long si, long st, T* __restrict__ wf,
                                   T* __restrict__ wd, int O, int I, int taps) {
  long n = (long)O * I * taps;
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (long)gridDim.x * blockDim.x) {
    int i = (int)(k % I);
    long r = k / I;
    int tap = (int)(r % taps);
    int o = (int)(r / taps);
    float v = src[o * so + i * si + tap * st];
    if (wf) Elem<T>::st(wf + k, v);
    if (wd) Elem<T>::st(wd + ((size_t)i * taps + (taps - 1 - tap)) * O + o, v);
  }
}

// all convs of a network in ONE launch: one block per table row = (conv, tap, 32-cout x 64-cin tile).
// The tile is read along cin (the master's contiguous axis for channels-last weights), written to the
// forward shadow in the same order, and transposed through LDS so the data-gradient shadow
// [cin][tap flipped][cout] is written along ITS contiguous axis too.
struct PackDesc {
  const float* src; void* wf; void* wd;
  long so, si, st;        // master strides of (o, i, tap)
  int O, I, taps;         // this source tensor
  int Ototal, o0;         // rows of the (possibly concatenated) shadow and this tensor's first row
  long e0;                // packed tile origin: tap | ot << 8 | it << 32
  void* wfrag;            // optional third shadow (3x3, O % 16 == 0, I % 64 == 0): the forward weights fragment-major for the
                          // image-resident ResBlock kernel (idf_resblock.hip): [I / 64][O / 16][tap][half][lane = fq * 16 + fr][8]
                          // with o = 16 * wave + fr, i = 64 * pair + 32 * half + 8 * fq + e -- a wave's A fragment is 1 KB of
                          // consecutive bytes
  void* wdfrag;           // optional fourth shadow (3x3, I % 16 == 0, Ototal % 64 == 0): the data-gradient weights [I][taps
                          // flipped][O] in the same fragment-major form (their "cout" is i, their "cin" is o)
};
template <typename T>
__global__ __launch_bounds__(256) void pack_batched_kernel(const PackDesc* __restrict__ tab) {
  const PackDesc d = tab[blockIdx.x];
  __shared__ float tile[32][65];
  T* wf = reinterpret_cast<T*>(d.wf);
  T* wd = reinterpret_cast<T*>(d.wd);
  const int tap = (int)(d.e0 & 0xff), ob = (int)((d.e0 >> 8) & 0xffffff) * 32, ib = (int)(d.e0 >> 32) * 64;
  const int tid = threadIdx.x;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    int o = ob + (tid >> 6) + k * 4, i = ib + (tid & 63);
    float v = 0.f;
    if (o < d.O && i < d.I) {
      v = d.src[o * d.so + i * d.si + tap * d.st];
      if (wf) Elem<T>::st(wf + ((size_t)(d.o0 + o) * d.taps + tap) * d.I + i, v);
    }
    tile[(tid >> 6) + k * 4][tid & 63] = v;
    if (d.wfrag && o < d.O && i < d.I) {
      const int oo = d.o0 + o, wave = oo >> 4, fr = oo & 15, pair = i >> 6, half = (i >> 5) & 1, fq = (i >> 3) & 3, e = i & 7;
      Elem<T>::st(reinterpret_cast<T*>(d.wfrag) +
                      ((((size_t)(pair * (d.Ototal >> 4) + wave) * d.taps + tap) * 2 + half) * 64 + fq * 16 + fr) * 8 + e, v);
    }
  }
  if (!wd) return;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    int i = ib + (tid >> 5) + k * 8, o = ob + (tid & 31);
    if (o < d.O && i < d.I) {
      const float v = tile[tid & 31][(tid >> 5) + k * 8];
      Elem<T>::st(wd + ((size_t)i * d.taps + (d.taps - 1 - tap)) * d.Ototal + d.o0 + o, v);
      if (d.wdfrag) {
        const int kk = d.o0 + o, pair = kk >> 6, half = (kk >> 5) & 1, fq = (kk >> 3) & 3, e = kk & 7, wave = i >> 4, fr = i & 15;
        Elem<T>::st(reinterpret_cast<T*>(d.wdfrag) +
                        ((((size_t)(pair * (d.I >> 4) + wave) * d.taps + (d.taps - 1 - tap)) * 2 + half) * 64 + fq * 16 + fr) * 8 + e, v);
      }
    }
  }
}

// dropout mask export (tests): mask[i] = keep ? scale : 0
__global__ void drop_mask_kernel(const uint64_t* seed, uint32_t salt, uint32_t thr, float scale, float* mask, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    mask[i] = idf_keep(*seed, salt, (uint64_t)i, thr) ? scale : 0.f;
}

inline int ew_blocks(long n, int per = 256) {
  long b = (n + per - 1) / per;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int idf_qsample(const float* x, const float* eps, const long* idx, const float* sqrt_ab,
                           const float* sqrt_1mab, float* xt32, void* xt, long per_sample, long n, int dtype,
                           void* stream) {
  if (n == 0) return IDF_OK;
  if (dtype == IDF_F32)
    hipLaunchKernelGGL(qsample_kernel<float>, dim3(ew_blocks(n)), dim3(256), 0, ST, x, eps, idx, sqrt_ab, sqrt_1mab,
                       xt32, (float*)xt, per_sample, n);
  else
    hipLaunchKernelGGL(qsample_kernel<bf16_t>, dim3(ew_blocks(n)), dim3(256), 0, ST, x, eps, idx, sqrt_ab, sqrt_1mab,
                       xt32, (bf16_t*)xt, per_sample, n);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}

extern "C" int idf_prep_u8(const uint8_t* src, const uint8_t* flip, float* dst, int B, int H, int W, int C,
                           void* stream) {
  long n = (long)B * H * W * C;
  if (n == 0) return IDF_OK;
  if (!src || !dst) IDF_FAIL(IDF_ERR_BADARG, "prep_u8: null buffer");
  hipLaunchKernelGGL(prep_u8_kernel, dim3(ew_blocks(n)), dim3(256), 0, ST, src, flip, dst, H, W, C, n);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}

extern "C" int idf_gather_rows(const float* table, const long* idx, float* out, int B, int D, void* stream) {
  long n = (long)B * D;
  if (n == 0) return IDF_OK;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(ew_blocks(n)), dim3(256), 0, ST, table, idx, out, D, n);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}

extern "C" int idf_silu_fwd(const float* x, float* y, long n, void* stream) {
  if (n == 0) return IDF_OK;
  hipLaunchKernelGGL(silu_fwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, ST, x, y, n);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}
extern "C" int idf_silu_bwd(const float* x, const float* dy, float* dx, long n, void* stream) {
  if (n == 0) return IDF_OK;
  hipLaunchKernelGGL(silu_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, ST, x, dy, dx, n);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}

// res[0] = mean((out-eps)^2), res[1] = mean((x0-x)^2)/T.  workspace: 2*1024 floats.
extern "C" int idf_loss_fwd(const void* out, const float* eps, const float* x, float c0, float c1, float inv_T,
                            float* res, float* workspace, long n, int dtype, void* stream) {
  int nb = ew_blocks(n, 1024);
  if (nb > 1024) nb = 1024;
  if (dtype == IDF_F32)
    hipLaunchKernelGGL(loss_partial_kernel<float>, dim3(nb), dim3(256), 0, ST, (const float*)out, eps, x, c0, c1,
                       (float2*)workspace, n);
  else
    hipLaunchKernelGGL(loss_partial_kernel<bf16_t>, dim3(nb), dim3(256), 0, ST, (const bf16_t*)out, eps, x, c0, c1,
                       (float2*)workspace, n);
  IDF_CHECK_LAUNCH();
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, ST, (const float2*)workspace, nb, 1.0f / (float)n, inv_T, res);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}
extern "C" int idf_loss_bwd(const void* out, const float* eps, const float* x, float c0, float c1, float inv_T,
                            const float* g, int g_stride, void* dout, long n, int dtype, void* stream) {
  if (g_stride != 0 && g_stride != 1) IDF_FAIL(IDF_ERR_BADARG, "loss_bwd: g_stride must be 0 or 1");
  if (dtype == IDF_F32)
    hipLaunchKernelGGL(loss_bwd_kernel<float>, dim3(ew_blocks(n)), dim3(256), 0, ST, (const float*)out, eps, x, c0, c1,
                       g, g_stride, 1.0f / (float)n, inv_T, (float*)dout, n);
  else
    hipLaunchKernelGGL(loss_bwd_kernel<bf16_t>, dim3(ew_blocks(n)), dim3(256), 0, ST, (const bf16_t*)out, eps, x, c0,
                       c1, g, g_stride, 1.0f / (float)n, inv_T, (bf16_t*)dout, n);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}

// res[4] = {denoise, recon, mmd, denoise + recon + w_mmd * mmd}: idf_loss_fwd + idf_mmd_fwd + the weighted sum in three
// launches (loss partials, MMD row sums, one tail block).  workspace: 2048 + 2n + m floats.
extern "C" int idf_objective_fwd(const void* out, const float* eps, const float* x, float c0, float c1, float inv_T,
                                 const float* prior, const float* lat, int n, int m, int D, float w_mmd, float* res,
                                 float* workspace, long numel, int dtype, void* stream) {
  if (n <= 0 || m <= 0) IDF_FAIL(IDF_ERR_BADARG, "objective: empty latent batch");
  int nb = ew_blocks(numel, 1024);
  if (nb > 1024) nb = 1024;
  if (dtype == IDF_F32)
    hipLaunchKernelGGL(loss_partial_kernel<float>, dim3(nb), dim3(256), 0, ST, (const float*)out, eps, x, c0, c1,
                       (float2*)workspace, numel);
  else
    hipLaunchKernelGGL(loss_partial_kernel<bf16_t>, dim3(nb), dim3(256), 0, ST, (const bf16_t*)out, eps, x, c0, c1,
                       (float2*)workspace, numel);
  IDF_CHECK_LAUNCH();
  float* rows = workspace + 2048;
  hipLaunchKernelGGL(mmd_rows_kernel, dim3(2 * n + m), dim3(256), 0, ST, prior, lat, n, m, D, rows);
  IDF_CHECK_LAUNCH();
  hipLaunchKernelGGL(objective_final_kernel, dim3(1), dim3(64), 0, ST, (const float2*)workspace, nb, 1.0f / (float)numel, inv_T,
                     rows, n, m, w_mmd, res);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}

extern "C" int idf_sampler_step(const float* x, const void* eps, const float* noise, float* xo, void* xo_t,
                                const long* idx, const float* coef, int mode, long n, int dtype, void* stream) {
  if (mode < 0 || mode > 2) IDF_FAIL(IDF_ERR_BADARG, "sampler_step: bad mode %d", mode);
  if (n == 0) return IDF_OK;
  if (dtype == IDF_F32)
    hipLaunchKernelGGL(sampler_step_kernel<float>, dim3(ew_blocks(n)), dim3(256), 0, ST, x, (const float*)eps, noise, xo,
                       (float*)xo_t, idx, coef, mode, n);
  else
    hipLaunchKernelGGL(sampler_step_kernel<bf16_t>, dim3(ew_blocks(n)), dim3(256), 0, ST, x, (const bf16_t*)eps, noise,
                       xo, (bf16_t*)xo_t, idx, coef, mode, n);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}

// workspace: (2n + m) floats
extern "C" int idf_mmd_fwd(const float* x, const float* y, int n, int m, int D, float* out, float* workspace,
                           void* stream) {
  if (n <= 0 || m <= 0) IDF_FAIL(IDF_ERR_BADARG, "mmd: empty input");
  hipLaunchKernelGGL(mmd_rows_kernel, dim3(2 * n + m), dim3(256), 0, ST, x, y, n, m, D, workspace);
  IDF_CHECK_LAUNCH();
  hipLaunchKernelGGL(mmd_final_kernel, dim3(1), dim3(64), 0, ST, workspace, n, m, out);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}
extern "C" int idf_mmd_bwd(const float* x, const float* y, int n, int m, int D, const float* g, float gscale, float* dy,
                           void* stream) {
  hipLaunchKernelGGL(mmd_bwd_kernel, dim3(m), dim3(256), 4 * D * sizeof(float), ST, x, y, n, m, D, g, gscale, dy);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}

// workspace: nblk*N floats where nblk = idf_colsum_blocks(R)
extern "C" int idf_colsum_blocks(long R) {
  long nb = (R + 255) / 256;
  return (int)(nb > 256 ? 256 : (nb < 1 ? 1 : nb));
}
extern "C" int idf_colsum(const void* in, float* out, float* workspace, long R, int N, int in_dtype, void* stream) {
  if (R <= 512) {
    if (in_dtype == IDF_F32)
      hipLaunchKernelGGL(colsum_small_kernel<float>, dim3((N + 63) / 64), dim3(64), 0, ST, (const float*)in, out, (int)R, N);
    else
      hipLaunchKernelGGL(colsum_small_kernel<bf16_t>, dim3((N + 63) / 64), dim3(64), 0, ST, (const bf16_t*)in, out, (int)R, N);
    IDF_CHECK_LAUNCH();
    return IDF_OK;
  }
  int nb = idf_colsum_blocks(R);
  long rpb = (R + nb - 1) / nb;
  int ncol = N < 256 ? N : 256;
  size_t lds = (size_t)(256 / ncol) * ncol * sizeof(float);
  if (in_dtype == IDF_F32)
    hipLaunchKernelGGL(colsum_partial_kernel<float>, dim3(nb), dim3(256), lds, ST, (const float*)in, workspace, R, N, rpb);
  else
    hipLaunchKernelGGL(colsum_partial_kernel<bf16_t>, dim3(nb), dim3(256), lds, ST, (const bf16_t*)in, workspace, R, N, rpb);
  IDF_CHECK_LAUNCH();
  hipLaunchKernelGGL(colsum_final_kernel, dim3((N + 3) / 4), dim3(256), 0, ST, workspace, nb, N, out);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}

extern "C" int idf_pool2_sum(const void* in, void* out, int B, int Ho, int Wo, int C, int dtype, void* stream) {
  long n = (long)B * Ho * Wo * C;
  if (n == 0) return IDF_OK;
  if (dtype == IDF_F32)
    hipLaunchKernelGGL(pool2_sum_kernel<float>, dim3(ew_blocks(n)), dim3(256), 0, ST, (const float*)in, (float*)out, B, Ho, Wo, C);
  else
    hipLaunchKernelGGL(pool2_sum_kernel<bf16_t>, dim3(ew_blocks(n)), dim3(256), 0, ST, (const bf16_t*)in, (bf16_t*)out, B, Ho, Wo, C);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}

extern "C" int idf_pack_conv_weight(const float* src, long so, long si, long st, void* w_fwd, void* w_dgrad, int O,
                                    int I, int taps, int dtype, void* stream) {
  long n = (long)O * I * taps;
  if (n == 0) return IDF_OK;
  if (dtype == IDF_F32)
    hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(ew_blocks(n)), dim3(256), 0, ST, src, so, si, st, (float*)w_fwd,
                       (float*)w_dgrad, O, I, taps);
  else
    hipLaunchKernelGGL(pack_weight_kernel<bf16_t>, dim3(ew_blocks(n)), dim3(256), 0, ST, src, so, si, st,
                       (bf16_t*)w_fwd, (bf16_t*)w_dgrad, O, I, taps);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}

// table: nrows x PackDesc {src*, wf*, wd*, long so, si, st, int O, I, taps, Ototal, o0, long tile, -} (device);
// tile = tap | (cout_tile32 << 8) | (cin_tile64 << 32)
extern "C" int idf_pack_conv_weights_batched(const void* table, int nrows, int dtype, void* stream) {
  if (nrows <= 0) return IDF_OK;
  if (dtype == IDF_F32)
    hipLaunchKernelGGL(pack_batched_kernel<float>, dim3(nrows), dim3(256), 0, ST, (const PackDesc*)table);
  else
    hipLaunchKernelGGL(pack_batched_kernel<bf16_t>, dim3(nrows), dim3(256), 0, ST, (const PackDesc*)table);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}

extern "C" int idf_dropout_mask(const uint64_t* seed, uint32_t salt, float p_drop, float* mask, long n, void* stream) {
  uint32_t thr = idf_drop_thresh(p_drop);
  float scale = 1.0f / (1.0f - (float)thr / 65536.0f);
  hipLaunchKernelGGL(drop_mask_kernel, dim3(ew_blocks(n)), dim3(256), 0, ST, seed, salt, thr, scale, mask, n);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}
