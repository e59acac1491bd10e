// Fused row kernels of the latent denoiser MLP (LatentUNet / MLPLNAct, models.py:147-163):
//   z = lin * (1 + cond);  y = Dropout(SiLU(LayerNorm(z) * g + b))
// one 256-thread block per row (width <= 4096), fp32.  Backward returns dlin, dcond and the
// per-row contributions to dg / db (summed over rows with idf_colsum).
#include "idf_common.h"

namespace {

__device__ __forceinline__ float block_sum(float v, float* sm) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  return sm[0] + sm[1] + sm[2] + sm[3];
}

__global__ __launch_bounds__(256) void ln_silu_fwd_kernel(const float* __restrict__ lin, const float* __restrict__ cond,
                                                          const float* __restrict__ g, const float* __restrict__ b,
                                                          float* __restrict__ y, float* __restrict__ stats, int Wd,
                                                          float eps, const uint64_t* seed, uint32_t salt, uint32_t thr,
                                                          float dscale) {
  __shared__ float sm[4];
  const int r = blockIdx.x;
  const float* lr = lin + (size_t)r * Wd;
  const float* cr = cond + (size_t)r * Wd;
  float s = 0.f;
  for (int i = threadIdx.x; i < Wd; i += 256) s += lr[i] * (1.f + cr[i]);
  const float mean = block_sum(s, sm) / Wd;
  float q = 0.f;
  for (int i = threadIdx.x; i < Wd; i += 256) { float d = lr[i] * (1.f + cr[i]) - mean; q += d * d; }
  const float rstd = rsqrtf(block_sum(q, sm) / Wd + eps);
  if (threadIdx.x == 0) { stats[2 * r] = mean; stats[2 * r + 1] = rstd; }
  for (int i = threadIdx.x; i < Wd; i += 256) {
    float u = (lr[i] * (1.f + cr[i]) - mean) * rstd * g[i] + b[i];
    u = silu_f(u);
    if (seed) u = idf_keep(*seed, salt, (uint64_t)r * Wd + i, thr) ? u * dscale : 0.f;
    y[(size_t)r * Wd + i] = u;
  }
}

__global__ __launch_bounds__(256) void ln_silu_bwd_kernel(const float* __restrict__ lin, const float* __restrict__ cond,
                                                          const float* __restrict__ g, const float* __restrict__ b,
                                                          const float* __restrict__ stats, const float* __restrict__ dy,
                                                          float* __restrict__ dlin, float* __restrict__ dcond,
                                                          float* __restrict__ dgb, int Wd, const uint64_t* seed,
                                                          uint32_t salt, uint32_t thr, float dscale) {
  __shared__ float sm[4];
  const int r = blockIdx.x;
  const size_t o = (size_t)r * Wd;
  const float mean = stats[2 * r], rstd = stats[2 * r + 1];
  float s1 = 0.f, s2 = 0.f;
  for (int i = threadIdx.x; i < Wd; i += 256) {
    float xh = (lin[o + i] * (1.f + cond[o + i]) - mean) * rstd;
    float u = xh * g[i] + b[i];
    float d = dy[o + i] * dsilu_f(u);
    if (seed) d = idf_keep(*seed, salt, o + i, thr) ? d * dscale : 0.f;
    dgb[(size_t)r * 2 * Wd + i] = d * xh;
    dgb[(size_t)r * 2 * Wd + Wd + i] = d;
    float dxh = d * g[i];
    s1 += dxh; s2 += dxh * xh;
  }
  const float m1 = block_sum(s1, sm) / Wd;
  const float m2 = block_sum(s2, sm) / Wd;
  for (int i = threadIdx.x; i < Wd; i += 256) {
    float xh = (lin[o + i] * (1.f + cond[o + i]) - mean) * rstd;
    float dxh = dgb[(size_t)r * 2 * Wd + Wd + i] * g[i];
    float dz = rstd * (dxh - m1 - xh * m2);
    dlin[o + i] = dz * (1.f + cond[o + i]);
    dcond[o + i] = dz * lin[o + i];
  }
}

}  // namespace

extern "C" int idf_ln_silu_fwd(const float* lin, const float* cond, const float* g, const float* b, float* y,
                               float* stats, int R, int Wd, float eps, const uint64_t* seed, uint32_t salt,
                               float p_drop, void* stream) {
  if (R == 0) return IDF_OK;
  uint32_t thr = idf_drop_thresh(p_drop);
  float dscale = 1.0f / (1.0f - (float)thr / 65536.0f);
  hipLaunchKernelGGL(ln_silu_fwd_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, lin, cond, g, b, y, stats, Wd, eps,
                     p_drop > 0.f ? seed : nullptr, salt, thr, dscale);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}

extern "C" int idf_ln_silu_bwd(const float* lin, const float* cond, const float* g, const float* b, const float* stats,
                               const float* dy, float* dlin, float* dcond, float* dgb, int R, int Wd,
                               const uint64_t* seed, uint32_t salt, float p_drop, void* stream) {
  if (R == 0) return IDF_OK;
  uint32_t thr = idf_drop_thresh(p_drop);
  float dscale = 1.0f / (1.0f - (float)thr / 65536.0f);
  hipLaunchKernelGGL(ln_silu_bwd_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, lin, cond, g, b, stats, dy, dlin,
                     dcond, dgb, Wd, p_drop > 0.f ? seed : nullptr, salt, thr, dscale);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}
