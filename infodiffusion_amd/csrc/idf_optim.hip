// Fused optimizer tail: global grad-norm clip (torch.nn.utils.clip_grad_norm_, run.py:199) +
// AdamW with decoupled weight decay (torch.optim.AdamW, run.py:177,200) over ALL parameter
// tensors in three launches, driven by a device-side chunk table (one block per chunk of
// <= 65536 contiguous elements).  The reference path issues ~745 vector-norm calls plus a
// dozen multi-tensor foreach launches per step; this reads g once for the norm and
// p, g, m, v once for the update (HBM-bound, ~28 B / parameter).
#include "idf_common.h"

namespace {

struct Chunk {
  float* p;
  float* g;
  float* m;
  float* v;
  long n;
};

__global__ __launch_bounds__(256) void sqnorm_kernel(const Chunk* __restrict__ tab, float* __restrict__ partial) {
  const Chunk c = tab[blockIdx.x];
  float s = 0.f;
  const float4* g4 = reinterpret_cast<const float4*>(c.g);
  const bool al = (((uintptr_t)c.g) & 15) == 0;
  long n4 = al ? c.n / 4 : 0;
  // four independent 16-byte loads in flight per lane (one per iteration left the kernel at 2.4 TB/s)
  long i = threadIdx.x;
  float s1 = 0.f, s2 = 0.f, s3 = 0.f;
  for (; i + 768 < n4; i += 1024) {
    float4 a = g4[i], b = g4[i + 256], c4 = g4[i + 512], d = g4[i + 768];
    s += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
    s1 += b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w;
    s2 += c4.x * c4.x + c4.y * c4.y + c4.z * c4.z + c4.w * c4.w;
    s3 += d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
  }
  for (; i < n4; i += 256) {
    float4 v = g4[i];
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  s += s1 + s2 + s3;
  for (long i = n4 * 4 + threadIdx.x; i < c.n; i += 256) s += c.g[i] * c.g[i];
  __shared__ float sm[4];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

// state: [0] step count (float), [1] clip coef, [2] 1-b1^t, [3] 1-b2^t, [4] total grad norm
// one 1024-thread block (thousands of partials: a single wave walked them in a 17-us chain of dependent loads)
__global__ __launch_bounds__(1024) void opt_scalars_kernel(const float* __restrict__ partial, int n, float max_norm, float b1,
                                                          float b2, float* __restrict__ state) {
  __shared__ float red[16];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) s += partial[i];
  float t = wave_sum((float)s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    t = 0.f;
    for (int w = 0; w < 16; ++w) t += red[w];
    float total = sqrtf(t);
    float coef = max_norm > 0.f ? fminf(1.0f, max_norm / (total + 1e-6f)) : 1.0f;
    float step = state[0] + 1.0f;
    state[0] = step;
    state[1] = coef;
    state[2] = 1.0f - powf(b1, step);
    state[3] = 1.0f - powf(b2, step);
    state[4] = total;
  }
}

__global__ __launch_bounds__(256) void adamw_kernel(const Chunk* __restrict__ tab, const float* __restrict__ state,
                                                    const float* __restrict__ lr_p, float b1, float b2, float eps,
                                                    float wd, int write_g) {
  const Chunk c = tab[blockIdx.x];
  const float coef = state[1], bc1 = state[2], bc2s = sqrtf(state[3]), lr = *lr_p;
  const float decay = 1.0f - lr * wd, step_size = lr / bc1;
  // chunk bases are 16-byte aligned for all but ragged tensors: 4 parameters per lane per access
  const bool al = ((((uintptr_t)c.p) | ((uintptr_t)c.g) | ((uintptr_t)c.m) | ((uintptr_t)c.v)) & 15) == 0;
  const long n4 = al ? c.n / 4 : 0;
  // two 4-element groups per lane per iteration: eight 16-byte loads in flight before the first dependent use
  auto upd = [&](long i, float4 g4, float4 p4, float4 m4, float4 v4) {
    float gg[4] = {g4.x, g4.y, g4.z, g4.w}, pp[4] = {p4.x, p4.y, p4.z, p4.w};
    float mm[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      gg[k] *= coef;
      mm[k] = b1 * mm[k] + (1.0f - b1) * gg[k];
      vv[k] = b2 * vv[k] + (1.0f - b2) * gg[k] * gg[k];
      pp[k] = pp[k] * decay - step_size * (mm[k] / (sqrtf(vv[k]) / bc2s + eps));
    }
    reinterpret_cast<float4*>(c.p)[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
    reinterpret_cast<float4*>(c.m)[i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
    reinterpret_cast<float4*>(c.v)[i] = make_float4(vv[0], vv[1], vv[2], vv[3]);
    if (write_g) reinterpret_cast<float4*>(c.g)[i] = make_float4(gg[0], gg[1], gg[2], gg[3]);
  };
  long i = threadIdx.x;
  for (; i + 256 < n4; i += 512) {
    const long j = i + 256;
    float4 ga = reinterpret_cast<float4*>(c.g)[i], pa = reinterpret_cast<float4*>(c.p)[i];
    float4 ma = reinterpret_cast<float4*>(c.m)[i], va = reinterpret_cast<float4*>(c.v)[i];
    float4 gb = reinterpret_cast<float4*>(c.g)[j], pb = reinterpret_cast<float4*>(c.p)[j];
    float4 mb = reinterpret_cast<float4*>(c.m)[j], vb = reinterpret_cast<float4*>(c.v)[j];
    upd(i, ga, pa, ma, va);
    upd(j, gb, pb, mb, vb);
  }
  for (; i < n4; i += 256)
    upd(i, reinterpret_cast<float4*>(c.g)[i], reinterpret_cast<float4*>(c.p)[i], reinterpret_cast<float4*>(c.m)[i],
        reinterpret_cast<float4*>(c.v)[i]);
  for (long i = n4 * 4 + threadIdx.x; i < c.n; i += 256) {
    float g = c.g[i] * coef;
    float p = c.p[i] * decay;
    float m = b1 * c.m[i] + (1.0f - b1) * g;
    float v = b2 * c.v[i] + (1.0f - b2) * g * g;
    float denom = sqrtf(v) / bc2s + eps;
    c.p[i] = p - step_size * (m / denom);
    c.m[i] = m;
    c.v[i] = v;
    if (write_g) c.g[i] = g;
  }
}

}  // namespace

// table: nchunks x {p*, g*, m*, v*, long n} in device memory; partial: nchunks floats; state: 8 floats
// (state[0] = step count, persistent across calls); lr: device float (so a captured graph follows LR schedules).
extern "C" int idf_clip_adamw(const void* table, int nchunks, float* partial, float* state, const float* lr,
                              float max_norm, float b1, float b2, float eps, float wd, int write_clipped_grads,
                              void* stream) {
  if (nchunks <= 0) return IDF_OK;
  hipStream_t st = (hipStream_t)stream;
  const Chunk* tab = reinterpret_cast<const Chunk*>(table);
  hipLaunchKernelGGL(sqnorm_kernel, dim3(nchunks), dim3(256), 0, st, tab, partial);
  IDF_CHECK_LAUNCH();
  hipLaunchKernelGGL(opt_scalars_kernel, dim3(1), dim3(1024), 0, st, partial, nchunks, max_norm, b1, b2, state);
  IDF_CHECK_LAUNCH();
  hipLaunchKernelGGL(adamw_kernel, dim3(nchunks), dim3(256), 0, st, tab, state, lr, b1, b2, eps, wd, write_clipped_grads);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}
