// Row softmax forward / backward for the single-head spatial self-attention
// (modules.py:154-156).  The two bmm's run on idf_bgemm; scores are N <= 1024 wide
// (N = H*W <= 256 on every dataset the reference configures), one wave per row,
// fp32 math, values kept in registers between the passes.
#include "idf_common.h"

namespace {

constexpr int MAXPL = 16;   // columns per lane: N <= 1024

template <typename T>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(T* __restrict__ s, long R, int N) {
  const int lane = threadIdx.x & 63;
  long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  T* row = s + r * N;
  float v[MAXPL];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < MAXPL; ++i) {
    int c = lane + i * 64;
    v[i] = c < N ? Elem<T>::ld(row + c) : -INFINITY;
    mx = fmaxf(mx, v[i]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < MAXPL; ++i) {
    int c = lane + i * 64;
    v[i] = c < N ? __expf(v[i] - mx) : 0.f;
    sum += v[i];
  }
  sum = wave_sum(sum);
  float inv = 1.0f / sum;
#pragma unroll
  for (int i = 0; i < MAXPL; ++i) {
    int c = lane + i * 64;
    if (c < N) Elem<T>::st(row + c, v[i] * inv);
  }
}

// dS = P * (dP - sum_j dP*P), written over dP
template <typename T>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const T* __restrict__ P, T* __restrict__ dP, long R, int N) {
  const int lane = threadIdx.x & 63;
  long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const T* prow = P + r * N;
  T* drow = dP + r * N;
  float pv[MAXPL], dv[MAXPL];
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < MAXPL; ++i) {
    int c = lane + i * 64;
    pv[i] = c < N ? Elem<T>::ld(prow + c) : 0.f;
    dv[i] = c < N ? Elem<T>::ld(drow + c) : 0.f;
    dot += pv[i] * dv[i];
  }
  dot = wave_sum(dot);
#pragma unroll
  for (int i = 0; i < MAXPL; ++i) {
    int c = lane + i * 64;
    if (c < N) Elem<T>::st(drow + c, pv[i] * (dv[i] - dot));
  }
}

}  // namespace

extern "C" int idf_softmax_fwd(void* s, long R, int N, int dtype, void* stream) {
  if (N > 64 * MAXPL) IDF_FAIL(IDF_ERR_UNSUPPORTED, "softmax: N=%d > %d", N, 64 * MAXPL);
  if (R == 0) return IDF_OK;
  dim3 g((unsigned)((R + 3) / 4));
  if (dtype == IDF_F32) hipLaunchKernelGGL(softmax_fwd_kernel<float>, g, dim3(256), 0, (hipStream_t)stream, (float*)s, R, N);
  else hipLaunchKernelGGL(softmax_fwd_kernel<bf16_t>, g, dim3(256), 0, (hipStream_t)stream, (bf16_t*)s, R, N);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}

extern "C" int idf_softmax_bwd(const void* P, void* dP, long R, int N, int dtype, void* stream) {
  if (N > 64 * MAXPL) IDF_FAIL(IDF_ERR_UNSUPPORTED, "softmax: N=%d > %d", N, 64 * MAXPL);
  if (R == 0) return IDF_OK;
  dim3 g((unsigned)((R + 3) / 4));
  if (dtype == IDF_F32)
    hipLaunchKernelGGL(softmax_bwd_kernel<float>, g, dim3(256), 0, (hipStream_t)stream, (const float*)P, (float*)dP, R, N);
  else
    hipLaunchKernelGGL(softmax_bwd_kernel<bf16_t>, g, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)P, (bf16_t*)dP, R, N);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}
