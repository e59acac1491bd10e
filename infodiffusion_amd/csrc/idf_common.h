// Shared device/host helpers for libinfodiff_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define IDF_OK 0
#define IDF_ERR_UNSUPPORTED 1001   // shape/dtype outside what the kernels cover
#define IDF_ERR_BADARG 1002
#define IDF_ERR_HIP 1003           // a HIP runtime call / launch failed (the hipError string is in idf_last_error())

#define IDF_F32 0
#define IDF_BF16 1

extern "C" const char* idf_last_error(void);
void idf_set_error(const char* fmt, ...);

// the library's environment switches (idf_capi.hip: read once)
struct IdfKnobs { int conv_rs, conv_rs_sync, conv_ps, wgrad_kr3, wgrad_tpb3, wgrad_ring; long conv_dlds_min; };
const IdfKnobs& idf_knobs();

#define IDF_FAIL(code, ...)            \
  do {                                 \
    idf_set_error(__VA_ARGS__);        \
    return (code);                     \
  } while (0)

#define IDF_CHECK_LAUNCH()                                        \
  do {                                                            \
    hipError_t _e = hipGetLastError();                            \
    if (_e != hipSuccess) {                                       \
      idf_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, \
                    hipGetErrorString(_e));                       \
      return IDF_ERR_HIP;                                         \
    }                                                             \
  } while (0)

// Dynamic LDS beyond 64 KB needs hipFuncAttributeMaxDynamicSharedMemorySize on the kernel -- a per-DEVICE attribute.  One
// `granted` array per kernel (a static of the launching function / template instantiation) remembers what each device
// already has, so the attribute is raised once per (kernel, device, larger request), not per launch.  A failure is
// returned: the launch that follows would run without the opt-in.  (A racing duplicate call is harmless.)
#include <atomic>
struct IdfLdsGrant { std::atomic<size_t> granted[32]; };
inline hipError_t idf_ensure_lds(const void* kern, size_t bytes, IdfLdsGrant& g) {
  if (bytes <= 64 * 1024) return hipSuccess;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev >= 0 && dev < 32 && g.granted[dev].load(std::memory_order_relaxed) >= bytes) return hipSuccess;
  e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess && dev >= 0 && dev < 32) g.granted[dev].store(bytes, std::memory_order_relaxed);
  return e;
}

typedef uint16_t bf16_t;  // raw storage

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
  return __uint_as_float(((uint32_t)v) << 16);
}
// round-to-nearest-even; a plain cast keeps NaN a NaN (MI355X_MICROARCH correctness table)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  __bf16 h = (__bf16)f;
  return *reinterpret_cast<bf16_t*>(&h);
}

// two floats -> one dword of bf16 (lo in bits 0..15), round-to-nearest-even, NaN stays NaN.  NOT as one two-source v_cvt_pk_bf16_f32
// (__builtin_convertvector of a float2): measured on MI355X / ROCm 7.2, that form in the 256-thread halo conv's du epilogue gave a few
// wrong elements per launch, different ones every launch (0 of 12 launches clean; 12 of 12 clean with the two single conversions
// below -- tools/_dbg2.py history in profiles/r05_cvt_pk_hazard.txt), so every pack goes through this one function.
__device__ __forceinline__ uint32_t idf_pack_bf16(float lo, float hi) {
  return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int VE = 4;  // elements per 16-byte vector
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
  static constexpr int VE = 8;
  __device__ static __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
  __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// 16-byte vector <-> VE floats
template <typename T> struct Vec16;
template <> struct Vec16<float> {
  __device__ static __forceinline__ void load(const float* p, float* o) {
    float4 v = *reinterpret_cast<const float4*>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  }
  __device__ static __forceinline__ void store(float* p, const float* o) {
    *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
  }
};
template <> struct Vec16<bf16_t> {
  __device__ static __forceinline__ void load(const bf16_t* p, float* o) {
    uint4 v = *reinterpret_cast<const uint4*>(p);
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o[2 * i] = __uint_as_float(w[i] << 16);
      o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  __device__ static __forceinline__ void store(bf16_t* p, const float* o) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      w[i] = idf_pack_bf16(o[2 * i], o[2 * i + 1]);
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
  }
};

// sigmoid via v_exp + v_rcp (1 ulp each): no IEEE division sequence in the HBM-bound passes
__device__ __forceinline__ float sigmoid_f(float u) { return __builtin_amdgcn_rcpf(1.0f + __expf(-u)); }
__device__ __forceinline__ float silu_f(float u) { return u * sigmoid_f(u); }
// d/du [u * sigmoid(u)]
__device__ __forceinline__ float dsilu_f(float u) {
  float s = sigmoid_f(u);
  return s * (1.0f + u * (1.0f - s));
}

// Counter-based dropout: keep decision for element `idx` of call site `salt` under step seed
// `seed`.  One well-mixed 32-bit hash per aligned group of 8 elements (= one 16-byte bf16 vector),
// from which the 8 per-element 16-bit draws are derived by one multiply-add each -- the mask costs
// ~4 VALU ops per element instead of ~9, which is what keeps the GroupNorm passes HBM-bound.
__device__ __forceinline__ uint32_t idf_hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t idf_vec_hash(uint64_t seed, uint32_t salt, uint64_t vec) {
  uint32_t h = idf_hash32((uint32_t)vec ^ (uint32_t)seed);
  return idf_hash32(h + salt * 0x9E3779B9u + (uint32_t)(seed >> 32) + (uint32_t)(vec >> 32) * 0x85ebca6bU);
}
// (round 5: a 24-bit multiply -- v_mad_u32_u24, full rate -- instead of the quarter-rate 32-bit v_mul_lo_u32: the per-element draw
// was a fifth of the cycles of the conv epilogues that recompute the mask; the upper half of the 32-bit product of the hash's low
// 24 bits with an odd 24-bit constant is the draw: uniform, pairwise correlation of the 8 draws of a vector < 1.2e-3 over 4e6
// vectors, keep rate 0.9000 +- 2e-4 per lane at p = 0.1.)
__device__ __forceinline__ bool idf_keep_h(uint32_t h, int lane, uint32_t thresh16) {
  constexpr uint32_t A[8] = {0x9E3779u, 0x85EBCBu, 0xC2B2AFu, 0x27D4EBu, 0x165667u, 0xD3A265u, 0xFD7047u, 0xB55A4Fu};
  return ((__umul24(h, A[lane & 7]) + (A[lane & 7] >> 5)) >> 16) >= thresh16;
}
// all-lanes sums across the halves / quarters / eighths of a wave (xor 32, 16, 8) without LDS traffic: gfx950's permlane swaps
// exchange 32- / 16-lane rows between two registers (both = v: the results hold the two partners side by side), row_ror:8 rotates
// inside a 16-lane row
__device__ __forceinline__ float idf_xor32_sum(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float idf_xor16_sum(float v) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float idf_xor8_sum(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));   // row_ror:8
}
__device__ __forceinline__ bool idf_keep(uint64_t seed, uint32_t salt, uint64_t idx, uint32_t thresh16) {
  return idf_keep_h(idf_vec_hash(seed, salt, idx >> 3), (int)(idx & 7), thresh16);
}
// act(x * sc + sh) (+ dropout) over one vector of N elements, straight-line.  The conditions are uniform (kernel
// arguments): resolving them once per VECTOR instead of once per element lets the N exp / rcp chains interleave --
// with a scalar branch around every element the compiler emits them strictly one after the other.
// Same operations per element as silu_f / idf_keep_h: results are bit-identical to the element-wise form.
template <int N, bool SILU, bool DROP>
__device__ __forceinline__ void idf_act_vec_t(float* v, const float* scv, const float* shv, uint32_t h, int l0,
                                              uint32_t thr, float dscale) {
#pragma unroll
  for (int e = 0; e < N; ++e) v[e] = v[e] * scv[e] + shv[e];
  if (SILU) {
    float t[N];
#pragma unroll
    for (int e = 0; e < N; ++e) t[e] = __expf(-v[e]);
#pragma unroll
    for (int e = 0; e < N; ++e) t[e] = __builtin_amdgcn_rcpf(1.0f + t[e]);
#pragma unroll
    for (int e = 0; e < N; ++e) v[e] = v[e] * t[e];
    if (DROP) {
#pragma unroll
      for (int e = 0; e < N; ++e) v[e] = idf_keep_h(h, l0 + e, thr) ? v[e] * dscale : 0.f;
    }
  }
}
template <int N>
__device__ __forceinline__ void idf_act_vec(float* v, const float* scv, const float* shv, int act, bool drop, uint32_t h,
                                            int l0, uint32_t thr, float dscale) {
  if (act == 2) {
    if (drop) idf_act_vec_t<N, true, true>(v, scv, shv, h, l0, thr, dscale);
    else idf_act_vec_t<N, true, false>(v, scv, shv, h, l0, thr, dscale);
  } else {
    idf_act_vec_t<N, false, false>(v, scv, shv, h, l0, thr, dscale);
  }
}
// backward companion: du = dA * act'(x * sc + sh) (* dropout mask / keep probability)
template <int N, bool SILU, bool DROP>
__device__ __forceinline__ void idf_dact_vec_t(const float* dav, const float* xv, const float* scv, const float* shv,
                                               uint32_t h, int l0, uint32_t thr, float dscale, float* du) {
  if (SILU) {
    float u[N], s[N];
#pragma unroll
    for (int e = 0; e < N; ++e) u[e] = xv[e] * scv[e] + shv[e];
#pragma unroll
    for (int e = 0; e < N; ++e) s[e] = __expf(-u[e]);
#pragma unroll
    for (int e = 0; e < N; ++e) s[e] = __builtin_amdgcn_rcpf(1.0f + s[e]);
#pragma unroll
    for (int e = 0; e < N; ++e) du[e] = dav[e] * (s[e] * (1.0f + u[e] * (1.0f - s[e])));
    if (DROP) {
#pragma unroll
      for (int e = 0; e < N; ++e) du[e] = idf_keep_h(h, l0 + e, thr) ? du[e] * dscale : 0.f;
    }
  } else {
#pragma unroll
    for (int e = 0; e < N; ++e) du[e] = dav[e];
  }
}

__host__ __device__ __forceinline__ uint32_t idf_drop_thresh(float p) {
  return (uint32_t)(p * 65536.0f + 0.5f);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

static inline int idf_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Zero n floats with a KERNEL.  Not hipMemsetAsync: a memset node recorded while the caller captures the stream
// into a hipGraph did not clear the buffer reliably on replay (ROCm 7.2, gfx950 -- the weight-gradient buffers
// of the image / epsilon convs kept garbage, the gradient norm overflowed and a replayed training step stopped
// learning), while kernel nodes replay exactly as recorded.
static __global__ void idf_zero_f32_kernel(float* __restrict__ p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = 0.f;
}
static inline hipError_t idf_zero_f32(float* p, size_t n, hipStream_t st) {
  if (n == 0) return hipSuccess;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(idf_zero_f32_kernel, dim3(blocks), dim3(256), 0, st, p, n);
  return hipGetLastError();
}

// Helper workgroups of an under-filled launch: one dword per 128-byte line of [ptr, ptr + bytes) into this XCD's L2, results
// discarded -- the launch's weights were last touched a training step ago and would arrive from HBM in front of every K chunk
// (resblock8: cold 29.1 -> 24.3 us, profiles/r05_resblock_warm.txt).  Only where the main workgroups leave CUs idle: helpers of a
// launch that fills the chip run on its tail and cost more than they bring.
__device__ __forceinline__ void idf_warm_lines(const void* ptr, int bytes, int tid, int nthreads) {
  const char* base = reinterpret_cast<const char*>(ptr);
  for (int off = tid * 128; off < bytes; off += nthreads * 128) {
    unsigned d;
    asm volatile("global_load_dword %0, %1, off" : "=v"(d) : "v"(base + off) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
constexpr int IDF_WARM_HELPERS = 8, IDF_WARM_MAX_MAIN = 128;     // helpers per launch; main workgroups up to which a launch gets them


// (mean, rstd) of a GroupNorm group from its (sum, sum of squares): mu and var in double (the divisions by the element count are
// multiplications by its reciprocal), 1 / sqrt as v_rsq_f32 + one Newton step in double (2e-14 relative: the float it rounds to is the
// exact quotient's except on rounding ties).  The double-precision divide and square-root sequences it replaces cost ~1 us of
// dependent arithmetic in front of a workgroup's first MFMA (round 4: idf_resblock.hip, round 5: idf_conv_rs.hip, round 6: every fold).
__device__ __forceinline__ void idf_group_stats(double a, double d, double inv_n, float eps, float* mean, float* rstd) {
  const double mu = a * inv_n;
  double var = d * inv_n - mu * mu;
  if (var < 0.0) var = 0.0;
  const double vd = var + (double)eps;
  const double r0 = (double)__builtin_amdgcn_rsqf((float)vd);
  *rstd = (float)(r0 * (1.5 - 0.5 * vd * r0 * r0));
  *mean = (float)mu;
}
