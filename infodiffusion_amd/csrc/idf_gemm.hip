// Batched MFMA GEMM for the true dense contractions on the path: the attention
// bmm's and their gradients (modules.py:152-159), the time/FiLM/latent linears and
// their gradients (modules.py:22-27, 269-276; models.py:244, 470-472), and the
// 1x1-conv weight gradient.
//
//   C[b][m][n] = alpha * sum_k opA(b)[m][k] * opB(b)[n][k]  (+ bias[n])
//
// opA is given either K-contiguous ([M][K], ta = 0) or transposed ([K][M], ta = 1);
// likewise opB ([N][K], tb = 0, or [K][N], tb = 1).  Tiles are staged to LDS in
// K-contiguous form either way (transposed sources are loaded along their
// contiguous dim and scattered), then consumed exactly like the conv kernel:
// MFMA operand A = the N side, operand B = the M side, so a lane owns 4
// consecutive n of one m.  Optional split-K (grid.z) accumulates with float atomics
// into a pre-zeroed fp32 C.
#include "idf_common.h"

namespace {

struct GemmP {
  const void* A; const void* B; void* C; const float* bias;
  const void* res;      // optional residual, same layout as C (storage dtype)
  long sA, sB, sC;      // batch strides (elements)
  int lda, ldb, ldc;
  int M, N, K;
  int ta, tb;
  int out_f32;          // C is float even when T is bf16
  int atomic;           // split-K: atomicAdd into fp32 C
  int kchunk;           // K range per grid.z slice (multiple of 32)
  float alpha;
};

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  using Frag = bf16x8_t;
  __device__ static __forceinline__ Frag ldfrag(const bf16_t* lds) { return *reinterpret_cast<const Frag*>(lds); }
  __device__ static __forceinline__ f32x4_t mma(const Frag& a, const Frag& b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
struct F32Frag { float v[8]; };
template <> struct Mma<float> {
  using Frag = F32Frag;
  __device__ static __forceinline__ Frag ldfrag(const float* lds) {
    Frag f;
    float4 a = *reinterpret_cast<const float4*>(lds);
    float4 b = *reinterpret_cast<const float4*>(lds + 4);
    f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w;
    f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
    return f;
  }
  __device__ static __forceinline__ f32x4_t mma(const Frag& a, const Frag& b, f32x4_t c) {
#pragma unroll
    for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], c, 0, 0, 0);
    return c;
  }
};

constexpr int BT = 64;   // 64 x 64 output tile; K step BKT = 32, or the whole (short) K at once: 128 (bf16) / 64 (fp32)

// A [64 rows][32 k] operand tile is staged into LDS (K-contiguous, padded pitch) in two halves:
// fetch_tile issues the global loads into registers (next tile, while the MFMAs of the current one
// run), commit_tile writes them to LDS.  src layout: trans = 0 -> element (r, k) at src[r*ld + k];
// trans = 1 -> src[k*ld + r] (loaded along its contiguous axis, scattered on commit).
template <typename T, int BKT> struct TileRegs { float v[BT * BKT / (Elem<T>::VE * 256)][Elem<T>::VE]; };

template <typename T, bool VEC, int BKT>
__device__ __forceinline__ void fetch_tile(TileRegs<T, BKT>& t, const T* src, int ld, int trans, int r0, int k0, int R,
                                           int Kend, int tid) {
  constexpr int VE = Elem<T>::VE;
  constexpr int NP = BT * BKT / (VE * 256);   // 16-byte vectors per thread per tile
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    int r, kk;
    const T* ptr;
    bool ok;
    if (!trans) {
      constexpr int VPR = BKT / VE, RPP = 256 / VPR;
      r = tid / VPR + i * RPP; kk = (tid % VPR) * VE;
      ok = r0 + r < R && k0 + kk < Kend;
      ptr = src + (size_t)(r0 + r) * ld + k0 + kk;
      if (VEC) {
        if (ok) Vec16<T>::load(ptr, t.v[i]);
      } else {
#pragma unroll
        for (int e = 0; e < VE; ++e) t.v[i][e] = (r0 + r < R && k0 + kk + e < Kend) ? Elem<T>::ld(ptr + e) : 0.f;
      }
    } else {
      constexpr int VPK = BT / VE, KPP = 256 / VPK;
      kk = tid / VPK + i * KPP; r = (tid % VPK) * VE;
      ok = k0 + kk < Kend && r0 + r < R;
      ptr = src + (size_t)(k0 + kk) * ld + r0 + r;
      if (VEC) {
        if (ok) Vec16<T>::load(ptr, t.v[i]);
      } else {
#pragma unroll
        for (int e = 0; e < VE; ++e) t.v[i][e] = (k0 + kk < Kend && r0 + r + e < R) ? Elem<T>::ld(ptr + e) : 0.f;
      }
    }
    if (VEC && !ok) {
#pragma unroll
      for (int e = 0; e < VE; ++e) t.v[i][e] = 0.f;
    }
  }
}

template <typename T, int BKT>
__device__ __forceinline__ void commit_tile(const TileRegs<T, BKT>& t, T* lds, int pitch, int trans, int tid) {
  constexpr int VE = Elem<T>::VE;
  constexpr int NP = BT * BKT / (VE * 256);
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    if (!trans) {
      constexpr int VPR = BKT / VE, RPP = 256 / VPR;
      Vec16<T>::store(lds + (tid / VPR + i * RPP) * pitch + (tid % VPR) * VE, t.v[i]);
    } else {
      constexpr int VPK = BT / VE, KPP = 256 / VPK;
      int k = tid / VPK + i * KPP, rr = (tid % VPK) * VE;
#pragma unroll
      for (int e = 0; e < VE; ++e) Elem<T>::st(lds + (rr + e) * pitch + k, t.v[i][e]);
    }
  }
}

template <typename T, bool VEC, int BKT>
__global__ __launch_bounds__(256) void bgemm_kernel(const GemmP p) {
  constexpr int VE = Elem<T>::VE;
  constexpr int PITCH = BKT + VE;
  __shared__ __attribute__((aligned(16))) T As[BT * PITCH];   // M side
  __shared__ __attribute__((aligned(16))) T Bs[BT * PITCH];   // N side
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntn = (p.N + BT - 1) / BT;
  const int m0 = (blockIdx.x / ntn) * BT, n0 = (blockIdx.x % ntn) * BT;
  const int bz = blockIdx.y;
  const int kbeg = blockIdx.z * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);
  const T* A = reinterpret_cast<const T*>(p.A) + (size_t)bz * p.sA;
  const T* B = reinterpret_cast<const T*>(p.B) + (size_t)bz * p.sB;
  const int wm0 = (wave & 1) * 32, wn0 = (wave >> 1) * 32;

  f32x4_t acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  TileRegs<T, BKT> ra, rb;
  if (kbeg < kend) {
    fetch_tile<T, VEC, BKT>(ra, A, p.lda, p.ta, m0, kbeg, p.M, kend, tid);
    fetch_tile<T, VEC, BKT>(rb, B, p.ldb, p.tb, n0, kbeg, p.N, kend, tid);
  }
  for (int k0 = kbeg; k0 < kend; k0 += BKT) {
    commit_tile<T, BKT>(ra, As, PITCH, p.ta, tid);
    commit_tile<T, BKT>(rb, Bs, PITCH, p.tb, tid);
    __syncthreads();
    if (k0 + BKT < kend) {      // next tile's loads fly during this tile's MFMAs
      fetch_tile<T, VEC, BKT>(ra, A, p.lda, p.ta, m0, k0 + BKT, p.M, kend, tid);
      fetch_tile<T, VEC, BKT>(rb, B, p.ldb, p.tb, n0, k0 + BKT, p.N, kend, tid);
    }
    const int fr = lane & 15, fk = (lane >> 4) * 8;
#pragma unroll
    for (int kk = 0; kk < BKT; kk += 32) {
      typename Mma<T>::Frag nf[2], mf[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) nf[a] = Mma<T>::ldfrag(Bs + (wn0 + a * 16 + fr) * PITCH + kk + fk);
#pragma unroll
      for (int b = 0; b < 2; ++b) mf[b] = Mma<T>::ldfrag(As + (wm0 + b * 16 + fr) * PITCH + kk + fk);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = Mma<T>::mma(nf[a], mf[b], acc[a][b]);
    }
    __syncthreads();
  }

  const T* R = reinterpret_cast<const T*>(p.res);
  const bool vec_out = !p.atomic && !(p.ldc & 3) && !(p.sC & 3);
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    int m = m0 + wm0 + b * 16 + (lane & 15);
    if (m >= p.M) continue;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      int n = n0 + wn0 + a * 16 + (lane >> 4) * 4;
      if (n >= p.N) continue;
      size_t e = (size_t)bz * p.sC + (size_t)m * p.ldc + n;
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        o[r] = acc[a][b][r] * p.alpha;
        if (n + r < p.N) {
          if (p.bias && blockIdx.z == 0) o[r] += p.bias[n + r];
          if (R) o[r] += Elem<T>::ld(R + e + r);
        }
      }
      if (vec_out && n + 3 < p.N) {
        if (p.out_f32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + e) = make_float4(o[0], o[1], o[2], o[3]);
        else {
          uint32_t lo = (uint32_t)f32_to_bf16(o[0]) | ((uint32_t)f32_to_bf16(o[1]) << 16);
          uint32_t hi = (uint32_t)f32_to_bf16(o[2]) | ((uint32_t)f32_to_bf16(o[3]) << 16);
          *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.C) + e) = make_uint2(lo, hi);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (n + r >= p.N) continue;
          if (p.atomic) atomicAdd(reinterpret_cast<float*>(p.C) + e + r, o[r]);
          else if (p.out_f32) reinterpret_cast<float*>(p.C)[e + r] = o[r];
          else Elem<T>::st(reinterpret_cast<T*>(p.C) + e + r, o[r]);
        }
      }
    }
  }
}

template <typename T>
int launch_gemm(const GemmP& p, int batch, int splitk, hipStream_t st) {
  constexpr int VE = Elem<T>::VE;
  constexpr int BIGK = VE == 8 ? 128 : 64;
  bool vec = (p.lda % VE == 0) && (p.ldb % VE == 0) && (p.sA % VE == 0) && (p.sB % VE == 0) &&
             (((uintptr_t)p.A | (uintptr_t)p.B) % 16 == 0) &&
             (p.ta ? (p.M % VE == 0) : (p.K % VE == 0)) && (p.tb ? (p.N % VE == 0) : (p.K % VE == 0));
  dim3 g(idf_cdiv(p.M, BT) * idf_cdiv(p.N, BT), batch, splitk);
  // short contractions (attention d = C, 1x1 convs, FiLM / time MLPs): take BIGK of K per barrier pair
  // so the load -> LDS -> MFMA chain is paid once or twice instead of K/32 times
  const bool big = vec && p.kchunk >= BIGK && (p.kchunk % BIGK) == 0;
  if (big) hipLaunchKernelGGL((bgemm_kernel<T, true, BIGK>), g, dim3(256), 0, st, p);
  else if (vec) hipLaunchKernelGGL((bgemm_kernel<T, true, 32>), g, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((bgemm_kernel<T, false, 32>), g, dim3(256), 0, st, p);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}

}  // namespace

extern "C" int idf_bgemm(const void* A, const void* B, void* C, const float* bias, const void* res, int batch,
                         long sA, long sB, long sC, int lda, int ldb, int ldc, int M, int N, int K, int ta, int tb, float alpha,
                         int out_f32, int splitk, int dtype, void* stream) {
  if (M <= 0 || N <= 0 || batch <= 0) return IDF_OK;
  GemmP p;
  memset(&p, 0, sizeof(p));
  p.A = A; p.B = B; p.C = C; p.bias = bias; p.res = res; p.sA = sA; p.sB = sB; p.sC = sC;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.ta = ta; p.tb = tb;
  p.alpha = alpha; p.out_f32 = out_f32 || dtype == IDF_F32;
  if (splitk < 1) splitk = 1;
  constexpr int BK = 32;
  int kchunk = idf_cdiv(idf_cdiv(K, splitk), BK) * BK;
  if (kchunk < BK) kchunk = BK;
  splitk = idf_cdiv(K, kchunk);
  if (splitk < 1) splitk = 1;
  p.kchunk = kchunk;
  p.atomic = splitk > 1;
  if (p.atomic && !p.out_f32) IDF_FAIL(IDF_ERR_BADARG, "bgemm: split-K needs an fp32 (pre-zeroed) C");
  if (p.atomic && res) IDF_FAIL(IDF_ERR_BADARG, "bgemm: residual with split-K is not supported");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == IDF_F32) return launch_gemm<float>(p, batch, splitk, st);
  if (dtype == IDF_BF16) return launch_gemm<bf16_t>(p, batch, splitk, st);
  IDF_FAIL(IDF_ERR_BADARG, "bgemm: bad dtype %d", dtype);
}
