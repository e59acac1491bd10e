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
          uint32_t lo = idf_pack_bf16(o[0], o[1]);
          uint32_t hi = idf_pack_bf16(o[2], o[3]);
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

// ---------------------------------------------------------------- several small fp32 GEMMs in ONE launch
// The conditioning path (modules.py:9-38 TimeEmbedding, models.py:298-301 fc_a, modules.py:269-276 the FiLM projections
// of every block) is a dependency chain of tiny dense products: as one launch per product (+ one per SiLU, per bias
// gradient, per split-K reduction) it was ~35 launches of 5-11 us each per training step.  Products of the same depth
// in the chain share a launch here: a block finds its job from the block index, then runs a 32 x 64 tile loop with the
// operand transforms the chain needs applied on the way into LDS:
//   a_op / b_op: 0 plain, 1 SiLU(v), 3 all ones (column sums = bias gradients as a product)
//   a_rows / b_rows: the operand's stored row r is row rows[r] of the source (the time-embedding table lookup)
//   c2: SiLU(C) stored beside C (what the next product contracts over);  cx: C *= SiLU'(cx) (the gradient through a SiLU)
//   nks > 1: the K range is cut into nks slices, slice s stores its partial product at C + s * c_kstride; a REDUCE job
//   (kind 1) of the next launch sums the slices in slice order (no atomics -- the gradient of the latent, and with it the
//   whole encoder backward pass, stays bit-reproducible):  C = (sum_s A[s]) * SiLU'(cx)
struct GemmJobD {
  const float* A; const float* B; float* C; float* c2; const float* cx; const float* bias;
  const long long* a_rows; const long long* b_rows;
  long c_kstride, a_pstride;
  int kind, lda, ldb, ldc, M, N, K, ta, tb, a_op, b_op, a_parts, kchunk, nks, ntn, ntiles;
};
constexpr int GM_MAXJ = 8;
struct GemmMultiP { GemmJobD j[GM_MAXJ]; int start[GM_MAXJ + 1]; int n; };

// Tile: 32 (M: the batch rows of most jobs) x 64 (N) outputs, 128 of K per step -- the FiLM widths (K = dim = 256) are two
// steps, the embedding products one.  These blocks are latency chains (a handful of blocks, each a sequence of dependent
// global loads at ~2 us a round trip when the weights were just rewritten by the optimizer), so a step is wide rather
// than the ring deep: twelve 16-byte loads per thread in flight, next step's issued before this step's MFMAs.
constexpr int GM_BM = 32, GM_BN = 64, GM_BK = 128, GM_PITCH = GM_BK + 4;

// one operand tile [TR rows][128 k] -> registers; layouts as fetch_tile.  LOADS ONLY, no branch and no use of the loaded
// value: a vector outside the operand (or past the K range) loads from the operand's first element, and gm_commit masks
// it when the tile goes to LDS.  (With a branch or a select next to the loads hipcc waits for them on the spot.)
// VEC: 16-byte loads; the host guarantees alignment AND that the contiguous extent is a multiple of 4 (a vector is wholly
// inside or wholly outside).  ROWS: stored row r is row rows[r] of the source.
template <int TR, bool VEC, bool ROWS>
__device__ __forceinline__ void gm_fetch(float (&v)[TR / 8][4], const float* __restrict__ src, const long long* __restrict__ rows,
                                         int ld, int trans, int op, int r0, int k0, int R, int Kend, int tid) {
#pragma unroll
  for (int i = 0; i < TR / 8; ++i) {
    // element e of the vector: (r, kk + e) when K-contiguous, (r + e, kk) when transposed
    const int r = trans ? (tid % (TR / 4)) * 4 : tid / 32 + i * 8;
    const int kk = trans ? tid / (TR / 4) + i * (1024 / TR) : (tid % 32) * 4;
    const int srow = trans ? k0 + kk : r0 + r, scol = trans ? r0 + r : k0 + kk;
    const int rowlim = trans ? Kend : R, collim = trans ? R : Kend;
    const bool ok = srow < rowlim && scol < collim && op != 3;
    long long row = srow;
    if (ROWS && rows) row = rows[ok ? srow : 0];
    const size_t base = ok ? (size_t)row * ld + scol : 0;
    if (VEC) {
      const float4 t4 = *reinterpret_cast<const float4*>(src + base);
      v[i][0] = t4.x; v[i][1] = t4.y; v[i][2] = t4.z; v[i][3] = t4.w;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[i][e] = src[(ok && scol + e < collim) ? base + e : 0];
    }
  }
}

// mask (outside the operand: 0; op 3: ones inside), transform (op 1: SiLU; SiLU(0) = 0 keeps the padding zero), store
template <int TR>
__device__ __forceinline__ void gm_commit(const float (&v)[TR / 8][4], float* lds, int trans, int op, int r0, int k0, int R,
                                          int Kend, int tid) {
#pragma unroll
  for (int i = 0; i < TR / 8; ++i) {
    const int r = trans ? (tid % (TR / 4)) * 4 : tid / 32 + i * 8;
    const int kk = trans ? tid / (TR / 4) + i * (1024 / TR) : (tid % 32) * 4;
    const int srow = trans ? k0 + kk : r0 + r, scol = trans ? r0 + r : k0 + kk;
    const int rowlim = trans ? Kend : R, collim = trans ? R : Kend;
    float x[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool in = srow < rowlim && scol + e < collim;
      const float t = op == 3 ? 1.f : (op == 1 ? silu_f(v[i][e]) : v[i][e]);
      x[e] = in ? t : 0.f;
    }
    if (!trans) *reinterpret_cast<float4*>(lds + r * GM_PITCH + kk) = make_float4(x[0], x[1], x[2], x[3]);
    else {
#pragma unroll
      for (int e = 0; e < 4; ++e) lds[(r + e) * GM_PITCH + kk] = x[e];
    }
  }
}

#ifdef IDF_GM_STAMP      // diagnostic build (tools/build_variant.sh gmstamp idf_gemm.hip -DIDF_GM_STAMP): phase cycle sums
__device__ unsigned long long g_gm_stamps[8];
__device__ __forceinline__ unsigned long long gm_now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  return t;
}
#define GM_STAMP(v) const unsigned long long v = gm_now()
#define GM_ADD(i, a, b) do { if (threadIdx.x == 0) atomicAdd(&g_gm_stamps[i], (b) - (a)); } while (0)
#else
#define GM_STAMP(v)
#define GM_ADD(i, a, b)
#endif

template <bool VEC, bool ROWS>
__global__ __launch_bounds__(256) void gemm_multi_kernel(const GemmMultiP mp) {
  __shared__ __attribute__((aligned(16))) float As[GM_BM * GM_PITCH];   // M side
  __shared__ __attribute__((aligned(16))) float Bs[GM_BN * GM_PITCH];   // N side
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  GM_STAMP(t_a);
  int ji = 0;
#pragma unroll
  for (int q = 1; q < GM_MAXJ; ++q)
    if (q < mp.n && (int)blockIdx.x >= mp.start[q]) ji = q;
  const GemmJobD p = mp.j[ji];      // by value: one block of scalar loads, not a reload per use
  const int local = blockIdx.x - mp.start[ji];
  GM_STAMP(t_b);
  GM_ADD(0, t_a, t_b);

  if (p.kind == 1) {      // C = (sum of a_parts slices of A) * SiLU'(cx), M * N elements (a multiple of 4), 1024 per block
    const size_t e = ((size_t)local * 256 + tid) * 4;
    if (e >= (size_t)p.M * p.N) return;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    for (int q0 = 0; q0 < p.a_parts; q0 += 8) {          // eight loads in flight, two chains in a fixed order
      float4 u[8];
#pragma unroll
      for (int q = 0; q < 8; ++q)
        u[q] = q0 + q < p.a_parts ? *reinterpret_cast<const float4*>(p.A + e + (size_t)(q0 + q) * p.a_pstride)
                                  : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int q = 0; q < 8; q += 2) {
        s0.x += u[q].x; s0.y += u[q].y; s0.z += u[q].z; s0.w += u[q].w;
        s1.x += u[q + 1].x; s1.y += u[q + 1].y; s1.z += u[q + 1].z; s1.w += u[q + 1].w;
      }
    }
    float4 o = make_float4(s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w);
    if (p.cx) {
      const float4 c = *reinterpret_cast<const float4*>(p.cx + e);
      o.x *= dsilu_f(c.x); o.y *= dsilu_f(c.y); o.z *= dsilu_f(c.z); o.w *= dsilu_f(c.w);
    }
    *reinterpret_cast<float4*>(p.C + e) = o;
    return;
  }

  const int ks = local / p.ntiles, tile = local - ks * p.ntiles;
  const int m0 = (tile / p.ntn) * GM_BM, n0 = (tile % p.ntn) * GM_BN;
  const int kbeg = ks * p.kchunk, kend = min(p.K, kbeg + p.kchunk);
  const int wm0 = (wave & 1) * 16, wn0 = (wave >> 1) * 32;       // a wave: 16 (M) x 32 (N)

  f32x4_t acc[2], acc2[2];      // two accumulator sets (even / odd 32-wide sub-steps): half the dependent MFMA chain
#pragma unroll
  for (int a = 0; a < 2; ++a) acc[a] = acc2[a] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  float ra[GM_BM / 8][4], rb[GM_BN / 8][4];
  gm_fetch<GM_BM, VEC, ROWS>(ra, p.A, p.a_rows, p.lda, p.ta, p.a_op, m0, kbeg, p.M, kend, tid);
  gm_fetch<GM_BN, VEC, ROWS>(rb, p.B, p.b_rows, p.ldb, p.tb, p.b_op, n0, kbeg, p.N, kend, tid);
  for (int k0 = kbeg; k0 < kend; k0 += GM_BK) {
    GM_STAMP(t_c);
    gm_commit<GM_BM>(ra, As, p.ta, p.a_op, m0, k0, p.M, kend, tid);
    gm_commit<GM_BN>(rb, Bs, p.tb, p.b_op, n0, k0, p.N, kend, tid);
    __syncthreads();
    GM_STAMP(t_d);
    GM_ADD(1, t_c, t_d);
    // the next step's tile (unconditional: past the K range it loads element 0 and is never committed)
    gm_fetch<GM_BM, VEC, ROWS>(ra, p.A, p.a_rows, p.lda, p.ta, p.a_op, m0, k0 + GM_BK, p.M, kend, tid);
    gm_fetch<GM_BN, VEC, ROWS>(rb, p.B, p.b_rows, p.ldb, p.tb, p.b_op, n0, k0 + GM_BK, p.N, kend, tid);
    const int fr = lane & 15, fk = (lane >> 4) * 8;
#pragma unroll
    for (int kk = 0; kk < GM_BK; kk += 32) {
      if (k0 + kk >= kend) break;
      Mma<float>::Frag nf[2], mf;
#pragma unroll
      for (int a = 0; a < 2; ++a) nf[a] = Mma<float>::ldfrag(Bs + (wn0 + a * 16 + fr) * GM_PITCH + kk + fk);
      mf = Mma<float>::ldfrag(As + (wm0 + fr) * GM_PITCH + kk + fk);
      if ((kk / 32) & 1) {
#pragma unroll
        for (int a = 0; a < 2; ++a) acc2[a] = Mma<float>::mma(nf[a], mf, acc2[a]);
      } else {
#pragma unroll
        for (int a = 0; a < 2; ++a) acc[a] = Mma<float>::mma(nf[a], mf, acc[a]);
      }
    }
    __syncthreads();
    GM_STAMP(t_e);
    GM_ADD(2, t_d, t_e);
  }
  GM_STAMP(t_f);
#pragma unroll
  for (int a = 0; a < 2; ++a) acc[a] += acc2[a];

  const int m = m0 + wm0 + (lane & 15);
  if (m >= p.M) return;
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int n = n0 + wn0 + a * 16 + (lane >> 4) * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (n + r >= p.N) continue;
      const size_t e = (size_t)m * p.ldc + n + r;
      float o = acc[a][r];
      if (p.bias && ks == 0) o += p.bias[n + r];
      if (p.nks > 1) { p.C[(size_t)ks * p.c_kstride + e] = o; continue; }
      if (p.cx) o *= dsilu_f(p.cx[e]);
      p.C[e] = o;
      if (p.c2) p.c2[e] = silu_f(o);
    }
  }
#ifdef IDF_GM_STAMP
  __builtin_amdgcn_s_waitcnt(0);
  GM_STAMP(t_g);
  GM_ADD(3, t_f, t_g);
  GM_ADD(4, t_a, t_g);
  if (threadIdx.x == 0) atomicAdd(&g_gm_stamps[5], 1ull);
#endif
}

struct GemmMulti {
  GemmMultiP mp;
  GemmMulti() { memset(&mp, 0, sizeof(mp)); }
  // C[M][N] = opA[M][K] * opB[N][K]^T (+ bias); returns the job for the optional fields
  GemmJobD* add(const float* A, int lda, int ta, int a_op, const float* B, int ldb, int tb, int b_op, float* C, int ldc, int M,
                int N, int K, const float* bias = nullptr, int splitk = 1) {
    if (mp.n >= GM_MAXJ || M <= 0 || N <= 0) return nullptr;
    GemmJobD& j = mp.j[mp.n];
    j.A = A; j.lda = lda; j.ta = ta; j.a_op = a_op; j.B = B; j.ldb = ldb; j.tb = tb; j.b_op = b_op; j.C = C; j.ldc = ldc;
    j.M = M; j.N = N; j.K = K; j.bias = bias; j.a_parts = 1;
    int kchunk = idf_cdiv(idf_cdiv(K, splitk < 1 ? 1 : splitk), 32) * 32;
    if (kchunk < 32) kchunk = 32;
    j.kchunk = kchunk;
    j.nks = K > 0 ? idf_cdiv(K, kchunk) : 1;
    j.ntn = idf_cdiv(N, GM_BN);
    j.ntiles = idf_cdiv(M, GM_BM) * j.ntn;
    if (a_op == 3) { j.A = B; j.lda = 0; }        // "ones": any valid address
    if (b_op == 3) { j.B = A; j.ldb = 0; }
    mp.start[mp.n + 1] = mp.start[mp.n] + j.ntiles * j.nks;
    return &mp.j[mp.n++];
  }
  // C[n] = (sum of `parts` slices of A, `pstride` apart) * SiLU'(cx), n elements (a multiple of 4; 16-byte aligned)
  void reduce(const float* A, int parts, long pstride, const float* cx, float* C, int n) {
    if (mp.n >= GM_MAXJ || n <= 0) return;
    GemmJobD& j = mp.j[mp.n];
    j.kind = 1; j.A = A; j.a_parts = parts; j.a_pstride = pstride; j.cx = cx; j.C = C; j.M = 1; j.N = n;
    mp.start[mp.n + 1] = mp.start[mp.n] + idf_cdiv(n, 1024);
    ++mp.n;
  }
  int launch(hipStream_t st) {
    if (mp.n == 0) return IDF_OK;
    bool vec = true, rows = false;
    for (int i = 0; i < mp.n; ++i) {
      const GemmJobD& j = mp.j[i];
      if (j.kind == 1) continue;
      rows = rows || j.a_rows || j.b_rows;
      // 16-byte loads: aligned bases and row pitches, contiguous extents that are multiples of 4
      if (j.a_op != 3) vec = vec && j.lda % 4 == 0 && (uintptr_t)j.A % 16 == 0 && (j.ta ? j.M : j.K) % 4 == 0;
      if (j.b_op != 3) vec = vec && j.ldb % 4 == 0 && (uintptr_t)j.B % 16 == 0 && (j.tb ? j.N : j.K) % 4 == 0;
    }
    const dim3 g(mp.start[mp.n]), b(256);
    if (vec && rows) hipLaunchKernelGGL((gemm_multi_kernel<true, true>), g, b, 0, st, mp);
    else if (vec) hipLaunchKernelGGL((gemm_multi_kernel<true, false>), g, b, 0, st, mp);
    else if (rows) hipLaunchKernelGGL((gemm_multi_kernel<false, true>), g, b, 0, st, mp);
    else hipLaunchKernelGGL((gemm_multi_kernel<false, false>), g, b, 0, st, mp);
    IDF_CHECK_LAUNCH();
    return IDF_OK;
  }
};

// K slices of the product dfilm [B][N] * W [N][dim] (N ~ 5k over 22 blocks): ~256 of K per slice
int film_splitk(int N) { return idf_cdiv(N, 256); }

}  // namespace

#ifdef IDF_GM_STAMP
extern "C" int idf_debug_gm_stamps(void** dev_addr) {
  return hipGetSymbolAddress(dev_addr, HIP_SYMBOL(g_gm_stamps)) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int idf_temb_film_parts(int N) {
  if (N <= 0) return 0;
  const int kchunk = idf_cdiv(idf_cdiv(N, film_splitk(N)), 32) * 32;
  return idf_cdiv(N, kchunk);
}

// ---- the conditioning path as one entry point per direction
// forward, three launches:
//   h1 = table[t] W1^T + b1, s1 = SiLU(h1)          | aemb = [SiLU](a) Wfc^T + bfc, sa = SiLU(aemb)
//   temb = s1 W2^T + b2, st = SiLU(temb)
//   film_t = st Wt^T + bt                            | film_a = sa Wa^T + ba
extern "C" int idf_temb_film_fwd(const long long* t, const float* table, int d_model, const float* W1, const float* b1,
                                 const float* W2, const float* b2, int dim, const float* a, int a_dim, const float* Wfc,
                                 const float* bfc, int fc_silu, const float* Wt, const float* bt, int Nt, const float* Wa,
                                 const float* ba, int Na, float* h1, float* s1, float* temb, float* st, float* aemb, float* sa,
                                 float* film_t, float* film_a, int B, void* stream) {
  if (B <= 0) return IDF_OK;
  if (!t || !table || !W1 || !W2 || !h1 || !s1 || !temb || !st) IDF_FAIL(IDF_ERR_BADARG, "temb_film_fwd: missing time-embedding operand");
  if (a && (!Wfc || !aemb || !sa)) IDF_FAIL(IDF_ERR_BADARG, "temb_film_fwd: missing latent-embedding operand");
  hipStream_t s = (hipStream_t)stream;
  {
    GemmMulti g;
    GemmJobD* j = g.add(table, d_model, 0, 0, W1, d_model, 0, 0, h1, dim, B, dim, d_model, b1);
    j->a_rows = t; j->c2 = s1;
    if (a) { j = g.add(a, a_dim, 0, fc_silu ? 1 : 0, Wfc, a_dim, 0, 0, aemb, dim, B, dim, a_dim, bfc); j->c2 = sa; }
    int rc = g.launch(s);
    if (rc) return rc;
  }
  {
    GemmMulti g;
    g.add(s1, dim, 0, 0, W2, dim, 0, 0, temb, dim, B, dim, dim, b2)->c2 = st;
    int rc = g.launch(s);
    if (rc) return rc;
  }
  GemmMulti g;
  if (Nt > 0) g.add(st, dim, 0, 0, Wt, dim, 0, 0, film_t, Nt, B, Nt, dim, bt);
  if (a && Na > 0) g.add(sa, dim, 0, 0, Wa, dim, 0, 0, film_a, Na, B, Na, dim, ba);
  return g.launch(s);
}

// backward, four launches (every gradient pointer optional):
//   K slices of dfilm_t Wt and dfilm_a Wa into dS | dWt = dfilm_t^T st, dbt = colsum(dfilm_t) | same for a
//   g_t = (sum of slices) * SiLU'(temb)           | g_a = (sum of slices) * SiLU'(aemb)          (in place of slice 0)
//   dW2 = g_t^T s1, db2 = colsum(g_t), g_1 = (g_t W2) * SiLU'(h1) | dWfc = g_a^T [SiLU](a), dbfc = colsum(g_a),
//                                                                   da = g_a Wfc [* SiLU'(a)]
//   dW1 = g_1^T table[t], db1 = colsum(g_1)
// scratch: dS, (idf_temb_film_parts(Nt) + idf_temb_film_parts(Na)) * B * dim floats; g1 [B][dim]
extern "C" int idf_temb_film_bwd(const float* dfilm_t, const float* dfilm_a, const long long* t, const float* table, int d_model,
                                 const float* W2, int dim, const float* a, int a_dim, const float* Wfc, int fc_silu,
                                 const float* Wt, int Nt, const float* Wa, int Na, const float* h1, const float* s1,
                                 const float* temb, const float* st, const float* aemb, const float* sa, float* dS,
                                 float* g1, float* dWt, float* dbt, float* dWa, float* dba, float* dW2, float* db2,
                                 float* dW1, float* db1, float* dWfc, float* dbfc, float* da, int B, void* stream) {
  if (B <= 0) return IDF_OK;
  if (!dS || !g1) IDF_FAIL(IDF_ERR_BADARG, "temb_film_bwd: missing scratch");
  if (((size_t)B * dim) % 4) IDF_FAIL(IDF_ERR_BADARG, "temb_film_bwd: B * dim must be a multiple of 4");
  hipStream_t s = (hipStream_t)stream;
  const int pt = idf_temb_film_parts(Nt), pa = idf_temb_film_parts(Na);
  const long ps = (long)B * dim;                     // one slice
  float* dSt = dS;
  float* dSa = dS + (size_t)pt * ps;
  const bool has_t = dfilm_t != nullptr && Nt > 0;
  const bool has_a = a != nullptr && dfilm_a != nullptr && Na > 0;
  {
    GemmMulti g;
    if (has_t) {
      g.add(dfilm_t, Nt, 0, 0, Wt, dim, 1, 0, dSt, dim, B, dim, Nt, nullptr, film_splitk(Nt))->c_kstride = ps;
      if (dWt) g.add(dfilm_t, Nt, 1, 0, st, dim, 1, 0, dWt, dim, Nt, dim, B);
      if (dbt) g.add(dfilm_t, Nt, 1, 0, nullptr, 0, 0, 3, dbt, 1, Nt, 1, B);
    }
    if (has_a) {
      g.add(dfilm_a, Na, 0, 0, Wa, dim, 1, 0, dSa, dim, B, dim, Na, nullptr, film_splitk(Na))->c_kstride = ps;
      if (dWa) g.add(dfilm_a, Na, 1, 0, sa, dim, 1, 0, dWa, dim, Na, dim, B);
      if (dba) g.add(dfilm_a, Na, 1, 0, nullptr, 0, 0, 3, dba, 1, Na, 1, B);
    }
    int rc = g.launch(s);
    if (rc) return rc;
  }
  {
    GemmMulti g;             // every element is read (all slices) and written (slice 0) by the same thread
    if (has_t) g.reduce(dSt, pt, ps, temb, dSt, (int)ps);
    if (has_a) g.reduce(dSa, pa, ps, aemb, dSa, (int)ps);
    int rc = g.launch(s);
    if (rc) return rc;
  }
  {
    GemmMulti g;
    if (has_t) {
      if (dW2) g.add(dSt, dim, 1, 0, s1, dim, 1, 0, dW2, dim, dim, dim, B);
      if (db2) g.add(dSt, dim, 1, 0, nullptr, 0, 0, 3, db2, 1, dim, 1, B);
      if (dW1 || db1) g.add(dSt, dim, 0, 0, W2, dim, 1, 0, g1, dim, B, dim, dim)->cx = h1;
    }
    if (has_a) {
      if (dWfc) g.add(dSa, dim, 1, 0, a, a_dim, 1, fc_silu ? 1 : 0, dWfc, a_dim, dim, a_dim, B);
      if (dbfc) g.add(dSa, dim, 1, 0, nullptr, 0, 0, 3, dbfc, 1, dim, 1, B);
      if (da) { GemmJobD* j = g.add(dSa, dim, 0, 0, Wfc, a_dim, 1, 0, da, a_dim, B, a_dim, dim); if (fc_silu) j->cx = a; }
    }
    int rc = g.launch(s);
    if (rc) return rc;
  }
  GemmMulti g;
  if (has_t && dW1) g.add(g1, dim, 1, 0, table, d_model, 1, 0, dW1, d_model, dim, d_model, B)->b_rows = t;
  if (has_t && db1) g.add(g1, dim, 1, 0, nullptr, 0, 0, 3, db1, 1, dim, 1, B);
  return g.launch(s);
}

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
