// Implicit-GEMM NHWC convolution on MFMA for gfx950: forward and data-gradient
// (one kernel, four gather modes), with the GroupNorm-apply + FiLM + SiLU +
// dropout prologue fused into the activation staging and bias / residual fused
// into the epilogue.  Replaces the reference's GroupNorm->SiLU->Dropout->Conv2d
// chains (modules.py:264-288, 312-320, 335-344), DownSample / UpSample convs
// (modules.py:63-93) and the 1x1 shortcut / attention projections
// (modules.py:133-136, 290-293).
//
//   out[m, n] = sum_{tap, c} act(x[gather(m, tap), c]) * w[n][tap][c] + bias[n] (+ res[m, n])
//
// GEMM view: M = B*Ho*Wo output pixels, N = Cout, K = taps*Cin.  MFMA operand A
// is the weight tile (rows = cout), operand B the pixel tile (cols = pixel), so
// each lane ends up with 4 consecutive couts of one pixel (one 8/16-byte store).
#include "idf_common.h"

namespace {

enum { MODE_S1 = 0, MODE_S2 = 1, MODE_UP2 = 2, MODE_T2 = 3 };

struct ConvP {
  const void* x;       // [B, Hs, Ws, Cin]
  const void* w;       // [Cout][taps][Cin]
  const float* bias;   // [Cout] or null
  const void* res;     // [B, Ho, Wo, Cout] or null
  void* y;             // [B, Ho, Wo, Cout]
  const float* sc;     // [B, Cin] prologue scale or null
  const float* sh;     // [B, Cin] prologue shift
  const uint64_t* seed;  // device pointer to the step seed, or null (no dropout)
  uint32_t salt;
  uint32_t drop_thresh;
  float drop_scale;
  int B, Hs, Ws, Cin, Ho, Wo, Cout;
  int mode, taps, act;   // act: 0 none, 1 affine, 2 affine + SiLU (+ dropout)
  int M;                 // B*Ho*Wo
  int n_tiles;           // ceil(Cout / BN)
};

// source pixel for output (oy, ox) and tap (ky, kx); returns false if padding
__device__ __forceinline__ bool gather(const ConvP& p, int oy, int ox, int ky, int kx, int& sy, int& sx) {
  int off = (p.taps == 9) ? 1 : 0;
  if (p.mode == MODE_S1) {
    sy = oy + ky - off; sx = ox + kx - off;
    return (unsigned)sy < (unsigned)p.Hs && (unsigned)sx < (unsigned)p.Ws;
  } else if (p.mode == MODE_S2) {
    sy = 2 * oy + ky - off; sx = 2 * ox + kx - off;
    return (unsigned)sy < (unsigned)p.Hs && (unsigned)sx < (unsigned)p.Ws;
  } else if (p.mode == MODE_UP2) {
    int iy = oy + ky - off, ix = ox + kx - off;
    sy = iy >> 1; sx = ix >> 1;
    return (unsigned)iy < (unsigned)(2 * p.Hs) && (unsigned)ix < (unsigned)(2 * p.Ws);
  } else {  // MODE_T2: transposed stride-2 (data gradient of MODE_S2), taps pre-flipped
    int ty = oy + ky - off, tx = ox + kx - off;
    sy = ty >> 1; sx = tx >> 1;
    return ty >= 0 && tx >= 0 && !(ty & 1) && !(tx & 1) && sy < p.Hs && sx < p.Ws;
  }
}

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  using Frag = bf16x8_t;
  __device__ static __forceinline__ Frag ldfrag(const bf16_t* lds) {
    return *reinterpret_cast<const Frag*>(lds);
  }
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
  // 8 k-steps of the exact-f32 MFMA; lane quarter q supplies k = 8q + j at step j
  __device__ static __forceinline__ f32x4_t mma(const Frag& a, const Frag& b, f32x4_t c) {
#pragma unroll
    for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], c, 0, 0, 0);
    return c;
  }
};

constexpr int BK = 32;

// BM pixels x BN couts per 256-thread block; waves 2 (pixel) x 2 (cout).
template <typename T, int BM, int BN, bool GENERIC>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvP p) {
  constexpr int VE = Elem<T>::VE;
  constexpr int VPR = BK / VE;            // 16-byte vectors per tile row
  constexpr int RPP = 256 / VPR;          // rows staged per pass
  constexpr int XP = BM / RPP;            // passes for the pixel tile
  constexpr int WP = (BN + RPP - 1) / RPP;
  constexpr int PITCH = BK + VE;          // padded row pitch (elements)
  constexpr int TM = BM / 2 / 16;         // pixel 16-tiles per wave
  constexpr int TN = BN / 2 / 16;         // cout 16-tiles per wave

  __shared__ __attribute__((aligned(16))) T Xs[BM * PITCH];
  __shared__ __attribute__((aligned(16))) T Ws[BN * PITCH];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile_n = blockIdx.x % p.n_tiles, tile_m = blockIdx.x / p.n_tiles;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int wm0 = (wave & 1) * (BM / 2), wn0 = (wave >> 1) * (BN / 2);

  const T* __restrict__ X = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ W = reinterpret_cast<const T*>(p.w);

  const int sub = tid % VPR, row0 = tid / VPR;
  // per-thread output-pixel coordinates of the rows it stages
  int pb[XP], poy[XP], pox[XP];
#pragma unroll
  for (int i = 0; i < XP; ++i) {
    int m = m0 + row0 + i * RPP;
    if (m < p.M) {
      int hw = p.Ho * p.Wo;
      pb[i] = m / hw;
      int r = m - pb[i] * hw;
      poy[i] = r / p.Wo;
      pox[i] = r - poy[i] * p.Wo;
    } else {
      pb[i] = -1; poy[i] = 0; pox[i] = 0;
    }
  }

  f32x4_t acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int cchunks = GENERIC ? 1 : p.Cin / BK;
  const int Ktot = p.taps * p.Cin;
  const int nk = GENERIC ? (Ktot + BK - 1) / BK : p.taps * cchunks;

  float xv[XP][VE];   // staged (already activated) pixel values
  float wv[WP][VE];

  auto stage_load = [&](int it) {
    if (!GENERIC) {
      const int tap = it / cchunks, c0 = (it - tap * cchunks) * BK + sub * VE;
      const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
      for (int i = 0; i < XP; ++i) {
        int sy, sx;
        bool ok = pb[i] >= 0 && gather(p, poy[i], pox[i], ky, kx, sy, sx);
        if (ok) {
          size_t e = ((size_t)(pb[i] * p.Hs + sy) * p.Ws + sx) * p.Cin + c0;
          Vec16<T>::load(X + e, xv[i]);
          if (p.act) {
            const float* scp = p.sc + (size_t)pb[i] * p.Cin + c0;
            const float* shp = p.sh + (size_t)pb[i] * p.Cin + c0;
#pragma unroll
            for (int v = 0; v < VE; ++v) {
              float u = xv[i][v] * scp[v] + shp[v];
              if (p.act == 2) {
                u = silu_f(u);
                if (p.seed) u = idf_keep(*p.seed, p.salt, e + v, p.drop_thresh) ? u * p.drop_scale : 0.f;
              }
              xv[i][v] = u;
            }
          }
        } else {
#pragma unroll
          for (int v = 0; v < VE; ++v) xv[i][v] = 0.f;
        }
      }
#pragma unroll
      for (int i = 0; i < WP; ++i) {
        int r = row0 + i * RPP, n = n0 + r;
        if (r < BN && n < p.Cout) {
          Vec16<T>::load(W + ((size_t)n * p.taps + tap) * p.Cin + c0, wv[i]);
        } else {
#pragma unroll
          for (int v = 0; v < VE; ++v) wv[i][v] = 0.f;
        }
      }
    } else {
      const int k0 = it * BK + sub * VE;
#pragma unroll
      for (int i = 0; i < XP; ++i) {
#pragma unroll
        for (int v = 0; v < VE; ++v) {
          int k = k0 + v;
          float val = 0.f;
          if (k < Ktot && pb[i] >= 0) {
            int tap = k / p.Cin, c = k - tap * p.Cin;
            int ky = tap / 3, kx = tap - ky * 3, sy, sx;
            if (gather(p, poy[i], pox[i], ky, kx, sy, sx))
              val = Elem<T>::ld(X + ((size_t)(pb[i] * p.Hs + sy) * p.Ws + sx) * p.Cin + c);
          }
          xv[i][v] = val;
        }
      }
#pragma unroll
      for (int i = 0; i < WP; ++i) {
        int r = row0 + i * RPP, n = n0 + r;
#pragma unroll
        for (int v = 0; v < VE; ++v) {
          int k = k0 + v;
          wv[i][v] = (r < BN && n < p.Cout && k < Ktot) ? Elem<T>::ld(W + (size_t)n * Ktot + k) : 0.f;
        }
      }
    }
  };

  auto stage_store = [&]() {
#pragma unroll
    for (int i = 0; i < XP; ++i) Vec16<T>::store(Xs + (row0 + i * RPP) * PITCH + sub * VE, xv[i]);
#pragma unroll
    for (int i = 0; i < WP; ++i) {
      int r = row0 + i * RPP;
      if (r < BN) Vec16<T>::store(Ws + r * PITCH + sub * VE, wv[i]);
    }
  };

  stage_load(0);
  for (int it = 0; it < nk; ++it) {
    stage_store();
    __syncthreads();
    if (it + 1 < nk) stage_load(it + 1);
    typename Mma<T>::Frag wf[TN], xf[TM];
    const int fr = lane & 15, fk = (lane >> 4) * 8;
#pragma unroll
    for (int a = 0; a < TN; ++a) wf[a] = Mma<T>::ldfrag(Ws + (wn0 + a * 16 + fr) * PITCH + fk);
#pragma unroll
    for (int b = 0; b < TM; ++b) xf[b] = Mma<T>::ldfrag(Xs + (wm0 + b * 16 + fr) * PITCH + fk);
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int b = 0; b < TM; ++b) acc[a][b] = Mma<T>::mma(wf[a], xf[b], acc[a][b]);
    __syncthreads();
  }

  // epilogue: lane holds couts n..n+3 of pixel m
  T* __restrict__ Y = reinterpret_cast<T*>(p.y);
  const T* __restrict__ R = reinterpret_cast<const T*>(p.res);
  const bool vec_ok = (p.Cout & 3) == 0;
#pragma unroll
  for (int b = 0; b < TM; ++b) {
    int m = m0 + wm0 + b * 16 + (lane & 15);
    if (m >= p.M) continue;
#pragma unroll
    for (int a = 0; a < TN; ++a) {
      int n = n0 + wn0 + a * 16 + (lane >> 4) * 4;
      if (n >= p.Cout) continue;
      float o[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
      size_t e = (size_t)m * p.Cout + n;
      if (vec_ok) {
        if (p.bias) {
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] += p.bias[n + r];
        }
        if (R) {
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] += Elem<T>::ld(R + e + r);
        }
        if (sizeof(T) == 4) {
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(Y) + e) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
          uint32_t lo = idf_pack_bf16(o[0], o[1]);
          uint32_t hi = idf_pack_bf16(o[2], o[3]);
          *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(Y) + e) = make_uint2(lo, hi);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (n + r < p.Cout) {
            float v = o[r] + (p.bias ? p.bias[n + r] : 0.f) + (R ? Elem<T>::ld(R + e + r) : 0.f);
            Elem<T>::st(Y + e + r, v);
          }
        }
      }
    }
  }
}

template <typename T>
int launch_conv(const ConvP& p0, hipStream_t st) {
  ConvP p = p0;
  bool generic = (p.Cin % BK) != 0;
  if (generic && p.act) IDF_FAIL(IDF_ERR_UNSUPPORTED, "conv: prologue needs Cin %% 32 == 0 (Cin=%d)", p.Cin);
  // small problems get the 64-pixel tile so the grid still covers the chip
  bool small = (long)p.M * p.Cout < (long)128 * 64 * 512;
  if (p.Cout <= 32) {
    p.n_tiles = idf_cdiv(p.Cout, 32);
    dim3 g(idf_cdiv(p.M, 64) * p.n_tiles);
    if (generic) hipLaunchKernelGGL((conv_igemm_kernel<T, 64, 32, true>), g, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((conv_igemm_kernel<T, 64, 32, false>), g, dim3(256), 0, st, p);
  } else if (small) {
    p.n_tiles = idf_cdiv(p.Cout, 64);
    dim3 g(idf_cdiv(p.M, 64) * p.n_tiles);
    if (generic) hipLaunchKernelGGL((conv_igemm_kernel<T, 64, 64, true>), g, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((conv_igemm_kernel<T, 64, 64, false>), g, dim3(256), 0, st, p);
  } else {
    p.n_tiles = idf_cdiv(p.Cout, 64);
    dim3 g(idf_cdiv(p.M, 128) * p.n_tiles);
    if (generic) hipLaunchKernelGGL((conv_igemm_kernel<T, 128, 64, true>), g, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((conv_igemm_kernel<T, 128, 64, false>), g, dim3(256), 0, st, p);
  }
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}


// ---------------------------------------------------------------------------
// Weight gradient:  dW[n][tap][c] = sum_m dy[m][n] * act(x[gather(m, tap)][c])
// GEMM view: rows = c (MFMA operand A), cols = n (operand B), reduction over the
// output pixels m.  Both operands are pixel-major in HBM, so tiles are loaded
// along channels and scattered into K(pixel)-contiguous LDS images.  All taps of a
// 32-pixel chunk are staged together (TG taps per phase) and share the dy tile.
// grid.x = c_tiles * n_tiles, grid.y = split over M; fp32 atomics into dW.
template <typename T, int TG>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const ConvP p, float* __restrict__ dW, int mchunk, int c_tiles) {
  constexpr int VE = Elem<T>::VE;
  constexpr int PITCH = BK + VE;
  constexpr int CT = 64, NT = 64;
  constexpr int NPH = 9 / TG;
  __shared__ __attribute__((aligned(16))) T Xs[TG * CT * PITCH];
  __shared__ __attribute__((aligned(16))) T Ds[NT * PITCH];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c0t = (blockIdx.x % c_tiles) * CT, n0t = (blockIdx.x / c_tiles) * NT;
  const int mbeg = blockIdx.y * mchunk, mend = min(p.M, mbeg + mchunk);
  const int wc0 = (wave & 1) * 32, wn0 = (wave >> 1) * 32;
  const T* __restrict__ X = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ DY = reinterpret_cast<const T*>(p.y);   // p.y carries dy here
  const bool vecx = (p.Cin % VE) == 0, vecn = (p.Cout % VE) == 0;
  const int ntaps = p.taps;
  const int nph = (ntaps == 1) ? 1 : NPH;
  const int tg = (ntaps == 1) ? 1 : TG;

  f32x4_t acc[9][2][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[t][a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int ml = tid % 32, vq = tid / 32;     // pixel within chunk, vector slot (0..7)
  const int hw = p.Ho * p.Wo;

  for (int mb = mbeg; mb < mend; mb += 32) {
    const int m = mb + ml;
    int b = -1, oy = 0, ox = 0;
    if (m < mend) { b = m / hw; int r = m - b * hw; oy = r / p.Wo; ox = r - oy * p.Wo; }
    // dy tile -> Ds[n][m]
    for (int v = vq; v < NT / VE; v += 8) {
      float dv[VE];
      int n = n0t + v * VE;
      if (b >= 0 && vecn && n < p.Cout) Vec16<T>::load(DY + (size_t)m * p.Cout + n, dv);
      else {
#pragma unroll
        for (int e = 0; e < VE; ++e) dv[e] = (b >= 0 && n + e < p.Cout) ? Elem<T>::ld(DY + (size_t)m * p.Cout + n + e) : 0.f;
      }
#pragma unroll
      for (int e = 0; e < VE; ++e) Elem<T>::st(Ds + (v * VE + e) * PITCH + ml, dv[e]);
    }
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) {
      if (ph >= nph) break;
      if (ph > 0) __syncthreads();   // previous phase's MFMA reads of Xs are done
      for (int tt = 0; tt < tg; ++tt) {
        const int tap = ph * tg + tt;
        const int ky = tap / 3, kx = tap - ky * 3;
        int sy, sx;
        bool ok = b >= 0 && gather(p, oy, ox, ky, kx, sy, sx);
        size_t pix = ok ? ((size_t)(b * p.Hs + sy) * p.Ws + sx) : 0;
        for (int v = vq; v < CT / VE; v += 8) {
          float xv[VE];
          int c = c0t + v * VE;
          if (ok && vecx && c < p.Cin) {
            size_t e0 = pix * p.Cin + c;
            Vec16<T>::load(X + e0, xv);
            if (p.act) {
#pragma unroll
              for (int e = 0; e < VE; ++e) {
                float u = xv[e] * p.sc[(size_t)b * p.Cin + c + e] + p.sh[(size_t)b * p.Cin + c + e];
                if (p.act == 2) {
                  u = silu_f(u);
                  if (p.seed) u = idf_keep(*p.seed, p.salt, e0 + e, p.drop_thresh) ? u * p.drop_scale : 0.f;
                }
                xv[e] = u;
              }
            }
          } else {
#pragma unroll
            for (int e = 0; e < VE; ++e)
              xv[e] = (ok && !vecx && c + e < p.Cin) ? Elem<T>::ld(X + pix * p.Cin + c + e) : 0.f;
          }
#pragma unroll
          for (int e = 0; e < VE; ++e) Elem<T>::st(Xs + (tt * CT + v * VE + e) * PITCH + ml, xv[e]);
        }
      }
      __syncthreads();
      typename Mma<T>::Frag nf[2];
      const int fr = lane & 15, fk = (lane >> 4) * 8;
#pragma unroll
      for (int bq = 0; bq < 2; ++bq) nf[bq] = Mma<T>::ldfrag(Ds + (wn0 + bq * 16 + fr) * PITCH + fk);
#pragma unroll
      for (int tt = 0; tt < TG; ++tt) {
        if (tt < tg) {
          typename Mma<T>::Frag cf[2];
#pragma unroll
          for (int a = 0; a < 2; ++a) cf[a] = Mma<T>::ldfrag(Xs + (tt * CT + wc0 + a * 16 + fr) * PITCH + fk);
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int bq = 0; bq < 2; ++bq) {
              acc[ph * TG + tt][a][bq] = Mma<T>::mma(cf[a], nf[bq], acc[ph * TG + tt][a][bq]);
            }
        }
      }
    }
    __syncthreads();   // before the next chunk overwrites Ds / Xs
  }

  // lane holds c..c+3 (rows) of column n
#pragma unroll
  for (int t = 0; t < 9; ++t) {
#pragma unroll
    for (int bq = 0; bq < 2; ++bq) {
      int n = n0t + wn0 + bq * 16 + (lane & 15);
      if (n >= p.Cout || t >= ntaps) continue;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        int c = c0t + wc0 + a * 16 + (lane >> 4) * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (c + r < p.Cin) atomicAdd(dW + ((size_t)n * ntaps + t) * p.Cin + c + r, acc[t][a][bq][r]);
      }
    }
  }
}

template <typename T, int TG>
int launch_wgrad(const ConvP& p, float* dW, hipStream_t st) {
  int c_tiles = idf_cdiv(p.Cin, 64), n_tiles = idf_cdiv(p.Cout, 64);
  int tiles = c_tiles * n_tiles;
  int split = idf_cdiv(768, tiles);
  int mchunk = idf_cdiv(idf_cdiv(p.M, split), 32) * 32;
  if (mchunk < 64) mchunk = 64;
  split = idf_cdiv(p.M, mchunk);
  hipLaunchKernelGGL((conv_wgrad_kernel<T, TG>), dim3(tiles, split), dim3(256), 0, st, p, dW, mchunk, c_tiles);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}

}  // namespace

extern "C" int idf_conv2d_fwd(const void* x, const void* w, const float* bias, const void* res, void* y,
                              const float* sc, const float* sh, const uint64_t* seed, uint32_t salt,
                              float p_drop, int B, int Hs, int Ws, int Cin, int Ho, int Wo, int Cout,
                              int mode, int taps, int act, int dtype, void* stream) {
  if (taps != 1 && taps != 9) IDF_FAIL(IDF_ERR_UNSUPPORTED, "conv: taps must be 1 or 9 (got %d)", taps);
  if (mode < 0 || mode > 3) IDF_FAIL(IDF_ERR_BADARG, "conv: bad mode %d", mode);
  if (act && (!sc || !sh)) IDF_FAIL(IDF_ERR_BADARG, "conv: act without sc/sh");
  ConvP p;
  memset(&p, 0, sizeof(p));
  p.x = x; p.w = w; p.bias = bias; p.res = res; p.y = y; p.sc = sc; p.sh = sh;
  p.seed = (act == 2 && p_drop > 0.f) ? seed : nullptr;
  p.salt = salt; p.drop_thresh = idf_drop_thresh(p_drop);
  p.drop_scale = 1.0f / (1.0f - (float)p.drop_thresh / 65536.0f);
  p.B = B; p.Hs = Hs; p.Ws = Ws; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.Cout = Cout;
  p.mode = mode; p.taps = taps; p.act = act; p.M = B * Ho * Wo;
  if (p.M == 0) return IDF_OK;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == IDF_F32) return launch_conv<float>(p, st);
  if (dtype == IDF_BF16) return launch_conv<bf16_t>(p, st);
  IDF_FAIL(IDF_ERR_BADARG, "conv: bad dtype %d", dtype);
}

// dW must be an fp32 [Cout][taps][Cin] buffer; it is zeroed here, then accumulated.
extern "C" int idf_conv2d_wgrad(const void* x, const void* dy, float* dW, const float* sc, const float* sh,
                                const uint64_t* seed, uint32_t salt, float p_drop, int B, int Hs, int Ws, int Cin,
                                int Ho, int Wo, int Cout, int mode, int taps, int act, int dtype, void* stream) {
  if (taps != 1 && taps != 9) IDF_FAIL(IDF_ERR_UNSUPPORTED, "wgrad: taps must be 1 or 9 (got %d)", taps);
  if (mode < 0 || mode > 2) IDF_FAIL(IDF_ERR_BADARG, "wgrad: bad mode %d", mode);
  if (act && (!sc || !sh)) IDF_FAIL(IDF_ERR_BADARG, "wgrad: act without sc/sh");
  int VE = dtype == IDF_F32 ? 4 : 8;
  if (act && (Cin % VE)) IDF_FAIL(IDF_ERR_UNSUPPORTED, "wgrad: prologue needs Cin %% %d == 0", VE);
  ConvP p;
  memset(&p, 0, sizeof(p));
  p.x = x; p.y = const_cast<void*>(dy); p.sc = sc; p.sh = sh;
  p.seed = (act == 2 && p_drop > 0.f) ? seed : nullptr;
  p.salt = salt; p.drop_thresh = idf_drop_thresh(p_drop);
  p.drop_scale = 1.0f / (1.0f - (float)p.drop_thresh / 65536.0f);
  p.B = B; p.Hs = Hs; p.Ws = Ws; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.Cout = Cout;
  p.mode = mode; p.taps = taps; p.act = act; p.M = B * Ho * Wo;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = idf_zero_f32(dW, (size_t)Cout * taps * Cin, st);     // a kernel, not a memset node (idf_common.h)
  if (e != hipSuccess) IDF_FAIL(IDF_ERR_HIP, "wgrad: memset failed: %s", hipGetErrorString(e));
  if (p.M == 0) return IDF_OK;
  if (dtype == IDF_F32) return launch_wgrad<float, 3>(p, dW, st);
  if (dtype == IDF_BF16) return launch_wgrad<bf16_t, 9>(p, dW, st);
  IDF_FAIL(IDF_ERR_BADARG, "wgrad: bad dtype %d", dtype);
}
