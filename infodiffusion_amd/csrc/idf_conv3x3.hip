// 3x3 convolution for bf16 NHWC activations on MFMA (gfx950), halo-tile form.
// Forward and data-gradient of every stride-1-shaped 3x3 conv on the path:
//   MODE 0  plain stride-1 conv (also the data gradient of one, with the flipped shadow)
//   MODE 1  stride-2 conv (DownSample, modules.py:63-75): the halo tile holds (2R+1) x (2W+1) source pixels
//   MODE 2  nearest-x2 upsample fused into the read (UpSample, modules.py:89-92)
//   MODE 3  zero-stuffed x2 input = transposed stride-2 (data gradient of DownSample)
//
//   y[pix][n] = sum_{tap, c} v[pix + tap][c] * w[n][tap][c] + bias[n] (+ res[pix][n])
// where v is the (virtual) input image the mode defines.  A block owns R output rows
// x W columns of one image (R*W = 64 or 128 pixels) x BN couts.  Per 32-channel chunk
// it stages ONE halo tile of (R+2) x (W+2) pixels plus the [9][BN][32] weight slab
// into LDS and runs all nine taps from LDS through shifted fragment addresses: the
// input is read from HBM/L2 once per chunk instead of once per tap, and one barrier
// pair covers 9 x the MFMA work of a per-tap implicit GEMM.  Next chunk's global
// loads are issued before the MFMAs and land in LDS after them.
// LDS rows are 64 B (32 bf16); the 16-byte chunk index is XOR-ed with
// 2*((row>>2)&1), which makes ds_read_b128 of any 16 consecutive rows conflict-free.
#include "idf_common.h"
#include <stdlib.h>

namespace {

struct C3P {
  const bf16_t* x;      // source activations [B, Hs, Ws, Cin]
  const bf16_t* x2;     // DUAL: channels C1.. of the input live here ([B, Hs, Ws, Cin - C1]); x holds [.., C1]
  int C1;
  const bf16_t* w;      // [Cout][9][Cin]
  const float* bias;    // [Cout] or null
  const bf16_t* res;    // [B, H, W, Cout] or null
  bf16_t* y;            // [B, H, W, Cout]
  int B, H, W, Hs, Ws, Cin, Cout;
  int R, tiles_per_img, n_tiles, wshift;
  unsigned wh_magic;    // (pix * wh_magic) >> 16 == pix / (W + 2*halo) for every halo pixel index (checked on the host)
};

constexpr int HALO_VEC_MAX_256 = 1280, HALO_VEC_MAX_512 = 2048, HALO_VEC_MAX_S2 = 1536;   // (R+2)*(W+2)*4 budget per block size
constexpr int CK = 32;

__device__ __forceinline__ int swz(int row, int q) { return q ^ (((row >> 2) & 1) << 1); }

// NWM = waves along the pixel axis (2 -> 256 threads; 4 -> 512 threads: a 256-pixel tile shares one
// weight slab, halving the slab re-reads from L2 and cutting the halo overhead from 2x to 1.5x).
// KS = 3 (3x3, pad 1) or 1 (1x1: the same pipeline without the halo -- the AttnBlock q/k/v and proj
// convs, ResBlock shortcuts and their data gradients; MODE 0 only).
// DUAL: the input is the never-materialised channel concatenation x | x2 (skip connection): a 32-channel
// chunk is fetched from the tensor it lies in (C1 % 32 == 0).
template <int MODE, int TM, int BN, int NWM, int KS = 3, bool DUAL = false>
__global__ __launch_bounds__(NWM * 128) void conv3x3_halo_bf16(const C3P p) {
  constexpr int NT = NWM * 128;               // threads (NWM x 2 waves)
  constexpr int TN = BN / 32;                 // cout 16-tiles per wave
  constexpr int TAPS = KS * KS, HALO = KS / 2;
  constexpr int WV = (BN * TAPS * 4 + NT - 1) / NT; // weight vectors per thread per chunk
  constexpr int BM = NWM * TM * 16;           // pixels per block
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int ST = MODE == 1 ? 2 : 1;       // output stride in the staged tile
  const int W = p.W, R = p.R;
  const int WH = MODE == 1 ? 2 * W + 1 : W + 2 * HALO;
  const int npix_h = (MODE == 1 ? 2 * R + 1 : R + 2 * HALO) * WH;
  const int KT = R * W;                       // valid pixels of the tile (<= BM)
  unsigned char* Xs = smem;                   // [npix_h][64 B]
  unsigned char* Ws = smem + (size_t)npix_h * 64;   // [TAPS][BN][64 B]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile = blockIdx.x / p.n_tiles, n0 = (blockIdx.x % p.n_tiles) * BN;
  const int b = tile / p.tiles_per_img, oy0 = (tile - b * p.tiles_per_img) * R;
  const int wm0 = (wave % NWM) * (TM * 16), wn0 = (wave / NWM) * (BN / 2);
  const int fr = lane & 15, fq = lane >> 4;

  // per-lane fragment bases
  int hbase[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    int pl = wm0 + i * 16 + fr;
    if (pl >= KT) pl = 0;
    int oy = pl >> p.wshift, ox = pl & (W - 1);
    hbase[i] = ST * (oy * WH + ox);
  }
  int wbase[TN];
#pragma unroll
  for (int a = 0; a < TN; ++a) {
    int n = wn0 + a * 16 + fr;
    wbase[a] = n * 64 + swz(n, fq) * 16;
  }

  constexpr int HV = (NWM == 4 ? HALO_VEC_MAX_512 : (MODE == 1 ? HALO_VEC_MAX_S2 : HALO_VEC_MAX_256)) / NT;   // halo vectors per thread
  f32x4_t acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int i = 0; i < TM; ++i) acc[a][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  uint4 hreg[HV], wreg[WV];
  const int nchunks = p.Cin / CK;

  // chunk-invariant staging plan: global element offsets (-1 = zero fill) and LDS byte offsets
  // (-1 = no slot), computed once so the chunk loop is loads + stores only
  int hoff[HV], woff[WV];      // element offsets (tensors < 2^31 elements: checked on the host)
  int hlds[HV], wlds[WV];
#pragma unroll
  for (int k = 0; k < HV; ++k) {
    int idx = tid + k * NT;
    hoff[k] = -1; hlds[k] = -1;
    if (idx < npix_h * 4) {
      int pix = idx >> 2, ch = idx & 3;
      int hy = (int)(((unsigned)pix * p.wh_magic) >> 16), hx = pix - hy * WH;
      int iy = ST * oy0 + hy - HALO, ix = hx - HALO;
      bool ok = (unsigned)iy < (unsigned)(ST * p.H) && (unsigned)ix < (unsigned)(ST * W);
      if (MODE == 3) ok = ok && !((iy | ix) & 1);
      if (MODE >= 2) { iy >>= 1; ix >>= 1; }
      if (ok) hoff[k] = DUAL ? (((b * p.Hs + iy) * p.Ws + ix) * 4 + ch)           // pixel index, vector slot
                             : ((b * p.Hs + iy) * p.Ws + ix) * p.Cin + ch * 8;
      hlds[k] = pix * 64 + swz(pix, ch) * 16;
    }
  }
#pragma unroll
  for (int k = 0; k < WV; ++k) {
    int idx = tid + k * NT;             // over [BN][TAPS][4]
    int ch = idx & 3, r = idx >> 2;
    int tap = r % TAPS, n = r / TAPS;
    woff[k] = -1; wlds[k] = -1;
    if (idx < BN * TAPS * 4) {
      if (n0 + n < p.Cout) woff[k] = ((n0 + n) * TAPS + tap) * p.Cin + ch * 8;
      wlds[k] = (tap * BN + n) * 64 + swz(n, ch) * 16;
    }
  }

  auto load_chunk = [&](int ck) {
    const int c0 = ck * CK;
    if (DUAL) {
      const bool first = c0 < p.C1;
      const bf16_t* src = first ? p.x + c0 : p.x2 + (c0 - p.C1);
      const int pitch = first ? p.C1 : p.Cin - p.C1;
#pragma unroll
      for (int k = 0; k < HV; ++k)
        hreg[k] = hoff[k] >= 0 ? *reinterpret_cast<const uint4*>(src + (size_t)(hoff[k] >> 2) * pitch + (hoff[k] & 3) * 8)
                               : make_uint4(0, 0, 0, 0);
    } else
#pragma unroll
    for (int k = 0; k < HV; ++k)
      hreg[k] = hoff[k] >= 0 ? *reinterpret_cast<const uint4*>(p.x + hoff[k] + c0) : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < WV; ++k)
      wreg[k] = woff[k] >= 0 ? *reinterpret_cast<const uint4*>(p.w + woff[k] + c0) : make_uint4(0, 0, 0, 0);
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int k = 0; k < HV; ++k)
      if (hlds[k] >= 0) *reinterpret_cast<uint4*>(Xs + hlds[k]) = hreg[k];
#pragma unroll
    for (int k = 0; k < WV; ++k)
      if (wlds[k] >= 0) *reinterpret_cast<uint4*>(Ws + wlds[k]) = wreg[k];
  };

  load_chunk(0);
  for (int ck = 0; ck < nchunks; ++ck) {
    store_chunk();
    __syncthreads();
    if (ck + 1 < nchunks) load_chunk(ck + 1);
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      const int toff = (tap / KS) * WH + (tap % KS);
      bf16x8_t wf[TN], xf[TM];
#pragma unroll
      for (int a = 0; a < TN; ++a) wf[a] = *reinterpret_cast<const bf16x8_t*>(Ws + tap * BN * 64 + wbase[a]);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        int h = hbase[i] + toff;
        xf[i] = *reinterpret_cast<const bf16x8_t*>(Xs + h * 64 + swz(h, fq) * 16);
      }
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int i = 0; i < TM; ++i)
          acc[a][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[a], xf[i], acc[a][i], 0, 0, 0);
    }
    __syncthreads();
  }

  // epilogue.  A lane holds couts n..n+3 of one pixel, i.e. 8-byte pieces scattered over 16 pixel
  // rows per store instruction.  When the cout tile is vector-aligned the fp32 tile goes through LDS
  // (the staging buffers are free now) and is written back as whole 16-byte chunks, consecutive lanes
  // covering one pixel's contiguous couts: full-line HBM writes, coalesced bias/residual reads.
  const int ncols = min(BN, p.Cout - n0);          // valid couts of this tile
  if ((p.Cout & 7) == 0) {
    constexpr int PF = BN + 4;                       // fp32 row pitch (floats)
    float* Os = reinterpret_cast<float*>(smem);      // [BM][PF]; fits: BM*(BN+4)*4 <= (halo + weights) bytes
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      int pl = wm0 + i * 16 + fr;
#pragma unroll
      for (int a = 0; a < TN; ++a) {
        int nl = wn0 + a * 16 + fq * 4;
        *reinterpret_cast<float4*>(Os + pl * PF + nl) = make_float4(acc[a][i][0], acc[a][i][1], acc[a][i][2], acc[a][i][3]);
      }
    }
    __syncthreads();
    constexpr int CPR = BN / 8;                      // 16-byte output chunks per pixel row
    for (int idx = tid; idx < BM * CPR; idx += NT) {
      int pl = idx / CPR, cc = (idx - pl * CPR) * 8;
      if (pl >= KT || cc >= ncols) continue;
      float o[8];
      float4 v0 = *reinterpret_cast<const float4*>(Os + pl * PF + cc);
      float4 v1 = *reinterpret_cast<const float4*>(Os + pl * PF + cc + 4);
      o[0] = v0.x; o[1] = v0.y; o[2] = v0.z; o[3] = v0.w; o[4] = v1.x; o[5] = v1.y; o[6] = v1.z; o[7] = v1.w;
      size_t e = ((size_t)(b * p.H + oy0) * W + pl) * p.Cout + n0 + cc;
      if (p.bias) {
        float4 b0 = *reinterpret_cast<const float4*>(p.bias + n0 + cc);
        float4 b1 = *reinterpret_cast<const float4*>(p.bias + n0 + cc + 4);
        o[0] += b0.x; o[1] += b0.y; o[2] += b0.z; o[3] += b0.w; o[4] += b1.x; o[5] += b1.y; o[6] += b1.z; o[7] += b1.w;
      }
      if (p.res) {
        float r[8];
        Vec16<bf16_t>::load(p.res + e, r);
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] += r[k];
      }
      Vec16<bf16_t>::store(p.y + e, o);
    }
    return;
  }
  // ragged cout counts (epsilon / latent heads): direct per-lane stores
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    int pl = wm0 + i * 16 + fr;
    if (pl >= KT) continue;
    size_t m = (size_t)(b * p.H + oy0) * W + pl;
#pragma unroll
    for (int a = 0; a < TN; ++a) {
      int n = n0 + wn0 + a * 16 + fq * 4;
      if (n >= p.Cout) continue;
      size_t e = m * p.Cout + n;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n + r < p.Cout) {
          float v = acc[a][i][r] + (p.bias ? p.bias[n + r] : 0.f) + (p.res ? bf16_to_f32(p.res[e + r]) : 0.f);
          p.y[e + r] = f32_to_bf16(v);
        }
    }
  }
}


// ---------------------------------------------------------------- direct-to-LDS variant
// Same tile, LDS image and MFMA loop as conv3x3_halo_bf16<MODE 0, TM 4, BN 64, NWM 4>, but a chunk is
// fetched with global_load_lds_dwordx4: each wave instruction moves one 1-KB group (16 LDS rows) from 64
// per-lane global addresses straight into LDS (lane l lands at group base + 16 l, so the lane picks the
// (row, channel-slot) whose swizzled home that is).  No staging registers and no ds_write phase: the kernel
// fits 128 VGPRs, TWO 512-thread blocks share a CU, and one block's MFMA phase covers the other's load
// latency (a block itself does not prefetch: its single LDS image is in use until the chunk's last read).
__device__ uint4 g_zero16;      // zero page: out-of-image halo pixels and padding rows load from here

template <int KS>
__global__ __launch_bounds__(512, 4) void conv_dlds_bf16(const C3P p) {
  constexpr int TM = 4, BN = 64, NWM = 4, NT = 512, TN = 2;
  constexpr int TAPS = KS * KS, HALO = KS / 2, BM = 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int W = p.W, R = p.R, WH = W + 2 * HALO;
  const int npix_h = (R + 2 * HALO) * WH;
  const int hgroups = (npix_h + 15) >> 4;            // 1-KB groups of the halo image (padded to 16 rows)
  constexpr int WGROUPS = TAPS * BN / 16;            // weight slab groups
  const int KT = R * W;
  unsigned char* Xs = smem;
  unsigned char* Ws = smem + (size_t)hgroups * 1024;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile = blockIdx.x / p.n_tiles, n0 = (blockIdx.x % p.n_tiles) * BN;
  const int b = tile / p.tiles_per_img, oy0 = (tile - b * p.tiles_per_img) * R;
  const int wm0 = (wave % NWM) * (TM * 16), wn0 = (wave / NWM) * (BN / 2);
  const int fr = lane & 15, fq = lane >> 4;

  int hbase[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    int pl = wm0 + i * 16 + fr;
    if (pl >= KT) pl = 0;
    int oy = pl >> p.wshift, ox = pl & (W - 1);
    hbase[i] = oy * WH + ox;
  }
  int wbase[TN];
#pragma unroll
  for (int a = 0; a < TN; ++a) {
    int n = wn0 + a * 16 + fr;
    wbase[a] = n * 64 + swz(n, fq) * 16;
  }
  f32x4_t acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int i = 0; i < TM; ++i) acc[a][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // group plan: wave w fetches groups w, w + 8, ...; per group this lane's global element offset
  // (relative to the chunk's first channel) or -1 for the zero page
  constexpr int MAXG = 9;                            // ceil((hgroups + WGROUPS) / 8) for every supported tile
  const int ngroups = hgroups + WGROUPS;
  int goff[MAXG];
  const bf16_t* gsrc[MAXG];
#pragma unroll
  for (int k = 0; k < MAXG; ++k) {
    const int gi = wave + k * 8;
    goff[k] = -1; gsrc[k] = p.x;
    if (gi < ngroups) {
      const int prow = lane >> 2, pslot = lane & 3;
      if (gi < hgroups) {
        const int pix = gi * 16 + prow;
        const int ch = pslot ^ (((pix >> 2) & 1) << 1);          // inverse of swz(): XOR is an involution
        if (pix < npix_h) {
          int hy = (int)(((unsigned)pix * p.wh_magic) >> 16), hx = pix - hy * WH;
          int iy = oy0 + hy - HALO, ix = hx - HALO;
          if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)W)
            goff[k] = ((b * p.H + iy) * W + ix) * p.Cin + ch * 8;
        }
      } else {
        const int r = (gi - hgroups) * 16 + prow;
        const int tap = r / BN, n = r - tap * BN;
        const int ch = pslot ^ (((n >> 2) & 1) << 1);
        gsrc[k] = p.w;
        if (n0 + n < p.Cout) goff[k] = ((n0 + n) * TAPS + tap) * p.Cin + ch * 8;
      }
    }
  }
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(&g_zero16);

  const int nchunks = p.Cin / CK;
  for (int ck = 0; ck < nchunks; ++ck) {
    const int c0 = ck * CK;
#pragma unroll
    for (int k = 0; k < MAXG; ++k) {
      const int gi = wave + k * 8;
      if (gi < ngroups) {
        const bf16_t* src = goff[k] >= 0 ? gsrc[k] + goff[k] + c0 : zero;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(smem + (size_t)gi * 1024), 16, 0, 0);
      }
    }
    __builtin_amdgcn_s_waitcnt(0);                   // the direct loads are counted by vmcnt
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      const int toff = (tap / KS) * WH + (tap % KS);
      bf16x8_t wf[TN], xf[TM];
#pragma unroll
      for (int a = 0; a < TN; ++a) wf[a] = *reinterpret_cast<const bf16x8_t*>(Ws + tap * BN * 64 + wbase[a]);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        int h = hbase[i] + toff;
        xf[i] = *reinterpret_cast<const bf16x8_t*>(Xs + h * 64 + swz(h, fq) * 16);
      }
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int i = 0; i < TM; ++i)
          acc[a][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[a], xf[i], acc[a][i], 0, 0, 0);
    }
    __syncthreads();
  }

  // epilogue through LDS (as conv3x3_halo_bf16; Cout % 8 == 0 is a launch condition)
  const int ncols = min(BN, p.Cout - n0);
  constexpr int OPF = BN + 4;
  float* Os = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    int pl = wm0 + i * 16 + fr;
#pragma unroll
    for (int a = 0; a < TN; ++a) {
      int nl = wn0 + a * 16 + fq * 4;
      *reinterpret_cast<float4*>(Os + pl * OPF + nl) = make_float4(acc[a][i][0], acc[a][i][1], acc[a][i][2], acc[a][i][3]);
    }
  }
  __syncthreads();
  constexpr int CPR = BN / 8;
  for (int idx = tid; idx < BM * CPR; idx += NT) {
    int pl = idx / CPR, cc = (idx - pl * CPR) * 8;
    if (pl >= KT || cc >= ncols) continue;
    float o[8];
    float4 v0 = *reinterpret_cast<const float4*>(Os + pl * OPF + cc);
    float4 v1 = *reinterpret_cast<const float4*>(Os + pl * OPF + cc + 4);
    o[0] = v0.x; o[1] = v0.y; o[2] = v0.z; o[3] = v0.w; o[4] = v1.x; o[5] = v1.y; o[6] = v1.z; o[7] = v1.w;
    size_t e = ((size_t)(b * p.H + oy0) * W + pl) * p.Cout + n0 + cc;
    if (p.bias) {
      float4 b0 = *reinterpret_cast<const float4*>(p.bias + n0 + cc);
      float4 b1 = *reinterpret_cast<const float4*>(p.bias + n0 + cc + 4);
      o[0] += b0.x; o[1] += b0.y; o[2] += b0.z; o[3] += b0.w; o[4] += b1.x; o[5] += b1.y; o[6] += b1.z; o[7] += b1.w;
    }
    if (p.res) {
      float r[8];
      Vec16<bf16_t>::load(p.res + e, r);
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] += r[k];
    }
    Vec16<bf16_t>::store(p.y + e, o);
  }
}

template <int KS>
void launch_dlds(const C3P& p, hipStream_t st) {
  const int HALO = KS / 2;
  const int npix_h = (p.R + 2 * HALO) * (p.W + 2 * HALO);
  size_t lds = ((size_t)((npix_h + 15) / 16) * 16 + KS * KS * 64) * 64;
  size_t olds = (size_t)256 * (64 + 4) * sizeof(float);
  if (olds > lds) lds = olds;
  auto kern = conv_dlds_bf16<KS>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, dim3(p.B * p.tiles_per_img * p.n_tiles), dim3(512), lds, st, p);
}

// magic multiplier for the division by the halo-row width; 0 when not exact over [0, npix)
inline unsigned wh_magic(int WH, int npix) {
  unsigned m = 65536u / (unsigned)WH + 1u;
  for (int i = 0; i < npix; ++i)
    if ((((unsigned)i * m) >> 16) != (unsigned)(i / WH)) return 0;
  return m;
}

template <int MODE, int TM, int BN, int NWM = 2, int KS = 3, bool DUAL = false>
void launch(const C3P& p, hipStream_t st) {
  size_t lds = ((MODE == 1 ? (size_t)(2 * p.R + 1) * (2 * p.W + 1) : (size_t)(p.R + 2 * (KS / 2)) * (p.W + 2 * (KS / 2))) +
                KS * KS * BN) * 64;
  size_t olds = (size_t)NWM * TM * 16 * (BN + 4) * sizeof(float);      // epilogue tile
  if (olds > lds) lds = olds;
  auto kern = conv3x3_halo_bf16<MODE, TM, BN, NWM, KS, DUAL>;
  if (lds > 64 * 1024) hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, dim3(p.B * p.tiles_per_img * p.n_tiles), dim3(NWM * 128), lds, st, p);
}

}  // namespace

// mode: 0 stride 1, 1 stride 2 (H, W <= 32 out), 2 nearest-x2-upsampled input, 3 zero-stuffed x2 input
// (transposed stride 2).
// H, W = OUTPUT dims.  Returns IDF_ERR_UNSUPPORTED for shapes it does not cover (the
// caller falls back to idf_conv2d_fwd).
extern "C" int idf_conv3x3_bf16(const void* x, const void* w, const float* bias, const void* res, void* y, int B,
                                int H, int W, int Cin, int Cout, int mode, void* stream) {
  if (mode < 0 || mode > 3 || (Cin % CK) || W < 4 || (W & (W - 1)) || W > 128 || (mode >= 2 && ((H | W) & 1)) ||
      (mode == 1 && (W > 32 || Cout <= 32)))
    IDF_FAIL(IDF_ERR_UNSUPPORTED, "conv3x3_bf16: B%d H%d W%d Cin%d Cout%d mode%d not covered", B, H, W, Cin, Cout, mode);
  if (B == 0) return IDF_OK;
  C3P p;
  p.x = (const bf16_t*)x; p.w = (const bf16_t*)w; p.bias = bias; p.res = (const bf16_t*)res; p.y = (bf16_t*)y;
  p.x2 = nullptr; p.C1 = Cin;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
  p.Hs = mode == 1 ? 2 * H : (mode ? H / 2 : H); p.Ws = mode == 1 ? 2 * W : (mode ? W / 2 : W);
  int ws = 0;
  while ((1 << ws) < W) ++ws;
  p.wshift = ws;
  // 128-pixel tiles when the problem is big enough to still fill the chip, else 64
  long M = (long)B * H * W;
  static const int force_bm = getenv("IDF_CONV_BM") ? atoi(getenv("IDF_CONV_BM")) : 0;
  int BM = ((M / 128) * idf_cdiv(Cout, 64) >= 256 && H * W >= 128) ? 128 : 64;
  // 512-thread blocks (256 pixels share one weight slab) once the grid still covers the chip
  if ((M / 256) * idf_cdiv(Cout, 64) >= 256 && H * W >= 256 && Cout > 32) BM = 256;
  if (force_bm && H * W >= force_bm && !(force_bm == 256 && Cout <= 32)) BM = force_bm;
  if (mode == 1) BM = 64;                       // the stride-2 halo tile is 4x the output tile
  int R;
  for (;;) {
    R = BM / W;
    if (R < 1) R = 1;
    if (R > H) R = H;
    while (H % R) --R;
    if (BM == 256 && (R + 2) * (W + 2) * 4 > HALO_VEC_MAX_512) { BM = 128; continue; }
    break;
  }
  const int hrows = mode == 1 ? 2 * R + 1 : R + 2, hcols = mode == 1 ? 2 * W + 1 : W + 2;
  if (hrows * hcols * 4 > (BM == 256 ? HALO_VEC_MAX_512 : (mode == 1 ? HALO_VEC_MAX_S2 : HALO_VEC_MAX_256)))
    IDF_FAIL(IDF_ERR_UNSUPPORTED, "conv3x3_bf16: halo too large (R%d W%d)", R, W);
  p.R = R; p.tiles_per_img = H / R;
  p.wh_magic = wh_magic(hcols, hrows * hcols);
  if (!p.wh_magic || (long)B * p.Hs * p.Ws * Cin >= (1L << 31) || (long)Cout * 9 * Cin >= (1L << 31))
    IDF_FAIL(IDF_ERR_UNSUPPORTED, "conv3x3_bf16: tensor too large for 32-bit offsets");
  hipStream_t st = (hipStream_t)stream;
  const bool bn32 = Cout <= 32;
  p.n_tiles = idf_cdiv(Cout, bn32 ? 32 : 64);
#define IDF_C3_LAUNCH(MODE)                                              \
  do {                                                                   \
    if (bn32) { if (BM == 128) launch<MODE, 4, 32>(p, st); else launch<MODE, 2, 32>(p, st); } \
    else { if (BM == 256) launch<MODE, 4, 64, 4>(p, st); else if (BM == 128) launch<MODE, 4, 64>(p, st); else launch<MODE, 2, 64>(p, st); } \
  } while (0)
  static const int dlds = getenv("IDF_CONV_DLDS") ? atoi(getenv("IDF_CONV_DLDS")) : 3;
  static const long dlds_min = getenv("IDF_CONV_DLDS_MIN") ? atol(getenv("IDF_CONV_DLDS_MIN")) : 1536;
  // direct-to-LDS variant: pays once two of its blocks share every CU (it does not prefetch within a block)
  if (mode == 0 && BM == 256 && !bn32 && (dlds & 1) && (Cout & 7) == 0 && (long)B * p.tiles_per_img * p.n_tiles >= dlds_min &&
      ((R + 2) * (W + 2) + 15) / 16 + 36 <= 72) launch_dlds<3>(p, st);
  else if (mode == 0) IDF_C3_LAUNCH(0);
  else if (mode == 1) launch<1, 2, 64>(p, st);
  else if (mode == 2) IDF_C3_LAUNCH(2);
  else IDF_C3_LAUNCH(3);
#undef IDF_C3_LAUNCH
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}

// 1x1 convolution (stride 1) forward / data gradient through the same pipeline (KS = 1): w [Cout][Cin].
// IDF_ERR_UNSUPPORTED for shapes it does not cover (the caller then uses idf_bgemm).
extern "C" int idf_conv1x1_bf16(const void* x, const void* x2, int C1, const void* w, const float* bias,
                                const void* res, void* y, int B, int H, int W, int Cin, int Cout, void* stream) {
  if (!x2) C1 = Cin;
  if ((Cin % CK) || W < 4 || (W & (W - 1)) || W > 128 || (Cout & 7) || (x2 && (C1 <= 0 || C1 >= Cin || (C1 % CK))))
    IDF_FAIL(IDF_ERR_UNSUPPORTED, "conv1x1_bf16: B%d H%d W%d Cin%d (C1 %d) Cout%d not covered", B, H, W, Cin, C1, Cout);
  if (B == 0) return IDF_OK;
  C3P p;
  p.x = (const bf16_t*)x; p.w = (const bf16_t*)w; p.bias = bias; p.res = (const bf16_t*)res; p.y = (bf16_t*)y;
  p.x2 = (const bf16_t*)x2; p.C1 = C1;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.Hs = H; p.Ws = W;
  int ws = 0;
  while ((1 << ws) < W) ++ws;
  p.wshift = ws;
  const long M = (long)B * H * W;
  const int nt = idf_cdiv(Cout, 64);
  int BM = 64;
  if ((M / 128) * nt >= 256 && H * W >= 128) BM = 128;
  if ((M / 256) * nt >= 256 && H * W >= 256) BM = 256;
  int R = BM / W;
  if (R < 1) R = 1;
  if (R > H) R = H;
  while (H % R) --R;
  if (R * W * 4 > (BM == 256 ? HALO_VEC_MAX_512 : HALO_VEC_MAX_256))
    IDF_FAIL(IDF_ERR_UNSUPPORTED, "conv1x1_bf16: tile too large (R%d W%d)", R, W);
  p.R = R; p.tiles_per_img = H / R; p.n_tiles = nt;
  p.wh_magic = wh_magic(W, R * W);
  if (!p.wh_magic || (long)B * H * W * Cin >= (1L << 31))
    IDF_FAIL(IDF_ERR_UNSUPPORTED, "conv1x1_bf16: tensor too large for 32-bit offsets");
  hipStream_t st = (hipStream_t)stream;
  static const int dlds = getenv("IDF_CONV_DLDS") ? atoi(getenv("IDF_CONV_DLDS")) : 3;
  static const long dlds_min = getenv("IDF_CONV_DLDS_MIN") ? atol(getenv("IDF_CONV_DLDS_MIN")) : 1536;
  if (!x2 && BM == 256 && (dlds & 2) && (long)B * p.tiles_per_img * nt >= dlds_min) launch_dlds<1>(p, st);
  else if (x2) {
    if (BM == 256) launch<0, 4, 64, 4, 1, true>(p, st);
    else if (BM == 128) launch<0, 4, 64, 2, 1, true>(p, st);
    else launch<0, 2, 64, 2, 1, true>(p, st);
  } else if (BM == 256) launch<0, 4, 64, 4, 1>(p, st);
  else if (BM == 128) launch<0, 4, 64, 2, 1>(p, st);
  else launch<0, 2, 64, 2, 1>(p, st);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}
