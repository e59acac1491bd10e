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
#include "idf_conv3x3_parts.h"
#include <atomic>
#include <stdlib.h>
#include <type_traits>

namespace {

// NWM = waves along the pixel axis (2 -> 256 threads; 4 -> 512 threads: a 256-pixel tile shares one
// weight slab, halving the slab re-reads from L2 and cutting the halo overhead from 2x to 1.5x).
// KS = 3 (3x3, pad 1) or 1 (1x1: the same pipeline without the halo -- the AttnBlock q/k/v and proj
// convs, ResBlock shortcuts and their data gradients; MODE 0 only).
// DUAL: the input is the never-materialised channel concatenation x | x2 (skip connection): a 32-channel
// chunk is fetched from the tensor it lies in (C1 % 32 == 0).
// PRO: GroupNorm / FiLM / SiLU / dropout applied to the staged tile (MODE 0 only), coefficients folded in-block.
// GNB: the epilogue is the GroupNorm backward (gnb_epilogue): a data-gradient conv whose tile is one whole image.
// BWD: the backward chain at the big maps (MODE 0, BN 64, Cout % 64 == 0) -- bit 0: dy prologue (the staged input is
// A * du + K1 * xg + K0, two tensors per vector), bit 1: du epilogue (due_epilogue).
template <int MODE, int TM, int BN, int NWM, int KS = 3, bool DUAL = false, bool PRO = false, bool GNB = false, int BWD = 0>
__global__ __launch_bounds__(NWM * 128) void conv3x3_halo_bf16(const C3P p_in) {
  constexpr bool DYP = (BWD & 1) != 0, DUE = (BWD & 2) != 0;
  constexpr bool AUX_OK = MODE == 0 && KS == 3 && BN == 64 && !GNB && !DYP;     // instantiations that may carry an auxiliary job
  if (p_in.hw_main > 0 && (int)blockIdx.x >= p_in.hw_main) {      // helper workgroups of an under-filled launch
    idf_warm_lines(p_in.w, p_in.Cout * KS * KS * p_in.Cin * 2, threadIdx.x, NWM * 128);
    return;
  }
  C3P p = p_in;
  int bid = p_in.aux_blocks > 0 ? (int)blockIdx.x : xcd_tile_id(blockIdx.x, p_in.hw_main > 0 ? p_in.hw_main : (int)gridDim.x);
  bool aux = false;
  if constexpr (AUX_OK) {
    if (p_in.aux_blocks > 0 && bid >= p_in.main_blocks) {       // block-uniform
      aux = true;
      bid -= p_in.main_blocks;
      p.x = p_in.aux_x; p.x2 = p_in.aux_x2; p.C1 = p_in.aux_C1; p.Cin = p_in.aux_Cin; p.w = p_in.aux_w; p.bias = p_in.aux_bias;
      p.res = nullptr; p.y = p_in.aux_y; p.Cout = p_in.aux_Cout; p.n_tiles = p_in.aux_n_tiles; p.st_out = nullptr;
      p.a_out = nullptr;
    }
  }
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
  const int tile = bid / p.n_tiles, n0 = (bid % p.n_tiles) * BN;
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

  // staging registers of one chunk in flight (PFD = 2: two, for the small-tile launches -- see IDF_SMALL_PFD)
  struct Stage { uint4 h[HV]; uint4 w[WV]; };
  constexpr int PFD = (MODE == 0 && !DYP && TM == 2 && NWM == 2 && BN == 64) ? IDF_SMALL_PFD : 1;
  Stage st0, st1;
  uint4 xreg[DYP ? HV : 1];    // DYP: the GroupNorm input xg beside du
  const int nchunks = p.Cin / CK;

  // chunk-invariant staging plan: global element offsets (-1 = zero fill) and LDS byte offsets
  // (-1 = no slot), computed once so the chunk loop is loads + stores only
  int hoff[HV], woff[WV];      // element offsets (tensors < 2^31 elements: checked on the host)
  int hlds[HV], wlds[WV];
  unsigned amask = 0;          // PRO: vectors of this block's own pixels (written to a_out by the first cout tile)
#pragma unroll
  for (int k = 0; k < HV; ++k) {
    int idx = tid + k * NT;
    hoff[k] = -1; hlds[k] = -1;
    if (idx < npix_h * 4) {
      int pix = idx >> 2, ch = idx & 3;
      int hy = (int)(((unsigned)pix * p.wh_magic) >> 16), hx = pix - hy * WH;
      int iy = ST * oy0 + hy - HALO, ix = hx - HALO;
      bool ok = (unsigned)iy < (unsigned)(ST * p.H) && (unsigned)ix < (unsigned)(ST * W);
      if (((PRO && p.a_out) || (DYP && p.dyp_out)) && ok && n0 == 0 && (unsigned)(hy - HALO) < (unsigned)R &&
          (unsigned)(hx - HALO) < (unsigned)W)
        amask |= 1u << k;
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
      if (AUX_OK && aux) {                     // 1x1 weights [Cout][Cin]: the centre tap's slot only
        woff[k] = (tap == TAPS / 2 && n0 + n < p.Cout) ? (n0 + n) * p.Cin + ch * 8 : -1;
        if (tap != TAPS / 2) wlds[k] = -1;
      }
    }
  }

  auto load_chunk = [&](int ck, Stage& S) {
    uint4 (&hreg)[HV] = S.h;
    uint4 (&wreg)[WV] = S.w;
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
    if constexpr (DYP) {
#pragma unroll
      for (int k = 0; k < HV; ++k)
        xreg[k] = hoff[k] >= 0 ? *reinterpret_cast<const uint4*>(p.dyp_x + hoff[k] + c0) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < WV; ++k)
      wreg[k] = woff[k] >= 0 ? *reinterpret_cast<const uint4*>(p.w + woff[k] + c0) : make_uint4(0, 0, 0, 0);
  };
  float* cof = reinterpret_cast<float*>(smem + p.aux_off);      // PRO: [Cin][2] (sc, sh)
  uint64_t seedv = 0;
  bool drop = false;
  if (PRO && !aux) { drop = p.act == 2 && p.seed != nullptr; if (drop) seedv = *p.seed; }
  auto store_chunk = [&](int ck, Stage& S) {
    uint4 (&hreg)[HV] = S.h;
    uint4 (&wreg)[WV] = S.w;
    if constexpr (DYP) {
      const int cb = ck * CK + (tid & 3) * 8;         // this thread's 8 channels of the chunk (idx & 3 == tid & 3)
      float av[8], k1v[8], k0v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float4 t4 = *reinterpret_cast<const float4*>(cof + 4 * (cb + e));
        av[e] = t4.x; k1v[e] = t4.y; k0v[e] = t4.z;
      }
#pragma unroll
      for (int k = 0; k < HV; ++k)
        if (hoff[k] >= 0) {                           // padding pixels stay zero
          const uint32_t d4[4] = {hreg[k].x, hreg[k].y, hreg[k].z, hreg[k].w}, x4[4] = {xreg[k].x, xreg[k].y, xreg[k].z, xreg[k].w};
          uint32_t o4[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float lo = av[2 * i] * __uint_as_float(d4[i] << 16) + k1v[2 * i] * __uint_as_float(x4[i] << 16) + k0v[2 * i];
            const float hi = av[2 * i + 1] * __uint_as_float(d4[i] & 0xffff0000u) + k1v[2 * i + 1] * __uint_as_float(x4[i] & 0xffff0000u) + k0v[2 * i + 1];
            o4[i] = idf_pack_bf16(lo, hi);
          }
          hreg[k] = make_uint4(o4[0], o4[1], o4[2], o4[3]);
          if ((amask >> k) & 1u) *reinterpret_cast<uint4*>(p.dyp_out + (unsigned)(hoff[k] + ck * CK)) = hreg[k];
        }
    }
    if (PRO && !aux) {
      const int cb = ck * CK + (tid & 3) * 8;         // this thread's 8 channels of the chunk (idx & 3 == tid & 3)
      float scv[8], shv[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float4 t4 = *reinterpret_cast<const float4*>(cof + 2 * cb + 4 * q);
        scv[2 * q] = t4.x; shv[2 * q] = t4.y; scv[2 * q + 1] = t4.z; shv[2 * q + 1] = t4.w;
      }
      // the activation / dropout switches are launch-uniform: resolved once per chunk, each case straight-line (inside pro_vec they
      // cost a branch, register shuffles and a partial copy of the body per vector: ~200 instructions per vector instead of ~110)
      auto xform = [&](auto silu_c, auto drop_c) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < HV; ++k)
          if (hoff[k] >= 0) {
            const unsigned e0 = DUAL ? (unsigned)(hoff[k] >> 2) * (unsigned)p.Cin + (unsigned)cb
                                     : (unsigned)(hoff[k] + ck * CK);
            hreg[k] = pro_vec_t<decltype(silu_c)::value, decltype(drop_c)::value, IDF_HALO_PRO_G>(hreg[k], scv, shv, seedv, p.salt, p.thr,
                                                                                                  p.dscale, e0 >> 3);
            if ((amask >> k) & 1u) *reinterpret_cast<uint4*>(p.a_out + e0) = hreg[k];
          }
      };
      if (p.act != 2) xform(std::false_type{}, std::false_type{});
      else if (drop) xform(std::true_type{}, std::true_type{});
      else xform(std::true_type{}, std::false_type{});
    }
#pragma unroll
    for (int k = 0; k < HV; ++k)
      if (hlds[k] >= 0) *reinterpret_cast<uint4*>(Xs + hlds[k]) = hreg[k];
#pragma unroll
    for (int k = 0; k < WV; ++k)
      if (wlds[k] >= 0) *reinterpret_cast<uint4*>(Ws + wlds[k]) = wreg[k];
  };

  load_chunk(0, st0);
  if (PFD == 2 && nchunks > 1) load_chunk(1, st1);
  GnbPre<GNB ? BM * (BN / 8) / NT : 1> gpre;
  if constexpr (GNB) gnb_prefetch<BM, BN, NT>(p, b, n0, KT, tid, gpre);
  if (PRO && !aux) pro_coefficients<NT>(p, b, oy0 == 0 && n0 == 0, cof, cof + 2 * p.Cin, tid);
  if constexpr (DYP) gn_bwd_fold<NT>(p.dyp_f, b, oy0 == 0 && n0 == 0, cof, cof + 4 * p.Cin, tid);
  uint4 due_xr[DUE ? BM * (BN / 8) / NT : 1];
  constexpr bool RES_PF = IDF_RES_PF && !DUE && !GNB && MODE == 0 && BN == 64;   // residual vectors fetched under the last MFMA phase
  uint4 res_rr[RES_PF ? (BM * (BN / 8) + NT - 1) / NT : 1];
  const bool res_pf = RES_PF && p.res != nullptr && (p.Cout & 7) == 0;
  auto step = [&](int ck, Stage& S) {
    store_chunk(ck, S);
    __syncthreads();
    if (ck + PFD < nchunks) load_chunk(ck + PFD, S);     // the stage's registers are free again: next chunk of this slot
    if constexpr (DUE) {
      // the du epilogue's x vectors fly during the last chunk's MFMAs (IDF_DUE_PF; issued earlier they sit in front of
      // every chunk's vmcnt wait)
      if (IDF_DUE_PF && ck == nchunks - 1 && !aux) due_fetch_x<BM, BN, NT>(p, due_xr, b, oy0, n0, KT, tid);
    }
    if constexpr (RES_PF) {
      if (res_pf && ck == nchunks - 1) res_fetch<BM, BN, NT>(p, res_rr, b, oy0, n0, KT, tid);
    }
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      if (AUX_OK && aux && tap != TAPS / 2) continue;       // (block-uniform)
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
  };
  for (int ck = 0; ck < nchunks; ck += PFD) {
    step(ck, st0);
    if (PFD == 2 && ck + 1 < nchunks) step(ck + 1, st1);
  }
  if constexpr (DUE) {
    if (!aux) {
      if (!IDF_DUE_PF) due_fetch_x<BM, BN, NT>(p, due_xr, b, oy0, n0, KT, tid);
      due_epilogue<TM, TN, BM, BN, NT>(p, acc, smem, b, oy0, n0, KT, tid, wm0, wn0, due_xr);
      return;
    }
  }

  // epilogue.  A lane holds couts n..n+3 of one pixel, i.e. 8-byte pieces scattered over 16 pixel
  // rows per store instruction.  When the cout tile is vector-aligned the fp32 tile goes through LDS
  // (the staging buffers are free now) and is written back as whole 16-byte chunks, consecutive lanes
  // covering one pixel's contiguous couts: full-line HBM writes, coalesced bias/residual reads.
  const int ncols = min(BN, p.Cout - n0);          // valid couts of this tile
  if constexpr (GNB) {
    gnb_epilogue<TM, TN, BM, BN, NT>(p, acc, smem, b, n0, KT, tid, wm0, wn0, gpre);
    return;
  }
  if ((p.Cout & 7) == 0) {
    if constexpr (RES_PF) lds_epilogue<TM, TN, BM, BN, NT>(p, acc, smem, b, oy0, n0, KT, tid, wm0, wn0, res_rr, res_pf);
    else {
      uint4 none[(BM * (BN / 8) + NT - 1) / NT];
      lds_epilogue<TM, TN, BM, BN, NT>(p, acc, smem, b, oy0, n0, KT, tid, wm0, wn0, none, false);
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
__device__ uint4 g_trash[16];   // where the persistent kernel's epilogue sends the stores of out-of-range elements (branch-free)

// Diagnostic build only (tools/build_variant.sh NAME idf_conv3x3.hip -DIDF_DLDS_STAMP; never in the shipped library): per-phase
// cycle sums of conv_dlds_bf16 -- wave 0 of every block stamps s_memtime at the phase boundaries and adds the differences
// here: [0] prologue (plan + coefficients), [1] load wait, [2] GroupNorm transform, [3] MFMA, [4] epilogue, [5] blocks.
#ifdef IDF_DLDS_STAMP
__device__ unsigned long long g_dlds_stamps[64 * 8];     // 64 shards (block id mod 64): no same-address atomic pile-up
__device__ __forceinline__ unsigned long long dlds_now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  return t;
}
#define DLDS_STAMP(var) const unsigned long long var = dlds_now()
#define DLDS_DECL unsigned long long dlds_sum[6] = {0, 0, 0, 0, 0, 1}
#define DLDS_ADD(i, a, b) dlds_sum[i] += (b) - (a)
#define DLDS_FLUSH do { if (threadIdx.x == 0) for (int i_ = 0; i_ < 6; ++i_) atomicAdd(&g_dlds_stamps[(blockIdx.x & 63) * 8 + i_], dlds_sum[i_]); } while (0)
// the diagnostic build's tool (tools/dlds_stamps.py) reads and clears the 64 x 8 counters through this device address
extern "C" int idf_debug_dlds_stamps(void** dev_addr) {
  return hipGetSymbolAddress(dev_addr, HIP_SYMBOL(g_dlds_stamps)) == hipSuccess ? 0 : 1;
}
#else
#define DLDS_STAMP(var)
#define DLDS_DECL
#define DLDS_ADD(i, a, b)
#define DLDS_FLUSH
#endif

// PRO: the GroupNorm prologue as an in-LDS pass -- once the chunk has landed every thread reads its own vectors
// back, applies act(x * sc + sh) and writes them in place (one more barrier per chunk); with two blocks per CU the
// other block's MFMA phase runs beside this VALU phase.
// DUAL: the input is the never-materialised concatenation x | x2 -- a chunk's per-lane source addresses point into
// the tensor the chunk lies in.
template <int KS, bool PRO = false, bool DUAL = false>
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

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform: group indices stay scalar
  const int bidx = xcd_tile_id(blockIdx.x, gridDim.x);
  const int tile = bidx / p.n_tiles, n0 = (bidx % p.n_tiles) * BN;
  const int b = tile / p.tiles_per_img, oy0 = (tile - b * p.tiles_per_img) * R;
  const int wm0 = (wave % NWM) * (TM * 16), wn0 = (wave / NWM) * (BN / 2);
  const int fr = lane & 15, fq = lane >> 4;
  DLDS_DECL;
  DLDS_STAMP(t_begin);

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

  // group plan: wave w fetches halo groups w, w + 8, ... and weight groups w, w + 8, ...; per halo group this
  // lane's global element offset (relative to the chunk's first channel) or -1 for the zero page is kept; the
  // weight offsets are recomputed per chunk (a handful of integer operations: registers matter more here)
  constexpr int HGM = 5, WGM = (WGROUPS + 7) / 8;    // halo groups per wave (<= 36 groups per tile), weight groups per wave
  const int prow = lane >> 2;
  const int chl = (lane & 3) ^ (((lane >> 4) & 1) << 1);   // logical 8-channel slot of this lane's 16 bytes: swz() inverted;
                                                           // (row >> 2) & 1 == (lane >> 4) & 1 for rows 16 g + (lane >> 2)
  int goff[HGM];
#pragma unroll
  for (int k = 0; k < HGM; ++k) {
    const int gi = wave + k * 8;
    goff[k] = -1;
    const int pix = gi * 16 + prow;
    if (gi < hgroups && pix < npix_h) {
      int hy = (int)(((unsigned)pix * p.wh_magic) >> 16), hx = pix - hy * WH;
      int iy = oy0 + hy - HALO, ix = hx - HALO;
      if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)W)
        goff[k] = DUAL ? (((b * p.H + iy) * W + ix) * 4 + chl)            // pixel index, vector slot
                       : ((b * p.H + iy) * W + ix) * p.Cin + chl * 8;
    }
  }
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(&g_zero16);

  // PRO: this thread's vectors of the halo image (logical channel slot tid & 3, as the register-staged kernel)
  constexpr int HVD = PRO ? 5 : 1;                   // ceil(576 * 4 / 512): the largest halo tile this kernel takes
  float* cof = reinterpret_cast<float*>(smem + p.aux_off);
  uint64_t seedv = 0;
  bool drop = false;
  if (PRO) {
    drop = p.act == 2 && p.seed != nullptr;
    if (drop) seedv = *p.seed;
    pro_coefficients<NT>(p, b, oy0 == 0 && n0 == 0, cof, cof + 2 * p.Cin, tid);
  }
  const bool keep_a = PRO && p.a_out && n0 == 0;
  DLDS_STAMP(t_pro);
  DLDS_ADD(0, t_begin, t_pro);

  const int nchunks = p.Cin / CK;
  for (int ck = 0; ck < nchunks; ++ck) {
    DLDS_STAMP(t0);
    const int c0 = ck * CK;
    const bf16_t* dbase = p.x;      // DUAL: the tensor this chunk lies in, advanced to the chunk's first channel
    int dpitch = p.Cin;
    if (DUAL) {
      if (c0 < p.C1) { dbase = p.x + c0; dpitch = p.C1; }
      else { dbase = p.x2 + (c0 - p.C1); dpitch = p.Cin - p.C1; }
    }
#pragma unroll
    for (int k = 0; k < HGM; ++k) {
      const int gi = wave + k * 8;
      if (gi < hgroups) {
        const bf16_t* src;
        if (DUAL)
          src = goff[k] >= 0 ? dbase + ((unsigned)(goff[k] >> 2) * (unsigned)dpitch + (unsigned)(goff[k] & 3) * 8u) : zero;
        else
          src = goff[k] >= 0 ? p.x + goff[k] + c0 : zero;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(smem + (size_t)gi * 1024), 16, 0, 0);
      }
    }
#pragma unroll
    for (int k = 0; k < WGM; ++k) {
      const int gw = wave + k * 8;
      if (gw < WGROUPS) {
        const int r = gw * 16 + prow, tap = r / BN, n = r - tap * BN;
        const bf16_t* src = (n0 + n < p.Cout) ? p.w + (((n0 + n) * TAPS + tap) * p.Cin + chl * 8 + c0) : zero;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(Ws + (size_t)gw * 1024), 16, 0, 0);
      }
    }
    __builtin_amdgcn_s_waitcnt(0);                   // the direct loads are counted by vmcnt
    __syncthreads();
    DLDS_STAMP(t1);
    DLDS_ADD(1, t0, t1);
    if (PRO) {
      const int cb = c0 + (tid & 3) * 8;
      float scv[8], shv[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float4 t4 = *reinterpret_cast<const float4*>(cof + 2 * cb + 4 * q);
        scv[2 * q] = t4.x; shv[2 * q] = t4.y; scv[2 * q + 1] = t4.z; shv[2 * q + 1] = t4.w;
      }
      // the plan is recomputed per chunk (a magic-number division per vector) rather than kept in registers:
      // two of these blocks must fit a CU
      // (the once-per-chunk switch of the halo kernel was tried here: at this kernel's 128-register budget -- two blocks per CU -- the
      // straight-line bodies spill, 28.2 -> 35.1 us at 64->64 @64x64; the element-wise runtime form stays)
#pragma unroll 1
      for (int k = 0; k < HVD; ++k) {
        const int pix = (tid + k * NT) >> 2;
        if (pix < npix_h) {
          const int hy = (int)(((unsigned)pix * p.wh_magic) >> 16), hx = pix - hy * WH;
          const int iy = oy0 + hy - HALO, ix = hx - HALO;
          if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)W) {      // padding pixels stay zero
            uint4* slot = reinterpret_cast<uint4*>(Xs + pix * 64 + swz(pix, tid & 3) * 16);
            const unsigned e0 = (unsigned)(((b * p.H + iy) * W + ix) * p.Cin + cb);
            const uint4 v = pro_vec<IDF_DLDS_PRO_G>(*slot, scv, shv, p.act, drop, seedv, p.salt, p.thr, p.dscale, e0 >> 3);
            *slot = v;
            if (keep_a && (unsigned)(hy - HALO) < (unsigned)R && (unsigned)(hx - HALO) < (unsigned)W)
              *reinterpret_cast<uint4*>(p.a_out + e0) = v;
          }
        }
      }
      __syncthreads();
    }
    DLDS_STAMP(t2);
    DLDS_ADD(2, t1, t2);
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
    DLDS_STAMP(t3);
    DLDS_ADD(3, t2, t3);
  }

  // epilogue through LDS (Cout % 8 == 0 is a launch condition)
  DLDS_STAMP(t4);
  {
    uint4 none[(BM * (BN / 8) + NT - 1) / NT];
    lds_epilogue<TM, TN, BM, BN, NT>(p, acc, smem, b, oy0, n0, KT, tid, wm0, wn0, none, false);
  }
  DLDS_STAMP(t5);
  DLDS_ADD(4, t4, t5);
  DLDS_FLUSH;
}

// ---------------------------------------------------------------- persistent, wave-specialised form
// One 512-thread workgroup per CU walks a contiguous range of (pixel tile, cout tile) work items; a pixel tile is
// 256 output pixels (NI images x R rows x W columns) x 64 couts.  The (tile, 32-channel chunk) pairs form one
// stream of STAGES that flows through two LDS rings, and the eight waves split into two roles:
//   waves 0-3, CONSUMERS (one per SIMD): all MFMAs.  Each owns 64 pixels x 64 couts (16 accumulator tiles) and per
//     stage runs 9 taps x 16 MFMAs from the stage's halo slot and weight slot (8 ds_read_b128 per 16 MFMAs).
//   waves 4-7, PRODUCERS (one per SIMD): all data movement and all GroupNorm arithmetic.  Per period each issues
//     the LDS-DMA loads (global_load_lds_dwordx4) of the halo chunk two stages ahead and of the weight chunk one
//     stage ahead, then -- PRO -- applies  dropout(SiLU(x * sc + sh))  in place to the halo chunk one stage ahead,
//     reading back exactly the 16-byte slots its own loads filled (so its own counted vmcnt is the only wait).
// A SIMD thus runs a matrix stream and a vector stream side by side; the loads of a stage are in flight for a whole
// period before anyone waits for them, across tile boundaries (the stream does not drain between tiles), and ONE
// raw s_barrier per period hands slots over:  halo ring 3 deep (being consumed / being transformed / landing),
// weight ring 2 deep.  The epilogue runs in the consumers' registers (bias, residual, bf16 rounding, statistics of
// the output for the next GroupNorm) and overlaps the producers' work on the next tile.
constexpr int PS_HG_MAX = 7;     // halo groups per producer wave (25 groups of 16 pixels -> 7, 6, 6, 6)

template <int N> __device__ __forceinline__ void wait_vmcnt_n() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_vmcnt(int n) {     // n wave-uniform, 0..16
  switch (n) {
    case 0: wait_vmcnt_n<0>(); break;   case 1: wait_vmcnt_n<1>(); break;   case 2: wait_vmcnt_n<2>(); break;
    case 3: wait_vmcnt_n<3>(); break;   case 4: wait_vmcnt_n<4>(); break;   case 5: wait_vmcnt_n<5>(); break;
    case 6: wait_vmcnt_n<6>(); break;   case 7: wait_vmcnt_n<7>(); break;   case 8: wait_vmcnt_n<8>(); break;
    case 9: wait_vmcnt_n<9>(); break;   case 10: wait_vmcnt_n<10>(); break; case 11: wait_vmcnt_n<11>(); break;
    case 12: wait_vmcnt_n<12>(); break; case 13: wait_vmcnt_n<13>(); break; case 14: wait_vmcnt_n<14>(); break;
    case 15: wait_vmcnt_n<15>(); break; default: wait_vmcnt_n<16>(); break;
  }
}
__device__ __forceinline__ void ps_barrier() {
  // this wave's LDS operations are done; LDS-DMA loads and the epilogue's global stores stay in flight across the
  // barrier (the builtin barrier would make hipcc drain vmcnt(0) first)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// LDS accesses the compiler must not see: beside LDS-DMA in flight hipcc puts s_waitcnt vmcnt(0) in front of every
// LDS read it knows of (it cannot tell which slot a pending global_load_lds writes), which would drain the producers'
// load-ahead every period.  The producers wait for exactly the loads that fill the slot they touch (wait_vmcnt) and
// for their own LDS operations here / in ps_barrier.
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)p;
}
template <int N>
__device__ __forceinline__ void lds_read16xN(u32x4_t (&v)[N], const unsigned (&addr)[N]) {
  static_assert(N == 4 || N == PS_HG_MAX, "instantiated sizes");
  if constexpr (N == 4)
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]) : "memory");
  else
    asm volatile("ds_read_b128 %0, %7\n\tds_read_b128 %1, %8\n\tds_read_b128 %2, %9\n\tds_read_b128 %3, %10\n\t"
                 "ds_read_b128 %4, %11\n\tds_read_b128 %5, %12\n\tds_read_b128 %6, %13\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6])
                 : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]), "v"(addr[4]), "v"(addr[5]), "v"(addr[6]) : "memory");
}
__device__ __forceinline__ void lds_write16(unsigned addr, const u32x4_t v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");     // retired by ps_barrier's lgkmcnt(0)
}
__device__ __forceinline__ float2 lds_read8(unsigned addr) {
  float2 v;
  asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
  return v;
}

// sum over the 16 lanes of a DPP row (every lane of the row ends with the total)
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
  return v;
}

// per-wave fold of the GroupNorm statistics into cof[c] = (sc, sh) for image b (pro_coefficients without
// workgroup barriers: every producer wave computes all channels and stores the same values)
__device__ __forceinline__ void ps_coefficients(const C3P& p, int b, bool writer, float* cof, float* chs, int lane) {
  const int C = p.Cin, cpg = C >> 5;
  if (p.cof_in) {
    for (int c = lane; c < 2 * C; c += 64) cof[c] = p.cof_in[(size_t)b * 2 * C + c];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return;
  }
  for (int c = lane; c < C; c += 64) {
    const float* st = p.st1;
    int T = p.T1, Cs = p.C1, cl = c;
    if (c >= p.C1) { st = p.st2; T = p.T2; Cs = C - p.C1; cl = c - p.C1; }
    const float2 S = idf_sum_partials(reinterpret_cast<const float2*>(st) + (size_t)b * T * Cs + cl, T, (size_t)Cs);
    chs[2 * c] = S.x; chs[2 * c + 1] = S.y;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // one wave: its own LDS writes are visible to its lanes
  const double inv_n = 1.0 / ((double)p.H * p.W * cpg);
  for (int c = lane; c < C; c += 64) {
    const int g = c / cpg;
    double a = 0.0, d = 0.0;
    for (int k = g * cpg; k < (g + 1) * cpg; ++k) { a += chs[2 * k]; d += chs[2 * k + 1]; }
    float r, mf;
    idf_group_stats(a, d, inv_n, p.eps, &mf, &r);
    float ga = p.gamma ? p.gamma[c] : 1.f, be = p.beta ? p.beta[c] : 0.f;
    float sc = r * ga, sh = be - mf * sc;
    if (p.film_t) { float f = 1.f + p.film_t[(size_t)b * p.ld_t + c]; sc *= f; sh = sh * f + p.film_t[(size_t)b * p.ld_t + C + c]; }
    if (p.film_a) { float f = 1.f + p.film_a[(size_t)b * p.ld_a + c]; sc *= f; sh = sh * f + p.film_a[(size_t)b * p.ld_a + C + c]; }
    cof[2 * c] = sc; cof[2 * c + 1] = sh;
    if (writer && p.sc_out) {
      p.sc_out[(size_t)b * C + c] = sc; p.sh_out[(size_t)b * C + c] = sh;
      if (c == g * cpg) { p.mean_out[b * 32 + g] = mf; p.rstd_out[b * 32 + g] = r; }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// position of a stream in the block's work-item range, decoded incrementally (no divisions in the period loop)
struct PsPos {
  int wi, nt, rt, b0, ck;       // work item, cout tile, row tile inside the image, first image, chunk
  __device__ __forceinline__ void init(const C3P& p, int w) {
    wi = w; ck = 0;
    const int pt = w / p.n_tiles;
    nt = w - pt * p.n_tiles;
    if (p.ps_NI > 1) { b0 = pt * p.ps_NI; rt = 0; }
    else { b0 = pt / p.tiles_per_img; rt = pt - b0 * p.tiles_per_img; }
  }
  __device__ __forceinline__ void next_item(const C3P& p) {
    ++wi;
    if (++nt == p.n_tiles) {
      nt = 0;
      if (p.ps_NI > 1) b0 += p.ps_NI;
      else if (++rt == p.tiles_per_img) { rt = 0; ++b0; }
    }
  }
  __device__ __forceinline__ bool next_chunk(const C3P& p, int nchunks) {     // true when a new work item starts
    if (++ck == nchunks) { ck = 0; next_item(p); return true; }
    return false;
  }
};

template <int KS, bool DUAL, bool PRO, bool RES>
__global__ __launch_bounds__(512) void conv_ps_bf16(const C3P p) {
  constexpr int TAPS = KS * KS, HALO = KS / 2, BN = 64;
  constexpr int WSZ = TAPS * BN * 64;                     // bytes of a weight ring slot
  constexpr int WG = TAPS * BN / 16 / 4;                  // weight groups per producer wave (9 / 1)
  constexpr int SROW = 144;                               // epilogue staging: bytes per pixel row (128 + pad, 16-byte aligned)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int HSZ = p.ps_hbytes;
  unsigned char* const Hring = smem;                      // [3][HSZ]
  unsigned char* const Wring = smem + 3 * HSZ;            // [2][WSZ]
  float* const aux = reinterpret_cast<float*>(smem + p.aux_off);   // PRO: cof [Cin][2] | chs [Cin][2]
  float* const stw = aux + (PRO ? 4 * p.Cin : 0);         // statistics [4 waves][64 couts][2]; then staging [4][16][SROW]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int W = p.W, R = p.R, WH = W + 2 * HALO;
  const int npix_h = p.ps_NI * p.ps_npi, hgroups = (npix_h + 15) >> 4;
  const int nchunks = p.Cin / CK;
  // this block's contiguous range of work items (pixel tile major, cout tile minor).
  // Blocks b and b + 8 share an XCD (observed round-robin placement; speed only): give each XCD one contiguous
  // stretch of the work so neighbouring tiles' halo rows and a pixel tile's cout tiles meet in one L2
  const int G = gridDim.x;
  const int bid = (G & 7) ? (int)blockIdx.x : (int)((blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3));
  const int w0 = (int)(((long)bid * p.ps_work) / G), w1 = (int)(((long)(bid + 1) * p.ps_work) / G);
  const int S = (w1 - w0) * nchunks;                      // stages of this block
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(&g_zero16);

  if (wave < 4) {
    // =============================================================== consumers
    const int fr = lane & 15, fq = lane >> 4;
    int wbase[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int n = a * 16 + fr;
      wbase[a] = n * 64 + swz(n, fq) * 16;
    }
    int hbase[4];                   // halo row of tap (0,0) of this lane's pixel in each 16-pixel slice
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pl = wave * 64 + i * 16 + fr;
      const int img = pl >> p.ps_rwshift, rem = pl & ((1 << p.ps_rwshift) - 1);
      hbase[i] = img * p.ps_npi + (rem >> p.wshift) * WH + (rem & (W - 1));
    }
    // row view of a 16-pixel slice (epilogue): lane -> pixel row (lane >> 2), 32 bytes = couts (lane & 3) * 16 .. + 15
    const int rrow = lane >> 2, rcol = (lane & 3) * 16;
    int rpix[4], rimg[4];           // that pixel's offset inside the tile's first image, and its image
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pl = wave * 64 + i * 16 + rrow;
      const int rem = pl & ((1 << p.ps_rwshift) - 1);
      rimg[i] = pl >> p.ps_rwshift;
      rpix[i] = (rimg[i] * p.H + (rem >> p.wshift)) * W + (rem & (W - 1));
    }
    unsigned char* const stg = reinterpret_cast<unsigned char*>(stw + 512) + wave * (16 * SROW);
    float* const swv = stw + wave * 128;                    // this wave's statistics [64 couts][2]
    const bool wants = p.st_out != nullptr;

    PsPos pos;
    pos.init(p, w0);
    // the accumulators start from the bias: no add in the epilogue
    auto bias4 = [&](int n0, int a) {
      const int n = n0 + a * 16 + fq * 4;                   // Cout % 8 == 0: the 4 couts are inside or outside together
      const float4 v = *reinterpret_cast<const float4*>((p.bias && n < p.Cout) ? p.bias + n : reinterpret_cast<const float*>(&g_zero16));
      return f32x4_t{v.x, v.y, v.z, v.w};
    };
    f32x4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const f32x4_t bv = bias4(pos.nt * BN, a);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[a][i] = bv;
    }

    ps_barrier();                                          // period -2
    ps_barrier();                                          // period -1
    int hs = 0;                                            // halo ring slot of the current stage (s % 3)
    for (int s = 0; s < S; ++s) {
      const unsigned char* Xs = Hring + hs * HSZ;
      const unsigned char* Ws = Wring + (s & 1) * WSZ;
      hs = hs == 2 ? 0 : hs + 1;
#pragma unroll
      for (int tap = 0; tap < TAPS; ++tap) {
        const int toff = (tap / KS) * WH + (tap % KS);
        bf16x8_t wf[4], xf[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) wf[a] = *reinterpret_cast<const bf16x8_t*>(Ws + tap * BN * 64 + wbase[a]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int h = hbase[i] + toff;
          xf[i] = *reinterpret_cast<const bf16x8_t*>(Xs + h * 64 + swz(h, fq) * 16);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            acc[a][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[a], xf[i], acc[a][i], 0, 0, 0);
      }
      if (pos.ck + 1 == nchunks) {
        // ------------------------------------------------ epilogue of work item pos.wi
        // A lane holds 4 couts of a pixel (8 bytes): stored as they stand, 32-byte pieces land in 16 different
        // lines per instruction and the memory side crawls (5.7 us per tile measured).  Each wave therefore passes
        // its 16-pixel x 64-cout slices through a private 16 x 144-byte LDS image and stores whole 128-byte pixel
        // rows, 16 bytes per lane; a residual takes the same road in the other direction first.  Wave-private:
        // no barrier, and plain LDS operations (the consumers have no LDS-DMA in flight).
        const int n0 = pos.nt * BN, oy0 = p.ps_NI > 1 ? 0 : pos.rt * R, b0 = pos.b0;
        const bool colok = n0 + rcol < p.Cout;
        size_t erow[4];
        bool rowok[4];
        u32x4_t rres0[4], rres1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          rowok[i] = (b0 + rimg[i]) < p.B && colok;
          erow[i] = ((size_t)(b0 * p.H + oy0) * W + rpix[i]) * p.Cout + n0 + rcol;
          if constexpr (RES) {                              // residual rows of all four slices first: one round trip
            const bf16_t* rp = rowok[i] ? p.res + erow[i] : zero;
            rres0[i] = *reinterpret_cast<const u32x4_t*>(rp);
            rres1[i] = *reinterpret_cast<const u32x4_t*>(rowok[i] ? rp + 8 : zero);
          }
        }
        f32x4_t ssum[4], ssq[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) ssum[a] = ssq[a] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          uint2 rr[4];
          if constexpr (RES) {                              // residual rows -> MFMA layout
            *reinterpret_cast<u32x4_t*>(stg + rrow * SROW + rcol * 2) = rres0[i];
            *reinterpret_cast<u32x4_t*>(stg + rrow * SROW + rcol * 2 + 16) = rres1[i];
#pragma unroll
            for (int a = 0; a < 4; ++a) rr[a] = *reinterpret_cast<const uint2*>(stg + fr * SROW + (a * 16 + fq * 4) * 2);
          }
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            f32x4_t o = acc[a][i];
            if constexpr (RES) {
              o[0] += __uint_as_float(rr[a].x << 16); o[1] += __uint_as_float(rr[a].x & 0xffff0000u);
              o[2] += __uint_as_float(rr[a].y << 16); o[3] += __uint_as_float(rr[a].y & 0xffff0000u);
            }
            const uint32_t lo = idf_pack_bf16(o[0], o[1]);
            const uint32_t hi = idf_pack_bf16(o[2], o[3]);
            *reinterpret_cast<uint2*>(stg + fr * SROW + (a * 16 + fq * 4) * 2) = make_uint2(lo, hi);
            if (wants) {                                    // wave-uniform: statistics of the rounded values
              const f32x4_t rv = {__uint_as_float(lo << 16), __uint_as_float(lo & 0xffff0000u),
                                  __uint_as_float(hi << 16), __uint_as_float(hi & 0xffff0000u)};
              ssum[a] += rv;
              ssq[a] += rv * rv;
            }
          }
          const u32x4_t v0 = *reinterpret_cast<const u32x4_t*>(stg + rrow * SROW + rcol * 2);
          const u32x4_t v1 = *reinterpret_cast<const u32x4_t*>(stg + rrow * SROW + rcol * 2 + 16);
          // hidden from hipcc's vmcnt bookkeeping (a counted store makes it drain vmcnt(0) before the next tile's
          // first LDS read); out-of-range rows go to a trash line
          const void* dst = rowok[i] ? (const void*)(p.y + erow[i]) : (const void*)&g_trash[(lane & 3) * 4];
          asm volatile("global_store_dwordx4 %0, %1, off\n\tglobal_store_dwordx4 %0, %2, off offset:16\n\ts_nop 1"
                       ::"v"(dst), "v"(v0), "v"(v1) : "memory");
        }
        if (wants) {
          // per-cout sums over this wave's 64 pixels (16 lanes of a DPP row hold the same couts) -> swv[cout][2];
          // a producer wave folds the four waves (fixed order) and stores the tile's partial one period later.
          // (LDS float adds from the 16 lanes instead: 14 us per tile -- same-address LDS atomics serialise badly.
          // Out-of-range pixels carry the bias: the host asks for statistics only when B is a multiple of the
          // images per tile.)
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float sv = row16_sum(ssum[a][r]), qv = row16_sum(ssq[a][r]);
              if (fr == 0) {
                swv[(a * 16 + fq * 4 + r) * 2] = sv;
                swv[(a * 16 + fq * 4 + r) * 2 + 1] = qv;
              }
            }
        }
        PsPos nx = pos;
        nx.next_item(p);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const f32x4_t bv = bias4(nx.nt * BN, a);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[a][i] = bv;
        }
      }
      pos.next_chunk(p, nchunks);
      ps_barrier();
    }
    ps_barrier();                                          // drain period (statistics of the last tile)
    return;
  }

  // ================================================================= producers
  const int pw = wave - 4;                                 // producer wave 0..3: halo groups pw, pw+4, ..; weight groups likewise
  const int prow = lane >> 2;
  const int chs_ = (lane & 3) ^ (((lane >> 4) & 1) << 1);  // logical 8-channel slot this lane's 16 bytes hold (swz inverse)
  const int nhg = (hgroups - pw + 3) >> 2;                 // halo groups of this wave (wave-uniform)
  int woff[WG];                                            // weight element offsets relative to cout tile 0, chunk 0
  bool wnok[WG];
#pragma unroll
  for (int k = 0; k < WG; ++k) {
    const int r = (pw + 4 * k) * 16 + prow, tap = r / BN, n = r - tap * BN;
    woff[k] = (n * TAPS + tap) * p.Cin + chs_ * 8;
    wnok[k] = true;
  }
  int poffL[PS_HG_MAX], poffT[PS_HG_MAX];                  // global pixel index of this lane's halo pixel per group (-1: zero page)
  unsigned amL = 0, amT = 0;                               // groups whose pixel belongs to the tile itself (a_out)
#pragma unroll
  for (int k = 0; k < PS_HG_MAX; ++k) poffL[k] = poffT[k] = -1;
  int bT = -1;                                             // image the coefficients in LDS belong to
  float scv[8], shv[8];
  uint64_t seedv = 0;
  bool drop = false;
  if (PRO) { drop = p.act == 2 && p.seed != nullptr; if (drop) seedv = *p.seed; }

  auto plan = [&](const PsPos& q) {    // poffL / amL for the work item at q
    const int oy0 = p.ps_NI > 1 ? 0 : q.rt * R;
    amL = 0;
#pragma unroll
    for (int k = 0; k < PS_HG_MAX; ++k) {
      poffL[k] = -1;
      const int pix = (pw + 4 * k) * 16 + prow;
      if (k < nhg && pix < npix_h) {
        const int img = (int)(((unsigned)pix * p.ps_magic_img) >> 16), qq = pix - img * p.ps_npi;
        const int hy = (int)(((unsigned)qq * p.wh_magic) >> 16), hx = qq - hy * WH;
        const int iy = oy0 + hy - HALO, ix = hx - HALO, b = q.b0 + img;
        if (b < p.B && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)W) {
          poffL[k] = (b * p.H + iy) * W + ix;
          if (PRO && p.a_out && q.nt == 0 && (unsigned)(hy - HALO) < (unsigned)R && (unsigned)(hx - HALO) < (unsigned)W) amL |= 1u << k;
        }
      }
    }
  };
  auto issue_halo = [&](int slot_i, int c0) {
    const bf16_t* src = p.x;
    int pitch = p.Cin, cc = c0;
    if (DUAL) { if (c0 < p.C1) pitch = p.C1; else { src = p.x2; pitch = p.Cin - p.C1; cc = c0 - p.C1; } }
    unsigned char* slot = Hring + slot_i * HSZ;
#pragma unroll
    for (int k = 0; k < PS_HG_MAX; ++k)
      if (k < nhg) {
        const bf16_t* g = poffL[k] >= 0 ? src + (size_t)((unsigned)poffL[k] * (unsigned)pitch + (unsigned)(cc + chs_ * 8)) : zero;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(slot + (pw + 4 * k) * 1024), 16, 0, 0);
      }
  };
  auto issue_weights = [&](int slot_i, int n0, int c0) {
    unsigned char* slot = Wring + slot_i * WSZ;
#pragma unroll
    for (int k = 0; k < WG; ++k) {
      const int n = ((pw + 4 * k) * 16 + prow) % BN;
      const bf16_t* g = (n0 + n < p.Cout) ? p.w + (size_t)((unsigned)woff[k] + (unsigned)(n0 * TAPS * p.Cin + c0)) : zero;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)(slot + (pw + 4 * k) * 1024), 16, 0, 0);
    }
  };
  auto transform = [&](int slot_i, int c0) {               // PRO: in-place activation of a halo chunk
    const unsigned slot = lds_addr(Hring + slot_i * HSZ) + (unsigned)(pw * 1024 + lane * 16);
    const int cb = c0 + chs_ * 8;
    {
      u32x4_t cv[4];
      const unsigned ca = lds_addr(aux + 2 * cb);
      const unsigned cad[4] = {ca, ca + 16, ca + 32, ca + 48};
      lds_read16xN<4>(cv, cad);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        scv[2 * q] = __uint_as_float(cv[q][0]); shv[2 * q] = __uint_as_float(cv[q][1]);
        scv[2 * q + 1] = __uint_as_float(cv[q][2]); shv[2 * q + 1] = __uint_as_float(cv[q][3]);
      }
    }
    u32x4_t hv[PS_HG_MAX];
    unsigned had[PS_HG_MAX];
#pragma unroll
    for (int k = 0; k < PS_HG_MAX; ++k) had[k] = slot + (unsigned)(k < nhg ? 4 * k * 1024 : 0);   // idle groups re-read group 0
    lds_read16xN<PS_HG_MAX>(hv, had);
#pragma unroll
    for (int k = 0; k < PS_HG_MAX; ++k)
      if (k < nhg && poffT[k] >= 0) {                      // padding pixels stay zero
        const unsigned e0 = (unsigned)poffT[k] * (unsigned)p.Cin + (unsigned)cb;
        const uint4 o = pro_vec(make_uint4(hv[k][0], hv[k][1], hv[k][2], hv[k][3]), scv, shv, p.act, drop, seedv, p.salt,
                                p.thr, p.dscale, e0 >> 3);
        lds_write16(had[k], u32x4_t{o.x, o.y, o.z, o.w});
        if ((amT >> k) & 1u) *reinterpret_cast<uint4*>(p.a_out + e0) = o;
      }
  };
  auto flush_stats = [&](const PsPos& q) {                 // the four consumer waves' sums of the work item at q -> st_out
    if (pw != 0) return;
    const int c = lane, n0 = q.nt * BN;
    const unsigned sb = lds_addr(stw) + (unsigned)c * 8;
    float2 v[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) v[w] = lds_read8(sb + (unsigned)w * 512);
    if (n0 + c >= p.Cout) return;
    if (p.ps_NI == 1) {                                    // one tile of one image: fold the waves in a fixed order
      const float a = ((v[0].x + v[1].x) + v[2].x) + v[3].x, qq = ((v[0].y + v[1].y) + v[2].y) + v[3].y;
      reinterpret_cast<float2*>(p.st_out)[((size_t)q.b0 * p.tiles_per_img + q.rt) * p.Cout + n0 + c] = make_float2(a, qq);
    } else {                                               // four 64-pixel images per tile: wave w = image b0 + w
#pragma unroll
      for (int w = 0; w < 4; ++w)
        if (q.b0 + w < p.B) reinterpret_cast<float2*>(p.st_out)[(size_t)(q.b0 + w) * p.Cout + n0 + c] = v[w];
    }
  };

  // stage s of the stream: work item w0 + s / nchunks, chunk s % nchunks
  // period -2: halo(0).   period -1: weights(0), halo(1); transform halo(0).
  // period t >= 0: weights(t+1), halo(t+2); transform halo(t+1); [statistics of the tile that ended at stage t-1]
  PsPos pL, pT, pW, pC;             // load / transform / weight stream positions; pC follows the consumers
  pL.init(p, w0); pT = pL; pW = pL; pC = pL;
  int sL = 0, sT = 0, sW = 0;
  int hL = 0, hT = 0;               // halo ring slots of the next load / transform stage
  bool stat_pending = false;
  PsPos stat_pos = pL;
  int wkey[2] = {-1, -1};           // what each weight ring slot holds
  for (int t = -2; t < S + 1; ++t) {
    int nissued = 0;
    // the transform stream enters a tile one period after the load stream planned it: take the plan over BEFORE the
    // load stream moves on (with one chunk per tile it plans the next tile in this very period)
    if (t >= -1 && sT < S && pT.ck == 0) {
#pragma unroll
      for (int k = 0; k < PS_HG_MAX; ++k) poffT[k] = poffL[k];
      amT = amL;
    }
    // ---- weights one stage ahead (oldest of this period's loads: the end-of-period wait retires them first).
    // With at most two chunks the ring holds the whole cout tile's weights: they are fetched once per cout tile
    // and stay for every pixel tile the block walks.
    if (t >= -1 && sW < S) {
      const int key = pW.nt * 64 + pW.ck;                  // (cout tile, chunk) the slot would hold
      if (wkey[sW & 1] != key) {
        issue_weights(sW & 1, pW.nt * BN, pW.ck * CK);
        nissued += WG;
        wkey[sW & 1] = key;
      }
      ++sW;
      pW.next_chunk(p, nchunks);
    }
    // ---- halo two stages ahead
    int nh_now = 0;
    if (sL < S) {
      if (pL.ck == 0) plan(pL);
      {
        issue_halo(hL, pL.ck * CK);
        nh_now = nhg;
        nissued += nhg;
      }
      hL = hL == 2 ? 0 : hL + 1;
      ++sL;
      pL.next_chunk(p, nchunks);
    }
    // ---- statistics of the tile whose last stage the consumers finished in the previous period
    if (stat_pending) { if (p.st_out) flush_stats(stat_pos); stat_pending = false; }
    // ---- transform the stage that is consumed next period
    if (t >= -1 && sT < S) {
      if (PRO) {
        if (pT.ck == 0 && pT.b0 != bT) {
          ps_coefficients(p, pT.b0, pw == 0 && pT.rt == 0 && pT.nt == 0, aux, aux + 2 * p.Cin, lane);
          bT = pT.b0;
        }
        wait_vmcnt(nissued);                               // everything older than this period's loads has landed
        transform(hT, pT.ck * CK);
      }
      hT = hT == 2 ? 0 : hT + 1;
      ++sT;
      pT.next_chunk(p, nchunks);
    }
    // the consumers finish stage t in this period: if it closes a tile, its statistics are ready next period
    if (t >= 0 && t < S) {
      if (pC.ck + 1 == nchunks) { stat_pending = true; stat_pos = pC; }
      pC.next_chunk(p, nchunks);
    }
    // ---- the weights of stage t+1 (and, without PRO, its halo chunk issued last period) must have landed
    wait_vmcnt(nh_now);
    ps_barrier();
  }
}


// ---------------------------------------------------------------- few input channels (the network's head conv)
// 3x3 stride-1 conv over an image of Cin <= 3 channels (modules / models.py:258, 300: head = Conv2d(3, ch)): the whole contraction
// is K = 9 * Cin <= 27, ONE MFMA K-step.  The generic implicit-GEMM kernel spent 111 us on it at B = 256 (a 3.6-GFLOP conv that
// writes 134 MB) and left no statistics for the first GroupNorm; here a block stages the (R + 2) x (W + 2) x Cin halo tile and
// the [64][32] zero-padded weight tile in LDS, each lane gathers its 8 k-values per pixel fragment with 2-byte LDS reads, and
// the epilogue is lds_epilogue (bias, full-line stores, statistics partials): HBM-bound on the output.
__global__ __launch_bounds__(512) void conv3x3_fewc_bf16(const C3P p) {
  constexpr int TM = 4, TN = 2, BM = 256, BN = 64, NT = 512, NWM = 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int W = p.W, R = p.R, WH = W + 2, npix_h = (R + 2) * WH, Cin = p.Cin;
  bf16_t* Xs = reinterpret_cast<bf16_t*>(smem);                              // [npix_h][4] (channel 3 unused)
  bf16_t* Ws = reinterpret_cast<bf16_t*>(smem + (size_t)npix_h * 8);         // [BN][32], k = tap * Cin + c, zero beyond 9 * Cin
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile = blockIdx.x / p.n_tiles, n0 = (blockIdx.x % p.n_tiles) * BN;
  const int b = tile / p.tiles_per_img, oy0 = (tile - b * p.tiles_per_img) * R;
  const int wm0 = (wave % NWM) * (TM * 16), wn0 = (wave / NWM) * (BN / 2);
  const int fr = lane & 15, fq = lane >> 4;
  for (int i = tid; i < npix_h * 4; i += NT) {
    const int pix = i >> 2, c = i & 3;
    const int hy = pix / WH, hx = pix - hy * WH, iy = oy0 + hy - 1, ix = hx - 1;
    bf16_t v = 0;
    if (c < Cin && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)W) v = p.x[((size_t)(b * p.H + iy) * W + ix) * Cin + c];
    Xs[i] = v;
  }
  for (int i = tid; i < BN * 32; i += NT) {
    const int n = i >> 5, k = i & 31;
    Ws[i] = (k < 9 * Cin && n0 + n < p.Cout) ? p.w[(size_t)(n0 + n) * 9 * Cin + k] : (bf16_t)0;
  }
  __syncthreads();
  // this lane's 8 k-values: k = fq * 8 + e -> (tap, c); their halo offsets relative to the pixel's window origin
  int koff[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = fq * 8 + e;
    if (k < 9 * Cin) { const int tap = k / Cin, c = k - tap * Cin; koff[e] = ((tap / 3) * WH + (tap % 3)) * 4 + c; }
    else koff[e] = -1;
  }
  f32x4_t acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int i = 0; i < TM; ++i) acc[a][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  bf16x8_t wf[TN];
#pragma unroll
  for (int a = 0; a < TN; ++a) wf[a] = *reinterpret_cast<const bf16x8_t*>(Ws + (wn0 + a * 16 + fr) * 32 + fq * 8);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    int pl = wm0 + i * 16 + fr;
    if (pl >= R * W) pl = 0;
    const int base = ((pl >> p.wshift) * WH + (pl & (W - 1))) * 4;
    uint32_t w4[4];
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      const uint32_t lo = koff[e] >= 0 ? Xs[base + koff[e]] : 0u, hi = koff[e + 1] >= 0 ? Xs[base + koff[e + 1]] : 0u;
      w4[e >> 1] = lo | (hi << 16);
    }
    const u32x4_t xv = {w4[0], w4[1], w4[2], w4[3]};
    const bf16x8_t xf = __builtin_bit_cast(bf16x8_t, xv);
#pragma unroll
    for (int a = 0; a < TN; ++a) acc[a][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[a], xf, acc[a][i], 0, 0, 0);
  }
  __syncthreads();
  uint4 none[(BM * (BN / 8) + NT - 1) / NT];
  lds_epilogue<TM, TN, BM, BN, NT>(p, acc, smem, b, oy0, n0, R * W, tid, wm0, wn0, none, false);
}

// magic multiplier for the division by the halo-row width; 0 when not exact over [0, npix)
inline unsigned wh_magic(int WH, int npix) {
  unsigned m = 65536u / (unsigned)WH + 1u;
  for (int i = 0; i < npix; ++i)
    if ((((unsigned)i * m) >> 16) != (unsigned)(i / WH)) return 0;
  return m;
}

inline size_t aux_bytes(const C3P& p, bool pro, int nwaves, int BN) {
  size_t a = pro ? (size_t)p.Cin * 16 : 0, s = p.st_out ? (size_t)nwaves * BN * 8 : 0;
  return a > s ? a : s;
}

// idf_ensure_lds with one grant record per kernel (a static of the calling launch<...> instantiation).  Should the opt-in
// fail, the launch is skipped and the error left for the caller's IDF_CHECK_LAUNCH.
#define IDF_ENSURE_LDS(kern, bytes)                                                              \
  do {                                                                                           \
    static IdfLdsGrant grant_;                                                                   \
    if (idf_ensure_lds((const void*)(kern), (size_t)(bytes), grant_) != hipSuccess) return;      \
  } while (0)

template <int KS, bool PRO = false, bool DUAL = false>
void launch_dlds(C3P& p, hipStream_t st) {
  const int HALO = KS / 2;
  const int npix_h = (p.R + 2 * HALO) * (p.W + 2 * HALO);
  size_t lds = ((size_t)((npix_h + 15) / 16) * 16 + KS * KS * 64) * 64;
  size_t olds = (size_t)256 * (64 + 4) * sizeof(float);
  if (olds > lds) lds = olds;
  p.aux_off = (int)lds;
  lds += aux_bytes(p, PRO, 8, 64);
  auto kern = conv_dlds_bf16<KS, PRO, DUAL>;
  IDF_ENSURE_LDS(kern, lds);
  hipLaunchKernelGGL(kern, dim3(p.B * p.tiles_per_img * p.n_tiles), dim3(512), lds, st, p);
}

#ifndef IDF_HALO_WARM
#define IDF_HALO_WARM 1        // helper workgroups for under-filled conv3x3_halo_bf16 launches (A/B: -DIDF_HALO_WARM=0)
#endif
template <int MODE, int TM, int BN, int NWM = 2, int KS = 3, bool DUAL = false, bool PRO = false>
void launch(C3P& p, hipStream_t st) {
  size_t lds = ((MODE == 1 ? (size_t)(2 * p.R + 1) * (2 * p.W + 1) : (size_t)(p.R + 2 * (KS / 2)) * (p.W + 2 * (KS / 2))) +
                KS * KS * BN) * 64;
  size_t olds = (size_t)NWM * TM * 16 * (BN + 4) * sizeof(float);      // epilogue tile
  if (olds > lds) lds = olds;
  p.aux_off = (int)lds;
  lds += aux_bytes(p, PRO, NWM * 2, BN);
  auto kern = conv3x3_halo_bf16<MODE, TM, BN, NWM, KS, DUAL, PRO>;
  IDF_ENSURE_LDS(kern, lds);
  p.main_blocks = p.B * p.tiles_per_img * p.n_tiles;
  if (!(MODE == 0 && KS == 3 && BN == 64)) p.aux_blocks = 0;           // (callers attach an auxiliary job only to these)
  const int real = p.main_blocks + p.aux_blocks;
  p.hw_main = (real <= IDF_WARM_MAX_MAIN && IDF_HALO_WARM) ? real : 0;
  hipLaunchKernelGGL(kern, dim3(real + (p.hw_main ? IDF_WARM_HELPERS : 0)), dim3(NWM * 128), lds, st, p);
}

template <int TM, int NWM, int KS, int BN = 64>
void launch_gnb(C3P& p, hipStream_t st) {
  size_t lds = ((size_t)(p.R + 2 * (KS / 2)) * (p.W + 2 * (KS / 2)) + KS * KS * BN) * 64;
  size_t olds = (size_t)NWM * TM * 16 * (BN + 4) * sizeof(float);      // epilogue tile
  if (olds > lds) lds = olds;
  p.aux_off = (int)lds;
  lds += (size_t)NWM * 2 * BN * 8 + BN * 8 + 32 * 8;                   // wave partials | per-channel products | per-group k1, k0
  auto kern = conv3x3_halo_bf16<0, TM, BN, NWM, KS, false, false, true>;
  IDF_ENSURE_LDS(kern, lds);
  const int real = p.B * p.n_tiles;
  p.aux_blocks = 0;
  p.hw_main = (real <= IDF_WARM_MAX_MAIN && IDF_HALO_WARM) ? real : 0;
  hipLaunchKernelGGL(kern, dim3(real + (p.hw_main ? IDF_WARM_HELPERS : 0)), dim3(NWM * 128), lds, st, p);
}

template <int TM, int NWM, int KS, int BWD>
void launch_bwd_chain(C3P& p, hipStream_t st) {
  constexpr int BN = 64;
  size_t lds = ((size_t)(p.R + 2 * (KS / 2)) * (p.W + 2 * (KS / 2)) + KS * KS * BN) * 64;
  size_t olds = (size_t)NWM * TM * 16 * (BN + 4) * sizeof(float);      // epilogue tile
  if (olds > lds) lds = olds;
  p.aux_off = (int)lds;
  const size_t a = (BWD & 1) ? (size_t)p.Cin * 24 : 0, s = p.st_out ? (size_t)NWM * 2 * BN * 8 : 0;    // (A, K1, K0, -) + scratch | wave partials
  lds += a > s ? a : s;
  auto kern = conv3x3_halo_bf16<0, TM, BN, NWM, KS, false, false, false, BWD>;
  IDF_ENSURE_LDS(kern, lds);
  p.main_blocks = p.B * p.tiles_per_img * p.n_tiles;
  if (!(KS == 3 && !(BWD & 1))) p.aux_blocks = 0;
  const int real = p.main_blocks + p.aux_blocks;
  p.hw_main = (real <= IDF_WARM_MAX_MAIN && IDF_HALO_WARM) ? real : 0;
  hipLaunchKernelGGL(kern, dim3(real + (p.hw_main ? IDF_WARM_HELPERS : 0)), dim3(NWM * 128), lds, st, p);
}

void clear_pro(C3P& p) {
  p.st_out = nullptr; p.aux_off = 0;
  p.st1 = p.st2 = nullptr; p.T1 = p.T2 = 0;
  p.gamma = p.beta = p.film_t = p.film_a = nullptr; p.ld_t = p.ld_a = 0; p.eps = 0.f;
  p.act = 0; p.seed = nullptr; p.salt = 0; p.thr = 0; p.dscale = 1.f;
  p.a_out = nullptr; p.mean_out = p.rstd_out = p.sc_out = p.sh_out = nullptr; p.cof_in = nullptr;
  p.gnb_x = p.gnb_res2 = nullptr; p.gnb_sc = p.gnb_sh = p.gnb_mean = p.gnb_rstd = nullptr;
  p.gnb_dfilm_t = p.gnb_dfilm_a = p.gnb_dgb = p.gnb_dgam = p.gnb_dbet = nullptr;
  p.due_x = p.due_x2 = nullptr; p.due_C1 = 0; p.due_sc = p.due_sh = nullptr;
  p.dyp_x = nullptr; p.dyp_out = nullptr; memset(&p.dyp_f, 0, sizeof(p.dyp_f));
  p.main_blocks = p.aux_blocks = 0; p.aux_x = p.aux_x2 = p.aux_w = nullptr; p.aux_C1 = p.aux_Cin = 0; p.aux_bias = nullptr;
  p.aux_y = nullptr; p.aux_Cout = p.aux_n_tiles = 0;
  p.ps_NI = 0; p.ps_rwshift = 0; p.ps_npi = 0; p.ps_magic_img = 0; p.ps_nptiles = p.ps_work = 0; p.ps_hbytes = 0;
}

// ---- persistent form: geometry and the decision to use it
#define g_ps (idf_knobs().conv_ps)
const int g_ps_min = 256;   // work items (one per CU) below which the small-tile kernels spread better
// The GroupNorm prologue in this form runs on the four producer waves alone -- one wave per SIMD issues a vector
// instruction every 4 cycles at best, beside a consumer wave whose MFMAs hold half the issue slots -- and is
// 2.5-3.6 us per stage against 1 us of MFMAs: measured equal to or slower than the two-blocks-per-CU kernels
// (64->64 @64^2: 38.7 vs 30.4 us at B = 32, 190 vs 196 us at B = 256).  Off unless asked for.

struct PsPlan { int R, NI, rwshift, npi, hgroups, nptiles, work, T; unsigned magic_img, magic_row; size_t lds; int aux_off; };

// stride-1 3x3 / 1x1 (mode 0) only.  pro: GroupNorm prologue; want_st: statistics epilogue
bool ps_plan(int B, int H, int W, int Cin, int Cout, int KS, bool pro, bool want_st, PsPlan* o) {
  if (!g_ps || (Cout & 7) || (Cin % CK) || W < 8 || W > 64 || (W & (W - 1)) || (H & (H - 1)) || H < 1) return false;
  if ((KS == 3 && !(g_ps & 1)) || (KS == 1 && !(g_ps & 2)) || pro) return false;      // the GroupNorm prologue in this form lost (below): never taken
  const int HW = H * W, halo = KS / 2;
  int R, NI;
  if (HW >= 256) { R = 256 / W; NI = 1; if (R < 1 || H % R) return false; }
  else { if (256 % HW) return false; NI = 256 / HW; R = H; }
  if (NI > 1 && (pro || (want_st && !(NI == 4 && HW == 64)))) return false;
  int rw = 0;
  while ((1 << rw) < R * W) ++rw;
  const int WH = W + 2 * halo, npi = (R + 2 * halo) * WH, npix = NI * npi, hg = (npix + 15) / 16;
  if (hg > 4 * PS_HG_MAX) return false;
  const size_t aux = (pro ? (size_t)Cin * 16 : 0) + 2048 + 4 * 16 * 144;      // coefficients, statistics, epilogue staging
  if (want_st && Cin == CK) return false;                  // one chunk per tile: the single statistics buffer would be rewritten too early
  const size_t main_ = (size_t)3 * hg * 1024 + (size_t)2 * KS * KS * 64 * 64;
  if (main_ + aux > 160 * 1024) return false;
  const int n_tiles = idf_cdiv(Cout, 64);
  const int nptiles = NI > 1 ? idf_cdiv(B, NI) : B * (HW / 256);
  if ((long)nptiles * n_tiles < g_ps_min) return false;
  const unsigned mi = NI > 1 ? wh_magic(npi, npix) : 1u, mr = wh_magic(WH, npi);     // one image: (pix * 1) >> 16 == 0
  if (!mi || !mr) return false;
  o->R = R; o->NI = NI; o->rwshift = rw; o->npi = npi; o->hgroups = hg; o->nptiles = nptiles; o->work = nptiles * n_tiles;
  o->T = NI > 1 ? 1 : HW / 256; o->magic_img = mi; o->magic_row = mr; o->lds = main_ + aux; o->aux_off = (int)main_;
  return true;
}

template <int KS, bool DUAL, bool PRO, bool RES>
void launch_ps_r(C3P& p, const PsPlan& pl, int G, hipStream_t st) {
  auto kern = conv_ps_bf16<KS, DUAL, PRO, RES>;
  IDF_ENSURE_LDS(kern, pl.lds);
  hipLaunchKernelGGL(kern, dim3(G), dim3(512), pl.lds, st, p);
}

template <int KS, bool DUAL, bool PRO>
void launch_ps(C3P& p, const PsPlan& pl, hipStream_t st) {
  p.R = pl.R; p.tiles_per_img = pl.T; p.n_tiles = idf_cdiv(p.Cout, 64); p.wh_magic = pl.magic_row;
  p.ps_NI = pl.NI; p.ps_rwshift = pl.rwshift; p.ps_npi = pl.npi; p.ps_magic_img = pl.magic_img;
  p.ps_nptiles = pl.nptiles; p.ps_work = pl.work; p.ps_hbytes = pl.hgroups * 1024; p.aux_off = pl.aux_off;
  // CU count of the current device, read once (a C++11 magic static: initialised exactly once under concurrent first calls;
  // one process drives one GPU here, so "the current device at first use" is the device)
  static const int ncu = [] {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    return prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }();
  const int per = idf_cdiv(pl.work, ncu);                  // work items per block, then an even spread
  const int G = idf_cdiv(pl.work, per);
  if (p.res) launch_ps_r<KS, DUAL, PRO, true>(p, pl, G, st);
  else launch_ps_r<KS, DUAL, PRO, false>(p, pl, G, st);
}

// pixel tile of a 3x3 launch: BM pixels = R rows x W columns; false when the halo tile does not fit
bool plan3(int B, int H, int W, int Cout, int mode, int* BM_, int* R_) {
  // 128-pixel tiles when the problem is big enough to still fill the chip, else 64
  long M = (long)B * H * W;
  int BM = ((M / 128) * idf_cdiv(Cout, 64) >= 256 && H * W >= 128) ? 128 : 64;
  // 512-thread blocks (256 pixels share one weight slab) once the grid still covers the chip
  if ((M / 256) * idf_cdiv(Cout, 64) >= 256 && H * W >= 256 && Cout > 32) BM = 256;
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
  *BM_ = BM; *R_ = R;
  return hrows * hcols * 4 <= (BM == 256 ? HALO_VEC_MAX_512 : (mode == 1 ? HALO_VEC_MAX_S2 : HALO_VEC_MAX_256));
}

bool plan1(int B, int H, int W, int Cout, int* BM_, int* R_) {
  const long M = (long)B * H * W;
  const int nt = idf_cdiv(Cout, 64);
  int BM = 64;
  if ((M / 128) * nt >= 256 && H * W >= 128) BM = 128;
  if ((M / 256) * nt >= 256 && H * W >= 256) BM = 256;
  int R = BM / W;
  if (R < 1) R = 1;
  if (R > H) R = H;
  while (H % R) --R;
  *BM_ = BM; *R_ = R;
  return R * W * 4 <= (BM == 256 ? HALO_VEC_MAX_512 : HALO_VEC_MAX_256);
}

bool shape3_ok(int H, int W, int Cin, int Cout, int mode) {
  return !(mode < 0 || mode > 3 || (Cin % CK) || W < 4 || (W & (W - 1)) || W > 128 || (mode >= 2 && ((H | W) & 1)) ||
           (mode == 1 && (W > 32 || Cout <= 32)));
}
bool shape1_ok(int W, int Cin, int Cout) {
  return !((Cin % CK) || W < 4 || (W & (W - 1)) || W > 128 || (Cout & 7));
}

const int g_dlds = 3;
#define g_dlds_min (idf_knobs().conv_dlds_min)
const long g_dlds_min_pro = 512;

// dispatch of a 3x3 launch whose C3P is filled in (PRO / DUAL only for mode 0)
template <bool DUAL, bool PRO>
void dispatch3(C3P& p, int mode, int BM, hipStream_t st) {
  PsPlan pl;
  if constexpr (!PRO) {       // (the persistent form with the GroupNorm prologue lost to the two-blocks-per-CU kernels: not instantiated)
    if (mode == 0 && ps_plan(p.B, p.H, p.W, p.Cin, p.Cout, 3, false, p.st_out != nullptr, &pl)) { launch_ps<3, DUAL, false>(p, pl, st); return; }
  }
  const bool bn32 = p.Cout <= 32;
  p.n_tiles = idf_cdiv(p.Cout, bn32 ? 32 : 64);
#define IDF_C3_LAUNCH(MODE)                                              \
  do {                                                                   \
    if (bn32) { if (BM == 128) launch<MODE, 4, 32, 2, 3, DUAL, PRO>(p, st); else launch<MODE, 2, 32, 2, 3, DUAL, PRO>(p, st); } \
    else { if (BM == 256) launch<MODE, 4, 64, 4, 3, DUAL, PRO>(p, st); else if (BM == 128) launch<MODE, 4, 64, 2, 3, DUAL, PRO>(p, st); else launch<MODE, 2, 64, 2, 3, DUAL, PRO>(p, st); } \
  } while (0)
  // direct-to-LDS variant: pays once two of its blocks share every CU (it does not prefetch within a block)
  const long blocks = (long)p.B * p.tiles_per_img * p.n_tiles;
  if (mode == 0 && BM == 256 && !bn32 && (g_dlds & 1) && (p.Cout & 7) == 0 && blocks >= (PRO ? g_dlds_min_pro : g_dlds_min) &&
      ((p.R + 2) * (p.W + 2) + 15) / 16 + 36 <= 72) launch_dlds<3, PRO, DUAL>(p, st);
  else if (mode == 0) IDF_C3_LAUNCH(0);
  else if constexpr (!DUAL && !PRO) {
    if (mode == 1) launch<1, 2, 64>(p, st);
    else if (mode == 2) IDF_C3_LAUNCH(2);
    else IDF_C3_LAUNCH(3);
  }
#undef IDF_C3_LAUNCH
}

template <bool DUAL, bool PRO>
void dispatch1(C3P& p, int BM, hipStream_t st) {
  PsPlan pl;
  if constexpr (!PRO) {
    if (ps_plan(p.B, p.H, p.W, p.Cin, p.Cout, 1, false, p.st_out != nullptr, &pl)) { launch_ps<1, DUAL, false>(p, pl, st); return; }
  }
  p.n_tiles = idf_cdiv(p.Cout, 64);
  const long blocks = (long)p.B * p.tiles_per_img * p.n_tiles;
  if (BM == 256 && (g_dlds & 2) && blocks >= (PRO ? g_dlds_min_pro : g_dlds_min)) launch_dlds<1, PRO, DUAL>(p, st);
  else if (BM == 256) launch<0, 4, 64, 4, 1, DUAL, PRO>(p, st);
  else if (BM == 128) launch<0, 4, 64, 2, 1, DUAL, PRO>(p, st);
  else launch<0, 2, 64, 2, 1, DUAL, PRO>(p, st);
}

int fill_common(C3P& p, int B, int H, int W, int Cin, int Cout, int mode, int KS, int* BM) {
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
  p.Hs = mode == 1 ? 2 * H : (mode >= 2 ? H / 2 : H); p.Ws = mode == 1 ? 2 * W : (mode >= 2 ? W / 2 : W);
  int ws = 0;
  while ((1 << ws) < W) ++ws;
  p.wshift = ws;
  int R;
  if (KS == 3) {
    if (!plan3(B, H, W, Cout, mode, BM, &R)) return 1;
    const int hrows = mode == 1 ? 2 * R + 1 : R + 2, hcols = mode == 1 ? 2 * W + 1 : W + 2;
    p.wh_magic = wh_magic(hcols, hrows * hcols);
  } else {
    if (!plan1(B, H, W, Cout, BM, &R)) return 1;
    p.wh_magic = wh_magic(W, R * W);
  }
  p.R = R; p.tiles_per_img = H / R;
  if (!p.wh_magic || (long)B * p.Hs * p.Ws * Cin >= (1L << 31) || (long)Cout * KS * KS * Cin >= (1L << 31) ||
      (long)B * H * W * Cout >= (1L << 31))
    return 2;
  return 0;
}

}  // namespace

// mode: 0 stride 1, 1 stride 2 (H, W <= 32 out), 2 nearest-x2-upsampled input, 3 zero-stuffed x2 input
// (transposed stride 2).
// H, W = OUTPUT dims.  Returns IDF_ERR_UNSUPPORTED for shapes it does not cover (the
// caller falls back to idf_conv2d_fwd).  st_out: optional statistics partials of y (idf_conv_tiles).
extern "C" int idf_conv3x3_bf16(const void* x, const void* w, const float* bias, const void* res, void* y, int B,
                                int H, int W, int Cin, int Cout, int mode, float* st_out, void* stream) {
  if (!shape3_ok(H, W, Cin, Cout, mode))
    IDF_FAIL(IDF_ERR_UNSUPPORTED, "conv3x3_bf16: B%d H%d W%d Cin%d Cout%d mode%d not covered", B, H, W, Cin, Cout, mode);
  if (st_out && (Cout & 7)) IDF_FAIL(IDF_ERR_UNSUPPORTED, "conv3x3_bf16: statistics need Cout %% 8 == 0");
  if (B == 0) return IDF_OK;
  C3P p;
  clear_pro(p);
  p.x = (const bf16_t*)x; p.w = (const bf16_t*)w; p.bias = bias; p.res = (const bf16_t*)res; p.y = (bf16_t*)y;
  p.x2 = nullptr; p.C1 = Cin; p.st_out = st_out;
  int BM;
  if (int e = fill_common(p, B, H, W, Cin, Cout, mode, 3, &BM))
    IDF_FAIL(IDF_ERR_UNSUPPORTED, e == 1 ? "conv3x3_bf16: halo too large (H%d W%d)" : "conv3x3_bf16: tensor too large for 32-bit offsets", H, W);
  dispatch3<false, false>(p, mode, BM, (hipStream_t)stream);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}

// Pixel tiles per image of idf_conv3x3_fewc_bf16 (= T of its statistics partials); 0: not covered (Cin <= 3, Cout % 8 == 0,
// W a power of two in [8, 64], H * W >= 256, 256-pixel tiles of whole rows).
extern "C" int idf_conv_fewc_tiles(int B, int H, int W, int Cin, int Cout) {
  if (B <= 0 || Cin < 1 || Cin > 3 || (Cout & 7) || W < 8 || W > 64 || (W & (W - 1)) || H * W < 256) return 0;
  const int R = 256 / W;
  if (H % R || (long)B * H * W * Cout >= (1L << 31)) return 0;
  return H / R;
}

// y = conv3x3(x, w) + bias, stride 1, for an input of Cin <= 3 channels (the head conv): x [B,H,W,Cin] bf16, w the forward
// shadow [Cout][9][Cin]; st_out (optional): statistics partials of y [B][idf_conv_fewc_tiles()][Cout][2].
extern "C" int idf_conv3x3_fewc_bf16(const void* x, const void* w, const float* bias, void* y, int B, int H, int W, int Cin,
                                     int Cout, float* st_out, void* stream) {
  const int T = idf_conv_fewc_tiles(B, H, W, Cin, Cout);
  if (!T) IDF_FAIL(IDF_ERR_UNSUPPORTED, "conv3x3_fewc_bf16: B%d H%d W%d Cin%d Cout%d not covered", B, H, W, Cin, Cout);
  if (!x || !w || !y) IDF_FAIL(IDF_ERR_BADARG, "conv3x3_fewc_bf16: null argument");
  C3P p;
  clear_pro(p);
  p.x = (const bf16_t*)x; p.x2 = nullptr; p.C1 = Cin; p.w = (const bf16_t*)w; p.bias = bias; p.res = nullptr; p.y = (bf16_t*)y;
  p.B = B; p.H = H; p.W = W; p.Hs = H; p.Ws = W; p.Cin = Cin; p.Cout = Cout; p.st_out = st_out;
  int ws = 0;
  while ((1 << ws) < W) ++ws;
  p.wshift = ws;
  p.R = 256 / W; p.tiles_per_img = T; p.n_tiles = idf_cdiv(Cout, 64); p.wh_magic = 1;
  size_t lds = (size_t)256 * (64 + 4) * sizeof(float);            // the epilogue's fp32 tile (>= halo + weight tiles)
  p.aux_off = (int)lds;
  lds += (size_t)8 * 64 * 8;
  static IdfLdsGrant grant;
  if (idf_ensure_lds((const void*)conv3x3_fewc_bf16, lds, grant) != hipSuccess) IDF_FAIL(IDF_ERR_HIP, "conv3x3_fewc_bf16: LDS request refused");
  hipLaunchKernelGGL(conv3x3_fewc_bf16, dim3(B * T * p.n_tiles), dim3(512), lds, (hipStream_t)stream, p);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}

// 1x1 convolution (stride 1) forward / data gradient through the same pipeline (KS = 1): w [Cout][Cin].
// IDF_ERR_UNSUPPORTED for shapes it does not cover (the caller then uses idf_bgemm).
extern "C" int idf_conv1x1_bf16(const void* x, const void* x2, int C1, const void* w, const float* bias,
                                const void* res, void* y, int B, int H, int W, int Cin, int Cout, float* st_out, void* stream) {
  if (!x2) C1 = Cin;
  if (!shape1_ok(W, Cin, Cout) || (x2 && (C1 <= 0 || C1 >= Cin || (C1 % CK))))
    IDF_FAIL(IDF_ERR_UNSUPPORTED, "conv1x1_bf16: B%d H%d W%d Cin%d (C1 %d) Cout%d not covered", B, H, W, Cin, C1, Cout);
  if (B == 0) return IDF_OK;
  C3P p;
  clear_pro(p);
  p.x = (const bf16_t*)x; p.w = (const bf16_t*)w; p.bias = bias; p.res = (const bf16_t*)res; p.y = (bf16_t*)y;
  p.x2 = (const bf16_t*)x2; p.C1 = C1; p.st_out = st_out;
  int BM;
  if (int e = fill_common(p, B, H, W, Cin, Cout, 0, 1, &BM))
    IDF_FAIL(IDF_ERR_UNSUPPORTED, e == 1 ? "conv1x1_bf16: tile too large (H%d W%d)" : "conv1x1_bf16: tensor too large for 32-bit offsets", H, W);
  hipStream_t st = (hipStream_t)stream;
  if (x2) dispatch1<true, false>(p, BM, st);
  else dispatch1<false, false>(p, BM, st);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}

// ---- data-gradient conv with the GroupNorm backward as its epilogue (the 16x16 / 8x8 / 4x4 levels: a tile = one image)
namespace {
const int g_dgn_maxhw = 256;
bool dgrad_gn_plan(int H, int W, int Cin, int Cout, int taps, int* BM) {       // coverage (not the policy)
  if ((taps != 9 && taps != 1) || H * W > 256 || (Cout % 64) || (Cin % CK) || W < 4 || (W & (W - 1))) return false;
  *BM = H * W <= 64 ? 64 : (H * W <= 128 ? 128 : 256);
  const int halo = taps == 9 ? 2 : 0;
  return (H + halo) * (W + halo) * 4 <= (*BM == 256 ? HALO_VEC_MAX_512 : HALO_VEC_MAX_256);
}
}  // namespace

// 1: covered AND expected to beat the data-gradient conv + idf_gn_fused_bwd pair; 2: covered only.  Measured in the B = 32
// train step: 3x3 at 8x8 13.6 vs 9.4 + 6.6 us, at 16x16 18.9 vs 10.4 + 8.9 us (a 256-pixel tile = 64 workgroups instead of
// 256); 1x1 (the AttnBlock's q/k/v conv, 384 -> 128 channels) loses -- 16x16: 20.9 vs 7.0 + 8.2 us, 8x8: 14.4 vs 5.7 + 6.6 us.
extern "C" int idf_conv_dgrad_gn_ok(int B, int H, int W, int Cin, int Cout, int taps) {
  int BM;
  (void)B;
  if (!dgrad_gn_plan(H, W, Cin, Cout, taps, &BM)) return 0;
  // the attention block's q/k/v data gradient (1x1, 3C -> C) with its GroupNorm backward as the epilogue: 12 launches fewer per
  // CelebA step, 9.376 -> 9.347 ms (same-box A/B, profiles/r04_conv_wr.txt); IDF_DGRAD_GN_1X1=0: du epilogue + apply pass
  static const int one = 1;
  return ((taps == 9 || one) && H * W <= g_dgn_maxhw) ? 1 : 2;
}

// dx = GroupNormBackward( conv(dy, w) ), stride 1, taps 9 or 1: dy [B,H,W,Cin] bf16 is the gradient of the forward conv's
// output, w its data-gradient weights [Cout][taps][Cin] (flipped taps, idf_pack_conv_weight), the conv result is dA, the
// gradient w.r.t. a = act(GroupNorm(x)) -- which the epilogue turns into dx [B,H,W,Cout] exactly as idf_gn_fused_bwd would
// (same arguments, same side outputs), without dA ever being written.
extern "C" int idf_conv_dgrad_gn_bf16(const void* dy, const void* w, const void* x, const void* dres, const void* dres2,
                                      void* dx, const float* gamma, const float* beta, const float* film_t,
                                      const float* film_a, int ld_t, int ld_a, const float* mean, const float* rstd,
                                      const float* sc, const float* sh, float* dfilm_t, float* dfilm_a, float* dgb,
                                      float* dgamma_acc, float* dbeta_acc, const uint64_t* seed, uint32_t salt,
                                      float p_drop, int act, int B, int H, int W, int Cin, int Cout, int taps,
                                      void* stream) {
  int BM;
  if (!dgrad_gn_plan(H, W, Cin, Cout, taps, &BM) || (act != 1 && act != 2))
    IDF_FAIL(IDF_ERR_UNSUPPORTED, "conv_dgrad_gn_bf16: B%d H%d W%d Cin%d Cout%d taps%d act%d not covered", B, H, W, Cin, Cout,
             taps, act);
  if (!dy || !w || !x || !dx || !mean || !rstd || !sc || !sh) IDF_FAIL(IDF_ERR_BADARG, "conv_dgrad_gn_bf16: null argument");
  if (B == 0) return IDF_OK;
  C3P p;
  clear_pro(p);
  p.x = (const bf16_t*)dy; p.x2 = nullptr; p.C1 = Cin; p.w = (const bf16_t*)w; p.bias = nullptr;
  p.res = (const bf16_t*)dres; p.y = (bf16_t*)dx;
  p.B = B; p.H = H; p.W = W; p.Hs = H; p.Ws = W; p.Cin = Cin; p.Cout = Cout;
  int ws = 0;
  while ((1 << ws) < W) ++ws;
  p.wshift = ws;
  p.R = H; p.tiles_per_img = 1; p.n_tiles = Cout / 64;
  const int halo = taps == 9 ? 2 : 0;
  p.wh_magic = wh_magic(W + halo, (H + halo) * (W + halo));
  if (!p.wh_magic || (long)B * H * W * (Cin > Cout ? Cin : Cout) >= (1L << 31))
    IDF_FAIL(IDF_ERR_UNSUPPORTED, "conv_dgrad_gn_bf16: tensor too large for 32-bit offsets");
  p.gnb_x = (const bf16_t*)x; p.gnb_res2 = (const bf16_t*)dres2;
  p.gnb_sc = sc; p.gnb_sh = sh; p.gnb_mean = mean; p.gnb_rstd = rstd;
  p.gamma = gamma; p.beta = beta; p.film_t = film_t; p.film_a = film_a;
  p.ld_t = ld_t ? ld_t : 2 * Cout; p.ld_a = ld_a ? ld_a : 2 * Cout;
  p.gnb_dfilm_t = dfilm_t; p.gnb_dfilm_a = dfilm_a; p.gnb_dgb = dgb; p.gnb_dgam = dgamma_acc; p.gnb_dbet = dbeta_acc;
  p.act = act; p.salt = salt;
  p.thr = idf_drop_thresh(p_drop);
  p.dscale = 1.0f / (1.0f - (float)p.thr / 65536.0f);
  p.seed = (act == 2 && p_drop > 0.f) ? seed : nullptr;
  hipStream_t st = (hipStream_t)stream;
  // whole-image tiles of 64 couts are B * Cout / 64 workgroups -- 64 or 128 of them at the benchmark's batch on 256 CUs: 32-cout
  // tiles (complete GroupNorm groups still: 32 % (Cout / 32) == 0) double the grid while it stays under one workgroup per CU
  // (IDF_GNB_BN32=0: off; profiles/r04_conv_wr.txt)
  static const int bn32 = 1;
  const int cpg = Cout >> 5;
  const bool half = bn32 && BM == 256 && (long)B * (Cout / 64) < 256 && cpg >= 1 && cpg <= 32 && 32 % cpg == 0;
  if (half) p.n_tiles = Cout / 32;
  if (taps == 9) {
    if (half) launch_gnb<4, 4, 3, 32>(p, st);
    else if (BM == 256) launch_gnb<4, 4, 3>(p, st); else if (BM == 128) launch_gnb<4, 2, 3>(p, st); else launch_gnb<2, 2, 3>(p, st);
  } else {
    if (half) launch_gnb<4, 4, 1, 32>(p, st);
    else if (BM == 256) launch_gnb<4, 4, 1>(p, st); else if (BM == 128) launch_gnb<4, 2, 1>(p, st); else launch_gnb<2, 2, 1>(p, st);
  }
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}

// Pixel tiles per image of the launch the entry points above / below make for this shape = the T of the
// statistics partials st_out [B][T][Cout][2] they write; -1 when the shape is not covered.  pro != 0: the launch is
// idf_conv_gn_bf16 (the choice of kernel, hence T, can depend on it).
extern "C" int idf_conv_tiles(int B, int H, int W, int Cin, int Cout, int mode, int taps, int pro) {
  int BM, R;
  PsPlan pl;
  if (taps == 9) {
    if (!shape3_ok(H, W, Cin, Cout, mode)) return -1;
    if (mode == 0 && ps_plan(B, H, W, Cin, Cout, 3, pro != 0, true, &pl)) return pl.T;
    if (!plan3(B, H, W, Cout, mode, &BM, &R)) return -1;
  } else {
    if (mode != 0 || !shape1_ok(W, Cin, Cout)) return -1;
    if (ps_plan(B, H, W, Cin, Cout, 1, pro != 0, true, &pl)) return pl.T;
    if (!plan1(B, H, W, Cout, &BM, &R)) return -1;
  }
  return H / R;
}

// 1 when the one-launch GroupNorm-prologue conv is expected to beat the GroupNorm kernel + plain conv pair for this
// shape, 0 when the pair wins (measured on MI355X, tools/bench_gnconv.py: the prologue's vector work is done once per
// staged element -- halo rows and every 64-cout tile repeat it -- so it loses where that repetition is large and the
// stand-alone GroupNorm pass runs near its bandwidth):
//   (small maps at small batches -- 16^2 / 8^2 at B = 32 -- measure the same either way in the train step: 10.98 vs
//   10.99 ms over repeated runs on one box; the one-launch form is kept there: 72 fewer launches.  IDF_GN_FUSE_MINPIX=512
//   restores the pair for them)
//   two or more cout tiles on >= 1536 pixel-tile blocks (128->128 @32^2 at B = 256: 130 vs 112 us),
//   ragged cout counts (epsilon / latent heads: 32-cout tiles of 128 pixels) on >= 2^19 pixels (263 vs 167 us).
extern "C" int idf_conv_gn_advice(int B, int H, int W, int Cin, int Cout, int taps) {
  const long M = (long)B * H * W;
  if (taps == 9) {
    if ((Cout & 7) && M >= (1L << 19)) return 0;
    if (idf_cdiv(Cout, 64) >= 2 && (M / 256) * idf_cdiv(Cout, 64) >= 1536) return 0;
  }
  return 1;
}

// y = conv(act(GroupNorm/FiLM(x))) + bias (+ res) in ONE launch (modules.py:264-288, 309-320, 145-150): stride-1 3x3
// (taps 9) or 1x1 (taps 1) over x [B,H,W,Cin] -- or over the never-materialised concatenation x | x2 (models.py:321).
// The GroupNorm(32) statistics are not computed here: the launches that produced x (and x2) left per-channel partial
// sums st1 [B][T1][C1][2] (st2 [B][T2][Cin-C1][2]) behind (st_out of the entry points of this file, or idf_gn_partials);
// every block folds them with gamma / beta and the FiLM pairs (layout as idf_gn_coef_fwd) into the per-(image, channel)
// affine and applies  act 1: u = x*sc+sh;  act 2: SiLU(u), then dropout(p_drop) when seed != NULL  while staging its tile.
// Optional outputs (training; NULL otherwise): a_out = the activated tensor [B,H,W,Cin] (kept for the weight gradient),
// mean / rstd [B,32] and sc / sh [B,Cin] (for idf_gn_fused_bwd / idf_gn_coef_bwd); st_out as above.
// coef_ws (optional): B * Cin * 2 floats of scratch -- launches that cut an image into many tiles fold the coefficients
// once per image into it with a small launch of their own instead of once per block.
static int conv_gn_impl(const void* x, const void* x2, int C1, const float* st1, int T1, const float* st2, int T2,
                        const float* gamma, const float* beta, const float* film_t, const float* film_a, int ld_t,
                        int ld_a, float eps, int act, const uint64_t* seed, uint32_t salt, float p_drop,
                        const void* w, const float* bias, const void* res, void* y, void* a_out, float* mean,
                        float* rstd, float* sc, float* sh, float* st_out, float* coef_ws, int B, int H, int W, int Cin,
                        int Cout, int taps, void* stream, const void* sc_w, const float* sc_bias, void* sc_y, int sc_Cout) {
  if (!x2) { C1 = Cin; st2 = nullptr; T2 = 0; }
  if (sc_w && (taps != 9 || Cout <= 32 || !sc_y || sc_Cout <= 0 || (sc_Cout & 7)))
    IDF_FAIL(IDF_ERR_UNSUPPORTED, "conv_gn_sc_bf16: the shortcut rides with 3x3 convs of > 32 couts (taps %d Cout %d sc_Cout %d)", taps, Cout, sc_Cout);
  const bool ok = taps == 9 ? shape3_ok(H, W, Cin, Cout, 0) : (taps == 1 && shape1_ok(W, Cin, Cout));
  if (!ok || (Cin % 32) || (x2 && (C1 <= 0 || C1 >= Cin || (C1 % CK))))
    IDF_FAIL(IDF_ERR_UNSUPPORTED, "conv_gn_bf16: B%d H%d W%d Cin%d (C1 %d) Cout%d taps%d not covered", B, H, W, Cin, C1, Cout, taps);
  if (act != 1 && act != 2) IDF_FAIL(IDF_ERR_BADARG, "conv_gn_bf16: act must be 1 or 2");
  if (!st1 || T1 < 1 || (x2 && (!st2 || T2 < 1))) IDF_FAIL(IDF_ERR_BADARG, "conv_gn_bf16: input statistics missing");
  if (st_out && (Cout & 7)) IDF_FAIL(IDF_ERR_UNSUPPORTED, "conv_gn_bf16: statistics need Cout %% 8 == 0");
  if ((sc != nullptr) != (sh != nullptr) || (sc != nullptr) != (mean != nullptr) || (sc != nullptr) != (rstd != nullptr))
    IDF_FAIL(IDF_ERR_BADARG, "conv_gn_bf16: mean / rstd / sc / sh go together");
  if (B == 0) return IDF_OK;
  C3P p;
  clear_pro(p);
  p.x = (const bf16_t*)x; p.w = (const bf16_t*)w; p.bias = bias; p.res = (const bf16_t*)res; p.y = (bf16_t*)y;
  p.x2 = (const bf16_t*)x2; p.C1 = C1; p.st_out = st_out;
  p.st1 = st1; p.T1 = T1; p.st2 = st2; p.T2 = T2;
  p.gamma = gamma; p.beta = beta; p.film_t = film_t; p.film_a = film_a;
  p.ld_t = ld_t ? ld_t : 2 * Cin; p.ld_a = ld_a ? ld_a : 2 * Cin; p.eps = eps;
  p.act = act; p.salt = salt; p.thr = idf_drop_thresh(p_drop);
  p.dscale = 1.0f / (1.0f - (float)p.thr / 65536.0f);
  p.seed = (act == 2 && p_drop > 0.f) ? seed : nullptr;
  p.a_out = (bf16_t*)a_out; p.mean_out = mean; p.rstd_out = rstd; p.sc_out = sc; p.sh_out = sh;
  int BM;
  if (int e = fill_common(p, B, H, W, Cin, Cout, 0, taps == 9 ? 3 : 1, &BM))
    IDF_FAIL(IDF_ERR_UNSUPPORTED, e == 1 ? "conv_gn_bf16: tile too large (H%d W%d)" : "conv_gn_bf16: tensor too large for 32-bit offsets", H, W);
  if ((long)B * H * W * Cin >= (1L << 31)) IDF_FAIL(IDF_ERR_UNSUPPORTED, "conv_gn_bf16: tensor too large for 32-bit offsets");
  hipStream_t st = (hipStream_t)stream;
  // many tiles per image: fold the coefficients once per image in a launch of their own
  static const long coef_min = 1024;
  if (coef_ws && (long)B * p.tiles_per_img * idf_cdiv(Cout, 64) >= coef_min && p.tiles_per_img >= 4) {
    hipLaunchKernelGGL(pro_coef_kernel, dim3(B), dim3(256), (size_t)Cin * 16, st, p, coef_ws);
    IDF_CHECK_LAUNCH();
    p.cof_in = coef_ws;
    p.mean_out = p.rstd_out = p.sc_out = p.sh_out = nullptr;      // written by the coefficient launch
  }
  if (sc_w) {
    // the block's 1x1 shortcut over the same (raw) input as extra blocks of this launch: the register-staged kernel only
    p.aux_x = p.x; p.aux_x2 = p.x2; p.aux_C1 = p.C1; p.aux_Cin = Cin; p.aux_w = (const bf16_t*)sc_w; p.aux_bias = sc_bias;
    p.aux_y = (bf16_t*)sc_y; p.aux_Cout = sc_Cout; p.aux_n_tiles = idf_cdiv(sc_Cout, 64);
    p.n_tiles = idf_cdiv(Cout, 64);
    p.aux_blocks = B * p.tiles_per_img * p.aux_n_tiles;
    if (x2) {
      if (BM == 256) launch<0, 4, 64, 4, 3, true, true>(p, st); else if (BM == 128) launch<0, 4, 64, 2, 3, true, true>(p, st); else launch<0, 2, 64, 2, 3, true, true>(p, st);
    } else {
      if (BM == 256) launch<0, 4, 64, 4, 3, false, true>(p, st); else if (BM == 128) launch<0, 4, 64, 2, 3, false, true>(p, st); else launch<0, 2, 64, 2, 3, false, true>(p, st);
    }
    IDF_CHECK_LAUNCH();
    return IDF_OK;
  }
  if (taps == 9) { if (x2) dispatch3<true, true>(p, 0, BM, st); else dispatch3<false, true>(p, 0, BM, st); }
  else { if (x2) dispatch1<true, true>(p, BM, st); else dispatch1<false, true>(p, BM, st); }
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}

// mean / rstd / (sc, sh) of a GroupNorm stage from the statistics partials its input carries (the fold idf_conv_gn_bf16 does
// in-block, as a launch of its own: grid B).  For the sites where the policy keeps GroupNorm and conv apart on big tensors
// (idf_conv_gn_advice = 0): coefficients from the producer's partials + the streaming idf_gn_apply instead of the one-launch
// GroupNorm, which re-reads nothing but runs its statistics -> apply phases serially per CU.  ws: [B][2C] floats scratch.
extern "C" int idf_gn_coef_from_stats(const float* st1, int T1, const float* st2, int T2, int C1, const float* gamma,
                                      const float* beta, const float* film_t, const float* film_a, int ld_t, int ld_a, float eps,
                                      float* mean, float* rstd, float* sc, float* sh, float* ws, int B, int HW, int C,
                                      void* stream) {
  if (!st1 || T1 < 1 || !mean || !rstd || !sc || !sh || !ws || C <= 0 || (C % 32) || HW <= 0)
    IDF_FAIL(IDF_ERR_BADARG, "gn_coef_from_stats: bad arguments (C %d HW %d T1 %d)", C, HW, T1);
  if (!st2) { C1 = C; T2 = 0; }
  if (st2 && (C1 <= 0 || C1 >= C || T2 < 1)) IDF_FAIL(IDF_ERR_BADARG, "gn_coef_from_stats: C1 %d of %d", C1, C);
  if (B == 0) return IDF_OK;
  C3P p;
  clear_pro(p);
  p.B = B; p.H = HW; p.W = 1; p.Cin = C; p.C1 = C1;
  p.st1 = st1; p.T1 = T1; p.st2 = st2; p.T2 = T2;
  p.gamma = gamma; p.beta = beta; p.film_t = film_t; p.film_a = film_a;
  p.ld_t = ld_t ? ld_t : 2 * C; p.ld_a = ld_a ? ld_a : 2 * C; p.eps = eps;
  p.mean_out = mean; p.rstd_out = rstd; p.sc_out = sc; p.sh_out = sh;
  hipLaunchKernelGGL(pro_coef_kernel, dim3(B), dim3(256), (size_t)C * 16, (hipStream_t)stream, p, ws);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}

extern "C" int idf_conv_gn_bf16(const void* x, const void* x2, int C1, const float* st1, int T1, const float* st2, int T2,
                                const float* gamma, const float* beta, const float* film_t, const float* film_a, int ld_t,
                                int ld_a, float eps, int act, const uint64_t* seed, uint32_t salt, float p_drop,
                                const void* w, const float* bias, const void* res, void* y, void* a_out, float* mean,
                                float* rstd, float* sc, float* sh, float* st_out, float* coef_ws, int B, int H, int W, int Cin,
                                int Cout, int taps, void* stream) {
  return conv_gn_impl(x, x2, C1, st1, T1, st2, T2, gamma, beta, film_t, film_a, ld_t, ld_a, eps, act, seed, salt, p_drop, w, bias,
                      res, y, a_out, mean, rstd, sc, sh, st_out, coef_ws, B, H, W, Cin, Cout, taps, stream, nullptr, nullptr,
                      nullptr, 0);
}

// The same launch also runs the block's 1x1 shortcut  sc_y = conv1x1(x | x2, sc_w [sc_Cout][Cin]) + sc_bias  over the RAW
// input (modules.py:228, 248, 281: `self.shortcut(x)` beside `self.block1(x)`): extra blocks of the 3x3 launch, one launch
// fewer per ResBlock whose channel count changes.  3x3 convs of more than 32 couts, sc_Cout % 8 == 0.
extern "C" int idf_conv_gn_sc_bf16(const void* x, const void* x2, int C1, const float* st1, int T1, const float* st2, int T2,
                                   const float* gamma, const float* beta, const float* film_t, const float* film_a, int ld_t,
                                   int ld_a, float eps, int act, const uint64_t* seed, uint32_t salt, float p_drop,
                                   const void* w, const float* bias, const void* res, void* y, void* a_out, float* mean,
                                   float* rstd, float* sc, float* sh, float* st_out, float* coef_ws, int B, int H, int W,
                                   int Cin, int Cout, int taps, void* stream, const void* sc_w, const float* sc_bias, void* sc_y,
                                   int sc_Cout) {
  if (!sc_w) IDF_FAIL(IDF_ERR_BADARG, "conv_gn_sc_bf16: shortcut weights missing");
  return conv_gn_impl(x, x2, C1, st1, T1, st2, T2, gamma, beta, film_t, film_a, ld_t, ld_a, eps, act, seed, salt, p_drop, w, bias,
                      res, y, a_out, mean, rstd, sc, sh, st_out, coef_ws, B, H, W, Cin, Cout, taps, stream, sc_w, sc_bias, sc_y,
                      sc_Cout);
}

// ---- the backward chain at the big maps (64x64 / 32x32: a tile is a slice of an image, so the GroupNorm backward's
// per-(sample, group) sums cannot close inside one block as they do in idf_conv_dgrad_gn_bf16)
//
//   out = conv(g, w)   with w the data-gradient weights [Cout][taps][Cin] (flipped taps, idf_pack_conv_weight), stride 1
//
// INPUT.  in_x == NULL: g = dy [B,H,W,Cin], an ordinary gradient tensor.  in_x != NULL ("dy prologue"): dy does not exist --
//   the conv that differentiated the GroupNorm stage BEHIND this conv left (du_in = `dy`, in_part [B][in_T][Cin][2]) and
//   g = A*du_in + K1*in_x + K0 is formed while the tile is staged (in_x = that GroupNorm's input; in_mean .. in_film_a its
//   saved statistics / parameters; coefficients: idf_gnfold.h).  That GroupNorm's parameter / FiLM gradients
//   (in_dfilm_t .. in_dbeta_acc, as idf_gn_fused_bwd) are stored by one block per image, and g is written once to dy_out
//   (optional) for the weight gradient of the conv in front of it.
// OUTPUT.  x == NULL: out = dA, a plain data gradient (+ no epilogue extras).  x != NULL ("du epilogue"): the conv input
//   was a = dropout(act(x*sc+sh)) (x possibly the pair x | x2, C1 % 64 == 0); out = du = dA * act'(x*sc+sh) * mask and
//   part_out [B][T][Cout][2] = per-tile (sum du, sum du*x), T = idf_conv_dgrad_chain_tiles(...): feed both to
//   idf_gn_bwd_apply, or to the next call of this function as (dy, in_part).
// bf16, Cin % 32 == 0, Cout % 64 == 0, W a power of two in [4, 128].
namespace {
// pixel tile of a chain launch: the forward kernels' plan (forcing smaller tiles -- 128 / 64 pixels, two or three 256-thread blocks per
// CU -- measured slower: 10.42 / 11.01 vs 10.33 ms per step, profiles/r03_e_ab_chain_knobs.txt; the switch is gone)
bool chain_plan(int B, int H, int W, int Cout, int taps, int* BM, int* R) {
  if (!(taps == 9 ? plan3(B, H, W, Cout, 0, BM, R) : plan1(B, H, W, Cout, BM, R))) return false;
  return true;
}
}  // namespace

extern "C" int idf_conv_dgrad_chain_tiles(int B, int H, int W, int Cin, int Cout, int taps) {
  int BM, R;
  if ((taps != 9 && taps != 1) || (Cout % 64) || (Cin % CK)) return -1;
  if (!(taps == 9 ? shape3_ok(H, W, Cin, Cout, 0) : shape1_ok(W, Cin, Cout)) || !chain_plan(B, H, W, Cout, taps, &BM, &R)) return -1;
  return H / R;
}

static int dgrad_chain_impl(const void* dy, const void* in_x, const float* in_part, int in_T,
                            const float* in_mean, const float* in_rstd, const float* in_sc,
                            const float* in_gamma, const float* in_beta, const float* in_film_t,
                            const float* in_film_a, int in_ld_t, int in_ld_a, float* in_dfilm_t,
                            float* in_dfilm_a, float* in_dgb, float* in_dgamma_acc, float* in_dbeta_acc,
                            void* dy_out, const void* w, const void* x, const void* x2, int C1,
                            const float* sc, const float* sh, const uint64_t* seed, uint32_t salt,
                            float p_drop, int act, void* out, float* part_out, int B, int H, int W, int Cin,
                            int Cout, int taps, void* stream, const void* sc_dy, const void* sc_w, void* sc_dx, int sc_Cin) {
  if (sc_dy && (taps != 9 || in_x || !sc_w || !sc_dx || sc_Cin <= 0 || (sc_Cin % 32)))
    IDF_FAIL(IDF_ERR_UNSUPPORTED, "conv_dgrad_chain_sc_bf16: the shortcut's data gradient rides with 3x3 launches without a dy prologue (taps %d sc_Cin %d)", taps, sc_Cin);
  if (idf_conv_dgrad_chain_tiles(B, H, W, Cin, Cout, taps) < 0)
    IDF_FAIL(IDF_ERR_UNSUPPORTED, "conv_dgrad_chain_bf16: B%d H%d W%d Cin%d Cout%d taps%d not covered", B, H, W, Cin, Cout, taps);
  if (!dy || !w || !out) IDF_FAIL(IDF_ERR_BADARG, "conv_dgrad_chain_bf16: null argument");
  if (in_x && (!in_part || in_T < 1 || !in_mean || !in_rstd || !in_sc))
    IDF_FAIL(IDF_ERR_BADARG, "conv_dgrad_chain_bf16: dy prologue needs partials, mean, rstd and sc");
  if (x && (!sc || !sh || !part_out || (act != 1 && act != 2))) IDF_FAIL(IDF_ERR_BADARG, "conv_dgrad_chain_bf16: du epilogue needs sc, sh, part_out, act 1|2");
  if (!x2) C1 = Cout;
  if (x && x2 && (C1 <= 0 || C1 >= Cout || (C1 % 64))) IDF_FAIL(IDF_ERR_UNSUPPORTED, "conv_dgrad_chain_bf16: C1 %d of %d", C1, Cout);
  if (!in_x && !x) IDF_FAIL(IDF_ERR_BADARG, "conv_dgrad_chain_bf16: neither prologue nor epilogue asked for (use idf_conv3x3_bf16)");
  if (B == 0) return IDF_OK;
  C3P p;
  clear_pro(p);
  p.x = (const bf16_t*)dy; p.x2 = nullptr; p.C1 = Cin; p.w = (const bf16_t*)w; p.bias = nullptr; p.res = nullptr;
  p.y = (bf16_t*)out;
  int BM;
  if (int e = fill_common(p, B, H, W, Cin, Cout, 0, taps == 9 ? 3 : 1, &BM))
    IDF_FAIL(IDF_ERR_UNSUPPORTED, e == 1 ? "conv_dgrad_chain_bf16: tile too large (H%d W%d)" : "conv_dgrad_chain_bf16: tensor too large for 32-bit offsets", H, W);
  {
    int R;
    chain_plan(B, H, W, Cout, taps, &BM, &R);
    const int halo = taps == 9 ? 2 : 0;
    p.R = R; p.tiles_per_img = H / R; p.wh_magic = wh_magic(W + halo, (R + halo) * (W + halo));
    if (!p.wh_magic) IDF_FAIL(IDF_ERR_UNSUPPORTED, "conv_dgrad_chain_bf16: tile not addressable (H%d W%d)", H, W);
  }
  p.n_tiles = Cout / 64;
  if (in_x) {
    p.dyp_x = (const bf16_t*)in_x; p.dyp_out = (bf16_t*)dy_out;
    GnFoldP& f = p.dyp_f;
    f.part = in_part; f.T = in_T; f.mean = in_mean; f.rstd = in_rstd; f.sc = in_sc; f.gamma = in_gamma; f.beta = in_beta;
    f.film_t = in_film_t; f.film_a = in_film_a; f.ld_t = in_ld_t ? in_ld_t : 2 * Cin; f.ld_a = in_ld_a ? in_ld_a : 2 * Cin;
    f.dfilm_t = in_dfilm_t; f.dfilm_a = in_dfilm_a; f.dgb = in_dgb; f.dgam = in_dgamma_acc; f.dbet = in_dbeta_acc;
    f.C = Cin; f.HW = H * W;
  }
  if (x) {
    p.due_x = (const bf16_t*)x; p.due_x2 = (const bf16_t*)x2; p.due_C1 = C1; p.due_sc = sc; p.due_sh = sh;
    p.st_out = part_out;
    p.act = act; p.salt = salt; p.thr = idf_drop_thresh(p_drop);
    p.dscale = 1.0f / (1.0f - (float)p.thr / 65536.0f);
    p.seed = (act == 2 && p_drop > 0.f) ? seed : nullptr;
  }
  if (sc_dy) {
    // sc_dx [B,H,W,Cout] = conv1x1(sc_dy [B,H,W,sc_Cin], sc_w [Cout][sc_Cin]): the shortcut's data gradient, extra blocks
    p.aux_x = (const bf16_t*)sc_dy; p.aux_x2 = nullptr; p.aux_C1 = sc_Cin; p.aux_Cin = sc_Cin; p.aux_w = (const bf16_t*)sc_w;
    p.aux_bias = nullptr; p.aux_y = (bf16_t*)sc_dx; p.aux_Cout = Cout; p.aux_n_tiles = Cout / 64;
    p.aux_blocks = B * p.tiles_per_img * p.aux_n_tiles;
  }
  hipStream_t st = (hipStream_t)stream;
  const int bwd = (in_x ? 1 : 0) | (x ? 2 : 0);
#define IDF_CHAIN(KS)                                                                                     \
  do {                                                                                                    \
    if (BM == 256) { if (bwd == 3) launch_bwd_chain<4, 4, KS, 3>(p, st); else if (bwd == 2) launch_bwd_chain<4, 4, KS, 2>(p, st); else launch_bwd_chain<4, 4, KS, 1>(p, st); } \
    else if (BM == 128) { if (bwd == 3) launch_bwd_chain<4, 2, KS, 3>(p, st); else if (bwd == 2) launch_bwd_chain<4, 2, KS, 2>(p, st); else launch_bwd_chain<4, 2, KS, 1>(p, st); } \
    else { if (bwd == 3) launch_bwd_chain<2, 2, KS, 3>(p, st); else if (bwd == 2) launch_bwd_chain<2, 2, KS, 2>(p, st); else launch_bwd_chain<2, 2, KS, 1>(p, st); } \
  } while (0)
  if (taps == 9) IDF_CHAIN(3); else IDF_CHAIN(1);
#undef IDF_CHAIN
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}

extern "C" int idf_conv_dgrad_chain_bf16(const void* dy, const void* in_x, const float* in_part, int in_T,
                                         const float* in_mean, const float* in_rstd, const float* in_sc,
                                         const float* in_gamma, const float* in_beta, const float* in_film_t,
                                         const float* in_film_a, int in_ld_t, int in_ld_a, float* in_dfilm_t,
                                         float* in_dfilm_a, float* in_dgb, float* in_dgamma_acc, float* in_dbeta_acc,
                                         void* dy_out, const void* w, const void* x, const void* x2, int C1,
                                         const float* sc, const float* sh, const uint64_t* seed, uint32_t salt,
                                         float p_drop, int act, void* out, float* part_out, int B, int H, int W, int Cin,
                                         int Cout, int taps, void* stream) {
  return dgrad_chain_impl(dy, in_x, in_part, in_T, in_mean, in_rstd, in_sc, in_gamma, in_beta, in_film_t, in_film_a, in_ld_t,
                          in_ld_a, in_dfilm_t, in_dfilm_a, in_dgb, in_dgamma_acc, in_dbeta_acc, dy_out, w, x, x2, C1, sc, sh, seed,
                          salt, p_drop, act, out, part_out, B, H, W, Cin, Cout, taps, stream, nullptr, nullptr, nullptr, 0);
}

// The same launch also runs the data gradient of the block's 1x1 shortcut,  sc_dx [B,H,W,Cout] = conv1x1(sc_dy [B,H,W,sc_Cin],
// sc_w [Cout][sc_Cin])  (sc_w = the shortcut's data-gradient weights, idf_pack_conv_weight): both gradients of a block
// entry in one launch; sc_dx then joins idf_gn_bwd_apply as `dres`.  3x3, du epilogue only (no dy prologue), sc_Cin % 32 == 0.
extern "C" int idf_conv_dgrad_chain_sc_bf16(const void* dy, const void* w, const void* x, const void* x2, int C1, const float* sc,
                                            const float* sh, const uint64_t* seed, uint32_t salt, float p_drop, int act,
                                            void* out, float* part_out, int B, int H, int W, int Cin, int Cout, void* stream,
                                            const void* sc_dy, const void* sc_w, void* sc_dx, int sc_Cin) {
  if (!sc_dy) IDF_FAIL(IDF_ERR_BADARG, "conv_dgrad_chain_sc_bf16: shortcut gradient missing");
  return dgrad_chain_impl(dy, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr,
                          nullptr, nullptr, nullptr, nullptr, nullptr, w, x, x2, C1, sc, sh, seed, salt, p_drop, act, out,
                          part_out, B, H, W, Cin, Cout, 9, stream, sc_dy, sc_w, sc_dx, sc_Cin);
}
