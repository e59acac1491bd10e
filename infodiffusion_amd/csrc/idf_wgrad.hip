// Stride-1 weight gradient (3x3 and 1x1) for bf16 activations on MFMA (gfx950):
//   dW[n][tap][c] += sum_{pixels} dy[pix][n] * a[pix + tap][c]       (+ db[n] += sum dy[pix][n])
// `a` is the already-activated conv input (idf_gn_apply output, or the raw input of
// an un-normalised conv).  Both operands are pixel-major in HBM and the contraction
// runs over pixels, so tiles are copied to LDS in their natural [pixel][channel]
// layout (16-byte vectors in, ds_write_b128) and the MFMA fragments are fetched with
// the gfx950 transposed read ds_read_b64_tr_b16 -- no scalar scatter, no per-tap
// re-staging: one R-row input tile (with a one-pixel column halo) serves the three
// taps of a kernel row through address offsets.
//
// Block = 64 couts x 64 cins x one kernel row (3 taps; 1 tap for 1x1).  grid.x =
// (c-tiles x n-tiles x kernel rows), grid.y splits the (image, row-group) tiles.
// Partial sums stay in registers across a block's tiles and are added to dW with
// fp32 atomics (64 contiguous bytes per row segment).  Splitting the kernel rows
// over blocks keeps the per-block partial small, so the whole chip is busy with
// 3x fewer atomic bytes than a 9-tap block would need.
// LDS pixel pitch is 80 bf16 (160 B): 8 consecutive pixels x 32 B then tile all
// 64 banks, so the transposed reads are conflict-free.
#include "idf_common.h"
#include <stdlib.h>

namespace {

struct WgP {
  const bf16_t* a;    // [B,H,W,Cin]  (or [.., C1] when a2 is set)
  const bf16_t* a2;   // channels C1.. of a never-materialised concatenation ([B,H,W,Cin-C1]) or null; C1 % 64 == 0
  int C1;
  const bf16_t* dy;   // [B,H,W,Cout]
  float* dW;          // [Cout][taps][Cin]
  float* db;          // [Cout] or null
  int B, H, W, Cin, Cout;   // H, W: output (dy) dims
  int Hs, Ws;               // source dims of `a` (S1: H,W; S2: 2H,2W; UP2: H/2,W/2)
  int R;              // output rows per tile (R*W in {32,64,128})
  int tiles;          // B*H/R
  int tiles_per_blk;
  int c_tiles, n_tiles;
  int Cw, Nw;         // dW is [Nw][taps][Cw] (Cw <= Cin, Nw <= Cout): channels the operands were zero-padded by get no gradient
  int wshift;         // log2(W) (W is a power of two); sub-pixel UP2 form: log2(W / 2)
  int upsub;          // MODE 2 in the sub-pixel form (wgrad_block_upsub): R = LOW-resolution rows per tile
  unsigned wh_magic;  // (pix * wh_magic) >> 16 == pix / WH over the staged tile's pixels (checked on the host)
  // two-stage, deterministic accumulation (batched launches): every pixel split writes ITS partial dW | db into its own slab of
  // `ws` with plain stores (slab s at ws + s * ws_stride floats: dW [Nw][taps][Cw], then db [Nw]; every element of a slab has
  // exactly one writer), and idf_wgrad_reduce_batched adds the slabs to dW / db in slab order.  ws == null: fp32 atomics into dW.
  float* ws;
  int ws_stride, ws_dboff;      // floats per slab (multiple of 4); offset of db inside a slab (= Nw * taps * Cw)
  int ring;                     // stride-1 3x3 on a 64- or 32-wide map with whole 64-channel tiles: the row-ring form (wgrad_block_ring)
};

#ifndef IDF_WGRAD_BLOCKS
#define IDF_WGRAD_BLOCKS 320   // target grid size: ~1.25 blocks per CU keeps the atomic bytes low
#endif
#ifndef IDF_WG_ABL
#define IDF_WG_ABL 0      // diagnostic builds (tools/build_variant.sh): 1 no global loads past the first tiles, 2 no LDS stores past the first,
#endif                    // 4 no MFMA loop, 8 MFMAs without their LDS reads, 16 no address arithmetic in the k loop (tools/bench_wgrad.py)
constexpr int PITCH = 80;                 // elements per LDS pixel row
constexpr int PITCHB = PITCH * 2;         // bytes

// Loads / atomics through pointers that came out of a descriptor TABLE are flat instructions to hipcc (address space unknown),
// and a flat load counts on lgkmcnt as well as vmcnt: every `s_waitcnt lgkmcnt(0)` in front of the MFMAs' LDS operands then
// also waits for the NEXT tile's prefetch.  Going through the global address space explicitly keeps the two queues apart.
typedef __attribute__((ext_vector_type(4))) unsigned wg_u32x4_t;
__device__ __forceinline__ uint4 gload16(const bf16_t* p) {      // (the cast goes through an integer: a pointer-to-pointer cast stays generic)
  const wg_u32x4_t v = *(const __attribute__((address_space(1))) wg_u32x4_t*)(unsigned long long)p;
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void gatomic_add(float* p, float v) {
  __hip_atomic_fetch_add((__attribute__((address_space(1))) float*)(unsigned long long)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void gstore(float* p, float v) {
  *(__attribute__((address_space(1))) float*)(unsigned long long)p = v;
}
// one partial-sum element: into the split's slab (deterministic path) or onto the gradient with an atomic
__device__ __forceinline__ void wg_emit(const WgP& p, int slab, size_t idx, float v) {
  if (p.ws) gstore(p.ws + (size_t)slab * p.ws_stride + idx, v);
  else gatomic_add(p.dW + idx, v);
}
__device__ __forceinline__ void wg_emit_db(const WgP& p, int slab, int n, float v) {
  if (p.ws) gstore(p.ws + (size_t)slab * p.ws_stride + p.ws_dboff + n, v);
  else gatomic_add(p.db + n, v);
}

__device__ __forceinline__ s16x4_t tr_read(const bf16_t* lds_ptr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4_t*)(lds_ptr));
}

__device__ __forceinline__ bf16x8_t mkfrag(s16x4_t lo, s16x4_t hi) {
  union { struct { s16x4_t a, b; } s; bf16x8_t v; } u;
  u.s.a = lo; u.s.b = hi;
  return u.v;
}

// KW = taps per block along x (3 for a 3x3 kernel row, 1 for 1x1).
// MODE 0: stride 1; 1: stride 2 (DownSample); 2: nearest-x2-upsampled input (UpSample):
// only the staging differs -- the LDS tile always holds the pixels the taps address.
// ROWS = 1: the block owns ONE kernel row (256 threads; the three rows of a (pixel range, cout tile, cin tile) are three
//   blocks, each staging its own a / dy tiles: every operand byte crosses the fabric three times).
// ROWS = 3 (KW 3, MODE 0): the three kernel rows are three 4-wave groups of ONE 768-thread block that share one staged
//   (R + 2)-row input tile and one dy tile -- a is fetched (R + 2) / R times instead of three, dy once; each group keeps
//   its own kernel row's accumulators (48 registers per thread, as before).
template <int KW, int MODE, int ROWS = 1>
__device__ __forceinline__ void wgrad_block(const WgP& p, int bx, const int by) {
  static_assert(ROWS == 1 || (ROWS == 3 && KW == 3 && MODE == 0), "the shared-tile form is the stride-1 3x3 one");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int HALO = KW / 2;
  constexpr int SX = (MODE == 1) ? 2 : 1;
  constexpr int NT = 256 * ROWS;
  const int W = p.W, R = p.R, WH = SX * W + 2 * HALO;
  const int npix_h = (R + (ROWS == 3 ? 2 : 0)) * WH;   // staged input pixels (ROWS 3: one halo row above and below)
  const int KT = R * W;                   // contraction length per tile
  bf16_t* Xs = reinterpret_cast<bf16_t*>(smem);
  bf16_t* Ds = Xs + npix_h * PITCH;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = (tid >> 6) & 3;        // wave inside its 4-wave group
  const int ky = (ROWS == 3) ? (tid >> 8) : ((KW == 3) ? bx % 3 : 0);
  if (KW == 3 && ROWS == 1) bx /= 3;
  const int c0 = (bx % p.c_tiles) * 64, n0 = (bx / p.c_tiles) * 64;
  const int wn0 = (wave >> 1) * 32, wc0 = (wave & 1) * 32;
  const int tiles_per_img = p.H / R;
  const int t_beg = by * p.tiles_per_blk, t_end = min(p.tiles, t_beg + p.tiles_per_blk);
  const int ntaps = KW * KW;

  f32x4_t acc[KW][2][2];
#pragma unroll
  for (int t = 0; t < KW; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[t][i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int v8 = tid & 7;                 // this thread's 8-channel vector slot (fixed)
  float dbs[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) dbs[e] = 0.f;
  const bool do_db = p.db != nullptr && c0 == 0 && (ROWS == 3 || ky == HALO);     // ROWS 3: dy is staged once, by every thread
  const bool cvalid = (c0 + v8 * 8) < p.Cin, nvalid = (n0 + v8 * 8) < p.Cout;

  // transposed-read lane geometry
  const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;

  // register double-buffering: tile t+1 is fetched while tile t is in the MFMAs
  constexpr int XV = ROWS == 3 ? 3 : 5, DV = ROWS == 3 ? 2 : 4;   // 16-byte vectors per thread (input rows / dy)
  struct TileRegs { uint4 x[XV], d[DV]; };
  TileRegs rg0, rg1;       // ROWS 3 (one block per CU: nobody else covers a tile's flight): TWO tiles in flight
  // input tile source: a cin tile lies entirely in `a` or in `a2`
  const bf16_t* asrc = p.a + c0;
  int apitch = p.Cin;
  if (p.a2) {
    if (c0 < p.C1) apitch = p.C1;
    else { asrc = p.a2 + (c0 - p.C1); apitch = p.Cin - p.C1; }
  }
  auto load_tile = [&](int t, TileRegs& rg) {
    uint4 (&xreg)[XV] = rg.x;
    uint4 (&dreg)[DV] = rg.d;
    if ((IDF_WG_ABL & 1) && ROWS == 3 && t > t_beg + 1) return;
    const int b = t / tiles_per_img, oy0 = (t - b * tiles_per_img) * R;
#pragma unroll
    for (int k = 0; k < XV; ++k) {
      int idx = tid + k * NT;
      uint4 val = make_uint4(0, 0, 0, 0);
      if (idx < npix_h * 8) {
        int pix = idx >> 3;
        int hy = (int)(((unsigned)pix * p.wh_magic) >> 16), hx = pix - hy * WH;      // no integer division in the tile loop
        int iy = (ROWS == 3) ? (oy0 + hy - 1) : (SX * (oy0 + hy) + ky - HALO), ix = hx - HALO;
        if (cvalid && (unsigned)iy < (unsigned)(SX * p.H) && (unsigned)ix < (unsigned)(SX * W)) {
          if (MODE == 2) { iy >>= 1; ix >>= 1; }
          val = gload16(asrc + ((size_t)(b * p.Hs + iy) * p.Ws + ix) * apitch + v8 * 8);
        }
      }
      xreg[k] = val;
    }
#pragma unroll
    for (int k = 0; k < DV; ++k) {
      int idx = tid + k * NT;
      uint4 val = make_uint4(0, 0, 0, 0);
      if (idx < KT * 8 && nvalid)
        val = gload16(p.dy + ((size_t)(b * p.H + oy0) * W + (idx >> 3)) * p.Cout + n0 + v8 * 8);
      dreg[k] = val;
    }
  };
  int abl_stores = 0;
  auto store_tile = [&](TileRegs& rg, bf16_t* Xs, bf16_t* Ds) {
    uint4 (&xreg)[XV] = rg.x;
    uint4 (&dreg)[DV] = rg.d;
    if ((IDF_WG_ABL & 2) && ROWS == 3 && abl_stores++ >= 2) return;
#pragma unroll
    for (int k = 0; k < XV; ++k) {
      int idx = tid + k * NT;
      if (idx < npix_h * 8) *reinterpret_cast<uint4*>(Xs + (idx >> 3) * PITCH + v8 * 8) = xreg[k];
    }
#pragma unroll
    for (int k = 0; k < DV; ++k) {
      int idx = tid + k * NT;
      if (idx < KT * 8) {
        *reinterpret_cast<uint4*>(Ds + (idx >> 3) * PITCH + v8 * 8) = dreg[k];
        if (do_db) {
          uint32_t w4[4] = {dreg[k].x, dreg[k].y, dreg[k].z, dreg[k].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            dbs[2 * e] += __uint_as_float(w4[e] << 16);
            dbs[2 * e + 1] += __uint_as_float(w4[e] & 0xffff0000u);
          }
        }
      }
    }
  };

  // ---- MFMA over the tile's pixels, 32 per step
  auto compute = [&](const bf16_t* Xs, const bf16_t* Ds) {
    // ---- MFMA over the tile's pixels, 32 per step.  Logical k slot (g, j) maps to
    // physical pixel 4g+j (j<4) / 16+4g+(j-4): consecutive pixels per read half.
    if ((IDF_WG_ABL & 4) && ROWS == 3) return;
    if ((IDF_WG_ABL & 8) && ROWS == 3) {
      bf16x8_t f0 = mkfrag(tr_read(Ds + lane * 4), tr_read(Ds + lane * 4 + 16)), f1 = mkfrag(tr_read(Xs + lane * 4), tr_read(Xs + lane * 4 + 16));
      for (int ks = 0; ks < KT; ks += 32)
#pragma unroll
        for (int kx = 0; kx < KW; ++kx)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[kx][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f0, f1, acc[kx][i][j], 0, 0, 0);
      return;
    }
    for (int ks = 0; ks < KT; ks += 32) {
      bf16x8_t nf[2];
      const int pixA = ((IDF_WG_ABL & 16) ? 0 : ks) + 4 * g + q, pixB = pixA + 16;
      {
        const bf16_t* d0 = Ds + pixA * PITCH + wn0 + 4 * pp;
        const bf16_t* d1 = Ds + pixB * PITCH + wn0 + 4 * pp;
#pragma unroll
        for (int i = 0; i < 2; ++i) nf[i] = mkfrag(tr_read(d0 + i * 16), tr_read(d1 + i * 16));
      }
      const int oyA = pixA >> p.wshift, oxA = pixA & (W - 1), oyB = pixB >> p.wshift, oxB = pixB & (W - 1);
      const int yo = (ROWS == 3) ? ky : 0;        // the group's kernel row selects the staged row
      const bf16_t* x0 = Xs + ((oyA + yo) * WH + SX * oxA) * PITCH + wc0 + 4 * pp;
      const bf16_t* x1 = Xs + ((oyB + yo) * WH + SX * oxB) * PITCH + wc0 + 4 * pp;
#pragma unroll
      for (int kx = 0; kx < KW; ++kx) {
        bf16x8_t cf[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) cf[j] = mkfrag(tr_read(x0 + kx * PITCH + j * 16), tr_read(x1 + kx * PITCH + j * 16));
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[kx][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(nf[i], cf[j], acc[kx][i][j], 0, 0, 0);
      }
    }
  };
  if constexpr (ROWS == 3) {
    // one block per CU: nobody else covers a tile's staging, so the tile lives in LDS twice.  Tile t is contracted from
    // one buffer while the waves that are done with it already write tile t + 1 (in registers since two tiles ago) into
    // the other: ONE barrier per tile, and the MFMA pipe no longer idles through every LDS write phase.
    bf16_t* X1 = Xs + (npix_h + KT) * PITCH;
    bf16_t* D1 = X1 + npix_h * PITCH;
    if (t_beg < t_end) load_tile(t_beg, rg0);
    if (t_beg + 1 < t_end) load_tile(t_beg + 1, rg1);
    if (t_beg < t_end) {
      store_tile(rg0, Xs, Ds);
      if (t_beg + 2 < t_end) load_tile(t_beg + 2, rg0);
    }
    __syncthreads();
    for (int t = t_beg; t < t_end; t += 2) {
      compute(Xs, Ds);                                  // tile t; rg1 = tile t + 1, rg0 = tile t + 2 (in flight)
      if (t + 1 < t_end) {
        store_tile(rg1, X1, D1);
        if (t + 3 < t_end) load_tile(t + 3, rg1);
      }
      __syncthreads();
      if (t + 1 >= t_end) break;
      compute(X1, D1);                                  // tile t + 1; rg0 = tile t + 2, rg1 = tile t + 3 (in flight)
      if (t + 2 < t_end) {
        store_tile(rg0, Xs, Ds);
        if (t + 4 < t_end) load_tile(t + 4, rg0);
      }
      __syncthreads();
    }
  } else {
    if (t_beg < t_end) load_tile(t_beg, rg0);
    for (int t = t_beg; t < t_end; ++t) {
      store_tile(rg0, Xs, Ds);
      __syncthreads();
      if (t + 1 < t_end) load_tile(t + 1, rg0);        // tile t + 1 is fetched while tile t is in the MFMAs
      compute(Xs, Ds);
      __syncthreads();
    }
  }

  // D: row = n (4 per lane), col = c (lane & 15)
#pragma unroll
  for (int kx = 0; kx < KW; ++kx)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int c = c0 + wc0 + j * 16 + (lane & 15);
        if (c >= p.Cw) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int n = n0 + wn0 + i * 16 + (lane >> 4) * 4 + r;
          if (n < p.Nw) wg_emit(p, by, ((size_t)n * ntaps + ky * KW + kx) * p.Cw + c, acc[kx][i][j][r]);
        }
      }
    }

  if (do_db) {
    // reduce the NT / 8 threads that share v8, then one atomic per cout
    float* red = reinterpret_cast<float*>(smem);   // [NT / 8][64]
#pragma unroll
    for (int e = 0; e < 8; ++e) red[(tid >> 3) * 64 + v8 * 8 + e] = dbs[e];
    __syncthreads();
    if (tid < 64 && n0 + tid < p.Nw) {
      float s = 0.f;
      for (int k = 0; k < NT / 8; ++k) s += red[k * 64 + tid];
      wg_emit_db(p, by, n0 + tid, s);
    }
  }
}

// ---- UpSample's weight gradient (MODE 2) in the sub-pixel form: 16 tap products per low-resolution pixel instead of 36.
// With oy = 2 Y + py, ox = 2 X + px the up-sampled input under tap (ky, kx) of output pixel (oy, ox) is the low-resolution pixel
// (Y + ty + py - 1, X + tx + px - 1) with ty = 0 for ky in S(py, 0), 1 for ky in S(py, 1) (S(0,0) = {0}, S(0,1) = {1,2}, S(1,0) = {0,1},
// S(1,1) = {2}; the same along x): taps that share (ty, tx) multiply dy with the SAME pixel, so
//   G[py][px][ty][tx] = sum_{Y,X} dy[2Y+py][2X+px] (x) x[Y+ty+py-1][X+tx+px-1]     and     dW[ky][kx] = sum of the G whose sets hold (ky, kx).
// A block owns (cin tile, cout tile, py): both column parities, 8 accumulator sets (2 px x 2 ty x 2 tx); per staged tile -- RL
// low-resolution rows of x with a one-pixel halo, the two dy parity planes of those rows -- every 32-pixel k-step runs 32 MFMAs.
// The epilogue adds the column-combined tiles (kx = 0: G[0][.][0] + G[1][.][0]; 1: G[0][.][1] + G[1][.][0]; 2: G[0][.][1] + G[1][.][1]) to
// every kernel row of S(py, ty) with fp32 atomics.
__device__ __forceinline__ void wgrad_block_upsub(const WgP& p, int bx, const int by) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NT = 256;
  const int Wl = p.W >> 1, Hl = p.H >> 1, RL = p.R, WHl = Wl + 2, W2 = p.W;
  const int npix_h = (RL + 2) * WHl, KT = RL * Wl;
  bf16_t* Xs = reinterpret_cast<bf16_t*>(smem);
  bf16_t* Ds = Xs + npix_h * PITCH;                 // [2 px][KT]
  const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3;
  const int py = bx & 1;
  bx >>= 1;
  const int c0 = (bx % p.c_tiles) * 64, n0 = (bx / p.c_tiles) * 64;
  const int wn0 = (wave >> 1) * 32, wc0 = (wave & 1) * 32;
  const int tiles_per_img = Hl / RL;
  const int t_beg = by * p.tiles_per_blk, t_end = min(p.tiles, t_beg + p.tiles_per_blk);

  f32x4_t acc[2][2][2][2][2];                       // [px][ty][tx][i][j]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[a][t][u][i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int v8 = tid & 7;
  float dbs[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) dbs[e] = 0.f;
  const bool do_db = p.db != nullptr && c0 == 0;
  const bool cvalid = (c0 + v8 * 8) < p.Cin, nvalid = (n0 + v8 * 8) < p.Cout;
  const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  constexpr int XV = 7, DV = 8;
  uint4 xreg[XV], dreg[DV];
  auto load_tile = [&](int t) {
    const int b = t / tiles_per_img, Y0 = (t - b * tiles_per_img) * RL;
#pragma unroll
    for (int k = 0; k < XV; ++k) {
      const int idx = tid + k * NT;
      uint4 val = make_uint4(0, 0, 0, 0);
      if (idx < npix_h * 8) {
        const int pix = idx >> 3;
        const int hy = (int)(((unsigned)pix * p.wh_magic) >> 16), hx = pix - hy * WHl;
        const int iy = Y0 + hy - 1, ix = hx - 1;
        if (cvalid && (unsigned)iy < (unsigned)Hl && (unsigned)ix < (unsigned)Wl)
          val = gload16(p.a + ((size_t)(b * Hl + iy) * Wl + ix) * p.Cin + c0 + v8 * 8);
      }
      xreg[k] = val;
    }
#pragma unroll
    for (int k = 0; k < DV; ++k) {
      const int idx = tid + k * NT;
      uint4 val = make_uint4(0, 0, 0, 0);
      if (idx < 2 * KT * 8 && nvalid) {
        const int pe = idx >> 3, px = pe >= KT ? 1 : 0, pl = pe - px * KT;
        const int Yl = pl >> p.wshift, Xl = pl & (Wl - 1);
        val = gload16(p.dy + ((size_t)(b * p.H + 2 * (Y0 + Yl) + py) * W2 + 2 * Xl + px) * p.Cout + n0 + v8 * 8);
      }
      dreg[k] = val;
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int k = 0; k < XV; ++k) {
      const int idx = tid + k * NT;
      if (idx < npix_h * 8) *reinterpret_cast<uint4*>(Xs + (idx >> 3) * PITCH + v8 * 8) = xreg[k];
    }
#pragma unroll
    for (int k = 0; k < DV; ++k) {
      const int idx = tid + k * NT;
      if (idx < 2 * KT * 8) {
        *reinterpret_cast<uint4*>(Ds + (idx >> 3) * PITCH + v8 * 8) = dreg[k];
        if (do_db) {
          const uint32_t w4[4] = {dreg[k].x, dreg[k].y, dreg[k].z, dreg[k].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            dbs[2 * e] += __uint_as_float(w4[e] << 16);
            dbs[2 * e + 1] += __uint_as_float(w4[e] & 0xffff0000u);
          }
        }
      }
    }
  };
  if (t_beg < t_end) load_tile(t_beg);
  for (int t = t_beg; t < t_end; ++t) {
    store_tile();
    __syncthreads();
    if (t + 1 < t_end) load_tile(t + 1);
    for (int ks = 0; ks < KT; ks += 32) {
      const int pixA = ks + 4 * g + q, pixB = pixA + 16;
      const int yA = pixA >> p.wshift, xA = pixA & (Wl - 1), yB = pixB >> p.wshift, xB = pixB & (Wl - 1);
      // x fragments at halo rows y + ty + py, columns x + cx for cx in {0, 1, 2} (cx = tx + px)
      bf16x8_t cf[2][3][2];
#pragma unroll
      for (int ty = 0; ty < 2; ++ty) {
        const bf16_t* x0 = Xs + ((yA + ty + py) * WHl + xA) * PITCH + wc0 + 4 * pp;
        const bf16_t* x1 = Xs + ((yB + ty + py) * WHl + xB) * PITCH + wc0 + 4 * pp;
#pragma unroll
        for (int cx = 0; cx < 3; ++cx)
#pragma unroll
          for (int j = 0; j < 2; ++j) cf[ty][cx][j] = mkfrag(tr_read(x0 + cx * PITCH + j * 16), tr_read(x1 + cx * PITCH + j * 16));
      }
#pragma unroll
      for (int px = 0; px < 2; ++px) {
        bf16x8_t nf[2];
        const bf16_t* d0 = Ds + (px * KT + pixA) * PITCH + wn0 + 4 * pp;
        const bf16_t* d1 = Ds + (px * KT + pixB) * PITCH + wn0 + 4 * pp;
#pragma unroll
        for (int i = 0; i < 2; ++i) nf[i] = mkfrag(tr_read(d0 + i * 16), tr_read(d1 + i * 16));
#pragma unroll
        for (int ty = 0; ty < 2; ++ty)
#pragma unroll
          for (int tx = 0; tx < 2; ++tx)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j)
                acc[px][ty][tx][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(nf[i], cf[ty][tx + px][j], acc[px][ty][tx][i][j], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // ---- epilogue: combine the column parities, add to every kernel row of S(py, ty)
#pragma unroll
  for (int ty = 0; ty < 2; ++ty) {
    const int ky0 = py == 0 ? (ty == 0 ? 0 : 1) : (ty == 0 ? 0 : 2);
    const int nky = (py == 0) == (ty == 1) ? 2 : 1;        // S(0,1) = {1,2}, S(1,0) = {0,1}: two rows; S(0,0), S(1,1): one
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int c = c0 + wc0 + j * 16 + (lane & 15);
          if (c >= p.Cw) continue;
          f32x4_t v;
          if (kx == 0) v = acc[0][ty][0][i][j] + acc[1][ty][0][i][j];
          else if (kx == 1) v = acc[0][ty][1][i][j] + acc[1][ty][0][i][j];
          else v = acc[0][ty][1][i][j] + acc[1][ty][1][i][j];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int n = n0 + wn0 + i * 16 + (lane >> 4) * 4 + r;
            if (n < p.Nw) {
              for (int kk = 0; kk < nky; ++kk) wg_emit(p, by * 2 + py, ((size_t)n * 9 + (ky0 + kk) * 3 + kx) * p.Cw + c, v[r]);      // slab per (split, row parity): S(py, 0) and S(py, 1) are disjoint, one writer per element
            }
          }
        }
  }
  if (do_db) {
    float* red = reinterpret_cast<float*>(smem);   // [NT / 8][64]
#pragma unroll
    for (int e = 0; e < 8; ++e) red[(tid >> 3) * 64 + v8 * 8 + e] = dbs[e];
    __syncthreads();
    if (tid < 64 && n0 + tid < p.Nw) {
      float sacc = 0.f;
      for (int k = 0; k < NT / 8; ++k) sacc += red[k * 64 + tid];
      wg_emit_db(p, by * 2 + py, n0 + tid, sacc);
    }
  }
}

// ---- Round 6: the stride-1 3x3 class rebuilt from what the ablations of the shared-tile kernel above measured (profiles/r06_wgrad.txt:
// of 1032 us back to back, 354 us are the fragment reads, 205 us the global loads, 80 us the LDS stores; MFMA pipe busy 32 %):
//  * ROW RING.  Consecutive tiles of a block are vertically adjacent, so 2 of a tile's R + 2 input rows were staged for the tile before.
//    The input lives in a ring of 2R + 4 LDS rows; a tile fetches and stores only its R NEW rows (all R + 2 at the top of an image / of
//    the block): half the input bytes at W = 64 (R = 2), two thirds at W = 32 (R = 4) -- fewer global loads, fewer ds_write_b128.
//  * WAVE TILE 64 couts x 16 cins x 9 taps, the contraction split over wave pairs: 8 waves = (k half, cin quarter), 144 accumulator
//    registers each at two waves per SIMD.  Per 32-pixel k-step a wave reads 4 dy + 9 input fragments for 36 MFMAs -- 0.72 transposed
//    reads per MFMA (the shared-tile kernel: 2 + 6 for 12, 1.33).  The two k halves meet once per block, through LDS.
//  * NO BRANCHES AROUND LOADS.  Every staging load is a buffer load whose offset is pushed out of range where the pixel lies outside
//    the image, the channel beyond the tensor's, or the tile has fewer rows (the range check returns zeros): straight-line code, so
//    hipcc counts vmcnt instead of draining it, and a tile's loads stay in flight across the tile before's contraction.
// Per iteration: store tile t + 1 (registers -> LDS), issue tile t + 2's loads, contract tile t, one barrier (LDS only).
// W in {64, 32, 16, 8}; channel counts multiples of 8 (partial 64-channel tiles contract zeros); tensors < 2 GB (32-bit offsets).
// (The body is a function of its own per map width -- each instantiation gets its own register allocation, 230-256 VGPRs; inlined into
// one kernel the four of them spilled.  Everything block-uniform is moved to SGPRs by hand: the descriptor is read through a pointer the
// compiler cannot prove uniform, and a non-uniform buffer descriptor turns every load into a waterfall loop.)
__device__ __forceinline__ int wg_sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T>
__device__ __forceinline__ T* wg_sgpr_ptr(T* q) {
  const unsigned long long v = (unsigned long long)q;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return (T*)(((unsigned long long)hi << 32) | lo);
}

template <int W>
__device__ __attribute__((noinline)) void wgrad_block_ring(const WgP* __restrict__ pp_, int bx_, int by_) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int R = W >= 16 ? 128 / W : 8, WH = W + 2, KT = R * W, NR = 2 * R + 4, NT = 512;
  constexpr int WSH = W == 64 ? 6 : (W == 32 ? 5 : (W == 16 ? 4 : 3));
  constexpr unsigned OOB = 0x80000000u;
  static_assert(KT == 128 || KT == 64, "tiles of 128 pixels (64 on the 8-wide map)");
  // the problem, in SGPRs
  struct { const bf16_t *a, *a2, *dy; float *dW, *db, *ws; int C1, B, H, Cin, Cout, tiles, tiles_per_blk, c_tiles, Cw, Nw, ws_stride, ws_dboff; } p;
  p.a = wg_sgpr_ptr(pp_->a); p.a2 = wg_sgpr_ptr(pp_->a2); p.dy = wg_sgpr_ptr(pp_->dy);
  p.dW = wg_sgpr_ptr(pp_->dW); p.db = wg_sgpr_ptr(pp_->db); p.ws = wg_sgpr_ptr(pp_->ws);
  p.C1 = wg_sgpr(pp_->C1); p.B = wg_sgpr(pp_->B); p.H = wg_sgpr(pp_->H); p.Cin = wg_sgpr(pp_->Cin); p.Cout = wg_sgpr(pp_->Cout);
  p.tiles = wg_sgpr(pp_->tiles); p.tiles_per_blk = wg_sgpr(pp_->tiles_per_blk); p.c_tiles = wg_sgpr(pp_->c_tiles);
  p.Cw = wg_sgpr(pp_->Cw); p.Nw = wg_sgpr(pp_->Nw); p.ws_stride = wg_sgpr(pp_->ws_stride); p.ws_dboff = wg_sgpr(pp_->ws_dboff);
  const int bx = wg_sgpr(bx_), by = wg_sgpr(by_);
  bf16_t* const Xs = reinterpret_cast<bf16_t*>(smem);          // [NR][WH] pixels
  bf16_t* const Ds = Xs + NR * WH * PITCH;                      // [2][KT] pixels
  const int tid = threadIdx.x, lane = tid & 63, wv = wg_sgpr(tid >> 6);
  const int kh = wv >> 2, cq = wv & 3;
  const int c0 = (bx % p.c_tiles) * 64, n0 = (bx / p.c_tiles) * 64;
  const int tiles_per_img = p.H / R;
  const int t_beg = by * p.tiles_per_blk, t_end = min(p.tiles, t_beg + p.tiles_per_blk);
  if (t_beg >= t_end) return;

  f32x4_t acc[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[t][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // ---- staging: a cin tile lies entirely in `a` or in `a2`
  const int v8 = tid & 7;
  const bf16_t* asrc = p.a;
  int apitch = p.Cin, acol = c0, alim = p.Cin;
  if (p.a2) {
    if (c0 < p.C1) { apitch = p.C1; alim = p.C1; }
    else { asrc = p.a2; apitch = p.Cin - p.C1; acol = c0 - p.C1; alim = apitch; }
  }
  const auto xrs = __builtin_amdgcn_make_buffer_rsrc((void*)asrc, 0, (int)((unsigned)p.B * p.H * W * apitch * 2u), 0x00020000);
  const auto drs = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (int)((unsigned)p.B * p.H * W * p.Cout * 2u), 0x00020000);
  // W = 64: the two top rows of an image's first window are fetched on the spot when that tile is stored (one exposed round trip
  // per 32 tiles) so that the tiles in flight hold 3 vectors per thread instead of 5 -- the registers the accumulators need
  constexpr bool SPLIT = false;      // (W == 64 measured: the branch around the on-the-spot fetch makes hipcc drain vmcnt at the top of every iteration)
  constexpr int XV = ((SPLIT ? R : R + 2) * WH * 8 + NT - 1) / NT, XVE = (2 * WH * 8 + NT - 1) / NT, DV = KT * 8 / NT;
  const bool cvalid = acol + v8 * 8 < alim, nvalid = n0 + v8 * 8 < p.Cout;
  const unsigned rowbytes = (unsigned)(W * apitch * 2);
  const bool do_db = p.db != nullptr && c0 == 0;
  float dbs[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) dbs[e] = 0.f;

  wg_u32x4_t xr[XV], dr[DV];
  // tile t: rows f .. R + 1 of its (R + 2)-row window are new (f = 0 at the top of an image / of the block, else 2)
  auto xload = [&](int k, int f, int b, int oy0, bool live) {
    const int pix = (tid + k * NT) >> 3, xrow = pix / WH, xcol = pix - xrow * WH;
    const int wr = f + xrow, y = oy0 - 1 + wr, ix = xcol - 1;              // window row, image row, image column
    const bool ok = live && cvalid && wr < R + 2 && (unsigned)y < (unsigned)p.H && (unsigned)ix < (unsigned)W;
    const unsigned ibase = (unsigned)(b * p.H) * rowbytes + (unsigned)((acol + v8 * 8) * 2);
    return __builtin_amdgcn_raw_buffer_load_b128(xrs, ok ? ibase + (unsigned)y * rowbytes + (unsigned)(ix * apitch * 2) : OOB, 0, 0);
  };
  auto xstore = [&](int k, int f, int rbase, const wg_u32x4_t& v) {
    const int pix = (tid + k * NT) >> 3, xrow = pix / WH, xcol = pix - xrow * WH;
    const int wr = f + xrow;
    if (wr < (SPLIT && f == 0 ? 2 : R + 2)) {
      int slot = rbase + wr;
      slot -= slot >= NR ? NR : 0;
      *reinterpret_cast<wg_u32x4_t*>(Xs + (slot * WH + xcol) * PITCH + v8 * 8) = v;
    }
  };
  // per-tile scalars of the staging (block-uniform)
  struct TileCtx { int b, oy0, f; bool live, fresh; };
  auto tile_ctx = [&](int t) {
    TileCtx c;
    c.live = t < t_end;
    const int tt = c.live ? t : t_beg;
    c.b = tt / tiles_per_img;
    const int ty = tt - c.b * tiles_per_img;
    c.oy0 = ty * R;
    c.fresh = tt == t_beg || ty == 0;
    c.f = (SPLIT || !c.fresh) ? 2 : 0;
    return c;
  };
  auto dload = [&](int k, const TileCtx& c) {
    const unsigned dbase = (unsigned)((c.b * p.H + c.oy0) * W) * (unsigned)(p.Cout * 2) + (unsigned)((n0 + v8 * 8) * 2);
    return __builtin_amdgcn_raw_buffer_load_b128(drs, (c.live && nvalid) ? dbase + (unsigned)(((tid + k * NT) >> 3) * p.Cout * 2) : OOB, 0, 0);
  };
  auto dstore = [&](int k, int dbuf, const wg_u32x4_t& v) {
    *reinterpret_cast<wg_u32x4_t*>(Ds + (dbuf * KT + ((tid + k * NT) >> 3)) * PITCH + v8 * 8) = v;
    if (do_db) {
      const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        dbs[2 * e] += __uint_as_float(w4[e] << 16);
        dbs[2 * e + 1] += __uint_as_float(w4[e] & 0xffff0000u);
      }
    }
  };
  auto issue = [&](const TileCtx& c) {
#pragma unroll
    for (int k = 0; k < XV; ++k) xr[k] = xload(k, c.f, c.b, c.oy0, c.live);
#pragma unroll
    for (int k = 0; k < DV; ++k) dr[k] = dload(k, c);
  };
  // the two top rows of an image's first window (W = 64): fetched on the spot, before the regular rows' registers are touched
  auto stash_top = [&](const TileCtx& c, int rbase) {
    if (SPLIT && c.fresh && c.live) {
      wg_u32x4_t xe[XVE];
#pragma unroll
      for (int k = 0; k < XVE; ++k) xe[k] = xload(k, 0, c.b, c.oy0, true);
#pragma unroll
      for (int k = 0; k < XVE; ++k) xstore(k, 0, rbase, xe[k]);
    }
  };
  // piece s of (store tile t + 1, issue tile t + 2): one staged vector leaves its register for LDS and the register is re-loaded at
  // once -- called between the taps of tile t's contraction, so the LDS stores and their address arithmetic issue in the MFMAs' shadow
  // instead of in a phase of their own (ablations: stores + load waits were 340 of 940 us, the MFMA work alone 525)
  auto piece = [&](int k, const TileCtx& c1, int rb1, int db1, const TileCtx& c2) {
    if (k < XV) {
      if (c1.live && !(IDF_WG_ABL & 2)) xstore(k, c1.f, rb1, xr[k]);
      if (!(IDF_WG_ABL & 1)) xr[k] = xload(k, c2.f, c2.b, c2.oy0, c2.live);
    } else if (k < XV + DV) {
      if (c1.live && !(IDF_WG_ABL & 2)) dstore(k - XV, db1, dr[k - XV]);
      if (!(IDF_WG_ABL & 1)) dr[k - XV] = dload(k - XV, c2);
    }
  };
  // transposed-read lane geometry: lane (g, q, pp) addresses pixel 4g + q of a 16-pixel run, channels 4pp .. 4pp + 3 of a 16-channel
  // fragment; k slot (g, j) of a 32-pixel step = pixel 4g + j (j < 4) / 16 + 4g + (j - 4)
  const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3, pl = 4 * g + q;
  const int lrow = W == 8 ? pl >> 3 : 0, lcol = W == 8 ? pl & 7 : pl;
  const int lane_x = lcol * PITCH + cq * 16 + 4 * pp, lane_d = pl * PITCH + 4 * pp;
  auto contract = [&](int rbase, int dbuf, const TileCtx& c1, int rb1, const TileCtx& c2) {
    constexpr int NP = XV + DV, NS = 9 * (KT / 64);          // staging pieces, tap slots of this wave's k-steps
    if (IDF_WG_ABL & 4) {
#pragma unroll
      for (int k = 0; k < NP; ++k) piece(k, c1, rb1, dbuf ^ 1, c2);
      return;
    }
    if (IDF_WG_ABL & 8) {
      const bf16x8_t f0 = mkfrag(tr_read(Ds + lane_d), tr_read(Ds + lane_d + 16)), f1 = mkfrag(tr_read(Xs + lane_x), tr_read(Xs + lane_x + 16));
#pragma unroll
      for (int i2 = 0; i2 < KT / 64; ++i2)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f0, f1, acc[t][i], 0, 0, 0);
          const int slot = i2 * 9 + t;
          if ((slot * NP) / NS != ((slot + 1) * NP) / NS) piece((slot * NP) / NS, c1, rb1, dbuf ^ 1, c2);
        }
      return;
    }
#pragma unroll
    for (int i2 = 0; i2 < KT / 64; ++i2) {
      const int ks = 32 * (2 * i2 + kh);
      const bf16_t* d0 = Ds + (dbuf * KT + ks) * PITCH + lane_d;
      const int rowA = ks >> WSH, colA = ks & (W - 1), rowB = (ks + 16) >> WSH, colB = (ks + 16) & (W - 1);
      const bf16_t *xA[3], *xB[3];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        int sA = rbase + rowA + ky + lrow, sB = rbase + rowB + ky + lrow;
        sA -= sA >= NR ? NR : 0;
        sB -= sB >= NR ? NR : 0;
        xA[ky] = Xs + (sA * WH + colA) * PITCH + lane_x;
        xB[ky] = Xs + (sB * WH + colB) * PITCH + lane_x;
      }
      bf16x8_t nf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) nf[i] = mkfrag(tr_read(d0 + i * 16), tr_read(d0 + 16 * PITCH + i * 16));
      // the next tap's fragment is in flight while this tap's four MFMAs issue (two waves per SIMD: nobody else hides an LDS round trip)
      bf16x8_t cf = mkfrag(tr_read(xA[0]), tr_read(xB[0]));
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        bf16x8_t cn = cf;
        if (t < 8) cn = mkfrag(tr_read(xA[(t + 1) / 3] + ((t + 1) % 3) * PITCH), tr_read(xB[(t + 1) / 3] + ((t + 1) % 3) * PITCH));
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(nf[i], cf, acc[t][i], 0, 0, 0);
        cf = cn;
        // NP staging pieces spread evenly over the NS tap slots
        const int slot = i2 * 9 + t;
        if ((slot * NP) / NS != ((slot + 1) * NP) / NS) piece((slot * NP) / NS, c1, rb1, dbuf ^ 1, c2);
      }
    }
  };
  auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

  int rb_c = 0;                                        // ring slot of the contracted tile's top window row
  {
    const TileCtx c0_ = tile_ctx(t_beg), c1_ = tile_ctx(t_beg + 1);
    issue(c0_);
    stash_top(c0_, rb_c);
#pragma unroll
    for (int k = 0; k < XV; ++k) xstore(k, c0_.f, rb_c, xr[k]);
#pragma unroll
    for (int k = 0; k < DV; ++k) dstore(k, 0, dr[k]);
    issue(c1_);
  }
  lds_barrier();
  for (int t = t_beg; t < t_end; ++t) {
    const int db_c = (t - t_beg) & 1;
    int rb_n = rb_c + (((t + 1) % tiles_per_img) == 0 ? R + 2 : R);       // the next tile's window: R rows down, or a fresh image behind this one
    rb_n -= rb_n >= NR ? NR : 0;
    const TileCtx c1_ = tile_ctx(t + 1), c2_ = tile_ctx(t + 2);
    stash_top(c1_, rb_n);
    contract(rb_c, db_c, c1_, rb_n, c2_);              // tile t contracted; tile t + 1 stored and tile t + 2 requested between its taps
    rb_c = rb_n;
    lds_barrier();
  }

  // ---- the two k halves meet: kh = 1 hands its accumulators over through LDS, one kernel row at a time
  float* const red = reinterpret_cast<float*>(smem);           // [4 cin quarters][12 fragments][64 lanes] float4 = 48 KB
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    if (kh) {
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4_t*>(red + ((cq * 12 + kx * 4 + i) * 64 + lane) * 4) = acc[ky * 3 + kx][i];
    }
    lds_barrier();
    if (!kh) {
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[ky * 3 + kx][i] += *reinterpret_cast<const f32x4_t*>(red + ((cq * 12 + kx * 4 + i) * 64 + lane) * 4);
    }
    lds_barrier();
  }
  if (!kh) {
    // D: row = n (4 per lane), col = c (lane & 15)
    const int c = c0 + cq * 16 + (lane & 15);
    if (c < p.Cw) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int n = n0 + i * 16 + (lane >> 4) * 4 + r;
            if (n < p.Nw) {
              const size_t idx = ((size_t)n * 9 + t) * p.Cw + c;
              if (p.ws) gstore(p.ws + (size_t)by * p.ws_stride + idx, acc[t][i][r]);
              else gatomic_add(p.dW + idx, acc[t][i][r]);
            }
          }
    }
  }
  if (do_db) {
#pragma unroll
    for (int e = 0; e < 8; ++e) red[(tid >> 3) * 64 + v8 * 8 + e] = dbs[e];
    lds_barrier();
    if (tid < 64 && n0 + tid < p.Nw) {
      float sacc = 0.f;
      for (int k = 0; k < NT / 8; ++k) sacc += red[k * 64 + tid];
      if (p.ws) gstore(p.ws + (size_t)by * p.ws_stride + p.ws_dboff + n0 + tid, sacc);
      else gatomic_add(p.db + n0 + tid, sacc);
    }
  }
}

template <int KW, int MODE>
__global__ __launch_bounds__(256, 3) void conv_wgrad_tr_bf16(const WgP p) {
  wgrad_block<KW, MODE>(p, blockIdx.x, blockIdx.y);
}

// One launch for many convolutions (all weight gradients of a backward pass, deferred to its end):
// blocks [blk0, blk0 + gx*gy) of the 1-D grid belong to table entry i.  Small problems no longer pay
// a launch each nor leave CUs idle, one problem's atomic tail overlaps its neighbours' loads, and each
// problem can use fewer pixel splits (= fewer atomic bytes) because the others fill the chip.
struct WgDesc {
  WgP p;
  int blk0, gx, gy, xcd;   // xcd = 1: blocks [blk0, blk0 + gx * roundup8(gy)) in XCD-major order
  int red_blk0, nslab;     // two-stage accumulation: first block of this entry in idf_wgrad_reduce_batched's grid, slabs to add
};

// `total` work items (the 1-D block ids of the table) over a grid of gridDim.x <= total blocks: block b takes items b, b + grid, ...
// A grid smaller than `total` keeps the launch on part of the chip (idf_conv_wgrad_bf16_batched_capped: weight gradients running
// on a side stream beside the data-gradient chain must leave CUs free for the chain's launches).
template <int KW, int MODE, int ROWS>
__device__ __forceinline__ void wgrad_item(const WgDesc* __restrict__ tab, int n, int bid) {
  int lo = 0, hi = n;
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (tab[mid].blk0 <= bid) lo = mid; else hi = mid;
  }
  const WgDesc* d = tab + lo;
  const WgP p = d->p;
  const int local = bid - d->blk0, gx = d->gx;
  if (d->xcd) {
    // workgroups go round-robin over the 8 XCDs: give XCD x the pixel splits x, x + 8, ... and ALL the
    // (kernel row, cout tile, cin tile) blocks of each, so the gx blocks that read the same a / dy tiles
    // share one L2 instead of fetching them over the fabric gx times
    const int x = local & 7, j = local >> 3;
    const int by = x + 8 * (j / gx);
    if (by >= d->gy) return;
    wgrad_block<KW, MODE, ROWS>(p, j % gx, by);
  } else {
    if (local >= gx * d->gy) return;      // alignment padding
    wgrad_block<KW, MODE, ROWS>(p, local % gx, local / gx);
  }
}

template <int KW, int MODE>
__global__ __launch_bounds__(256, 3) void conv_wgrad_tr_bf16_batched(const WgDesc* __restrict__ tab, int n, int total) {
  for (int bid = blockIdx.x; bid < total; bid += gridDim.x) {
    if (bid != (int)blockIdx.x) __syncthreads();          // the previous item's last LDS reads
    wgrad_item<KW, MODE, 1>(tab, n, bid);
  }
}

// the sub-pixel form of the UpSample class (wgrad_block_upsub: 8 accumulator sets, ~250 registers -- its own kernel, so the other
// classes keep their three-blocks-per-CU budgets)
__global__ __launch_bounds__(256) void conv_wgrad_up_sub_batched(const WgDesc* __restrict__ tab, int n, int total) {
  for (int bid = blockIdx.x; bid < total; bid += gridDim.x) {
    if (bid != (int)blockIdx.x) __syncthreads();
    int lo = 0, hi = n;
    while (hi - lo > 1) {
      int mid = (lo + hi) >> 1;
      if (tab[mid].blk0 <= bid) lo = mid; else hi = mid;
    }
    const WgDesc* d = tab + lo;
    const WgP p = d->p;
    const int local = bid - d->blk0, gx = d->gx;
    if (d->xcd) {
      const int x = local & 7, j = local >> 3;
      const int by = x + 8 * (j / gx);
      if (by < d->gy) wgrad_block_upsub(p, j % gx, by);
    } else if (local < gx * d->gy) wgrad_block_upsub(p, local % gx, local / gx);
  }
}

// the shared-tile form of the stride-1 3x3 class (768 threads, one block per CU)
__global__ __launch_bounds__(768) void conv_wgrad_tr_bf16_batched_kr3(const WgDesc* __restrict__ tab, int n, int total) {
  for (int bid = blockIdx.x; bid < total; bid += gridDim.x) {
    if (bid != (int)blockIdx.x) __syncthreads();
    wgrad_item<3, 0, 3>(tab, n, bid);
  }
}

// the row-ring form of the stride-1 3x3 class (512 threads: 8 waves = k half x cin quarter, one or two blocks per CU)
__global__ __launch_bounds__(512) void conv_wgrad_ring_batched(const WgDesc* __restrict__ tab, int n, int total) {
  for (int bid = blockIdx.x; bid < total; bid += gridDim.x) {
    if (bid != (int)blockIdx.x) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    int lo = 0, hi = n;
    while (hi - lo > 1) {
      int mid = (lo + hi) >> 1;
      if (tab[mid].blk0 <= bid) lo = mid; else hi = mid;
    }
    const WgDesc* d = tab + lo;
    const int local = bid - d->blk0, gx = d->gx;
    int bx, by;
    if (d->xcd) {
      const int x = local & 7, j = local >> 3;
      by = x + 8 * (j / gx); bx = j % gx;
    } else { bx = local % gx; by = local / gx; }
    if (by >= d->gy) continue;
    switch (wg_sgpr(d->p.W)) {
      case 64: wgrad_block_ring<64>(&d->p, bx, by); break;
      case 32: wgrad_block_ring<32>(&d->p, bx, by); break;
      case 16: wgrad_block_ring<16>(&d->p, bx, by); break;
      default: wgrad_block_ring<8>(&d->p, bx, by); break;
    }
  }
}

// Second stage of the deterministic accumulation: dW[i] += sum over this entry's slabs of ws[slab][i], in slab order (and db likewise):
// a fixed order of fp32 additions, so two runs of a step give the same bits (the reference's convolution_backward under
// --deterministic: utils.py:64-71).  One workgroup = 1024 consecutive floats of one entry's slab; float4 slab loads.
constexpr int WG_RED = 1024;
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WgDesc* __restrict__ tab, int n) {
  const int bid = blockIdx.x;
  int lo = 0, hi = n;
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (tab[mid].red_blk0 <= bid) lo = mid; else hi = mid;
  }
  const WgDesc* d = tab + lo;
  const WgP& p = d->p;
  if (!p.ws) return;
  const int i0 = (bid - d->red_blk0) * WG_RED + threadIdx.x * 4;
  if (i0 >= p.ws_stride) return;
  f32x4_t s = {0.f, 0.f, 0.f, 0.f};
  const float* src = p.ws + i0;
  for (int k = 0; k < d->nslab; ++k) {
    const f32x4_t v = *(const __attribute__((address_space(1))) f32x4_t*)(unsigned long long)(src + (size_t)k * p.ws_stride);
    s += v;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int i = i0 + e;
    if (i < p.ws_dboff) p.dW[i] += s[e];
    else if (p.db && i < p.ws_dboff + p.Nw) p.db[i - p.ws_dboff] += s[e];
  }
}

// IDF_WGRAD_KR3 (default 1): the batched stride-1 3x3 class runs in the shared-tile form
#define g_kr3 (idf_knobs().wgrad_kr3)

// IDF_WGRAD_RING (default 1): stride-1 3x3 problems on 64 / 32 / 16 / 8-wide maps run in the row-ring form (wgrad_block_ring)
bool ring_fits(int B, int H, int W, int Cin, int Cout, int taps, int mode) {
  if (!idf_knobs().wgrad_ring || taps != 9 || mode != 0 || B <= 0 || (Cin % 8) || (Cout % 8)) return false;
  if (W != 64 && W != 32 && W != 16 && W != 8) return false;
  const int R = W >= 16 ? 128 / W : 8;
  if (H < R || (H % R)) return false;
  return (long)B * H * W * (Cin > Cout ? Cin : Cout) * 2 < (1L << 31);      // 32-bit buffer offsets
}

// Shape checks + tiling plan of one problem.  target_blocks <= 0: the stand-alone heuristic.
int wg_plan(WgP& p, int& gx, int& gy, size_t& lds, const void* a, const void* dy, float* dW, float* db, int B, int H,
            int W, int Cin, int Cout, int taps, int mode, int target_blocks, const void* a2 = nullptr, int C1 = 0,
            int tiles_per_block = 0, int min_blocks = 0, int Cin_w = 0, int Cout_w = 0, bool kr3 = false, bool want_upsub = false,
            bool want_ring = false) {
  if ((taps != 9 && taps != 1) || (Cin % 8) || (Cout % 8) || H <= 0 || W < 4 || (W & (W - 1)) || mode < 0 ||
      mode > 2 || (mode && taps != 9) || (mode == 2 && ((H | W) & 1)))
    IDF_FAIL(IDF_ERR_UNSUPPORTED, "wgrad_bf16: shape B%d H%d W%d Cin%d Cout%d taps%d mode%d not covered", B, H, W, Cin,
             Cout, taps, mode);
  // MODE 2 in the sub-pixel form (wgrad_block_upsub) where the low-resolution map tiles: W / 2 in {8, 16, 32}
  const int Wl_ = W / 2, Hl_ = H / 2;
  int RLs = Wl_ > 0 ? 128 / Wl_ : 0;
  if (RLs > Hl_) RLs = Hl_;
  const bool upsub = want_upsub && mode == 2 && taps == 9 && Wl_ >= 8 && Wl_ <= 32 && RLs >= 1 && (Hl_ % RLs) == 0 && ((RLs * Wl_) % 32) == 0 &&
                     (RLs + 2) * (Wl_ + 2) * 8 <= 7 * 256;
  if (want_upsub && !upsub) IDF_FAIL(IDF_ERR_UNSUPPORTED, "wgrad_bf16: H%d W%d not covered by the sub-pixel form", H, W);
  int R = (mode == 1 ? 64 : 128) / W;
  if (R > H) R = H;
  if (R < 1 || (H % R) || ((R * W) % 32) || R * ((mode == 1 ? 2 : 1) * W + 2) > 160 || R * W > 128)
    IDF_FAIL(IDF_ERR_UNSUPPORTED, "wgrad_bf16: H%d W%d not tileable", H, W);
  if (a2 && (C1 <= 0 || C1 >= Cin || (C1 % 64) || mode != 0))
    IDF_FAIL(IDF_ERR_UNSUPPORTED, "wgrad_bf16: two-source input needs C1 %% 64 == 0 (C1 %d) and stride 1", C1);
  p.a = (const bf16_t*)a; p.dy = (const bf16_t*)dy; p.dW = dW; p.db = db;
  p.ws = nullptr; p.ws_stride = 0; p.ws_dboff = 0;
  p.a2 = (const bf16_t*)a2; p.C1 = a2 ? C1 : Cin;
  if (Cin_w < 0 || Cin_w > Cin || Cout_w < 0 || Cout_w > Cout)
    IDF_FAIL(IDF_ERR_BADARG, "wgrad_bf16: gradient extents %d x %d exceed the operands' %d x %d", Cout_w, Cin_w, Cout, Cin);
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.R = upsub ? RLs : R;
  p.upsub = upsub ? 1 : 0;
  p.Cw = Cin_w > 0 ? Cin_w : Cin; p.Nw = Cout_w > 0 ? Cout_w : Cout;
  p.Hs = mode == 1 ? 2 * H : (mode == 2 ? H / 2 : H);
  p.Ws = mode == 1 ? 2 * W : (mode == 2 ? W / 2 : W);
  p.tiles = upsub ? B * (Hl_ / RLs) : B * (H / R);
  {
    int ws = 0;
    while ((1 << ws) < (upsub ? Wl_ : W)) ++ws;
    p.wshift = ws;
    const int WHh = upsub ? Wl_ + 2 : (mode == 1 ? 2 : 1) * W + 2 * (taps == 9 ? 1 : 0), np = upsub ? (RLs + 2) * WHh : (R + (kr3 ? 2 : 0)) * WHh;
    unsigned m = 65536u / (unsigned)WHh + 1u;
    for (int i = 0; i < np; ++i)
      if ((((unsigned)i * m) >> 16) != (unsigned)(i / WHh)) IDF_FAIL(IDF_ERR_UNSUPPORTED, "wgrad_bf16: tile of %d pixels not addressable", np);
    p.wh_magic = m;
  }
  p.c_tiles = idf_cdiv(Cin, 64);
  p.n_tiles = idf_cdiv(Cout, 64);
  // the row-ring form of the stride-1 3x3 class (wgrad_block_ring; the caller asked: idf_wgrad_ring_ok)
  p.ring = want_ring ? 1 : 0;
  if (want_ring && !ring_fits(B, H, W, Cin, Cout, taps, mode))
    IDF_FAIL(IDF_ERR_UNSUPPORTED, "wgrad_bf16: B%d H%d W%d Cin%d Cout%d not covered by the row-ring form", B, H, W, Cin, Cout);
  const int kh = taps == 9 ? 3 : 1;
  gx = p.c_tiles * p.n_tiles * (upsub ? 2 : ((kr3 || want_ring) ? 1 : kh));
  if (target_blocks <= 0) {
    // grid size trades chip occupancy against fp32-atomic bytes (= blocks/gx * |dW|): measured optimum on
    // MI355X is ~1 block per CU for small weight tiles and 2-3 per CU once a block does enough MFMA work
    // per atomic byte (profiles/r01_wgrad_grid_sweep.txt)
    static const int forced = 0;
    const long M = (long)B * H * W, cc = (long)Cin * Cout;
    target_blocks = 512;
    if (cc <= 64 * 64 || (M <= 32768 && cc <= 128 * 128)) target_blocks = 256;
    else if (M >= 131072 && cc >= 128 * 128) target_blocks = 768;
    if (forced > 0) target_blocks = forced;
  }
  int split = idf_cdiv(target_blocks, gx);
  if (tiles_per_block > 0) {
    // batched launch: every problem shares the chip, so blocks are sized by WORK (pixel tiles per block) instead of a
    // per-problem block count -- the fp32-atomic bytes of a problem are split * |dW|, and only the large-image problems
    // (small dW, many pixels) need many splits
    split = (p.tiles + tiles_per_block / 2) / tiles_per_block;
    if (gx * split < min_blocks) split = idf_cdiv(min_blocks, gx);
  }
  if (split > p.tiles) split = p.tiles;
  if (split < 1) split = 1;
  p.tiles_per_blk = idf_cdiv(p.tiles > 0 ? p.tiles : 1, split);
  gy = idf_cdiv(p.tiles, p.tiles_per_blk);
  const int sx = mode == 1 ? 2 : 1;
  lds = ((size_t)(R + (kr3 ? 2 : 0)) * (sx * W + 2 * (kh / 2)) + (size_t)R * W) * PITCHB * (kr3 ? 2 : 1);     // kr3: two LDS tiles
  if (p.ring) lds = ((size_t)(2 * R + 4) * (W + 2) + (size_t)2 * R * W) * PITCHB;                                  // ring rows + two dy tiles
  if (upsub) lds = ((size_t)(RLs + 2) * (Wl_ + 2) + (size_t)2 * RLs * Wl_) * PITCHB;
  const size_t red = p.ring ? (size_t)4 * 12 * 64 * 16 : (size_t)(kr3 ? 96 : 32) * 64 * sizeof(float);
  if (lds < red) lds = red;
  return IDF_OK;
}

// the shared-tile form stages (R + 2)(W + 2) input pixels with 3 vectors and R * W dy pixels with 2 vectors per thread of a
// 768-thread block, twice (two LDS tiles): shapes beyond that (W = 128) stay with the row-split kernel
bool kr3_fits(int H, int W) {
  int R = 128 / W;
  if (R > H) R = H;
  if (R < 1 || (H % R)) return false;
  const size_t lds = ((size_t)(R + 2) * (W + 2) + (size_t)R * W) * PITCHB * 2;
  return (R + 2) * (W + 2) * 8 <= 3 * 768 && R * W * 8 <= 2 * 768 && lds <= 160 * 1024;
}

}  // namespace

// 1 when the batched stride-1 3x3 weight gradient of an H x W map runs in the shared-tile form (kr3); the host keeps problems
// that do not fit in a class of their own (mode | IDF_WGRAD_ROWSPLIT in idf_wgrad_desc_fill / idf_conv_wgrad_bf16_batched)
// 1 when the UpSample weight gradient of an H x W output map runs in the sub-pixel form (mode | IDF_WGRAD_UPSUB in
// idf_wgrad_desc_fill / idf_conv_wgrad_bf16_batched): W / 2 in {8, 16, 32}, whole 128-pixel (64 at 8x8) low-resolution tiles
extern "C" int idf_wgrad_upsub_ok(int H, int W) {
  static const int on = 1;
  if (!on || H <= 0 || W < 16 || (W & (W - 1)) || ((H | W) & 1)) return 0;
  const int Wl = W / 2, Hl = H / 2;
  int RL = 128 / Wl;
  if (RL > Hl) RL = Hl;
  return (Wl >= 8 && Wl <= 32 && RL >= 1 && (Hl % RL) == 0 && ((RL * Wl) % 32) == 0 && (RL + 2) * (Wl + 2) * 8 <= 7 * 256) ? 1 : 0;
}

// 1 when the batched stride-1 3x3 weight gradient of this shape runs in the row-ring form (round 6): the host puts such problems
// in a class of their own (mode | IDF_WGRAD_RING = 64 in idf_wgrad_desc_fill / idf_conv_wgrad_bf16_batched)
extern "C" int idf_wgrad_ring_ok(int B, int H, int W, int Cin, int Cout) { return ring_fits(B, H, W, Cin, Cout, 9, 0) ? 1 : 0; }
extern "C" int idf_wgrad_kr3_ok(int H, int W) { return (g_kr3 && H > 0 && W >= 4 && !(W & (W - 1)) && kr3_fits(H, W)) ? 1 : 0; }

// taps = 9 (3x3, pad 1) or 1 (1x1); mode 0 stride 1, 1 stride 2, 2 nearest-x2-upsampled input
// (3x3 only).  H, W are the OUTPUT (dy) dims.  Returns IDF_ERR_UNSUPPORTED (without touching
// outputs) for shapes this kernel does not cover; the caller then uses idf_conv2d_wgrad.
// dW / db are zeroed inside.
extern "C" int idf_conv_wgrad_bf16(const void* a, const void* dy, float* dW, float* db, int B, int H, int W,
                                   int Cin, int Cout, int Cin_w, int Cout_w, int taps, int mode, int accumulate,
                                   void* stream) {
  WgP p;
  int gx, gy;
  size_t lds;
  int rc = wg_plan(p, gx, gy, lds, a, dy, dW, db, B, H, W, Cin, Cout, taps, mode, 0, nullptr, 0, 0, 0, Cin_w, Cout_w);
  if (rc != IDF_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  const size_t nW = (size_t)p.Nw * taps * p.Cw;
  hipError_t e = hipSuccess;
  if (accumulate) {}                     // dW / db already hold zeros or a running sum (gradient arena)
  else if (db == dW + nW) e = idf_zero_f32(dW, nW + p.Nw, st);   // caller packed dW | db: one launch
  else {
    e = idf_zero_f32(dW, nW, st);
    if (e == hipSuccess && db) e = idf_zero_f32(db, (size_t)p.Nw, st);
  }
  if (e != hipSuccess) IDF_FAIL(IDF_ERR_HIP, "wgrad_bf16: zero fill failed: %s", hipGetErrorString(e));
  if (B == 0) return IDF_OK;
  dim3 g(gx, gy);
  if (taps == 1) hipLaunchKernelGGL((conv_wgrad_tr_bf16<1, 0>), g, dim3(256), lds, st, p);
  else if (mode == 0) hipLaunchKernelGGL((conv_wgrad_tr_bf16<3, 0>), g, dim3(256), lds, st, p);
  else if (mode == 1) hipLaunchKernelGGL((conv_wgrad_tr_bf16<3, 1>), g, dim3(256), lds, st, p);
  else hipLaunchKernelGGL((conv_wgrad_tr_bf16<3, 2>), g, dim3(256), lds, st, p);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}

// ---- batched form: host fills a table entry per problem, uploads the table, one launch per
// (taps, mode) class.  Accumulating only (dW / db pre-zeroed: the gradient arena).
extern "C" int idf_wgrad_desc_bytes(void) { return (int)sizeof(WgDesc); }

extern "C" int idf_wgrad_desc_fill(void* host_table, int index, const void* a, const void* a2, int C1, const void* dy,
                                   float* dW, float* db, int B, int H, int W, int Cin, int Cout, int Cin_w, int Cout_w,
                                   int taps, int mode, int target_blocks, int blk0, int* blocks_out, int* lds_out, float* ws,
                                   int red_blk0, long* ws_floats_out, int* red_blocks_out) {
  if (!host_table || index < 0 || !blocks_out || !lds_out) IDF_FAIL(IDF_ERR_BADARG, "wgrad_desc_fill: null argument");
  if (B <= 0) IDF_FAIL(IDF_ERR_BADARG, "wgrad_desc_fill: empty batch");
  WgDesc d;
  memset(&d, 0, sizeof(d));
  size_t lds;
  static const int forced = 0;
  if (forced > 0) target_blocks = forced;
  static const int tpb = 16;      // re-swept for the 1x1 / stride-2 / up-sampling classes: profiles/r03_wgrad_tpb_sweep.txt
  static const int minb = 96;
  static const int tpb_up = 16;   // sub-pixel UpSample class: tiles are 128 LOW-resolution pixels
  // pixel tiles per block of the shared-tile 3x3 class: 128 (round 4; 64 before) halves the pixel splits of the 64x64 / 32x32 problems
  // at a slightly shorter step (9.088 -> 9.065 ms; 192 / 256: the same; profiles/r04_wgrad_tpb3.txt).  The launch's atomic bytes
  // barely move (291 -> 281 MB by the WRITE_SIZE counter): they are the 16x16 / 8x8 problems' (590 KB - 1.2 MB of dW each, split by
  // the 16-block minimum, IDF_WGRAD_MINB3), not the big maps' (147 KB each)
  static const int tpb3 = idf_knobs().wgrad_tpb3;
  static const int minb3 = 16;
  const bool rowsplit = (mode & 16) != 0;         // IDF_WGRAD_ROWSPLIT: the caller keeps this problem out of the shared-tile class
  const bool upsub = (mode & 32) != 0;            // IDF_WGRAD_UPSUB: the UpSample class in its sub-pixel form
  const bool ring = (mode & 64) != 0;             // IDF_WGRAD_RING: the stride-1 3x3 class in its row-ring form
  mode &= 15;
  const bool kr3 = g_kr3 && taps == 9 && mode == 0 && !rowsplit && !ring;
  if (kr3 && !kr3_fits(H, W))
    IDF_FAIL(IDF_ERR_UNSUPPORTED, "wgrad_desc_fill: H%d W%d does not fit the shared-tile form (class it with IDF_WGRAD_ROWSPLIT)", H, W);
  int rc = wg_plan(d.p, d.gx, d.gy, lds, a, dy, dW, db, B, H, W, Cin, Cout, taps, mode, target_blocks > 0 ? target_blocks : 96,
                   a2, C1, target_blocks > 0 ? 0 : ((kr3 || ring) ? tpb3 : (upsub ? tpb_up : tpb)), (kr3 || ring) ? minb3 : minb, Cin_w, Cout_w, kr3, upsub, ring);
  if (rc != IDF_OK) return rc;
  d.blk0 = blk0;
  {
    // two-stage accumulation: one slab per pixel split (two per split in the sub-pixel UpSample form: one per row parity)
    const long nW = (long)d.p.Nw * taps * d.p.Cw;
    d.p.ws_dboff = (int)nW;
    d.p.ws_stride = (int)((nW + d.p.Nw + 3) / 4 * 4);
    d.nslab = d.gy * (d.p.upsub ? 2 : 1);
    d.p.ws = ws;
    d.red_blk0 = red_blk0;
    if (nW + d.p.Nw >= (1L << 31)) IDF_FAIL(IDF_ERR_UNSUPPORTED, "wgrad_desc_fill: gradient too large");
    if (ws_floats_out) *ws_floats_out = (long)d.nslab * d.p.ws_stride;
    if (red_blocks_out) *red_blocks_out = (d.p.ws_stride + WG_RED - 1) / WG_RED;
  }
  static const int xcd = 1;
  d.xcd = (xcd && (blk0 % 8) == 0 && d.gy >= 8) ? 1 : 0;
  memcpy((char*)host_table + (size_t)index * sizeof(WgDesc), &d, sizeof(d));
  *blocks_out = d.xcd ? d.gx * ((d.gy + 7) / 8 * 8) : d.gx * d.gy;
  if (!d.xcd && (*blocks_out % 8)) *blocks_out += 8 - *blocks_out % 8;     // keep later entries 8-aligned
  *lds_out = (int)lds;
  return IDF_OK;
}

extern "C" int idf_conv_wgrad_bf16_batched(const void* dev_table, int n, int total_blocks, int lds_bytes, int taps,
                                           int mode, void* stream) {
  if (n <= 0 || total_blocks <= 0) return IDF_OK;
  const bool rowsplit = (mode & 16) != 0, upsub = (mode & 32) != 0, ring = (mode & 64) != 0;
  mode &= 15;
  if (!dev_table || (taps != 9 && taps != 1) || mode < 0 || mode > 2 || (mode && taps != 9) || (ring && (taps != 9 || mode)))
    IDF_FAIL(IDF_ERR_BADARG, "wgrad_bf16_batched: bad arguments (taps %d mode %d)", taps, mode);
  hipStream_t st = (hipStream_t)stream;
  const WgDesc* tab = (const WgDesc*)dev_table;
  const bool kr3 = taps == 9 && mode == 0 && g_kr3 && !rowsplit && !ring;
  const int grid = total_blocks;                     // one workgroup per work item
  dim3 g(grid);
  if (ring) {
    static IdfLdsGrant grant;
    if (hipError_t e = idf_ensure_lds((const void*)conv_wgrad_ring_batched, (size_t)lds_bytes, grant); e != hipSuccess)
      IDF_FAIL(IDF_ERR_HIP, "wgrad_bf16_batched: %d bytes of LDS refused: %s", lds_bytes, hipGetErrorString(e));
    hipLaunchKernelGGL(conv_wgrad_ring_batched, g, dim3(512), lds_bytes, st, tab, n, total_blocks);
  }
  else if (taps == 1) hipLaunchKernelGGL((conv_wgrad_tr_bf16_batched<1, 0>), g, dim3(256), lds_bytes, st, tab, n, total_blocks);
  else if (kr3) {
    static IdfLdsGrant grant;
    if (hipError_t e = idf_ensure_lds((const void*)conv_wgrad_tr_bf16_batched_kr3, (size_t)lds_bytes, grant); e != hipSuccess)
      IDF_FAIL(IDF_ERR_HIP, "wgrad_bf16_batched: %d bytes of LDS refused: %s", lds_bytes, hipGetErrorString(e));
    hipLaunchKernelGGL(conv_wgrad_tr_bf16_batched_kr3, g, dim3(768), lds_bytes, st, tab, n, total_blocks);
  }
  else if (mode == 0) hipLaunchKernelGGL((conv_wgrad_tr_bf16_batched<3, 0>), g, dim3(256), lds_bytes, st, tab, n, total_blocks);
  else if (mode == 1) hipLaunchKernelGGL((conv_wgrad_tr_bf16_batched<3, 1>), g, dim3(256), lds_bytes, st, tab, n, total_blocks);
  else if (upsub) {
    static IdfLdsGrant grant;
    if (hipError_t e = idf_ensure_lds((const void*)conv_wgrad_up_sub_batched, (size_t)lds_bytes, grant); e != hipSuccess)
      IDF_FAIL(IDF_ERR_HIP, "wgrad_bf16_batched: %d bytes of LDS refused: %s", lds_bytes, hipGetErrorString(e));
    hipLaunchKernelGGL(conv_wgrad_up_sub_batched, g, dim3(256), lds_bytes, st, tab, n, total_blocks);
  }
  else hipLaunchKernelGGL((conv_wgrad_tr_bf16_batched<3, 2>), g, dim3(256), lds_bytes, st, tab, n, total_blocks);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}

// Second stage of the batched weight gradients when idf_wgrad_desc_fill was given workspaces: adds every entry's slabs to its dW / db
// in slab order (deterministic).  dev_table: ALL entries of the flush (every class), n of them; total_red_blocks = sum of the
// red_blocks_out values.  Entries filled with ws = NULL (fp32 atomics) are skipped.
extern "C" int idf_wgrad_reduce_batched(const void* dev_table, int n, int total_red_blocks, void* stream) {
  if (n <= 0 || total_red_blocks <= 0) return IDF_OK;
  if (!dev_table) IDF_FAIL(IDF_ERR_BADARG, "wgrad_reduce_batched: null table");
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(total_red_blocks), dim3(256), 0, (hipStream_t)stream, (const WgDesc*)dev_table, n);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}
