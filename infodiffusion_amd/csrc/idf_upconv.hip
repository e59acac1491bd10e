// UpSample (modules.py:78-93: nearest x2, then conv3x3 pad 1) as FOUR 2x2 convolutions on the low-resolution input -- the
// sub-pixel form of the same sum.  With oy = 2 y + py the three kernel rows of output row oy read the up-sampled rows
// oy - 1, oy, oy + 1, i.e. the low-resolution rows
//     py = 0:  y - 1 (kernel row 0),  y (kernel rows 1 + 2)          py = 1:  y (kernel rows 0 + 1),  y + 1 (kernel row 2)
// and the same along x: each output parity (py, px) is a 2x2-tap conv whose weights are sums of the 3x3 weights
//     W'[py][px][ty][tx] = sum_{ky in S(py, ty)} sum_{kx in S(px, tx)} W[ky][kx],   S(0,0) = {0}, S(0,1) = {1,2}, S(1,0) = {0,1}, S(1,1) = {2}
// reading low-resolution pixel (y + ty + py - 1, x + tx + px - 1); the zero padding of the low-resolution image is exactly the zero
// padding of the up-sampled one.  16 MFMA tap-products per four outputs instead of 36: the conv that held the nearest-x2 image in
// LDS and ran all nine taps over it (conv3x3_halo_bf16 MODE 2) spent 572 us per DDIM evaluation at B = 256 on three such layers.
// The sums are formed in fp32 from the master weights and rounded to bf16 once (the host packs them fragment-major: 1 KB per
// wave instruction, straight into registers; upconv_pack_kernel re-packs them with the other weight shadows).  Forward and data
// gradient (upconv_dgrad_bf16_kernel below); the weight gradient in the same form is idf_wgrad.hip's wgrad_block_upsub.
//
// One 512-thread workgroup = 256 output pixels (R rows x W columns of one image) x 64 couts.  Wave w owns parity w & 3 and the
// cout half w >> 2: 64 pixels of its parity x 32 couts, 4 x 2 accumulator tiles; per 32-channel chunk 4 taps x 8 MFMAs.  The
// low-resolution halo tile of a chunk -- (R / 2 + 2) x (W / 2 + 2) pixels, pixel pitch 96 B (conflict-free ds_read_b128) -- is
// double-buffered: one barrier per chunk.  Epilogue: bias, bf16, the tile through LDS in pixel order, full-line stores, the
// statistics partials of y for the next GroupNorm in a fixed order.
#include "idf_common.h"
#include <stdlib.h>

namespace {

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

struct UpP {
  const bf16_t* x;        // [B][Hl][Wl][Cin]
  const bf16_t* w;        // fragment-major [Cin / 64][Cout / 16][16 taps = (py, px, ty, tx)][2][64][8]
  const float* bias;      // [Cout] or null
  bf16_t* y;              // [B][2 Hl][2 Wl][Cout]
  float* st_out;          // [B][tiles_per_img][Cout][2] or null
  int B, Hl, Wl, Cin, Cout;
  int R, tiles_per_img, n_tiles, wlshift;     // output rows per tile (even), 2 Hl / R, Cout / 64, log2(Wl)
};

constexpr int UP_PPB = 96;                    // bytes per low-resolution pixel of a 32-channel chunk image
constexpr int UP_TP = 72;                     // bf16 per pixel row of the output tile in LDS (64 couts + 8: 144 B, 16-byte aligned)

__device__ __forceinline__ void up_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(512) void upconv_bf16_kernel(const UpP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int Wl = p.Wl, W = 2 * Wl, RL = p.R >> 1, WH = Wl + 2, npl = (RL + 2) * WH;
  const int img_bytes = ((npl * UP_PPB + 15) >> 4) << 4;
  unsigned char* img0 = smem;                                   // two chunk images
  bf16_t* tileo = reinterpret_cast<bf16_t*>(smem + 2 * img_bytes);             // [256][UP_TP]
  float* part = reinterpret_cast<float*>(smem);                                 // [512][16] statistics partials (over the images and the tile, once they are read)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r16 = lane & 15;
  const int tile = blockIdx.x / p.n_tiles, n0 = (blockIdx.x % p.n_tiles) * 64;
  const int b = tile / p.tiles_per_img, t_in = tile - b * p.tiles_per_img, oy0 = t_in * p.R, ly0 = oy0 >> 1;
  const int py = (wave >> 1) & 1, px = wave & 1, ch = wave >> 2;
  const int nchunks = p.Cin >> 5;

  // staging plan: this thread's (up to two) 16-byte vectors of a chunk image
  int goff[2], loff[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int idx = tid + k * 512;
    goff[k] = -1; loff[k] = -1;
    if (idx < npl * 4) {
      const int pix = idx >> 2, q = idx & 3;
      const int hy = pix / WH, hx = pix - hy * WH;
      const int iy = ly0 + hy - 1, ix = hx - 1;
      loff[k] = pix * UP_PPB + q * 16;
      if ((unsigned)iy < (unsigned)p.Hl && (unsigned)ix < (unsigned)Wl) goff[k] = ((b * p.Hl + iy) * Wl + ix) * p.Cin + q * 8;
    }
  }
  // this lane's pixel of every fragment: q = i * 16 + r16 over the RL x Wl low-resolution positions of the tile
  int pbase[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = i * 16 + r16, ly = q >> p.wlshift, lx = q & (Wl - 1);
    pbase[i] = ((ly + py) * WH + lx + px) * UP_PPB + g * 16;
  }
  const bf16_t* wb = p.w + ((size_t)((n0 >> 4) + 2 * ch) * 16 + (py * 2 + px) * 4) * 2 * 512 + lane * 8;
  // element offset of (pair, cout fragment a, tap t, half): ((pair * (Cout / 16) + nf) * 16 + tap) * 2 + half) * 512
  const size_t pair_stride = (size_t)(p.Cout >> 4) * 16 * 2 * 512;

  f32x4_t acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[a][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  u32x4_t xr[2];
  auto load_chunk = [&](int c) {
#pragma unroll
    for (int k = 0; k < 2; ++k)
      xr[k] = goff[k] >= 0 ? *reinterpret_cast<const u32x4_t*>(p.x + goff[k] + c * 32) : u32x4_t{0u, 0u, 0u, 0u};
  };
  auto store_chunk = [&](int c) {
    unsigned char* img = img0 + (c & 1) * img_bytes;
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (loff[k] >= 0) *reinterpret_cast<u32x4_t*>(img + loff[k]) = xr[k];
  };
  load_chunk(0);
#pragma unroll 1
  for (int c = 0; c < nchunks; ++c) {
    // the chunk's weights: 4 taps x 2 cout fragments, 1 KB per wave instruction
    bf16x8_t wf[4][2];
    const bf16_t* wc = wb + (size_t)(c >> 1) * pair_stride + (c & 1) * 512;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int a = 0; a < 2; ++a)
        wf[t][a] = *reinterpret_cast<const bf16x8_t*>(wc + ((size_t)a * 16 + t) * 2 * 512);
    store_chunk(c);
    if (c + 1 < nchunks) load_chunk(c + 1);
    up_barrier();                 // image c complete; image c - 1's readers are past it (they stored image c after reading it)
    const unsigned char* img = img0 + (c & 1) * img_bytes;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int toff = ((t >> 1) * WH + (t & 1)) * UP_PPB;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bf16x8_t xf = *reinterpret_cast<const bf16x8_t*>(img + pbase[i] + toff);
#pragma unroll
        for (int a = 0; a < 2; ++a) acc[a][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t][a], xf, acc[a][i], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: bias, bf16, tile in pixel order through LDS
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int col = 32 * ch + 16 * a + 4 * g;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) bv = *reinterpret_cast<const float4*>(p.bias + n0 + col);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = i * 16 + r16, ly = q >> p.wlshift, lx = q & (Wl - 1);
      const int pl = (2 * ly + py) * W + 2 * lx + px;
      const uint32_t lo = idf_pack_bf16(acc[a][i][0] + bv.x, acc[a][i][1] + bv.y);
      const uint32_t hi = idf_pack_bf16(acc[a][i][2] + bv.z, acc[a][i][3] + bv.w);
      *reinterpret_cast<uint2*>(tileo + pl * UP_TP + col) = make_uint2(lo, hi);
    }
  }
  __syncthreads();
  const int v = tid & 7;
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int pl = (tid >> 3) + 64 * k;
    const uint4 o = *reinterpret_cast<const uint4*>(tileo + pl * UP_TP + v * 8);
    const int oy = oy0 + pl / W, ox = pl - (pl / W) * W;
    *reinterpret_cast<uint4*>(p.y + ((size_t)(b * 2 * p.Hl + oy) * W + ox) * p.Cout + n0 + v * 8) = o;
    const uint32_t w4[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float lo = __uint_as_float(w4[j] << 16), hi = __uint_as_float(w4[j] & 0xffff0000u);
      s1[2 * j] += lo; s2[2 * j] += lo * lo; s1[2 * j + 1] += hi; s2[2 * j + 1] += hi * hi;
    }
  }
  if (p.st_out) {
    __syncthreads();              // every thread has read its tile vectors
#pragma unroll
    for (int e = 0; e < 8; ++e) { part[tid * 16 + e] = s1[e]; part[tid * 16 + 8 + e] = s2[e]; }
    __syncthreads();
    if (tid < 128) {
      const int c = tid >> 1, which = tid & 1, cv = c >> 3, e = c & 7;
      float s = 0.f;
      for (int t = 0; t < 64; ++t) s += part[(t * 8 + cv) * 16 + which * 8 + e];      // the 64 threads of cout vector cv, fixed order
      p.st_out[(((size_t)b * p.tiles_per_img + t_in) * p.Cout + n0 + c) * 2 + which] = s;
    }
  }
}

// Summed sub-pixel weights of every UpSample conv of a network in ONE launch (the training step re-packs them with the other
// weight shadows after each optimizer step).  One thread per (cout, cin) pair: nine master weights in, sixteen sums out.
struct UpPackDesc {
  const float* src;       // master weight, logical (o, i, tap) at o * so + i * si + tap * st
  bf16_t* dst;            // fragment-major [I / 64][O / 16][16][2][64][8]: rows = couts, k = cin (forward)
  bf16_t* dstd;           // optional, fragment-major [O / 64][I / 16][16][2][64][8]: rows = cins, k = cout (data gradient; O % 64 == 0)
  long so, si, st;
  int O, I;
};
__global__ __launch_bounds__(256) void upconv_pack_kernel(const UpPackDesc* __restrict__ tab) {
  const UpPackDesc d = tab[blockIdx.y];
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)d.O * d.I) return;
  const int o = (int)(idx / d.I), i = (int)(idx - (long)o * d.I);
  float w[3][3];
#pragma unroll
  for (int t = 0; t < 9; ++t) w[t / 3][t % 3] = d.src[o * d.so + i * d.si + t * d.st];
  // rows / columns a low-resolution tap stands for: S(0,0) = {0}, S(0,1) = {1,2}, S(1,0) = {0,1}, S(1,1) = {2}
  float rs[2][2][3];      // [py][ty][kx]: kernel rows summed
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    rs[0][0][kx] = w[0][kx]; rs[0][1][kx] = w[1][kx] + w[2][kx];
    rs[1][0][kx] = w[0][kx] + w[1][kx]; rs[1][1][kx] = w[2][kx];
  }
  const size_t base = ((size_t)(i >> 6) * (d.O >> 4) + (o >> 4)) * 16;
  const int inner = (((i >> 3) & 3) * 16 + (o & 15)) * 8 + (i & 7), half = (i >> 5) & 1;
  const size_t based = ((size_t)(o >> 6) * (d.I >> 4) + (i >> 4)) * 16;
  const int innerd = (((o >> 3) & 3) * 16 + (i & 15)) * 8 + (o & 7), halfd = (o >> 5) & 1;
#pragma unroll
  for (int py = 0; py < 2; ++py)
#pragma unroll
    for (int px = 0; px < 2; ++px)
#pragma unroll
      for (int ty = 0; ty < 2; ++ty)
#pragma unroll
        for (int tx = 0; tx < 2; ++tx) {
          const float* r = rs[py][ty];
          const float v = px == 0 ? (tx == 0 ? r[0] : r[1] + r[2]) : (tx == 0 ? r[0] + r[1] : r[2]);
          const int tap = (py * 2 + px) * 4 + ty * 2 + tx;
          d.dst[((base + tap) * 2 + half) * 512 + inner] = f32_to_bf16(v);
          if (d.dstd) d.dstd[((based + tap) * 2 + halfd) * 512 + innerd] = f32_to_bf16(v);
        }
}

// ---- the data gradient of the same layer, in the same form: dx (low resolution) = sum over the four output parities of a 2x2-tap
// product of dy's parity plane with the transposed summed weights -- 16 tap products per low-resolution pixel instead of a 3x3 conv
// over the 4x larger dy followed by a 2x2 sum-pool pass:
//   dx[Y][X] = sum_{py,px,ty,tx} W'[py][px][ty][tx]^T dy[2 (Y + 1 - ty - py) + py][2 (X + 1 - tx - px) + px]
// One 512-thread workgroup = 64 low-resolution pixels (RL rows x Wl columns) x 64 cins.  Wave w takes parity w & 3 and the cin half
// w >> 2: its 4 taps over all 64 pixels x 32 cins (4 x 2 accumulator tiles, the forward kernel's shape), and the four parity waves'
// partial sums meet in LDS at the end.  The dy tile of a 32-cout chunk -- (2 RL + 2) x (2 Wl + 2) pixels, pitch 96 B -- is staged once
// per chunk (two barriers: the tile is 38 KB, not double-buffered).
struct UpDP {
  const bf16_t* dy;       // [B][2 Hl][2 Wl][Cout]
  const bf16_t* w;        // fragment-major [Cout / 64][Cin / 16][16][2][64][8]
  bf16_t* dx;             // [B][Hl][Wl][Cin]
  int B, Hl, Wl, Cin, Cout;
  int RL, tiles_per_img, n_tiles, wlshift;
};

__global__ __launch_bounds__(512) void upconv_dgrad_bf16_kernel(const UpDP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int Wl = p.Wl, RL = p.RL, WH = 2 * Wl + 2, nph = (2 * RL + 2) * WH;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r16 = lane & 15;
  const int tile = blockIdx.x / p.n_tiles, n0 = (blockIdx.x % p.n_tiles) * 64;
  const int b = tile / p.tiles_per_img, t_in = tile - b * p.tiles_per_img, Y0 = t_in * RL;
  const int py = (wave >> 1) & 1, px = wave & 1, ch = wave >> 2;
  const int nchunks = p.Cout >> 5, H2 = 2 * p.Hl, W2 = 2 * Wl;

  constexpr int NV = 4;           // staged vectors per thread: (2 RL + 2)(2 Wl + 2) x 4 <= 2048 (planned on the host)
  int goff[NV], loff[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int idx = tid + k * 512;
    goff[k] = -1; loff[k] = -1;
    if (idx < nph * 4) {
      const int pix = idx >> 2, q = idx & 3;
      const int hy = pix / WH, hx = pix - hy * WH;
      const int iy = 2 * Y0 - 1 + hy, ix = hx - 1;
      loff[k] = pix * UP_PPB + q * 16;
      if ((unsigned)iy < (unsigned)H2 && (unsigned)ix < (unsigned)W2) goff[k] = ((b * H2 + iy) * W2 + ix) * p.Cout + q * 8;
    }
  }
  int pbase[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = i * 16 + r16, Yl = q >> p.wlshift, Xl = q & (Wl - 1);
    pbase[i] = ((2 * Yl + 3 - py) * WH + 2 * Xl + 3 - px) * UP_PPB + g * 16;      // tap (0, 0); tap (ty, tx): - 2 (ty WH + tx) pixels
  }
  const bf16_t* wb = p.w + ((size_t)((n0 >> 4) + 2 * ch) * 16 + (py * 2 + px) * 4) * 2 * 512 + lane * 8;
  const size_t pair_stride = (size_t)(p.Cin >> 4) * 16 * 2 * 512;

  f32x4_t acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[a][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  u32x4_t xr[NV];
  auto load_chunk = [&](int c) {
#pragma unroll
    for (int k = 0; k < NV; ++k)
      xr[k] = goff[k] >= 0 ? *reinterpret_cast<const u32x4_t*>(p.dy + goff[k] + c * 32) : u32x4_t{0u, 0u, 0u, 0u};
  };
  load_chunk(0);
#pragma unroll 1
  for (int c = 0; c < nchunks; ++c) {
    bf16x8_t wf[4][2];
    const bf16_t* wc = wb + (size_t)(c >> 1) * pair_stride + (c & 1) * 512;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int a = 0; a < 2; ++a)
        wf[t][a] = *reinterpret_cast<const bf16x8_t*>(wc + ((size_t)a * 16 + t) * 2 * 512);
    if (c > 0) up_barrier();        // the previous chunk's reads are done
#pragma unroll
    for (int k = 0; k < NV; ++k)
      if (loff[k] >= 0) *reinterpret_cast<u32x4_t*>(smem + loff[k]) = xr[k];
    if (c + 1 < nchunks) load_chunk(c + 1);
    up_barrier();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int toff = -2 * ((t >> 1) * WH + (t & 1)) * UP_PPB;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bf16x8_t xf = *reinterpret_cast<const bf16x8_t*>(smem + pbase[i] + toff);
#pragma unroll
        for (int a = 0; a < 2; ++a) acc[a][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t][a], xf, acc[a][i], 0, 0, 0);
      }
    }
  }
  // ---- the four parities' partial sums meet in LDS: red [8 waves][64 pixels][32 cins + 4] fp32
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);
  constexpr int RP = 36;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<float4*>(red + ((size_t)wave * 64 + i * 16 + r16) * RP + 16 * a + 4 * g) =
          make_float4(acc[a][i][0], acc[a][i][1], acc[a][i][2], acc[a][i][3]);
  __syncthreads();
  {
    // 64 pixels x 64 cins = 512 vectors of 8 cins: one per thread
    const int pxl = tid >> 3, v = tid & 7, half = v >> 2, c8 = (v & 3) * 8;
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float* src = red + ((size_t)(half * 4 + w) * 64 + pxl) * RP + c8;
      const float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 4);
      o[0] += lo.x; o[1] += lo.y; o[2] += lo.z; o[3] += lo.w; o[4] += hi.x; o[5] += hi.y; o[6] += hi.z; o[7] += hi.w;
    }
    const int Yl = pxl >> p.wlshift, Xl = pxl & (Wl - 1);
    uint32_t w4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) w4[j] = idf_pack_bf16(o[2 * j], o[2 * j + 1]);
    *reinterpret_cast<uint4*>(p.dx + ((size_t)(b * p.Hl + Y0 + Yl) * Wl + Xl) * p.Cin + n0 + v * 8) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
  }
}

// ---- DownSample's data gradient (modules.py:63-75: conv3x3 stride 2, pad 1) by output parity.  The transposed stride-2 conv over a
// zero-stuffed dy multiplies three zeros out of four; per parity of the high-resolution pixel only the taps that land on a dy pixel
// remain:  i = 2 Y + py needs ky = 1 (dy row Y) for py = 0 and ky = 0 (row Y + 1), ky = 2 (row Y) for py = 1 -- 1, 2, 2 and 4 taps for
// the four parities, the 3x3 weights used as they are (the data-gradient fragment-major shadow, taps stored flipped).
// The workgroup shape is upconv_bf16_kernel's: 256 high-resolution pixels x 64 cins, wave = (parity, cin half).
struct DnP {
  const bf16_t* dy;       // [B][Hl][Wl][Cout]
  const bf16_t* w;        // data-gradient fragment-major shadow [Cout / 64][Cin / 16][9 taps flipped][2][64][8]
  const bf16_t* res;      // [B][2 Hl][2 Wl][Cin] or null: a gradient arriving over another branch of the same input
  bf16_t* dx;             // [B][2 Hl][2 Wl][Cin]
  int B, Hl, Wl, Cin, Cout;
  int R, tiles_per_img, n_tiles, wlshift;
};

__global__ __launch_bounds__(512) void downconv_dgrad_bf16_kernel(const DnP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int Wl = p.Wl, W = 2 * Wl, RL = p.R >> 1, WH = Wl + 2, npl = (RL + 2) * WH;
  const int img_bytes = ((npl * UP_PPB + 15) >> 4) << 4;
  unsigned char* img0 = smem;
  bf16_t* tileo = reinterpret_cast<bf16_t*>(smem + 2 * img_bytes);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r16 = lane & 15;
  const int tile = blockIdx.x / p.n_tiles, n0 = (blockIdx.x % p.n_tiles) * 64;
  const int b = tile / p.tiles_per_img, t_in = tile - b * p.tiles_per_img, oy0 = t_in * p.R, ly0 = oy0 >> 1;
  const int py = (wave >> 1) & 1, px = wave & 1, ch = wave >> 2;
  const int nchunks = p.Cout >> 5;

  int goff[2], loff[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int idx = tid + k * 512;
    goff[k] = -1; loff[k] = -1;
    if (idx < npl * 4) {
      const int pix = idx >> 2, q = idx & 3;
      const int hy = pix / WH, hx = pix - hy * WH;
      const int iy = ly0 + hy - 1, ix = hx - 1;
      loff[k] = pix * UP_PPB + q * 16;
      if ((unsigned)iy < (unsigned)p.Hl && (unsigned)ix < (unsigned)Wl) goff[k] = ((b * p.Hl + iy) * Wl + ix) * p.Cout + q * 8;
    }
  }
  int pbase[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = i * 16 + r16, ly = q >> p.wlshift, lx = q & (Wl - 1);
    pbase[i] = ((ly + 1) * WH + lx + 1) * UP_PPB + g * 16;        // dy pixel (Y, X) of this lane's output; taps add (dY WH + dX) pixels
  }
  const bf16_t* wb = p.w + ((size_t)((n0 >> 4) + 2 * ch) * 9) * 2 * 512 + lane * 8;
  const size_t pair_stride = (size_t)(p.Cin >> 4) * 9 * 2 * 512;

  f32x4_t acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[a][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  u32x4_t xr[2];
  auto load_chunk = [&](int c) {
#pragma unroll
    for (int k = 0; k < 2; ++k)
      xr[k] = goff[k] >= 0 ? *reinterpret_cast<const u32x4_t*>(p.dy + goff[k] + c * 32) : u32x4_t{0u, 0u, 0u, 0u};
  };
  load_chunk(0);
#pragma unroll 1
  for (int c = 0; c < nchunks; ++c) {
    bf16x8_t wf[2][2][2];           // [row tap][column tap][cin fragment]; a parity-0 axis has one tap
    const bf16_t* wc = wb + (size_t)(c >> 1) * pair_stride + (c & 1) * 512;
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int tj = 0; tj < 2; ++tj) {
        if (ti > py || tj > px) continue;                       // (wave-uniform)
        const int ky = py ? 2 * ti : 1, kx = px ? 2 * tj : 1, tf = 8 - (ky * 3 + kx);
#pragma unroll
        for (int a = 0; a < 2; ++a)
          wf[ti][tj][a] = *reinterpret_cast<const bf16x8_t*>(wc + ((size_t)a * 9 + tf) * 2 * 512);
      }
    unsigned char* img = img0 + (c & 1) * img_bytes;
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (loff[k] >= 0) *reinterpret_cast<u32x4_t*>(img + loff[k]) = xr[k];
    if (c + 1 < nchunks) load_chunk(c + 1);
    up_barrier();
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int tj = 0; tj < 2; ++tj) {
        if (ti > py || tj > px) continue;
        // ky = 0 reads dy row Y + 1, ky = 2 (and the lone ky = 1) row Y
        const int dY = (py && ti == 0) ? 1 : 0, dX = (px && tj == 0) ? 1 : 0;
        const int toff = (dY * WH + dX) * UP_PPB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bf16x8_t xf = *reinterpret_cast<const bf16x8_t*>(img + pbase[i] + toff);
#pragma unroll
          for (int a = 0; a < 2; ++a) acc[a][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ti][tj][a], xf, acc[a][i], 0, 0, 0);
        }
      }
  }
  // ---- epilogue: (+ res), bf16, the tile in pixel order through LDS, full-line stores
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int col = 32 * ch + 16 * a + 4 * g;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = i * 16 + r16, ly = q >> p.wlshift, lx = q & (Wl - 1);
      const int pl = (2 * ly + py) * W + 2 * lx + px;
      float v0 = acc[a][i][0], v1 = acc[a][i][1], v2 = acc[a][i][2], v3 = acc[a][i][3];
      if (p.res) {
        const uint2 rs = *reinterpret_cast<const uint2*>(p.res + ((size_t)(b * 2 * p.Hl + oy0) * W + pl) * p.Cin + n0 + col);
        v0 += __uint_as_float(rs.x << 16); v1 += __uint_as_float(rs.x & 0xffff0000u);
        v2 += __uint_as_float(rs.y << 16); v3 += __uint_as_float(rs.y & 0xffff0000u);
      }
      const uint32_t lo = idf_pack_bf16(v0, v1);
      const uint32_t hi = idf_pack_bf16(v2, v3);
      *reinterpret_cast<uint2*>(tileo + pl * UP_TP + col) = make_uint2(lo, hi);
    }
  }
  __syncthreads();
  const int v = tid & 7;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int pl = (tid >> 3) + 64 * k;
    const uint4 o = *reinterpret_cast<const uint4*>(tileo + pl * UP_TP + v * 8);
    *reinterpret_cast<uint4*>(p.dx + ((size_t)(b * 2 * p.Hl + oy0) * W + pl) * p.Cin + n0 + v * 8) = o;
  }
}

inline bool up_dgrad_plan(int Hl, int Wl, int Cin, int Cout, int* RL) {
  if (Wl < 8 || Wl > 32 || (Wl & (Wl - 1)) || Hl < 1 || (Cin % 64) || (Cout % 64)) return false;
  const int r = 64 / Wl;                          // low-resolution rows per 64-pixel tile: 8 / 4 / 2
  if (r < 1 || Hl % r || (2 * r + 2) * (2 * Wl + 2) * 4 > 2048) return false;
  *RL = r;
  return true;
}

inline bool up_plan(int Hl, int Wl, int Cin, int Cout, int* R) {
  if (Wl < 8 || Wl > 32 || (Wl & (Wl - 1)) || Hl < 1 || (Cin % 64) || (Cout % 64)) return false;
  const int W = 2 * Wl, r = 256 / W;            // output rows per 256-pixel tile: 4 / 8 / 16
  if (r < 2 || (r & 1) || (2 * Hl) % r) return false;
  *R = r;
  return true;
}

}  // namespace

// tiles per image of idf_upconv_bf16's statistics partials (st_out [B][tiles][Cout][2]); 0: shape not covered
// (low-resolution width 8, 16 or 32; Cin, Cout multiples of 64)
extern "C" int idf_upconv_tiles(int Hl, int Wl, int Cin, int Cout) {
  int R;
  return up_plan(Hl, Wl, Cin, Cout, &R) ? 2 * Hl / R : 0;
}

extern "C" int idf_upconv_bf16(const void* x, const void* w_sub_frag, const float* bias, void* y, float* st_out, int B, int Hl,
                               int Wl, int Cin, int Cout, void* stream) {
  int R;
  if (!up_plan(Hl, Wl, Cin, Cout, &R)) IDF_FAIL(IDF_ERR_UNSUPPORTED, "upconv_bf16: Hl%d Wl%d Cin%d Cout%d not covered", Hl, Wl, Cin, Cout);
  if (!x || !w_sub_frag || !y) IDF_FAIL(IDF_ERR_BADARG, "upconv_bf16: null argument");
  if (B == 0) return IDF_OK;
  if ((long)B * 4 * Hl * Wl * (Cin > Cout ? Cin : Cout) >= (1L << 31)) IDF_FAIL(IDF_ERR_UNSUPPORTED, "upconv_bf16: tensor too large for 32-bit offsets");
  UpP p;
  p.x = (const bf16_t*)x; p.w = (const bf16_t*)w_sub_frag; p.bias = bias; p.y = (bf16_t*)y; p.st_out = st_out;
  p.B = B; p.Hl = Hl; p.Wl = Wl; p.Cin = Cin; p.Cout = Cout;
  p.R = R; p.tiles_per_img = 2 * Hl / R; p.n_tiles = Cout / 64;
  int ws = 0;
  while ((1 << ws) < Wl) ++ws;
  p.wlshift = ws;
  const int npl = (R / 2 + 2) * (Wl + 2);
  const size_t img = (((size_t)npl * UP_PPB + 15) / 16) * 16;
  size_t lds = 2 * img + (size_t)256 * UP_TP * 2;
  if (lds < (size_t)512 * 16 * sizeof(float)) lds = (size_t)512 * 16 * sizeof(float);
  static IdfLdsGrant grant;
  if (hipError_t e = idf_ensure_lds((const void*)upconv_bf16_kernel, lds, grant); e != hipSuccess)
    IDF_FAIL(IDF_ERR_HIP, "upconv_bf16: %d bytes of LDS refused: %s", (int)lds, hipGetErrorString(e));
  hipLaunchKernelGGL(upconv_bf16_kernel, dim3((unsigned)(B * p.tiles_per_img * p.n_tiles)), dim3(512), lds, (hipStream_t)stream, p);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}

// table (device): nrows x {src*, dst*, dstd*, long so, si, st, int O, I} (56 bytes), one row per UpSample conv; max_pairs = the
// largest O * I of the rows.  dst: the w_sub_frag operand of idf_upconv_bf16; dstd (optional): that of idf_upconv_dgrad_bf16.
extern "C" int idf_upconv_pack_batched(const void* table, int nrows, long max_pairs, void* stream) {
  if (nrows <= 0 || max_pairs <= 0) return IDF_OK;
  if (!table) IDF_FAIL(IDF_ERR_BADARG, "upconv_pack_batched: null table");
  static_assert(sizeof(UpPackDesc) == 56, "the host builds 56-byte rows");
  hipLaunchKernelGGL(upconv_pack_kernel, dim3((unsigned)((max_pairs + 255) / 256), (unsigned)nrows), dim3(256), 0, (hipStream_t)stream,
                     (const UpPackDesc*)table);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}

// dx [B, Hl, Wl, Cin] = the data gradient of idf_upconv_bf16's layer w.r.t. its low-resolution input, from dy [B, 2 Hl, 2 Wl, Cout];
// w_sub_dgrad_frag: the summed weights with rows = cins, k = couts (idf_upconv_pack_batched's `dstd`).  idf_upconv_dgrad_ok: 1 when
// the shape is covered (Wl in {8, 16, 32}; Cin, Cout % 64 == 0; Hl a multiple of 64 / Wl).
extern "C" int idf_upconv_dgrad_ok(int Hl, int Wl, int Cin, int Cout) {
  int RL;
  return up_dgrad_plan(Hl, Wl, Cin, Cout, &RL) ? 1 : 0;
}

extern "C" int idf_upconv_dgrad_bf16(const void* dy, const void* w_sub_dgrad_frag, void* dx, int B, int Hl, int Wl, int Cin,
                                     int Cout, void* stream) {
  int RL;
  if (!up_dgrad_plan(Hl, Wl, Cin, Cout, &RL)) IDF_FAIL(IDF_ERR_UNSUPPORTED, "upconv_dgrad_bf16: Hl%d Wl%d Cin%d Cout%d not covered", Hl, Wl, Cin, Cout);
  if (!dy || !w_sub_dgrad_frag || !dx) IDF_FAIL(IDF_ERR_BADARG, "upconv_dgrad_bf16: null argument");
  if (B == 0) return IDF_OK;
  if ((long)B * 4 * Hl * Wl * (Cin > Cout ? Cin : Cout) >= (1L << 31)) IDF_FAIL(IDF_ERR_UNSUPPORTED, "upconv_dgrad_bf16: tensor too large for 32-bit offsets");
  UpDP p;
  p.dy = (const bf16_t*)dy; p.w = (const bf16_t*)w_sub_dgrad_frag; p.dx = (bf16_t*)dx;
  p.B = B; p.Hl = Hl; p.Wl = Wl; p.Cin = Cin; p.Cout = Cout;
  p.RL = RL; p.tiles_per_img = Hl / RL; p.n_tiles = Cin / 64;
  int ws = 0;
  while ((1 << ws) < Wl) ++ws;
  p.wlshift = ws;
  size_t lds = (size_t)(2 * RL + 2) * (2 * Wl + 2) * UP_PPB, red = (size_t)8 * 64 * 36 * sizeof(float);
  if (lds < red) lds = red;
  static IdfLdsGrant grant;
  if (hipError_t e = idf_ensure_lds((const void*)upconv_dgrad_bf16_kernel, lds, grant); e != hipSuccess)
    IDF_FAIL(IDF_ERR_HIP, "upconv_dgrad_bf16: %d bytes of LDS refused: %s", (int)lds, hipGetErrorString(e));
  hipLaunchKernelGGL(upconv_dgrad_bf16_kernel, dim3((unsigned)(B * p.tiles_per_img * p.n_tiles)), dim3(512), lds, (hipStream_t)stream, p);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}

// dx [B, 2 Hl, 2 Wl, Cin] (+ res) = the data gradient of a stride-2 3x3 conv (pad 1) w.r.t. its input, from dy [B, Hl, Wl, Cout]:
// w_dgrad_frag = the conv's data-gradient weights fragment-major (idf_pack_conv_weights_batched's `wdfrag`).
// idf_downconv_dgrad_ok: 1 when covered (Wl in {8, 16, 32}; Cin, Cout % 64 == 0; 256-pixel tiles of whole rows).
extern "C" int idf_downconv_dgrad_ok(int Hl, int Wl, int Cin, int Cout) {
  int R;
  return up_plan(Hl, Wl, Cout, Cin, &R) ? 1 : 0;
}

extern "C" int idf_downconv_dgrad_bf16(const void* dy, const void* w_dgrad_frag, const void* res, void* dx, int B, int Hl, int Wl,
                                       int Cin, int Cout, void* stream) {
  int R;
  if (!up_plan(Hl, Wl, Cout, Cin, &R)) IDF_FAIL(IDF_ERR_UNSUPPORTED, "downconv_dgrad_bf16: Hl%d Wl%d Cin%d Cout%d not covered", Hl, Wl, Cin, Cout);
  if (!dy || !w_dgrad_frag || !dx) IDF_FAIL(IDF_ERR_BADARG, "downconv_dgrad_bf16: null argument");
  if (B == 0) return IDF_OK;
  if ((long)B * 4 * Hl * Wl * (Cin > Cout ? Cin : Cout) >= (1L << 31)) IDF_FAIL(IDF_ERR_UNSUPPORTED, "downconv_dgrad_bf16: tensor too large for 32-bit offsets");
  DnP p;
  p.dy = (const bf16_t*)dy; p.w = (const bf16_t*)w_dgrad_frag; p.res = (const bf16_t*)res; p.dx = (bf16_t*)dx;
  p.B = B; p.Hl = Hl; p.Wl = Wl; p.Cin = Cin; p.Cout = Cout;
  p.R = R; p.tiles_per_img = 2 * Hl / R; p.n_tiles = Cin / 64;
  int ws = 0;
  while ((1 << ws) < Wl) ++ws;
  p.wlshift = ws;
  const int npl = (R / 2 + 2) * (Wl + 2);
  const size_t img = (((size_t)npl * UP_PPB + 15) / 16) * 16;
  const size_t lds = 2 * img + (size_t)256 * UP_TP * 2;
  static IdfLdsGrant grant;
  if (hipError_t e = idf_ensure_lds((const void*)downconv_dgrad_bf16_kernel, lds, grant); e != hipSuccess)
    IDF_FAIL(IDF_ERR_HIP, "downconv_dgrad_bf16: %d bytes of LDS refused: %s", (int)lds, hipGetErrorString(e));
  hipLaunchKernelGGL(downconv_dgrad_bf16_kernel, dim3((unsigned)(B * p.tiles_per_img * p.n_tiles)), dim3(512), lds, (hipStream_t)stream, p);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}
