// Image-resident ResBlock forward for the 8x8 maps (bf16, gfx950): ONE launch computes a whole
//   AuxResBlock / ResBlock    (modules.py:261-328, 206-258):  h1 = conv1(SiLU(GN1(x)));  h2 = conv2(drop(SiLU(FiLM(GN2(h1)))));
//                                                            y  = conv3(drop(SiLU(GN3(h2)))) + shortcut(x)
//   ResBlock_encoder          (modules.py:331-366):          h1 = conv1(SiLU(GN1(x)));  y = conv2(drop(SiLU(GN2(h1)))) + shortcut(x)
// with one 512-thread workgroup per IMAGE.  At 8x8 a whole image is 64 pixels: the workgroup owns every pixel of every
// channel, so each GroupNorm's statistics close inside the workgroup and the activated tensor of the next stage is written
// straight into the LDS halo image its conv reads -- no launch boundary, no statistics round trip through memory, no
// re-staging between the stages.  The per-op path needs one launch per stage (17 us each at B = 32: launch floor + one
// memory round trip + coefficient fold + epilogue) on 64 of 256 CUs; this kernel runs the stages back to back on 32 CUs
// at the pace of its MFMA / weight stream.
//
// Work split: wave w of the 8 owns couts 16 w .. 16 w + 15 of ALL 64 pixels (4 MFMA tiles of 16 pixels x 16 couts).
//  * Weights never touch LDS: a wave's A fragments (16 couts x 32 channels of one tap = 16 B per lane) are loaded from
//    global memory straight into registers, one 64-channel chunk pair (18 fragments) ahead of the MFMAs that use them -- each
//    fragment register is re-loaded right behind its last MFMA, across stage boundaries too -- so the conv loop has NO
//    workgroup barrier and no LDS write phase (the first form of this kernel staged
//    a [9][128][64 B] slab per chunk through LDS: 930 cycles of ds_write_b128 + two barriers per chunk, 34 us per block; this
//    form: see profiles/r04_resblock_small.txt).
//  * A wave's 16 couts are 4 whole GroupNorm groups (128 channels / 32 groups = 4 per group = the 4 couts one lane holds), and
//    the 16 lanes of a DPP row hold the same couts: each GroupNorm's statistics, its coefficient fold and the activation of
//    the next stage's input happen in the wave's registers -- two workgroup barriers per stage in all (the LDS image is
//    free / the LDS image is written).
// LDS:  Abuf [Cin/32][10 rows][16][96 B]       the activated conv input, 32-channel chunk major, as a halo image (10 x 10 pixels
//                                              used, 64 of the 96 bytes of a pixel used: the padding makes the fragment reads
//                                              bank-conflict-free, see WH / PPB below); the border pixels are zeroed once and
//                                              stay zero (the reference pads the ACTIVATED tensor)
//       cof / chs                              coefficients and channel sums of the first GroupNorm (its statistics come
//                                              from the producers' partials)
// Arithmetic per element is the per-op path's (pro_vec: fold, SiLU, dropout keyed by the element's index in the dense
// activated tensor; statistics of the bf16-ROUNDED outputs; the shortcut rounded to bf16 before it joins): results agree
// with the per-op launches up to the summation order of the statistics.
#include "idf_common.h"
#include "idf_gnfold.h"
#include "../../include/infodiff_hip.h"
#include <type_traits>

namespace {

constexpr int CK = 32, NPX = 64, BN = 128, NT = 512;
// LDS image of one 32-channel chunk: 10 halo rows of 16 pixel slots (10 used), 96 bytes per pixel (64 used): with a row pitch of
// 16 pixels and a pixel pitch of 96 B the ds_read_b128 of an MFMA fragment -- 16 lanes = two image rows of 8 pixels, 4 lane
// groups = the 4 channel slots -- is bank-conflict-free (brute-forced over the gfx950 lane groups; the 10-pixel row / 64-byte
// pitch of idf_conv3x3.hip, with or without its XOR swizzle, is 2-way here because an 8-pixel row breaks the run of
// consecutive halo rows), and every tap / chunk displacement is an instruction immediate.
constexpr int WH = 16, HROWS = 10, PPB = 96, NPH = HROWS * WH, CHB = NPH * PPB;      // bytes of a chunk image: 15360
constexpr int MAXC = 256;

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

// byte offset of the 16-byte channel slot q of halo pixel h inside a chunk image
__device__ __forceinline__ int aoff(int h, int q) { return h * PPB + q * 16; }

// sum over the 16 lanes of a DPP row (every lane of the row ends with the total)
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
  return v;
}

// a = act(x * sc + sh) on one 16-byte vector (8 bf16 channels), dropout keyed by the vector's index in the dense activated
// tensor: idf_conv3x3.hip's pro_vec
__device__ __forceinline__ uint4 act_vec(const uint4 raw, const float (&scv)[8], const float (&shv)[8], bool drop, uint64_t seedv,
                                         uint32_t salt, uint32_t thr, float dscale, uint32_t vec) {
  const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
  float v[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
  const uint32_t h = drop ? idf_vec_hash(seedv, salt, vec) : 0u;
  idf_act_vec<8>(v, scv, shv, 2, drop, h, 0, thr, dscale);
  uint32_t o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = idf_pack_bf16(v[2 * i], v[2 * i + 1]);
  return make_uint4(o[0], o[1], o[2], o[3]);
}

// (mean, rstd) of a group from its (sum, sum of squares) over n = 2^k elements.  mu and var in double as
// idf_conv3x3.hip's pro_coefficients forms them (the divisions by n are exact multiplications); 1 / sqrt as v_rsq_f32 + one
// Newton step in double (2e-14 relative: the float it rounds to is pro_coefficients' except on rounding ties) -- the
// double-precision divide and square root sequences were 1 us per GroupNorm here.
__device__ __forceinline__ void group_stats(double a, double d, double inv_n, float eps, float* mean, float* rstd) {
  const double mu = a * inv_n;
  double var = d * inv_n - mu * mu;
  if (var < 0.0) var = 0.0;
  const double vd = var + (double)eps;
  const double r0 = (double)__builtin_amdgcn_rsqf((float)vd);
  *rstd = (float)(r0 * (1.5 - 0.5 * vd * r0 * r0));
  *mean = (float)mu;
}

// (sc, sh) of channel c from the per-channel sums chs [C][2] (sum, sum of squares over the image's 64 pixels): the fold of
// idf_conv3x3.hip's pro_coefficients
__device__ __forceinline__ void fold_channel(const float* chs, int c, int C, int b, const IdfResblockStage& s, float eps,
                                             float* cof) {
  const int cpg = C >> 5, g = c / cpg;
  double a = 0.0, d = 0.0;
  for (int k = g * cpg; k < (g + 1) * cpg; ++k) { a += chs[2 * k]; d += chs[2 * k + 1]; }
  float mf, r;
  group_stats(a, d, 1.0 / ((double)NPX * cpg), eps, &mf, &r);
  const float ga = s.gamma ? s.gamma[c] : 1.f, be = s.beta ? s.beta[c] : 0.f;
  float sc = r * ga, sh = be - mf * sc;
  if (s.film_t) { const float f = 1.f + s.film_t[(size_t)b * s.ld_t + c]; sc *= f; sh = sh * f + s.film_t[(size_t)b * s.ld_t + C + c]; }
  if (s.film_a) { const float f = 1.f + s.film_a[(size_t)b * s.ld_a + c]; sc *= f; sh = sh * f + s.film_a[(size_t)b * s.ld_a + C + c]; }
  cof[2 * c] = sc; cof[2 * c + 1] = sh;
  if (s.sc) {
    s.sc[(size_t)b * C + c] = sc; s.sh[(size_t)b * C + c] = sh;
    if (c == g * cpg) { s.mean[b * 32 + g] = mf; s.rstd[b * 32 + g] = r; }
  }
}

#ifndef IDF_RB_WARM
#define IDF_RB_WARM 8        // helper workgroups per launch that pull the block's weights into L2 (0: none); launches of up to 128 images
                             // only: beyond that the image workgroups fill the chip and helpers would run on the launch's tail
#endif
struct RbK { IdfResblockArgs a; uint32_t thr; float dscale; };

// Workgroup barrier that orders LDS traffic only.  __syncthreads() makes hipcc drain vmcnt(0) first, i.e. wait for every
// global store of the epilogue (h, a: ~2 us of write latency) and for the weight fragments prefetched for the next stage;
// nothing in global memory is shared between the waves of this kernel (stamps: the two barriers of a stage cost 3.3 us of its
// 4.5-us epilogue, stage 0 7.5 us -- profiles/r04_resblock_small.txt).
__device__ __forceinline__ void rb_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Diagnostic build only (tools/build_variant.sh rbstamp idf_resblock.hip -DIDF_RB_STAMP; never in the shipped library):
// wave 0 of every block stamps s_memtime at the phase boundaries and adds the differences to g_rb_stamps
// [0] stage 0 (input, first fold, shortcut, activation), [1 + 2 k] conv loop of stage k, [2 + 2 k] its epilogue, [7] blocks
#ifdef IDF_RB_STAMP
__device__ unsigned long long g_rb_stamps[8];
__device__ __forceinline__ unsigned long long rb_now() {
  unsigned long long t;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  return t;
}
#define RB_STAMP(var) const unsigned long long var = rb_now()
#define RB_ADD(i, a, b) do { if (threadIdx.x == 0) atomicAdd(&g_rb_stamps[i], (b) - (a)); } while (0)
#else
#define RB_STAMP(var)
#define RB_ADD(i, a, b)
#endif

// one 64-channel PAIR of chunks of this wave's weights: 9 taps x 2 x (16 couts x 32 channels), 16 B per lane, tap and chunk.
// Pairs, because a weight row [cout][tap][Cin] is read in 128-byte lines = 64 channels: fetching the two halves of a line with
// back-to-back loads uses every line once (chunk by chunk each line crossed the L2 -> L1 path twice, which is what bounds a
// CU that streams 0.9 MB of weights per block)
struct WPair { bf16x8_t t[9][2]; };

__global__ __launch_bounds__(NT) void resblock8_fwd_kernel(const RbK k_in) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const IdfResblockArgs& p = k_in.a;
  const bf16_t* const px = reinterpret_cast<const bf16_t*>(p.x);
  const bf16_t* const px2 = reinterpret_cast<const bf16_t*>(p.x2);
  const uint32_t thr = k_in.thr;
  const float dscale = k_in.dscale;
  const int Cin = p.Cin, nck1 = Cin / CK;
  unsigned char* const Abuf = smem;                                               // [nck1][HROWS][WH][PPB]
  float* const cof = reinterpret_cast<float*>(smem + (size_t)nck1 * CHB);    // [MAXC][2]
  float* const chs = cof + 2 * MAXC;                                              // [MAXC][2]

  const int tid = threadIdx.x, lane = tid & 63, b = blockIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (b >= p.B) {
    // helper workgroups (IDF_RB_WARM of them behind the images', one per XCD when B % 8 == 0; they land on CUs the 32 image workgroups
    // leave idle): pull the block's weights into this XCD's L2 -- one dword per 128-byte line, results discarded -- while the image
    // workgroups run their input stage.  In a training step the 0.9 MB were last touched a step ago: cold block 29.1 us, warm 23.8
    // (tools/bench_resblock_cold.py).  (Warm-up loads issued by the image workgroups themselves lost: the vector memory pipe returns in
    // order, so every real load queued behind the warm-up's HBM round trips.)
    auto warm_all = [&](const void* w, int bytes) __attribute__((always_inline)) {
      const char* base = reinterpret_cast<const char*>(w);
      for (int off = tid * 128; off < bytes; off += NT * 128) {
        unsigned d;
        asm volatile("global_load_dword %0, %1, off" : "=v"(d) : "v"(base + off) : "memory");
      }
    };
    warm_all(p.s[0].w, 128 * 9 * Cin * 2);
    warm_all(p.s[1].w, 128 * 9 * BN * 2);
    if (p.nstage == 3) warm_all(p.s[2].w, 128 * 9 * BN * 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  RB_STAMP(t_begin);
  const int fr = lane & 15, fq = lane >> 4;
  const int wn0 = wave * 16;                                 // this wave's 16 couts
  const int c0 = wn0 + fq * 4;                               // the 4 couts this lane holds in the MFMA output = one GroupNorm group
  const bool drop_any = p.seed != nullptr;
  const uint64_t seedv = drop_any ? *p.seed : 0;

  int hbase[4];                                              // halo row of tap (0, 0) of this lane's pixel in each 16-pixel slice
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int pl = i * 16 + fr;
    hbase[i] = (pl >> 3) * WH + (pl & 7);
  }
  // this lane's piece of the wave's weight fragments: row (cout) wn0 + fr, channels fq * 8 .. + 7 of a chunk
  // Where fragment (tap, half) of chunk pair cp lives for this lane.  Layout 0: the forward shadow [cout][tap][cin] (a wave
  // instruction gathers 16 rows x 64 B).  Layout 1 (idf_resblock_pack_weight): fragment-major [pair][wave][tap][half][lane][8]
  // -- every wave instruction reads 1 KB of consecutive bytes.
  const bool wfrag = p.w_layout == 1;
  auto wptr = [&](const bf16_t* w, int cin, int cp, int tap, int half) __attribute__((always_inline)) -> const bf16_t* {
    if (wfrag) return w + ((size_t)((cp * 8 + wave) * 18 + tap * 2 + half) * 64 + lane) * 8;
    return w + (size_t)(wn0 + fr) * 9 * cin + (size_t)tap * cin + (cp * 2 + half) * CK + fq * 8;
  };
  auto load_w = [&](WPair& W, const bf16_t* w, int cin, int cp) __attribute__((always_inline)) {     // chunks 2 cp, 2 cp + 1
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      W.t[tap][0] = *reinterpret_cast<const bf16x8_t*>(wptr(w, cin, cp, tap, 0));
      W.t[tap][1] = *reinterpret_cast<const bf16x8_t*>(wptr(w, cin, cp, tap, 1));
    }
  };

  // ---- stage 0: the block input.  Its vectors and the statistics partials are fetched together (one round trip).
  // element-wise thread map over a 64-pixel x C-channel tensor: vector v = tid + k * NT -> pixel v / (C / 8), slot v % (C / 8)
  const int vpp1 = Cin / 8;                                  // vectors per pixel of the input
  const int nv1 = NPX * vpp1 / NT;                           // vectors per thread: 2 (128 channels) or 4 (256)
  u32x4_t xraw[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    xraw[k] = u32x4_t{0, 0, 0, 0};
    if (k < nv1) {
      const int v = tid + k * NT, pl = v / vpp1, c = (v - pl * vpp1) * 8;
      const bf16_t* src = (px2 && c >= p.C1) ? px2 + ((size_t)(b * NPX + pl) * (Cin - p.C1) + (c - p.C1))
                                             : px + ((size_t)(b * NPX + pl) * (px2 ? p.C1 : Cin) + c);
      xraw[k] = *reinterpret_cast<const u32x4_t*>(src);
    }
  }
  for (int c = tid; c < Cin; c += NT) {
    const float* st = p.st1;
    int T = p.T1, Cs = p.x2 ? p.C1 : Cin, cl = c;
    if (p.x2 && c >= p.C1) { st = p.st2; T = p.T2; Cs = Cin - p.C1; cl = c - p.C1; }
    const float2 S = idf_sum_partials(reinterpret_cast<const float2*>(st) + (size_t)b * T * Cs + cl, T, (size_t)Cs);
    chs[2 * c] = S.x; chs[2 * c + 1] = S.y;
  }
  WPair wA;                  // ONE pair in registers: every fragment is re-loaded for the next pair right behind its last MFMA
  load_w(wA, reinterpret_cast<const bf16_t*>(p.s[0].w), Cin, 0);     // conv1's first chunk pair flies through all of stage 0
  // zero the halo border of every chunk image (36 border pixels x 4 slots per chunk)
  for (int i = tid; i < nck1 * 36 * 4; i += NT) {
    const int ck = i / 144, r = i - ck * 144, bp = r >> 2, q = r & 3;
    // border pixel bp: top row 0..9, bottom row 10..19, left column rows 1..8 (20..27), right column (28..35)
    int hy, hx;
    if (bp < 10) { hy = 0; hx = bp; } else if (bp < 20) { hy = 9; hx = bp - 10; }
    else if (bp < 28) { hy = bp - 19; hx = 0; } else { hy = bp - 27; hx = 9; }
    const int h = hy * WH + hx;
    *reinterpret_cast<u32x4_t*>(Abuf + (size_t)ck * CHB + aoff(h, q)) = u32x4_t{0, 0, 0, 0};
  }
  rb_barrier();   
  for (int c = tid; c < Cin; c += NT) fold_channel(chs, c, Cin, b, p.s[0], p.eps, cof);

  uint2 res[4];              // the residual branch in the MFMA output layout (couts c0 .. c0 + 3 of pixels i * 16 + fr), bf16:
                             // the raw input (identity) or the 1x1 shortcut + bias, rounded as the tensor it is in the per-op path
  auto a_slot = [&](int pl, int c) -> unsigned char* {      // LDS home of the 8 channels (c & ~7).. of pixel pl
    const int h = ((pl >> 3) + 1) * WH + (pl & 7) + 1;
    return Abuf + (size_t)(c >> 5) * CHB + aoff(h, (c & 31) >> 3);
  };
  if (p.w_sc) {
    // raw input -> Abuf; centre-tap MFMAs against the [128][Cin] shortcut weight, fragments straight from global memory
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < nv1) {
        const int v = tid + k * NT, pl = v / vpp1, c = (v - pl * vpp1) * 8;
        *reinterpret_cast<u32x4_t*>(a_slot(pl, c)) = xraw[k];
      }
    bf16x8_t ws[8];
    const bf16_t* wsb = reinterpret_cast<const bf16_t*>(p.w_sc) + (size_t)(wn0 + fr) * Cin + fq * 8;
#pragma unroll
    for (int ck = 0; ck < 8; ++ck)
      if (ck < nck1) ws[ck] = *reinterpret_cast<const bf16x8_t*>(wsb + ck * CK);
    float4 bsc4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.b_sc) bsc4 = *reinterpret_cast<const float4*>(p.b_sc + c0);
    rb_barrier();   
    f32x4_t sacc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) sacc[i] = f32x4_t{bsc4.x, bsc4.y, bsc4.z, bsc4.w};
#pragma unroll
    for (int ck = 0; ck < 8; ++ck)
      if (ck < nck1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int h = hbase[i] + WH + 1;
          const bf16x8_t xf = *reinterpret_cast<const bf16x8_t*>(Abuf + (size_t)ck * CHB + aoff(h, fq));
          sacc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ws[ck], xf, sacc[i], 0, 0, 0);
        }
      }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      res[i].x = idf_pack_bf16(sacc[i][0], sacc[i][1]);
      res[i].y = idf_pack_bf16(sacc[i][2], sacc[i][3]);
    }
    rb_barrier();            // the raw image has been read (and cof is complete)
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) res[i] = *reinterpret_cast<const uint2*>(px + (size_t)(b * NPX + i * 16 + fr) * BN + c0);
    rb_barrier();            // cof complete
  }
  {
    const IdfResblockStage& s = p.s[0];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < nv1) {
        const int v = tid + k * NT, pl = v / vpp1, c = (v - pl * vpp1) * 8;
        float scv[8], shv[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 t4 = *reinterpret_cast<const float4*>(cof + 2 * c + 4 * q);
          scv[2 * q] = t4.x; shv[2 * q] = t4.y; scv[2 * q + 1] = t4.z; shv[2 * q + 1] = t4.w;
        }
        const unsigned e0 = (unsigned)((b * NPX + pl) * Cin + c);
        const uint4 o = act_vec(make_uint4(xraw[k][0], xraw[k][1], xraw[k][2], xraw[k][3]), scv, shv, s.drop && drop_any, seedv,
                                s.salt, thr, dscale, e0 >> 3);
        *reinterpret_cast<uint4*>(a_slot(pl, c)) = o;
        if (s.a_out) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(s.a_out) + e0) = o;
      }
  }
  rb_barrier();              // the activated image of conv1 is in place
  RB_STAMP(t_s0);
  RB_ADD(0, t_begin, t_s0);
  RB_ADD(7, 0ull, 1ull);

  // ---- the conv stages.  (The stage index is a compile-time constant in each copy of the body: p.s[] is then read from
  // the kernel arguments, not from a scratch copy of the struct.)
  auto run_stage = [&](auto ST) __attribute__((always_inline)) {
    constexpr int st = decltype(ST)::value;
    const IdfResblockStage& s = p.s[st];
    const IdfResblockStage& nx = p.s[st + 1 < 3 ? st + 1 : 2];
    const int cin = st == 0 ? Cin : BN, nck = cin / CK;      // 4 or 8 chunks: always even
    const bool last = st + 1 == p.nstage;
    const bf16_t* const wp = reinterpret_cast<const bf16_t*>(s.w);
    // what the epilogue needs from memory does not depend on the conv: fetched now, used after the MFMAs
    const float4 bias4 = *reinterpret_cast<const float4*>(s.bias + c0);
    float4 gam4 = make_float4(1.f, 1.f, 1.f, 1.f), bet4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 ft0 = make_float4(0.f, 0.f, 0.f, 0.f), ft1 = ft0, fa0 = ft0, fa1 = ft0;
    if (!last) {
      if (nx.gamma) gam4 = *reinterpret_cast<const float4*>(nx.gamma + c0);
      if (nx.beta) bet4 = *reinterpret_cast<const float4*>(nx.beta + c0);
      if (nx.film_t) { ft0 = *reinterpret_cast<const float4*>(nx.film_t + (size_t)b * nx.ld_t + c0); ft1 = *reinterpret_cast<const float4*>(nx.film_t + (size_t)b * nx.ld_t + BN + c0); }
      if (nx.film_a) { fa0 = *reinterpret_cast<const float4*>(nx.film_a + (size_t)b * nx.ld_a + c0); fa1 = *reinterpret_cast<const float4*>(nx.film_a + (size_t)b * nx.ld_a + BN + c0); }
    }
    f32x4_t acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    RB_STAMP(t_c0);
    const int ncp = nck / 2;                   // 2 or 4 chunk pairs
    for (int cp = 0; cp < ncp; ++cp) {
      // where this lane's fragments of the NEXT pair live (the next stage's first pair behind this stage's last)
      const bf16_t* nb = nullptr;
      int ncin = cin, ncp_i = cp + 1;
      if (cp + 1 < ncp) nb = wp;
      else if (!last) { ncin = BN; ncp_i = 0; nb = reinterpret_cast<const bf16_t*>(nx.w); }
      // 18 steps (9 taps x 2 chunks); the pixel fragments of step k + 1 are requested before the MFMAs of step k are issued
      // (left to itself hipcc issued read -> wait -> MFMA one at a time: the LDS latency sat in front of every MFMA)
      const unsigned char* X0 = Abuf + (size_t)(2 * cp) * CHB;
      bf16x8_t xfA[4], xfB[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) xfA[i] = *reinterpret_cast<const bf16x8_t*>(X0 + aoff(hbase[i], fq));
#pragma unroll
      for (int k = 0; k < 18; ++k) {
        const int half = k & 1, tap = k >> 1;      // tap major: the two halves of a 128-byte weight line are loaded back to back
        bf16x8_t (&cur)[4] = (k & 1) ? xfB : xfA;
        bf16x8_t (&nxt)[4] = (k & 1) ? xfA : xfB;
        if (k + 1 < 18) {
          const int nh = (k + 1) & 1, nt = (k + 1) >> 1;
#pragma unroll
          for (int i = 0; i < 4; ++i)
            nxt[i] = *reinterpret_cast<const bf16x8_t*>(X0 + (size_t)nh * CHB + aoff(hbase[i] + (nt / 3) * WH + (nt % 3), fq));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wA.t[tap][half], cur[i], acc[i], 0, 0, 0);
        if (nb) wA.t[tap][half] = *reinterpret_cast<const bf16x8_t*>(wptr(nb, ncin, ncp_i, tap, half));
        // pin the step: without this the scheduler sinks each weight re-load down to its use in the next pair (a full memory
        // latency in front of every tap) and undoes the read-ahead of the pixel fragments
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    RB_STAMP(t_c1);
    RB_ADD(1 + 2 * st, t_c0, t_c1);
    // ---- epilogue, in the wave's registers.  Additions in the per-op path's order: (acc + bias) + residual, the shortcut
    // rounded to bf16 first (it was a tensor of its own there).  A lane holds couts c0 .. c0 + 3 of pixels i * 16 + fr.
    const float bb[4] = {bias4.x, bias4.y, bias4.z, bias4.w};
    float hr[4][4];            // the stage's output as a reader of the tensor sees it (rounded to bf16)
    uint2 hp[4];
    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = acc[i][r] + bb[r];
      if (last) {
        o[0] += __uint_as_float(res[i].x << 16); o[1] += __uint_as_float(res[i].x & 0xffff0000u);
        o[2] += __uint_as_float(res[i].y << 16); o[3] += __uint_as_float(res[i].y & 0xffff0000u);
      }
      hp[i].x = idf_pack_bf16(o[0], o[1]);
      hp[i].y = idf_pack_bf16(o[2], o[3]);
      hr[i][0] = __uint_as_float(hp[i].x << 16); hr[i][1] = __uint_as_float(hp[i].x & 0xffff0000u);
      hr[i][2] = __uint_as_float(hp[i].y << 16); hr[i][3] = __uint_as_float(hp[i].y & 0xffff0000u);
#pragma unroll
      for (int r = 0; r < 4; ++r) { ssum[r] += hr[i][r]; ssq[r] += hr[i][r] * hr[i][r]; }
      if (s.h_out) *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(s.h_out) + (size_t)(b * NPX + i * 16 + fr) * BN + c0) = hp[i];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { ssum[r] = row16_sum(ssum[r]); ssq[r] = row16_sum(ssq[r]); }      // over the wave's 64 pixels
    if (last) {
      if (p.st_out && fr == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) reinterpret_cast<float2*>(p.st_out)[(size_t)b * BN + c0 + r] = make_float2(ssum[r], ssq[r]);
      }
      RB_STAMP(t_e1);
      RB_ADD(2 + 2 * st, t_c1, t_e1);
      return;
    }
    // the next stage's GroupNorm: this lane's 4 couts ARE one group (128 channels / 32 groups)
    float mf, rs;
    group_stats((double)ssum[0] + (double)ssum[1] + (double)ssum[2] + (double)ssum[3],
                (double)ssq[0] + (double)ssq[1] + (double)ssq[2] + (double)ssq[3], 1.0 / 256.0, p.eps, &mf, &rs);
    const float ga[4] = {gam4.x, gam4.y, gam4.z, gam4.w}, be[4] = {bet4.x, bet4.y, bet4.z, bet4.w};
    const float t0[4] = {ft0.x, ft0.y, ft0.z, ft0.w}, t1[4] = {ft1.x, ft1.y, ft1.z, ft1.w};
    const float a0[4] = {fa0.x, fa0.y, fa0.z, fa0.w}, a1[4] = {fa1.x, fa1.y, fa1.z, fa1.w};
    float scv[4], shv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float sc = rs * ga[r], sh = be[r] - mf * sc;
      if (nx.film_t) { const float f = 1.f + t0[r]; sc *= f; sh = sh * f + t1[r]; }
      if (nx.film_a) { const float f = 1.f + a0[r]; sc *= f; sh = sh * f + a1[r]; }
      scv[r] = sc; shv[r] = sh;
    }
    if (nx.sc && fr == 0) {
      *reinterpret_cast<float4*>(nx.sc + (size_t)b * BN + c0) = make_float4(scv[0], scv[1], scv[2], scv[3]);
      *reinterpret_cast<float4*>(nx.sh + (size_t)b * BN + c0) = make_float4(shv[0], shv[1], shv[2], shv[3]);
      nx.mean[b * 32 + (c0 >> 2)] = mf; nx.rstd[b * 32 + (c0 >> 2)] = rs;
    }
    const bool drop = nx.drop && drop_any;
    uint2 ap[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned e0 = (unsigned)((b * NPX + i * 16 + fr) * BN + c0);     // index in the dense activated tensor
      const uint32_t h = drop ? idf_vec_hash(seedv, nx.salt, e0 >> 3) : 0u;
      float v[4] = {hr[i][0], hr[i][1], hr[i][2], hr[i][3]};
      idf_act_vec<4>(v, scv, shv, 2, drop, h, (int)(c0 & 7), thr, dscale);
      ap[i].x = idf_pack_bf16(v[0], v[1]);
      ap[i].y = idf_pack_bf16(v[2], v[3]);
      if (nx.a_out) *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(nx.a_out) + e0) = ap[i];
    }
    rb_barrier();              // every wave is past its last read of the activated image
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<uint2*>(a_slot(i * 16 + fr, c0) + (c0 & 4) * 2) = ap[i];
    rb_barrier();              // the next stage's image is in place
    RB_STAMP(t_e2);
    RB_ADD(2 + 2 * st, t_c1, t_e2);
  };
  run_stage(std::integral_constant<int, 0>{});
  run_stage(std::integral_constant<int, 1>{});
  if (p.nstage == 3) run_stage(std::integral_constant<int, 2>{});
}

// =====================================================================================================================
// Image-resident ResBlock BACKWARD for the 8x8 maps: the data-gradient convs of stages n-1 .. first and the GroupNorm / FiLM /
// SiLU / dropout backward behind each (idf_conv_wr_dgrad_gn_bf16's arithmetic) chained in ONE launch, one workgroup per image:
// the gradient a stage hands to the stage before it is written to memory (the weight gradient of that stage's conv reads it)
// AND into the LDS image the next data-gradient conv reads.  Same work split as the forward kernel: wave w owns channels
// 16 w .. 16 w + 15, a lane's 4 channels are one GroupNorm group (128 channels), weights fragment-major into registers.
struct RbBK { IdfResblockBwdArgs a; uint32_t thr; float dscale; };

__global__ __launch_bounds__(NT) void resblock8_bwd_kernel(const RbBK k_in) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const IdfResblockBwdArgs& p = k_in.a;
  unsigned char* const Abuf = smem;                         // [4 chunks][HROWS][WH][PPB]: the current stage's dy
  const int tid = threadIdx.x, lane = tid & 63, b = blockIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (b >= p.B) {                                            // helper workgroups: the stages' weights into this XCD's L2 (see the forward kernel)
    auto warm_all = [&](const void* w) __attribute__((always_inline)) {
      const char* base = reinterpret_cast<const char*>(w);
      for (int off = tid * 128; off < 128 * 9 * BN * 2; off += NT * 128) {
        unsigned d;
        asm volatile("global_load_dword %0, %1, off" : "=v"(d) : "v"(base + off) : "memory");
      }
    };
    if (p.nstage == 3) warm_all(p.s[2].w_frag);
    if (p.nstage >= 2 && p.first <= 1) warm_all(p.s[1].w_frag);
    if (p.first == 0) warm_all(p.s[0].w_frag);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  const int fr = lane & 15, fq = lane >> 4;
  const int c0 = wave * 16 + fq * 4;
  const bool drop_any = p.seed != nullptr;
  const uint64_t seedv = drop_any ? *p.seed : 0;
  const bf16_t* const pdy = reinterpret_cast<const bf16_t*>(p.dy);

  auto wptr = [&](const void* w, int cp, int tap, int half) __attribute__((always_inline)) -> const bf16_t* {
    return reinterpret_cast<const bf16_t*>(w) + ((size_t)((cp * 8 + wave) * 18 + tap * 2 + half) * 64 + lane) * 8;
  };
  auto a_slot = [&](int pl, int c) -> unsigned char* {
    const int h = ((pl >> 3) + 1) * WH + (pl & 7) + 1;
    return Abuf + (size_t)(c >> 5) * CHB + aoff(h, (c & 31) >> 3);
  };
  // dy -> LDS (raw), halo border zeroed once
  u32x4_t dyr[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int v = tid + k * NT, pl = v >> 4, c = (v & 15) * 8;
    dyr[k] = *reinterpret_cast<const u32x4_t*>(pdy + (size_t)(b * NPX + pl) * BN + c);
  }
  WPair wA;
  {
    const void* w0 = p.s[p.nstage - 1].w_frag;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      wA.t[tap][0] = *reinterpret_cast<const bf16x8_t*>(wptr(w0, 0, tap, 0));
      wA.t[tap][1] = *reinterpret_cast<const bf16x8_t*>(wptr(w0, 0, tap, 1));
    }
  }
  for (int i = tid; i < 4 * 36 * 4; i += NT) {
    const int ck = i / 144, r = i - ck * 144, bp = r >> 2, q = r & 3;
    int hy, hx;
    if (bp < 10) { hy = 0; hx = bp; } else if (bp < 20) { hy = 9; hx = bp - 10; }
    else if (bp < 28) { hy = bp - 19; hx = 0; } else { hy = bp - 27; hx = 9; }
    *reinterpret_cast<u32x4_t*>(Abuf + (size_t)ck * CHB + aoff(hy * WH + hx, q)) = u32x4_t{0, 0, 0, 0};
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int v = tid + k * NT, pl = v >> 4, c = (v & 15) * 8;
    *reinterpret_cast<u32x4_t*>(a_slot(pl, c)) = dyr[k];
  }
  rb_barrier();

  const int lbase = ((fr >> 3) * 16 + (fr & 7)) * PPB + fq * 16;
  auto run_stage = [&](auto ST) __attribute__((always_inline)) {
    constexpr int st = decltype(ST)::value;
    if (st >= p.nstage || st < p.first) return;
    const IdfResblockBwdStage& s = p.s[st];
    const IdfResblockBwdStage& nx = p.s[st > 0 ? st - 1 : 0];
    const bool more_stages = st > p.first;
    const bf16_t* const gx = reinterpret_cast<const bf16_t*>(s.x);
    // operands of the epilogue, fetched ahead of the conv
    uint2 gxr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) gxr[i] = *reinterpret_cast<const uint2*>(gx + (size_t)(b * NPX + i * 16 + fr) * BN + c0);
    const float4 sc4 = *reinterpret_cast<const float4*>(s.sc + (size_t)b * BN + c0), sh4 = *reinterpret_cast<const float4*>(s.sh + (size_t)b * BN + c0);
    const float mu = s.mean[b * 32 + (c0 >> 2)], rs = s.rstd[b * 32 + (c0 >> 2)];
    float4 g4 = make_float4(1.f, 1.f, 1.f, 1.f), b4 = make_float4(0.f, 0.f, 0.f, 0.f), t0 = b4, t1 = b4, a0 = b4;
    if (s.gamma) g4 = *reinterpret_cast<const float4*>(s.gamma + c0);
    if (s.beta) b4 = *reinterpret_cast<const float4*>(s.beta + c0);
    if (s.film_t) { t0 = *reinterpret_cast<const float4*>(s.film_t + (size_t)b * s.ld_t + c0); t1 = *reinterpret_cast<const float4*>(s.film_t + (size_t)b * s.ld_t + BN + c0); }
    if (s.film_a) a0 = *reinterpret_cast<const float4*>(s.film_a + (size_t)b * s.ld_a + c0);

    f32x4_t acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int cp = 0; cp < 2; ++cp) {
      const void* nb = nullptr;
      int ncp_i = cp + 1;
      if (cp == 0) nb = s.w_frag;
      else if (more_stages) { nb = nx.w_frag; ncp_i = 0; }
      const unsigned char* X0 = Abuf + (size_t)(2 * cp) * CHB + lbase;
      bf16x8_t xfA[4], xfB[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) xfA[j] = *reinterpret_cast<const bf16x8_t*>(X0 + (size_t)(j * 32) * PPB);
#pragma unroll
      for (int k = 0; k < 18; ++k) {
        const int half = k & 1, tap = k >> 1;
        bf16x8_t (&cur)[4] = (k & 1) ? xfB : xfA;
        bf16x8_t (&nxt)[4] = (k & 1) ? xfA : xfB;
        if (k + 1 < 18) {
          const int nh = (k + 1) & 1, nt = (k + 1) >> 1;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            nxt[j] = *reinterpret_cast<const bf16x8_t*>(X0 + (size_t)nh * CHB + (size_t)(j * 32 + (nt / 3) * WH + (nt % 3)) * PPB);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wA.t[tap][half], cur[j], acc[j], 0, 0, 0);
        if (nb) wA.t[tap][half] = *reinterpret_cast<const bf16x8_t*>(wptr(nb, ncp_i, tap, half));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- GroupNorm backward in the wave's registers (conv_wr_kernel's GNB epilogue, C = 128: a lane's 4 channels = one group)
    const float scv[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, shv[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
    const float ga[4] = {g4.x, g4.y, g4.z, g4.w}, be[4] = {b4.x, b4.y, b4.z, b4.w};
    const float stv[4] = {t0.x, t0.y, t0.z, t0.w}, btv[4] = {t1.x, t1.y, t1.z, t1.w}, sav[4] = {a0.x, a0.y, a0.z, a0.w};
    const bool drop = s.drop && drop_any;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float xv[4] = {__uint_as_float(gxr[i].x << 16), __uint_as_float(gxr[i].x & 0xffff0000u),
                     __uint_as_float(gxr[i].y << 16), __uint_as_float(gxr[i].y & 0xffff0000u)};
      float dav[4] = {acc[i][0], acc[i][1], acc[i][2], acc[i][3]}, du[4];
      const size_t e0 = (size_t)(b * NPX + i * 16 + fr) * BN + c0;
      const uint32_t h = drop ? idf_vec_hash(seedv, s.salt, e0 >> 3) : 0u;
      if (drop) idf_dact_vec_t<4, true, true>(dav, xv, scv, shv, h, (int)(c0 & 7), k_in.thr, k_in.dscale, du);
      else idf_dact_vec_t<4, true, false>(dav, xv, scv, shv, h, (int)(c0 & 7), k_in.thr, k_in.dscale, du);
#pragma unroll
      for (int r = 0; r < 4; ++r) { s1[r] += du[r]; s2[r] += du[r] * xv[r]; acc[i][r] = du[r]; }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[r] = row16_sum(s1[r]); s2[r] = row16_sum(s2[r]); }
    float P1 = 0.f, P2 = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float D1 = s1[r], D2 = rs * (s2[r] - mu * s1[r]);
      const float f = (1.f + stv[r]) * (1.f + sav[r]);
      const float Gf = ga[r] * D2 + be[r] * D1, Ge = D1;
      if (fr == 0) {
        const int c = c0 + r;
        if (s.dfilm_t) { s.dfilm_t[(size_t)b * 2 * BN + c] = Gf * (1.f + sav[r]); s.dfilm_t[(size_t)b * 2 * BN + BN + c] = Ge * (1.f + sav[r]); }
        if (s.dfilm_a) { s.dfilm_a[(size_t)b * 2 * BN + c] = Gf * (1.f + stv[r]) + Ge * btv[r]; s.dfilm_a[(size_t)b * 2 * BN + BN + c] = Ge; }
        if (s.dgb) { s.dgb[((size_t)b * 2 + 0) * BN + c] = f * D2; s.dgb[((size_t)b * 2 + 1) * BN + c] = f * D1; }
        if (s.dgamma_acc) atomicAdd(s.dgamma_acc + c, f * D2);
        if (s.dbeta_acc) atomicAdd(s.dbeta_acc + c, f * D1);
      }
      P1 += ga[r] * f * D1; P2 += ga[r] * f * D2;
    }
    const float invN = 1.f / ((float)NPX * 4);
    const float k1 = -rs * rs * P2 * invN, k0 = (-rs * P1 + rs * rs * mu * P2) * invN;
    uint2 ov[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const size_t e0 = (size_t)(b * NPX + i * 16 + fr) * BN + c0;
      const float xv[4] = {__uint_as_float(gxr[i].x << 16), __uint_as_float(gxr[i].x & 0xffff0000u),
                           __uint_as_float(gxr[i].y << 16), __uint_as_float(gxr[i].y & 0xffff0000u)};
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = scv[r] * acc[i][r] + k1 * xv[r] + k0;
      if (st == 0) {          // the residual branch (identity) and the skip alias join at the block input
        const uint2 rv = *reinterpret_cast<const uint2*>(pdy + e0);
        o[0] += __uint_as_float(rv.x << 16); o[1] += __uint_as_float(rv.x & 0xffff0000u);
        o[2] += __uint_as_float(rv.y << 16); o[3] += __uint_as_float(rv.y & 0xffff0000u);
        if (p.dres2) {
          const uint2 r2 = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(p.dres2) + e0);
          o[0] += __uint_as_float(r2.x << 16); o[1] += __uint_as_float(r2.x & 0xffff0000u);
          o[2] += __uint_as_float(r2.y << 16); o[3] += __uint_as_float(r2.y & 0xffff0000u);
        }
      }
      ov[i].x = idf_pack_bf16(o[0], o[1]);
      ov[i].y = idf_pack_bf16(o[2], o[3]);
      *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(s.dx) + e0) = ov[i];
    }
    if (!more_stages) return;
    rb_barrier();            // every wave is past its last read of this stage's dy image
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<uint2*>(a_slot(i * 16 + fr, c0) + (c0 & 4) * 2) = ov[i];
    rb_barrier();            // the next stage's dy image is in place
  };
  run_stage(std::integral_constant<int, 2>{});
  run_stage(std::integral_constant<int, 1>{});
  run_stage(std::integral_constant<int, 0>{});
}

// =====================================================================================================================
// Per-op 3x3 convs of the small maps (16x16, 8x8) in the same form: a workgroup = 4 waves x 16 couts (a 64-cout tile) over a
// pixel tile of 64 pixels (4 rows of a 16x16 map / a whole 8x8 map) or 256 pixels (a whole 16x16 map); the weights come
// fragment-major straight into registers, the activated input of ALL channel chunks lies in LDS (pitch-96 image, immediates
// for every tap), the conv loop has no workgroup barrier, the epilogue works in the wave's registers:
//   PRO:  y = conv(dropout(SiLU(FiLM(GroupNorm(x | x2))))) + bias (+ res), statistics partials of y   (idf_conv_gn_bf16's job)
//   GNB:  dx = GroupNormBackward(conv(dy, w_dgrad)) (+ dres + dres2) on a whole-image tile           (idf_conv_dgrad_gn_bf16's)
// Replaces the register-staged halo kernel's 64-pixel launches at these levels (19.8 / 13.5 us per launch at B = 32).
struct WrP {
  const bf16_t* x; const bf16_t* x2; int C1, Cin;         // conv input (x | x2), [B, H, W, .]
  const bf16_t* w; int Cout;                              // fragment-major weights, all couts
  int B, H, tiles_per_img, n_tiles;
  // PRO
  const float* st1; const float* st2; int T1, T2;
  const float* gamma; const float* beta; const float* film_t; const float* film_a; int ld_t, ld_a;
  float eps; const uint64_t* seed; uint32_t salt, thr; float dscale;
  bf16_t* a_out; float* mean_out; float* rstd_out; float* sc_out; float* sh_out;
  // plain epilogue
  const float* bias; const bf16_t* res; bf16_t* y; float* st_out;
  // GNB epilogue: y = dx, res / res2 = branch gradients, gamma .. dscale as above describe the GroupNorm being differentiated
  const bf16_t* gx; const bf16_t* res2; const float* gsc; const float* gsh; const float* gmean; const float* grstd;
  float* dfilm_t; float* dfilm_a; float* dgb; float* dgam; float* dbet;
};

template <bool W16, int NF, bool PRO, bool GNB>
__global__ __launch_bounds__(256) void conv_wr_kernel(const WrP p) {
  constexpr int W = W16 ? 16 : 8, WHP = W16 ? 18 : 16, R = NF * 16 / W, HR = R + 2, WH2 = W + 2;
  constexpr int CHBW = HR * WHP * PPB;                      // bytes of one chunk image
  constexpr int FRS = W16 ? WHP : 32;                       // halo pixels between the first pixels of consecutive fragments
  constexpr int PB = NF * 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int Cin = p.Cin, nck = Cin / CK;
  unsigned char* const Abuf = smem;
  float* const cof = reinterpret_cast<float*>(smem + (size_t)nck * CHBW);     // PRO: [Cin][2] | chs [Cin][2]
  float* const chs = cof + 2 * MAXC;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int tile = blockIdx.x / p.n_tiles, n0 = (blockIdx.x - tile * p.n_tiles) * 64;
  const int b = tile / p.tiles_per_img, oy0 = (tile - b * p.tiles_per_img) * R;
  const int c0 = n0 + wave * 16 + fq * 4;                   // the 4 couts this lane holds in the MFMA output
  const int wg = (n0 >> 4) + wave;                          // this wave's 16-cout slice of the fragment-major weights
  const bool drop = (PRO || GNB) && p.seed != nullptr;
  const uint64_t seedv = drop ? *p.seed : 0;

  auto wptr = [&](int cp, int tap, int half) __attribute__((always_inline)) -> const bf16_t* {
    return p.w + ((size_t)((cp * (p.Cout >> 4) + wg) * 18 + tap * 2 + half) * 64 + lane) * 8;
  };
  WPair wA;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    wA.t[tap][0] = *reinterpret_cast<const bf16x8_t*>(wptr(0, tap, 0));
    wA.t[tap][1] = *reinterpret_cast<const bf16x8_t*>(wptr(0, tap, 1));
  }
  // GNB: the GroupNorm input in the MFMA output layout (8 bytes per lane and fragment), fetched ahead of the conv
  uint2 gxr[GNB ? NF : 1];
  if constexpr (GNB) {
#pragma unroll
    for (int i = 0; i < NF; ++i)
      gxr[i] = *reinterpret_cast<const uint2*>(p.gx + ((size_t)tile * PB + i * 16 + fr) * p.Cout + c0);
  }

  // ---- the input tile -> LDS, every chunk.  PRO: statistics fold first, act(x * sc + sh) on the way in.
  if constexpr (PRO) {
    for (int c = tid; c < Cin; c += 256) {
      const float* st = p.st1;
      int T = p.T1, Cs = p.x2 ? p.C1 : Cin, cl = c;
      if (p.x2 && c >= p.C1) { st = p.st2; T = p.T2; Cs = Cin - p.C1; cl = c - p.C1; }
      const float2 S = idf_sum_partials(reinterpret_cast<const float2*>(st) + (size_t)b * T * Cs + cl, T, (size_t)Cs);
      chs[2 * c] = S.x; chs[2 * c + 1] = S.y;
    }
  }
  const int vshift = Cin == 256 ? 5 : (Cin == 128 ? 4 : 3);      // log2(Cin / 8) (Cin 64, 128 or 256)
  const int nitems = (HR * WH2) << vshift;
  auto item = [&](int v, int& hp, int& c, bool& ok, const bf16_t*& src) __attribute__((always_inline)) {
    const int pix = v >> vshift;
    c = (v - (pix << vshift)) * 8;
    const int hy = pix / WH2, hx = pix - hy * WH2;
    hp = hy * WHP + hx;
    const int iy = oy0 + hy - 1, ix = hx - 1;
    ok = v < nitems && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)W;
    const size_t gp = (size_t)(b * p.H + iy) * W + ix;
    src = (p.x2 && c >= p.C1) ? p.x2 + gp * (Cin - p.C1) + (c - p.C1) : p.x + gp * (p.x2 ? p.C1 : Cin) + c;
  };
  constexpr int BATCH = 8;
  u32x4_t xr[BATCH];
  {     // first batch in flight across the fold
#pragma unroll
    for (int k = 0; k < BATCH; ++k) {
      int hp, c; bool ok; const bf16_t* src;
      item(tid + k * 256, hp, c, ok, src);
      xr[k] = ok ? *reinterpret_cast<const u32x4_t*>(src) : u32x4_t{0, 0, 0, 0};
    }
  }
  if constexpr (PRO) {
    rb_barrier();
    const int cpg = Cin >> 5;
    for (int c = tid; c < Cin; c += 256) {
      const int g = c / cpg;
      double a = 0.0, d = 0.0;
      for (int k = g * cpg; k < (g + 1) * cpg; ++k) { a += chs[2 * k]; d += chs[2 * k + 1]; }
      float mf, r;
      group_stats(a, d, 1.0 / ((double)p.H * W * cpg), p.eps, &mf, &r);
      const float ga = p.gamma ? p.gamma[c] : 1.f, be = p.beta ? p.beta[c] : 0.f;
      float sc = r * ga, sh = be - mf * sc;
      if (p.film_t) { const float f = 1.f + p.film_t[(size_t)b * p.ld_t + c]; sc *= f; sh = sh * f + p.film_t[(size_t)b * p.ld_t + Cin + c]; }
      if (p.film_a) { const float f = 1.f + p.film_a[(size_t)b * p.ld_a + c]; sc *= f; sh = sh * f + p.film_a[(size_t)b * p.ld_a + Cin + c]; }
      cof[2 * c] = sc; cof[2 * c + 1] = sh;
      if (p.sc_out && oy0 == 0 && n0 == 0) {
        p.sc_out[(size_t)b * Cin + c] = sc; p.sh_out[(size_t)b * Cin + c] = sh;
        if (c == g * cpg) { p.mean_out[b * 32 + g] = mf; p.rstd_out[b * 32 + g] = r; }
      }
    }
    rb_barrier();
  }
  for (int v0 = 0; v0 < nitems; v0 += BATCH * 256) {
    u32x4_t nx[BATCH];
#pragma unroll
    for (int k = 0; k < BATCH; ++k) {
      int hp, c; bool ok; const bf16_t* src;
      item(v0 + BATCH * 256 + tid + k * 256, hp, c, ok, src);
      nx[k] = ok ? *reinterpret_cast<const u32x4_t*>(src) : u32x4_t{0, 0, 0, 0};
    }
#pragma unroll
    for (int k = 0; k < BATCH; ++k) {
      int hp, c; bool ok; const bf16_t* src;
      const int v = v0 + tid + k * 256;
      item(v, hp, c, ok, src);
      if (v < nitems) {
        u32x4_t o = xr[k];
        if (PRO && ok) {
          float scv[8], shv[8];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 t4 = *reinterpret_cast<const float4*>(cof + 2 * c + 4 * q);
            scv[2 * q] = t4.x; shv[2 * q] = t4.y; scv[2 * q + 1] = t4.z; shv[2 * q + 1] = t4.w;
          }
          const int pix = v >> vshift, hy = pix / WH2, hx = pix - hy * WH2;
          const unsigned e0 = (unsigned)(((b * p.H + oy0 + hy - 1) * W + hx - 1) * Cin + c);
          const uint4 a4 = act_vec(make_uint4(o[0], o[1], o[2], o[3]), scv, shv, drop, seedv, p.salt, p.thr, p.dscale, e0 >> 3);
          o = u32x4_t{a4.x, a4.y, a4.z, a4.w};
          if (p.a_out && n0 == 0 && (unsigned)(hy - 1) < (unsigned)R) *reinterpret_cast<u32x4_t*>(p.a_out + e0) = o;
        }
        *reinterpret_cast<u32x4_t*>(Abuf + (size_t)(c >> 5) * CHBW + aoff(hp, (c & 31) >> 3)) = o;
      }
    }
#pragma unroll
    for (int k = 0; k < BATCH; ++k) xr[k] = nx[k];
  }
  rb_barrier();

  // ---- the conv: 18 steps per 64-channel pair, 4 fragments per sub-step, pixel fragments read one sub-step ahead
  f32x4_t acc[NF];
#pragma unroll
  for (int i = 0; i < NF; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int lbase = (W16 ? fr : ((fr >> 3) * 16 + (fr & 7))) * PPB + fq * 16;
  const int ncp = nck / 2;
  constexpr int NSUB = NF / 4;
  for (int cp = 0; cp < ncp; ++cp) {
    const unsigned char* X0 = Abuf + (size_t)(2 * cp) * CHBW + lbase;
    const bool more = cp + 1 < ncp;
    bf16x8_t xfA[4], xfB[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) xfA[j] = *reinterpret_cast<const bf16x8_t*>(X0 + (size_t)(j * FRS) * PPB);
#pragma unroll
    for (int k = 0; k < 18 * NSUB; ++k) {
      const int step = k / NSUB, sub = k % NSUB, half = step & 1, tap = step >> 1;
      bf16x8_t (&cur)[4] = (k & 1) ? xfB : xfA;
      bf16x8_t (&nxt)[4] = (k & 1) ? xfA : xfB;
      if (k + 1 < 18 * NSUB) {
        const int ns = (k + 1) / NSUB, nsub = (k + 1) % NSUB, nh = ns & 1, nt = ns >> 1;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          nxt[j] = *reinterpret_cast<const bf16x8_t*>(X0 + (size_t)nh * CHBW + (size_t)((nsub * 4 + j) * FRS + (nt / 3) * WHP + (nt % 3)) * PPB);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[sub * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wA.t[tap][half], cur[j], acc[sub * 4 + j], 0, 0, 0);
      if (sub == NSUB - 1 && more) wA.t[tap][half] = *reinterpret_cast<const bf16x8_t*>(wptr(cp + 1, tap, half));
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  const size_t gp0 = (size_t)tile * PB;                      // first pixel of the tile in [B * H * W] (tiles are whole rows)
  if constexpr (!GNB) {
    // ---- plain epilogue: (acc + bias) + residual, rounded; statistics partial of the tile
    const float4 b4 = p.bias ? *reinterpret_cast<const float4*>(p.bias + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      const size_t e = (gp0 + i * 16 + fr) * p.Cout + c0;
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = acc[i][r] + bb[r];
      if (p.res) {
        const uint2 rv = *reinterpret_cast<const uint2*>(p.res + e);
        o[0] += __uint_as_float(rv.x << 16); o[1] += __uint_as_float(rv.x & 0xffff0000u);
        o[2] += __uint_as_float(rv.y << 16); o[3] += __uint_as_float(rv.y & 0xffff0000u);
      }
      uint2 hp2;
      hp2.x = idf_pack_bf16(o[0], o[1]);
      hp2.y = idf_pack_bf16(o[2], o[3]);
      *reinterpret_cast<uint2*>(p.y + e) = hp2;
      const float h0 = __uint_as_float(hp2.x << 16), h1 = __uint_as_float(hp2.x & 0xffff0000u);
      const float h2 = __uint_as_float(hp2.y << 16), h3 = __uint_as_float(hp2.y & 0xffff0000u);
      ssum[0] += h0; ssq[0] += h0 * h0; ssum[1] += h1; ssq[1] += h1 * h1;
      ssum[2] += h2; ssq[2] += h2 * h2; ssum[3] += h3; ssq[3] += h3 * h3;
    }
    if (p.st_out) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { ssum[r] = row16_sum(ssum[r]); ssq[r] = row16_sum(ssq[r]); }
      if (fr == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) reinterpret_cast<float2*>(p.st_out)[(size_t)tile * p.Cout + c0 + r] = make_float2(ssum[r], ssq[r]);
      }
    }
  } else {
    // ---- GroupNorm backward on the accumulator tile (the tile is a whole image): gnb_epilogue's algebra (idf_conv3x3.hip)
    // in the wave's registers.  A lane holds channels c0 .. c0 + 3 of pixels i * 16 + fr; a GroupNorm group is cpg = C / 32
    // channels: this lane's 4 (C = 128), or those of two neighbouring lane groups (C = 256).
    const int C = p.Cout, cpg = C >> 5, g = c0 / cpg;
    const float4 sc4 = *reinterpret_cast<const float4*>(p.gsc + (size_t)b * C + c0), sh4 = *reinterpret_cast<const float4*>(p.gsh + (size_t)b * C + c0);
    const float scv[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, shv[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
    const float mu = p.gmean[b * 32 + g], rs = p.grstd[b * 32 + g];
    float ga[4] = {1.f, 1.f, 1.f, 1.f}, be[4] = {0.f, 0.f, 0.f, 0.f}, stv[4] = {0.f, 0.f, 0.f, 0.f}, btv[4] = {0.f, 0.f, 0.f, 0.f},
          sav[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.gamma) { const float4 t = *reinterpret_cast<const float4*>(p.gamma + c0); ga[0] = t.x; ga[1] = t.y; ga[2] = t.z; ga[3] = t.w; }
    if (p.beta) { const float4 t = *reinterpret_cast<const float4*>(p.beta + c0); be[0] = t.x; be[1] = t.y; be[2] = t.z; be[3] = t.w; }
    if (p.film_t) {
      const float4 t = *reinterpret_cast<const float4*>(p.film_t + (size_t)b * p.ld_t + c0), u = *reinterpret_cast<const float4*>(p.film_t + (size_t)b * p.ld_t + C + c0);
      stv[0] = t.x; stv[1] = t.y; stv[2] = t.z; stv[3] = t.w; btv[0] = u.x; btv[1] = u.y; btv[2] = u.z; btv[3] = u.w;
    }
    if (p.film_a) { const float4 t = *reinterpret_cast<const float4*>(p.film_a + (size_t)b * p.ld_a + c0); sav[0] = t.x; sav[1] = t.y; sav[2] = t.z; sav[3] = t.w; }
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      float xv[4] = {__uint_as_float(gxr[i].x << 16), __uint_as_float(gxr[i].x & 0xffff0000u),
                     __uint_as_float(gxr[i].y << 16), __uint_as_float(gxr[i].y & 0xffff0000u)};
      float dav[4] = {acc[i][0], acc[i][1], acc[i][2], acc[i][3]}, du[4];
      const size_t e0 = (gp0 + i * 16 + fr) * C + c0;
      const uint32_t h = drop ? idf_vec_hash(seedv, p.salt, e0 >> 3) : 0u;
      if (drop) idf_dact_vec_t<4, true, true>(dav, xv, scv, shv, h, (int)(c0 & 7), p.thr, p.dscale, du);
      else idf_dact_vec_t<4, true, false>(dav, xv, scv, shv, h, (int)(c0 & 7), p.thr, p.dscale, du);
#pragma unroll
      for (int r = 0; r < 4; ++r) { s1[r] += du[r]; s2[r] += du[r] * xv[r]; acc[i][r] = du[r]; }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[r] = row16_sum(s1[r]); s2[r] = row16_sum(s2[r]); }
    float P1 = 0.f, P2 = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float D1 = s1[r], D2 = rs * (s2[r] - mu * s1[r]);
      const float f = (1.f + stv[r]) * (1.f + sav[r]);
      const float Gf = ga[r] * D2 + be[r] * D1, Ge = D1;
      if (fr == 0) {
        const int c = c0 + r;
        if (p.dfilm_t) { p.dfilm_t[(size_t)b * 2 * C + c] = Gf * (1.f + sav[r]); p.dfilm_t[(size_t)b * 2 * C + C + c] = Ge * (1.f + sav[r]); }
        if (p.dfilm_a) { p.dfilm_a[(size_t)b * 2 * C + c] = Gf * (1.f + stv[r]) + Ge * btv[r]; p.dfilm_a[(size_t)b * 2 * C + C + c] = Ge; }
        if (p.dgb) { p.dgb[((size_t)b * 2 + 0) * C + c] = f * D2; p.dgb[((size_t)b * 2 + 1) * C + c] = f * D1; }
        if (p.dgam) atomicAdd(p.dgam + c, f * D2);
        if (p.dbet) atomicAdd(p.dbet + c, f * D1);
      }
      P1 += ga[r] * f * D1; P2 += ga[r] * f * D2;
    }
    if (cpg == 8) { P1 += __shfl_xor(P1, 16, 64); P2 += __shfl_xor(P2, 16, 64); }       // the group's other 4 channels
    const float invN = 1.f / ((float)PB * cpg);
    const float k1 = -rs * rs * P2 * invN, k0 = (-rs * P1 + rs * rs * mu * P2) * invN;
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      const size_t e0 = (gp0 + i * 16 + fr) * C + c0;
      const float xv[4] = {__uint_as_float(gxr[i].x << 16), __uint_as_float(gxr[i].x & 0xffff0000u),
                           __uint_as_float(gxr[i].y << 16), __uint_as_float(gxr[i].y & 0xffff0000u)};
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = scv[r] * acc[i][r] + k1 * xv[r] + k0;
      if (p.res) {
        const uint2 rv = *reinterpret_cast<const uint2*>(p.res + e0);
        o[0] += __uint_as_float(rv.x << 16); o[1] += __uint_as_float(rv.x & 0xffff0000u);
        o[2] += __uint_as_float(rv.y << 16); o[3] += __uint_as_float(rv.y & 0xffff0000u);
      }
      if (p.res2) {
        const uint2 rv = *reinterpret_cast<const uint2*>(p.res2 + e0);
        o[0] += __uint_as_float(rv.x << 16); o[1] += __uint_as_float(rv.x & 0xffff0000u);
        o[2] += __uint_as_float(rv.y << 16); o[3] += __uint_as_float(rv.y & 0xffff0000u);
      }
      uint2 ov;
      ov.x = idf_pack_bf16(o[0], o[1]);
      ov.y = idf_pack_bf16(o[2], o[3]);
      *reinterpret_cast<uint2*>(p.y + e0) = ov;
    }
  }
}

template <bool W16, int NF, bool PRO, bool GNB>
int launch_wr(const WrP& p, hipStream_t st) {
  constexpr int WHP = W16 ? 18 : 16, R = NF * 16 / (W16 ? 16 : 8);
  const size_t lds = (size_t)(p.Cin / CK) * (R + 2) * WHP * PPB + (PRO ? (size_t)4 * MAXC * sizeof(float) : 0);
  if (lds > 160 * 1024) return 1;
  auto kern = conv_wr_kernel<W16, NF, PRO, GNB>;
  static IdfLdsGrant grant;
  if (idf_ensure_lds((const void*)kern, lds, grant) != hipSuccess) return 2;
  hipLaunchKernelGGL(kern, dim3(p.B * p.tiles_per_img * p.n_tiles), dim3(256), lds, st, p);
  return 0;
}

// coverage of the forms above: tiles per image (= T of the statistics partials) or 0.  whole != 0: the tile must be the image
int wr_tiles(int B, int H, int W, int Cin, int Cout, int whole) {
  if (B <= 0 || H != W || (W != 8 && W != 16) || (Cin != 64 && Cin != 128 && Cin != 256) || (Cout % 64) || Cout > 256) return 0;
  if ((long)B * H * W * (Cin > Cout ? Cin : Cout) >= (1L << 31)) return 0;
  if (W == 8) return 1;
  if (whole) return 0;        // whole 16x16 images in this form measured slower than the 512-thread halo kernel (25.8 vs 19.7 us,
                              // profiles/r04_conv_wr.txt): the form was removed in round 5
  return 4;
}

}  // namespace

extern "C" int idf_conv_wr_tiles(int B, int H, int W, int Cin, int Cout, int whole) { return wr_tiles(B, H, W, Cin, Cout, whole); }

// y = conv3x3(dropout(SiLU(FiLM(GroupNorm(x | x2))))) + bias (+ res): idf_conv_gn_bf16's contract (act 2, taps 9) on 16x16 / 8x8
// maps with the weights fragment-major (idf_pack_conv_weights_batched's w_frag); st_out [B][idf_conv_wr_tiles(..., 0)][Cout][2].
extern "C" int idf_conv_wr_gn_bf16(const void* x, const void* x2, int C1, const float* st1, int T1, const float* st2, int T2,
                                   const float* gamma, const float* beta, const float* film_t, const float* film_a, int ld_t,
                                   int ld_a, float eps, const uint64_t* seed, uint32_t salt, float p_drop, const void* w_frag,
                                   const float* bias, const void* res, void* y, void* a_out, float* mean, float* rstd, float* sc,
                                   float* sh, float* st_out, int B, int H, int W, int Cin, int Cout, void* stream) {
  const int T = wr_tiles(B, H, W, Cin, Cout, 0);
  if (!T) IDF_FAIL(IDF_ERR_UNSUPPORTED, "conv_wr_gn_bf16: B%d H%d W%d Cin%d Cout%d not covered", B, H, W, Cin, Cout);
  if (!x2) { C1 = Cin; st2 = nullptr; T2 = 0; }
  if (!x || !w_frag || !y || !st1 || T1 < 1 || (x2 && (!st2 || T2 < 1 || C1 <= 0 || C1 >= Cin || (C1 % CK))))
    IDF_FAIL(IDF_ERR_BADARG, "conv_wr_gn_bf16: bad arguments");
  if ((sc != nullptr) != (sh != nullptr) || (sc != nullptr) != (mean != nullptr) || (sc != nullptr) != (rstd != nullptr))
    IDF_FAIL(IDF_ERR_BADARG, "conv_wr_gn_bf16: mean / rstd / sc / sh go together");
  WrP p;
  memset(&p, 0, sizeof(p));
  p.x = (const bf16_t*)x; p.x2 = (const bf16_t*)x2; p.C1 = C1; p.Cin = Cin; p.w = (const bf16_t*)w_frag; p.Cout = Cout;
  p.B = B; p.H = H; p.tiles_per_img = T; p.n_tiles = Cout / 64;
  p.st1 = st1; p.st2 = st2; p.T1 = T1; p.T2 = T2; p.gamma = gamma; p.beta = beta; p.film_t = film_t; p.film_a = film_a;
  p.ld_t = ld_t ? ld_t : 2 * Cin; p.ld_a = ld_a ? ld_a : 2 * Cin; p.eps = eps;
  p.salt = salt; p.thr = idf_drop_thresh(p_drop); p.dscale = 1.0f / (1.0f - (float)p.thr / 65536.0f);
  p.seed = p_drop > 0.f ? seed : nullptr;
  p.a_out = (bf16_t*)a_out; p.mean_out = mean; p.rstd_out = rstd; p.sc_out = sc; p.sh_out = sh;
  p.bias = bias; p.res = (const bf16_t*)res; p.y = (bf16_t*)y; p.st_out = st_out;
  hipStream_t s = (hipStream_t)stream;
  const int rc = W == 16 ? launch_wr<true, 4, true, false>(p, s) : launch_wr<false, 4, true, false>(p, s);
  if (rc) IDF_FAIL(IDF_ERR_HIP, "conv_wr_gn_bf16: LDS request refused (%d)", rc);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}

// dx = GroupNormBackward(conv3x3(dy, w_dgrad)) (+ dres + dres2): idf_conv_dgrad_gn_bf16's contract (act 2, taps 9) on whole
// 16x16 / 8x8 images, w_frag = the data-gradient weights fragment-major; Cin = channels of dy, Cout = channels of x / dx.
extern "C" int idf_conv_wr_dgrad_gn_bf16(const void* dy, const void* w_frag, const void* x, const void* dres, const void* dres2,
                                         void* dx, const float* gamma, const float* beta, const float* film_t,
                                         const float* film_a, int ld_t, int ld_a, const float* mean, const float* rstd,
                                         const float* sc, const float* sh, float* dfilm_t, float* dfilm_a, float* dgb,
                                         float* dgamma_acc, float* dbeta_acc, const uint64_t* seed, uint32_t salt,
                                         float p_drop, int B, int H, int W, int Cin, int Cout, void* stream) {
  if (!wr_tiles(B, H, W, Cin, Cout, 1) || (Cout != 128 && Cout != 256))
    IDF_FAIL(IDF_ERR_UNSUPPORTED, "conv_wr_dgrad_gn_bf16: B%d H%d W%d Cin%d Cout%d not covered", B, H, W, Cin, Cout);
  if (!dy || !w_frag || !x || !dx || !mean || !rstd || !sc || !sh) IDF_FAIL(IDF_ERR_BADARG, "conv_wr_dgrad_gn_bf16: null argument");
  WrP p;
  memset(&p, 0, sizeof(p));
  p.x = (const bf16_t*)dy; p.C1 = Cin; p.Cin = Cin; p.w = (const bf16_t*)w_frag; p.Cout = Cout;
  p.B = B; p.H = H; p.tiles_per_img = 1; p.n_tiles = Cout / 64;
  p.gamma = gamma; p.beta = beta; p.film_t = film_t; p.film_a = film_a;
  p.ld_t = ld_t ? ld_t : 2 * Cout; p.ld_a = ld_a ? ld_a : 2 * Cout;
  p.salt = salt; p.thr = idf_drop_thresh(p_drop); p.dscale = 1.0f / (1.0f - (float)p.thr / 65536.0f);
  p.seed = p_drop > 0.f ? seed : nullptr;
  p.res = (const bf16_t*)dres; p.res2 = (const bf16_t*)dres2; p.y = (bf16_t*)dx;
  p.gx = (const bf16_t*)x; p.gsc = sc; p.gsh = sh; p.gmean = mean; p.grstd = rstd;
  p.dfilm_t = dfilm_t; p.dfilm_a = dfilm_a; p.dgb = dgb; p.dgam = dgamma_acc; p.dbet = dbeta_acc;
  hipStream_t s = (hipStream_t)stream;
  const int rc = launch_wr<false, 4, false, true>(p, s);          // whole images: 8x8 only (wr_tiles)
  if (rc) IDF_FAIL(IDF_ERR_HIP, "conv_wr_dgrad_gn_bf16: LDS request refused (%d)", rc);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}

#ifdef IDF_RB_STAMP
extern "C" int idf_debug_rb_stamps(void** dev_addr) {
  return hipGetSymbolAddress(dev_addr, HIP_SYMBOL(g_rb_stamps)) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int idf_resblock_small_ok(int B, int H, int W, int Cin, int C1, int Cout, int nstage) {
  if (B <= 0 || H != 8 || W != 8 || Cout != BN || (Cin != 128 && Cin != 256) || (nstage != 2 && nstage != 3)) return 0;
  if (C1 && (C1 <= 0 || C1 >= Cin || (C1 % CK))) return 0;
  if ((long)B * NPX * Cin >= (1L << 31)) return 0;
  return 1;
}

extern "C" int idf_resblock_small_fwd(const IdfResblockArgs* args, void* stream) {
  if (!args) IDF_FAIL(IDF_ERR_BADARG, "resblock_small_fwd: null arguments");
  const IdfResblockArgs& p = *args;
  if (!idf_resblock_small_ok(p.B, 8, 8, p.Cin, p.x2 ? p.C1 : 0, BN, p.nstage))
    IDF_FAIL(IDF_ERR_UNSUPPORTED, "resblock_small_fwd: B%d Cin%d (C1 %d) nstage%d not covered (8x8 maps, 128 couts)", p.B, p.Cin,
             p.C1, p.nstage);
  if (!p.x || !p.st1 || p.T1 < 1 || (p.x2 && (!p.st2 || p.T2 < 1)) || !p.y) IDF_FAIL(IDF_ERR_BADARG, "resblock_small_fwd: null tensor");
  if (!p.w_sc && p.Cin != BN) IDF_FAIL(IDF_ERR_BADARG, "resblock_small_fwd: an identity residual needs Cin == 128");
  if (!p.w_sc && p.x2) IDF_FAIL(IDF_ERR_BADARG, "resblock_small_fwd: a two-source input needs the 1x1 shortcut");
  for (int i = 0; i < p.nstage; ++i) {
    const IdfResblockStage& s = p.s[i];
    if (!s.w || !s.bias) IDF_FAIL(IDF_ERR_BADARG, "resblock_small_fwd: stage %d weights missing", i);
    if ((s.sc != nullptr) != (s.sh != nullptr) || (s.sc != nullptr) != (s.mean != nullptr) || (s.sc != nullptr) != (s.rstd != nullptr))
      IDF_FAIL(IDF_ERR_BADARG, "resblock_small_fwd: mean / rstd / sc / sh go together (stage %d)", i);
  }
  if (p.s[p.nstage - 1].h_out != p.y) IDF_FAIL(IDF_ERR_BADARG, "resblock_small_fwd: the last stage's output is y");
  const size_t lds = (size_t)(p.Cin / CK) * CHB + (size_t)(4 * MAXC) * sizeof(float);
  static IdfLdsGrant grant;
  if (hipError_t e = idf_ensure_lds((const void*)resblock8_fwd_kernel, lds, grant); e != hipSuccess)
    IDF_FAIL(IDF_ERR_HIP, "resblock_small_fwd: %zu bytes of LDS refused: %s", lds, hipGetErrorString(e));
  RbK k;
  k.a = p;
  k.thr = idf_drop_thresh(p.p_drop);
  k.dscale = 1.0f / (1.0f - (float)k.thr / 65536.0f);
  if (!(p.p_drop > 0.f)) k.a.seed = nullptr;
  hipLaunchKernelGGL(resblock8_fwd_kernel, dim3(p.B + (p.w_layout == 1 && p.B <= 128 ? IDF_RB_WARM : 0)), dim3(NT), lds, (hipStream_t)stream, k);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}

// Backward of the block's stages nstage-1 .. first in one launch (see resblock8_bwd_kernel): per stage the data-gradient conv with
// fragment-major weights + the GroupNorm backward of idf_conv_wr_dgrad_gn_bf16; stage i's dx is the gradient of stage i-1's conv
// output (written for that conv's weight gradient too).  first == 0 needs a one-source block input with an identity residual.
extern "C" int idf_resblock_small_bwd(const IdfResblockBwdArgs* args, void* stream) {
  if (!args) IDF_FAIL(IDF_ERR_BADARG, "resblock_small_bwd: null arguments");
  const IdfResblockBwdArgs& p = *args;
  if (p.B <= 0 || (p.nstage != 2 && p.nstage != 3) || p.first < 0 || p.first >= p.nstage || (long)p.B * NPX * BN >= (1L << 31))
    IDF_FAIL(IDF_ERR_UNSUPPORTED, "resblock_small_bwd: B%d nstage%d first%d not covered", p.B, p.nstage, p.first);
  if (!p.dy) IDF_FAIL(IDF_ERR_BADARG, "resblock_small_bwd: null dy");
  for (int i = p.first; i < p.nstage; ++i) {
    const IdfResblockBwdStage& s = p.s[i];
    if (!s.w_frag || !s.x || !s.mean || !s.rstd || !s.sc || !s.sh || !s.dx)
      IDF_FAIL(IDF_ERR_BADARG, "resblock_small_bwd: stage %d: null argument", i);
  }
  const size_t lds = (size_t)4 * CHB;
  static IdfLdsGrant grant;
  if (hipError_t e = idf_ensure_lds((const void*)resblock8_bwd_kernel, lds, grant); e != hipSuccess)
    IDF_FAIL(IDF_ERR_HIP, "resblock_small_bwd: %zu bytes of LDS refused: %s", lds, hipGetErrorString(e));
  RbBK k;
  k.a = p;
  k.thr = idf_drop_thresh(p.p_drop);
  k.dscale = 1.0f / (1.0f - (float)k.thr / 65536.0f);
  if (!(p.p_drop > 0.f)) k.a.seed = nullptr;
  hipLaunchKernelGGL(resblock8_bwd_kernel, dim3(p.B + (p.B <= 128 ? IDF_RB_WARM : 0)), dim3(NT), lds, (hipStream_t)stream, k);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}
