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
  for (int i = 0; i < 4; ++i) o[i] = (uint32_t)f32_to_bf16(v[2 * i]) | ((uint32_t)f32_to_bf16(v[2 * i + 1]) << 16);
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
      res[i].x = (uint32_t)f32_to_bf16(sacc[i][0]) | ((uint32_t)f32_to_bf16(sacc[i][1]) << 16);
      res[i].y = (uint32_t)f32_to_bf16(sacc[i][2]) | ((uint32_t)f32_to_bf16(sacc[i][3]) << 16);
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
      hp[i].x = (uint32_t)f32_to_bf16(o[0]) | ((uint32_t)f32_to_bf16(o[1]) << 16);
      hp[i].y = (uint32_t)f32_to_bf16(o[2]) | ((uint32_t)f32_to_bf16(o[3]) << 16);
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
      ap[i].x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
      ap[i].y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
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

}  // namespace
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
  hipLaunchKernelGGL(resblock8_fwd_kernel, dim3(p.B), dim3(NT), lds, (hipStream_t)stream, k);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}
