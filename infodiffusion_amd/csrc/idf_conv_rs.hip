// 3x3 stride-1 convolution of the 64x64 / 32x32 maps (bf16 NHWC, gfx950), "register weights, row reuse" form.
//
// The ResBlock convs of the big maps -- the GroupNorm-prologue forward conv (modules.py:264-268, 283-288, 312-320) and the
// data-gradient conv with the `du` epilogue of its backward -- were a chain of phases per 256-pixel tile in the halo kernels
// (idf_conv3x3.hip): stage 32 channels of the halo tile AND a 36-KB weight slab, barrier, 72 MFMAs per wave, barrier, next
// chunk ..., epilogue; one block per CU, the block's life 14 us for 2 us of matrix work.  This form removes the per-chunk
// structure altogether:
//
//  * WEIGHTS NEVER TOUCH LDS.  Wave w owns couts 16 (w & 3) .. + 15 of the 64-cout tile and keeps ALL its A fragments of a
//    64-channel pair -- 9 taps x 2 chunks x 16 B per lane = 72 VGPRs -- in registers, loaded once per workgroup from the
//    fragment-major shadow (1 KB of consecutive bytes per wave instruction, idf_pack_conv_weights_batched) and kept across the
//    tiles the workgroup walks.  No weight slab staging, no barrier inside the K loop.
//  * THE WHOLE K RANGE OF THE TILE IS RESIDENT: the halo image of every 32-channel chunk lies in LDS at once (64-byte pixel rows,
//    16-byte slots XOR-swizzled by the pixel's column as in idf_conv3x3.hip, so ds_read_b128 of 16 consecutive pixels is
//    conflict-free and every row / chunk displacement is an instruction immediate).
//  * ROW REUSE.  A wave's pixels are a 16-pixel-wide column block over all R rows of the tile.  For an input row it reads the
//    three horizontally shifted B fragments once and uses them for the (up to) three output rows they feed (ky = 0, 1, 2):
//    (R + 2) * 3 LDS reads per R * 9 MFMAs = 0.5 per MFMA at R = 4, 0.42 at R = 8, with one operand per MFMA in registers.
//  * PERSISTENT over consecutive tiles of the same cout tile: the next tile's rows are in flight (registers) during this tile's
//    MFMAs and epilogue and are written into the image right behind the last fragment read; the GroupNorm fold runs once per
//    image and workgroup with its partial sums fetched AHEAD of the row loads (vmcnt is in order: a fold behind the rows waits
//    for every row).
//  * Workgroup barriers order LDS only (`s_waitcnt lgkmcnt(0); s_barrier`): __syncthreads() would drain the prefetch.
//
// Epilogues are the halo kernels' (idf_conv3x3_parts.h): the fp32 tile through LDS, full-line stores, statistics partials of
// the bf16-rounded outputs (forward) or du + (sum du, sum du x) partials (backward chain).  Same tiles (256 pixels = R rows x
// W), same partial layout [B][H / R][Cout][2], same arithmetic per element: results differ from the halo kernels' only by the
// summation order inside the MFMA accumulators.
#include "idf_conv3x3_parts.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Diagnostic build only (tools/build_variant.sh rsstamp idf_conv_rs.hip -DIDF_RS_STAMP; never in the shipped library): wave 0 of
// every workgroup stamps s_memtime at the phase boundaries and adds the differences to g_rs_stamps (tools/rs_stamps.py):
// [0] issue (plan, fold + row loads) [1] fold [2] rows landed, transformed, written [3] first barrier [4] per-tile issue
// [5] MFMA loop [6] barrier behind it [7] accumulators -> LDS + next rows -> image [8] barrier [9] epilogue tail [10] workgroup
// [11] tiles [12] workgroups
#ifdef IDF_RS_STAMP
__device__ unsigned long long g_rs_stamps[64 * 16];
__device__ __forceinline__ unsigned long long rs_now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  return t;
}
#define RS_STAMP(var) const unsigned long long var = rs_now()
#define RS_DECL unsigned long long rs_sum[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0}
#define RS_ADD(i, a, b) rs_sum[i] += (b) - (a)
#define RS_FLUSH do { if (threadIdx.x == 0) for (int i_ = 0; i_ < 16; ++i_) atomicAdd(&g_rs_stamps[(blockIdx.x & 63) * 16 + i_], rs_sum[i_]); } while (0)
#else
#define RS_STAMP(var)
#define RS_DECL
#define RS_ADD(i, a, b)
#define RS_FLUSH
#endif

// NPH = pixel halves per workgroup: 2 -> a 512-thread workgroup owns the whole 256-pixel tile (R rows x W columns), one per CU;
// 1 -> a 256-thread workgroup owns one half-width tile (R rows x W / 2 columns = 128 pixels), TWO independent workgroups per CU:
// while one sits in its epilogue (vector pipe) the other runs its MFMA loop on the same SIMDs, and a barrier stalls four waves,
// not eight.
#ifndef IDF_RS_REUSE
#define IDF_RS_REUSE 1      // (A/B: tools/build_variant.sh noreuse idf_conv_rs.hip -DIDF_RS_REUSE=0)
#endif
// SH ("shared image", Cout = 128 on the 32x32 maps): a workgroup computes BOTH 64-cout tiles of its pixel tile from ONE halo image
// -- the GroupNorm prologue transforms the tile once instead of once per cout tile (the transform is 40 % of such a launch) --
// on half-height tiles (R = 4 rows), so the launch keeps its number of workgroups.
template <int W, int NPH, bool SH = false>
struct RsGeo {
  static constexpr int R = (SH ? 128 : 256) / W, TW = W * NPH / 2, WH = TW + 2, HR = R + 2, NPH_ = NPH, NPHW = HR * WH;
  static constexpr int CHB = NPHW * 64 + ((NPHW * 64) % 128 == 0 ? 64 : 0);     // bytes of one 32-channel chunk image; == 64 (mod 128): the
                                                // two chunks a ds_write_b128 lane group covers fall into different halves of the 32 store banks
  static constexpr int NCB = W / 32;            // 16-pixel column blocks per wave
  static constexpr int TWS = TW == 64 ? 6 : (TW == 32 ? 5 : 4);
  static constexpr bool ALIAS_OS(int cin) {
    return NPH == 1 && (size_t)(cin / 32) * CHB + (size_t)R * TW * 68 * 4 + 4 * NPH * 64 * 8 + (size_t)cin * 16 > 80 * 1024;
  }
  static_assert(W == 64 || W == 32, "64x64 / 32x32 maps");
  static_assert(NPH == 1 || NPH == 2, "one or two pixel halves");
  static_assert(!SH || (W == 32 && NPH == 1), "shared-image form: 32x32 maps, two workgroups per CU");
};

// (mean, rstd) of a group from (sum, sum of squares): idf_resblock.hip's group_stats (mu / var in double, 1 / sqrt as v_rsq_f32 +
// one Newton step in double -- the double-precision divide and square root sequences cost ~1 us in front of the first MFMA)
__device__ __forceinline__ void rs_group_stats(double a, double d, double inv_n, float eps, float* mean, float* rstd) {
  const double mu = a * inv_n;
  double var = d * inv_n - mu * mu;
  if (var < 0.0) var = 0.0;
  const double vd = var + (double)eps;
  const double r0 = (double)__builtin_amdgcn_rsqf((float)vd);
  *rstd = (float)(r0 * (1.5 - 0.5 * vd * r0 * r0));
  *mean = (float)mu;
}

// EPI: 0 plain (bias, residual, statistics partials of y), 2 the backward chain's du epilogue (idf_conv_dgrad_chain_bf16's),
// 3 the du epilogue + the GroupNorm backward itself, GROUP-SYNCHRONISED: the workgroups that hold the tiles of one (image, 64-channel
// slice) publish their (sum du, sum du x) partials, meet at a counter, fold everybody's partials into (K1, K0) and write
// dx = A du + K1 x + K0 (+ dres + dres2) from the du and x they still hold in registers -- du never goes to memory and the streaming
// pass idf_gn_bwd_apply (3 reads + 1 write of the tensor) does not exist.  Workgroups walk the items in lock step (item = round *
// grid + workgroup; the grid is a whole number of groups and resident at once: host), so a group's members always sit in the same
// round; the hand-off is MI355X_MICROARCH.md's counter form: `sc1` partial stores by wave 0, its vmcnt(0), one agent-scope add,
// an `sc1` poll by that wave, the workgroup's barrier, `sc1` loads.  The counter only ever grows (64 per completed group, whatever
// the group size), so no launch has to zero it and a replayed graph needs no per-launch state; the spin is bounded (a workgroup
// that gives up raises *rs_sync_err and the host reports it: results of that launch are garbage, the process does not hang).
template <int W, int CIN, bool PRO, int EPI, int NPH, bool SH = false>
__global__ __launch_bounds__(256 * NPH, 2) void conv_rs_bf16(const C3P p) {
  using G = RsGeo<W, NPH, SH>;
  static_assert(!SH || EPI == 0, "shared-image form: the plain epilogue");
  constexpr int R = G::R, TW = G::TW, WH = G::WH, HR = G::HR, CHB = G::CHB, NCB = G::NCB, TWS = G::TWS;
  constexpr int NT = 256 * NPH, BM = R * TW, BN = 64, KP = CIN / 64, NCH = CIN / 32;
  constexpr int PIECES = CIN / 8, PXK = NT / PIECES, HV = HR * TW * PIECES / NT;     // 16-byte vectors per pixel / pixels per round / rounds
  constexpr int RPK = PXK / TW;                                                       // halo rows per round (1, or 2 for 64 channels at 32x32)
  constexpr int NEV = HR * 2 * PIECES, NE = (NEV + NT - 1) / NT;                      // the two halo columns: vectors, rounds
  constexpr int TWP = NPH == 2 ? 0 : TWS;                                             // the tails' pixel map: whole rows / half-width tiles
  constexpr bool DUE = EPI == 2 || EPI == 3, SYN = EPI == 3;
  // 64x64: a workgroup owns TWO vertically adjacent tiles (B = 32: 1024 half-width tiles on 512 resident workgroups) and waits ONCE,
  // behind the second: the first tile's du waits in LDS (packed, 16 KB), its x is read again at the apply.  (Lock-step rounds with a
  // wait per round measured 50.6 us against 39.5 for conv + apply: the wait in the middle re-aligns the two workgroups of a CU, whose
  // drift is what lets one's MFMA loop run under the other's epilogue.)
  constexpr bool PAIR = SYN && W == 64, LOCK = SYN && !PAIR;
  static_assert(!SYN || NPH == 1, "the synchronised form is built for two workgroups per CU");
  // two workgroups per CU have 80 KB of LDS each: where the chunk images and the fp32 epilogue tile do not fit side by side the
  // tile takes the images' place (the next tile's rows then wait in registers until the epilogue has read it)
  constexpr bool ALIAS = G::ALIAS_OS(CIN);
  static_assert(CIN == 64 || CIN == 128, "one or two 64-channel pairs");
  static_assert(!SH || !G::ALIAS_OS(CIN), "shared-image form: the image must survive the epilogue");
  static_assert(PXK == RPK * TW && HV * RPK == HR, "whole halo rows per staging round");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int cg = wave & 3, ph = wave >> 2;            // cout group of 16, pixel half (NPH = 1: always 0)
  const int wgid = xcd_tile_id(blockIdx.x, gridDim.x);
  const int istep = LOCK ? (int)gridDim.x : 1;        // LOCK: lock-step rounds over the grid; else rs_per consecutive items
  int item = LOCK ? wgid : wgid * p.rs_per;
  const int item_end = LOCK ? p.rs_total : min(item + p.rs_per, p.rs_total);
  int prev_oy0 = -1;                                  // PAIR: the tile whose du waits in the stash
  bool prev_writer = false;
  if (item >= item_end) return;
  bool first_item = true;
  const int TR = p.H / R, halves = p.rs_halves;      // row strips per image; tiles per strip (1 or 2)
  const int npt = p.B * halves * TR;                  // pixel tiles: ((image, half), strip) -- a workgroup's consecutive items are
                                                      // vertically adjacent tiles of one image half

  unsigned char* const Os = smem + p.rs_os_off;
  float* const cof = reinterpret_cast<float*>(smem + p.rs_cof_off);     // PRO: (sc, sh) [CIN][2] | channel sums [CIN][2]
  float* const chs = cof + 2 * CIN;
  C3P pe = p;                                          // the epilogue tails address their scratch relative to the fp32 tile
  pe.aux_off = p.aux_off - p.rs_os_off;

  // ---- this thread's slot in the row staging: round k = halo row(s) RPK k (+ prl), pixel ppx of the tile's TW columns, 8-channel
  // piece tid % PIECES (the same piece in every round: its coefficients stay in registers)
  const int piece = tid & (PIECES - 1), prl = (tid / PIECES) / TW, ppx = (tid / PIECES) % TW;
  const int pch = piece >> 2, pq = piece & 3;
  // ---- this lane's fragment addresses: column blocks x0 = 16 * (NCB * ph + jj), taps kx = 0..2
  int lp[NCB][3];
#pragma unroll
  for (int jj = 0; jj < NCB; ++jj)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int hx = (NCB * ph + jj) * 16 + fr + kx;
      lp[jj][kx] = hx * 64 + ((fq ^ (((hx >> 2) & 1) << 1)) << 4);
    }

  // item -> (cout tile, image, half, strip)
  int nt, b, half, oy0, n0;
  auto decode = [&](int it, int& nt_, int& b_, int& half_, int& oy_, int& n0_) __attribute__((always_inline)) {
    int pt;
    if constexpr (SH) { pt = it >> 1; nt_ = it & 1; }          // the two cout tiles of a pixel tile are consecutive items
    else { nt_ = it / npt; pt = it - nt_ * npt; }
    const int bh = pt / TR;
    oy_ = (pt - bh * TR) * R;
    b_ = halves == 2 ? bh >> 1 : bh; half_ = halves == 2 ? bh & 1 : 0;
    n0_ = nt_ * BN;
  };
  decode(item, nt, b, half, oy0, n0);

  // ---- weights: this wave's 16 couts, one 64-channel pair = 18 fragments
  bf16x8_t Wf[2][9];
  auto wsrc = [&](int kp, int n16, int c, int tap) __attribute__((always_inline)) -> const bf16x8_t* {
    return reinterpret_cast<const bf16x8_t*>(p.w + ((size_t)((kp * (p.Cout >> 4) + n16) * 18 + tap * 2 + c) * 64 + lane) * 8);
  };

  // ---- GroupNorm fold.  Phase 1 (first image of the workgroup): the T1 <= 32 partial sums and this channel's parameters on
  // their way, every load issued before anything waits (clamped addresses + a 0 / 1 factor: no branch, no per-load wait)
  constexpr int FT = 32;
  struct FoldRegs { float2 fpart[PRO ? FT : 1]; float fpar[6]; };
  auto fold_params = [&](int bb, float (&fpar)[6]) __attribute__((always_inline)) {
    fpar[0] = 1.f; fpar[1] = fpar[2] = fpar[3] = fpar[4] = fpar[5] = 0.f;
    if (p.gamma) fpar[0] = p.gamma[tid];
    if (p.beta) fpar[1] = p.beta[tid];
    if (p.film_t) { fpar[2] = p.film_t[(size_t)bb * p.ld_t + tid]; fpar[3] = p.film_t[(size_t)bb * p.ld_t + CIN + tid]; }
    if (p.film_a) { fpar[4] = p.film_a[(size_t)bb * p.ld_a + tid]; fpar[5] = p.film_a[(size_t)bb * p.ld_a + CIN + tid]; }
  };
  auto fold_issue = [&](int bb, FoldRegs& fr_) __attribute__((always_inline)) {
    if constexpr (PRO) {
      if (tid < CIN) {
        const float2* src = reinterpret_cast<const float2*>(p.st1) + (size_t)bb * p.T1 * CIN + tid;
#pragma unroll
        for (int j = 0; j < FT; ++j) fr_.fpart[j] = src[(size_t)min(j, p.T1 - 1) * CIN];
        fold_params(bb, fr_.fpar);
      }
    }
  };
  // phase 2: (sc, sh) per channel into LDS (idf_conv3x3_parts.h's pro_coefficients; the sums in idf_sum_partials' order: batches
  // of 16, two interleaved chains).  fast: the sums come from phase 1's registers; else (the workgroup crosses into another
  // image: rare) a plain loop
  auto fold_finish = [&](int bb, bool writer, bool fast, FoldRegs& fr_) __attribute__((always_inline)) {
    if constexpr (PRO) {
      float (&fpar)[6] = fr_.fpar;
      float2 (&fpart)[PRO ? FT : 1] = fr_.fpart;
      if (tid < CIN) {
        float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
        if (fast) {
#pragma unroll
          for (int j = 0; j < FT; j += 2) {
            const float m0 = j < p.T1 ? 1.f : 0.f, m1 = j + 1 < p.T1 ? 1.f : 0.f;
            s0 += m0 * fpart[j].x; q0 += m0 * fpart[j].y; s1 += m1 * fpart[j + 1].x; q1 += m1 * fpart[j + 1].y;
          }
        } else {
          const float2* src = reinterpret_cast<const float2*>(p.st1) + (size_t)bb * p.T1 * CIN + tid;
#pragma unroll 1
          for (int j = 0; j < p.T1; j += 2) {
            const float2 v0 = src[(size_t)j * CIN], v1 = j + 1 < p.T1 ? src[(size_t)(j + 1) * CIN] : make_float2(0.f, 0.f);
            s0 += v0.x; q0 += v0.y; s1 += v1.x; q1 += v1.y;
          }
          fold_params(bb, fpar);
        }
        chs[2 * tid] = s0 + s1; chs[2 * tid + 1] = q0 + q1;
      }
      lds_barrier();
      if (tid < CIN) {
        constexpr int cpg = CIN >> 5;
        const int c = tid, g = c / cpg;
        double a = 0.0, d = 0.0;
#pragma unroll
        for (int k = 0; k < cpg; ++k) { a += chs[2 * (g * cpg + k)]; d += chs[2 * (g * cpg + k) + 1]; }
        float mf, r;
        rs_group_stats(a, d, 1.0 / ((double)p.H * W * cpg), p.eps, &mf, &r);
        float sc = r * fpar[0], sh = fpar[1] - mf * sc;
        if (p.film_t) { const float f = 1.f + fpar[2]; sc *= f; sh = sh * f + fpar[3]; }
        if (p.film_a) { const float f = 1.f + fpar[4]; sc *= f; sh = sh * f + fpar[5]; }
        cof[2 * c] = sc; cof[2 * c + 1] = sh;
        if (writer && p.sc_out) {
          p.sc_out[(size_t)bb * CIN + c] = sc; p.sh_out[(size_t)bb * CIN + c] = sh;
          if (c == g * cpg) { p.mean_out[bb * 32 + g] = mf; p.rstd_out[bb * 32 + g] = r; }
        }
      }
      lds_barrier();
    }
  };

  // ---- rows of a tile: global -> registers (in flight), registers -> (transform) -> LDS image.  rows[k]: halo row k, the tile's own
  // TW columns; edge[j]: the two halo columns (image border: zero; half-width tiles: the neighbouring half's first / last column)
  u32x4_t rows[HV], edge[NE];
  auto edge_slot = [&](int j, int& ly, int& hx, int& ix, int& epiece, int x0) __attribute__((always_inline)) -> bool {
    const int ev = tid + NT * j;
    epiece = ev & (PIECES - 1);
    const int side = (ev / PIECES) & 1;
    ly = ev / (2 * PIECES);
    hx = side ? WH - 1 : 0;
    ix = x0 + (side ? TW : -1);
    return ev < NEV;
  };
  // reuse: the tile lies right below the one in LDS (same image half): its halo rows 0 and 1 ARE that image's rows R and R + 1 --
  // already transformed -- and move inside LDS (copy_rows); only rows 2 .. R + 1 are fetched and transformed (4 of 6 at 64x64)
  auto issue_rows = [&](int bb, int oy, int x0, bool reuse = false) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < HV; ++k) {
      const int ly = k * RPK + prl, iy = oy + ly - 1;
      rows[k] = u32x4_t{0, 0, 0, 0};
      if ((unsigned)iy < (unsigned)p.H && !(reuse && ly < 2))
        rows[k] = *reinterpret_cast<const u32x4_t*>(p.x + (unsigned)(((bb * p.H + iy) * W + x0 + ppx) * CIN + piece * 8));
    }
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      int ly, hx, ix, ep;
      const bool in = edge_slot(j, ly, hx, ix, ep, x0);
      const int iy = oy + ly - 1;
      edge[j] = u32x4_t{0, 0, 0, 0};
      if (in && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)W && !(reuse && ly < 2))
        edge[j] = *reinterpret_cast<const u32x4_t*>(p.x + (unsigned)(((bb * p.H + iy) * W + ix) * CIN + ep * 8));
    }
  };
  // rows R, R + 1 of every chunk image -> rows 0, 1 (whole rows incl. the halo columns: the swizzle depends on the column only).
  // Every wave is past its last fragment read (the caller's barrier); the barrier inside separates the reads of rows R, R + 1 from
  // the caller's write_rows, which overwrites them
  auto copy_rows = [&]() __attribute__((always_inline)) {
    constexpr int VR = 2 * WH * 4, CV = NCH * VR, NCV = (CV + NT - 1) / NT;      // 16-byte vectors per chunk / in all / per thread
    u32x4_t cp[NCV];
#pragma unroll
    for (int j = 0; j < NCV; ++j) {
      const int idx = tid + j * NT, c = idx / VR, r = idx - c * VR;
      if (idx < CV) cp[j] = *reinterpret_cast<const u32x4_t*>(smem + c * CHB + R * WH * 64 + r * 16);
    }
    lds_barrier();
#pragma unroll
    for (int j = 0; j < NCV; ++j) {
      const int idx = tid + j * NT, c = idx / VR, r = idx - c * VR;
      if (idx < CV) *reinterpret_cast<u32x4_t*>(smem + c * CHB + r * 16) = cp[j];
    }
  };
  uint64_t seedv = 0;
  bool drop = false;
  if (PRO) { drop = p.act == 2 && p.seed != nullptr; if (drop) seedv = *p.seed; }
  auto write_rows_t = [&](auto silu_c, auto drop_c, int bb, int oy, int x0, bool keep_a, bool reuse) __attribute__((always_inline)) {
    constexpr bool SILU = decltype(silu_c)::value, DROP = decltype(drop_c)::value;
    float scv[8], shv[8];
    auto coefs = [&](int pc) __attribute__((always_inline)) {
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const float4 t4 = *reinterpret_cast<const float4*>(cof + 2 * (pc * 8) + 4 * q4);
        scv[2 * q4] = t4.x; shv[2 * q4] = t4.y; scv[2 * q4 + 1] = t4.z; shv[2 * q4 + 1] = t4.w;
      }
    };
    if constexpr (PRO) {
      // the halo columns first (their piece differs from the thread's row piece)
#pragma unroll
      for (int j = 0; j < NE; ++j) {
        int ly, hx, ix, ep;
        const bool in = edge_slot(j, ly, hx, ix, ep, x0);
        const int iy = oy + ly - 1;
        if (in && !(reuse && ly < 2)) {
          u32x4_t v = edge[j];
          if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)W) {
            coefs(ep);
            const unsigned e0 = (unsigned)(((bb * p.H + iy) * W + ix) * CIN + ep * 8);
            const uint4 a4 = pro_vec_t<SILU, DROP>(make_uint4(v[0], v[1], v[2], v[3]), scv, shv, seedv, p.salt, p.thr, p.dscale, e0 >> 3);
            v = u32x4_t{a4.x, a4.y, a4.z, a4.w};
          }
          *reinterpret_cast<u32x4_t*>(smem + (ep >> 2) * CHB + (ly * WH + hx) * 64 + (((ep & 3) ^ (((hx >> 2) & 1) << 1)) << 4)) = v;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      coefs(piece);
    } else {
#pragma unroll
      for (int j = 0; j < NE; ++j) {
        int ly, hx, ix, ep;
        if (edge_slot(j, ly, hx, ix, ep, x0) && !(reuse && ly < 2))
          *reinterpret_cast<u32x4_t*>(smem + (ep >> 2) * CHB + (ly * WH + hx) * 64 + (((ep & 3) ^ (((hx >> 2) & 1) << 1)) << 4)) = edge[j];
      }
    }
#pragma unroll
    for (int k = 0; k < HV; ++k) {
      const int ly = k * RPK + prl, iy = oy + ly - 1, hx = ppx + 1;
      if (reuse && ly < 2) continue;
      u32x4_t v = rows[k];
      if constexpr (PRO) {
        if ((unsigned)iy < (unsigned)p.H) {            // rows outside the image stay zero (the reference pads the ACTIVATED tensor)
          const unsigned e0 = (unsigned)(((bb * p.H + iy) * W + x0 + ppx) * CIN + piece * 8);
          const uint4 a4 = pro_vec_t<SILU, DROP>(make_uint4(v[0], v[1], v[2], v[3]), scv, shv, seedv, p.salt, p.thr, p.dscale, e0 >> 3);
          v = u32x4_t{a4.x, a4.y, a4.z, a4.w};
          if (keep_a && ly >= 1 && ly <= R) *reinterpret_cast<u32x4_t*>(p.a_out + e0) = v;
        }
      }
      *reinterpret_cast<u32x4_t*>(smem + pch * CHB + (ly * WH + hx) * 64 + ((pq ^ (((hx >> 2) & 1) << 1)) << 4)) = v;
      if constexpr (PRO) __builtin_amdgcn_sched_barrier(0);      // one vector's exp / rcp chains at a time: interleaving all HV of them spills
    }
  };
  // the activation / dropout switches are launch-uniform: one branch per tile, straight-line bodies
  auto write_rows = [&](int bb, int oy, int x0, bool keep_a, bool reuse = false) __attribute__((always_inline)) {
    using T = std::true_type;
    using F = std::false_type;
    if (!PRO || p.act != 2) write_rows_t(F{}, F{}, bb, oy, x0, keep_a, reuse);
    else if (drop) write_rows_t(T{}, T{}, bb, oy, x0, keep_a, reuse);
    else write_rows_t(T{}, F{}, bb, oy, x0, keep_a, reuse);
  };

  RS_DECL;
  RS_STAMP(ts0);
  {
  FoldRegs f0;
  if (PRO) fold_issue(b, f0);
  issue_rows(b, oy0, half * TW);
  RS_STAMP(ts1);
  RS_ADD(0, ts0, ts1);
  if (PRO) fold_finish(b, oy0 == 0 && n0 == 0 && half == 0, true, f0);
  RS_STAMP(ts2);
  RS_ADD(1, ts1, ts2);
  }
  RS_STAMP(ts2b);
  // the weights land behind the rows (and behind the fold's registers): their latency hides under the first tile's transform
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) { Wf[0][tap] = *wsrc(0, (n0 >> 4) + cg, 0, tap); Wf[1][tap] = *wsrc(0, (n0 >> 4) + cg, 1, tap); }
  write_rows(b, oy0, half * TW, PRO && p.a_out != nullptr && n0 == 0);
  RS_STAMP(ts3);
  RS_ADD(2, ts2b, ts3);
  lds_barrier();
  RS_STAMP(ts4);
  RS_ADD(3, ts3, ts4);

  // ---- the group-synchronised GroupNorm backward (EPI 3) behind a tile's tail: publish (wave 0 stored the partials), meet the
  // other tiles of (image b, channels n0 .. n0 + 63), fold, apply.  LOCK: per tile, inside the loop; PAIR: once, behind the loop
  constexpr int DNI = BM * (BN / 8) / NT;
  uint4 due_xr[DUE ? DNI : 1];
  float dscv[8], dshv[8];
  uint64_t dseed = 0;
  uint4* const stash = reinterpret_cast<uint4*>(smem + p.rs_stash_off);      // PAIR: [DNI][NT] vectors, thread-private slots
  auto sync_apply = [&](const bool have_prev) __attribute__((always_inline)) {
    if constexpr (SYN) {
      uint4 drr[DNI];                                     // the residual-branch gradient of this thread's vectors
      float gpar[7];                                      // this channel's fold parameters (threads < 64)
      // what the apply reads besides du and x: issued behind the tail (its arithmetic needs every register), in front of the wait
      auto fetch_apply_operands = [&]() __attribute__((always_inline)) {
        const int cc = (tid & 7) * 8;
#pragma unroll
        for (int k = 0; k < DNI; ++k) {
          drr[k] = make_uint4(0, 0, 0, 0);
          if (p.res) drr[k] = *reinterpret_cast<const uint4*>(p.res + (unsigned)(tile_pix<TWP>(pe, b, oy0, (tid + k * NT) >> 3) * p.Cout + n0 + cc));
        }
        if (tid < 64) {
          const GnFoldP& f = p.dyp_f;
          const int c = n0 + tid, g = c / (p.Cout >> 5);
          gpar[0] = f.mean[b * 32 + g]; gpar[1] = f.rstd[b * 32 + g];
          gpar[2] = f.gamma ? f.gamma[c] : 1.f; gpar[3] = f.beta ? f.beta[c] : 0.f;
          gpar[4] = gpar[5] = gpar[6] = 0.f;
          if (f.film_t) { gpar[4] = f.film_t[(size_t)b * f.ld_t + c]; gpar[5] = f.film_t[(size_t)b * f.ld_t + p.Cout + c]; }
          if (f.film_a) gpar[6] = f.film_a[(size_t)b * f.ld_a + c];
        }
      };
      uint4 pxr[PAIR ? DNI : 1], pdr[PAIR ? DNI : 1];          // PAIR: x and the residual gradient of the stashed tile
      auto fetch_prev_operands = [&]() __attribute__((always_inline)) {
        if constexpr (PAIR) {
          if (have_prev) {
            due_fetch_x<BM, BN, NT, TWP>(pe, pxr, b, prev_oy0, n0, BM, tid);
            const int cc = (tid & 7) * 8;
#pragma unroll
            for (int k = 0; k < DNI; ++k) {
              pdr[k] = make_uint4(0, 0, 0, 0);
              if (p.res) pdr[k] = *reinterpret_cast<const uint4*>(p.res + (unsigned)(tile_pix<TWP>(pe, b, prev_oy0, (tid + k * NT) >> 3) * p.Cout + n0 + cc));
            }
          }
        }
      };
      // ---- publish (wave 0 stored the partials), meet the other tiles of (image b, channels n0 .. n0 + 63)
      const GnFoldP& f = p.dyp_f;
      const int C = p.Cout, T = p.tiles_per_img;
      if (wave == 0) {
        auto* cnt = reinterpret_cast<__attribute__((address_space(1))) unsigned*>(reinterpret_cast<uintptr_t>(p.rs_sync + b * p.n_tiles + nt));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(cnt, (unsigned)(64 / T) * (have_prev ? 2u : 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        fetch_apply_operands();
        fetch_prev_operands();
        const unsigned target = ((unsigned)__builtin_amdgcn_readfirstlane((int)old) & ~63u) + 64u;
        unsigned spins = 0;
        while ((int)(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > p.rs_spin_max) { if (lane == 0) atomicAdd(p.rs_sync_err, 1u); break; }
        }
      } else { fetch_apply_operands(); fetch_prev_operands(); }
      lds_barrier();
      // ---- fold: S1, S2 of this channel over the group's T tiles (thread = channel x tile quarter; fixed order), then the
      // per-channel / per-group arithmetic of gn_bwd_fold (idf_gnfold.h)
      float* red = reinterpret_cast<float*>(smem + p.aux_off);        // [4][64][2] | pc [64][2] | kk [64][2] (du stays in the fp32 tile's place)
      float* pc = red + 512;
      float* kk = pc + 128;
      {
        const int c = tid & 63, tq = tid >> 6;
        auto* src = reinterpret_cast<__attribute__((address_space(1))) const unsigned*>(
            reinterpret_cast<uintptr_t>(p.st_out + (((size_t)b * T) * C + n0 + c) * 2));
        unsigned vs[8], vq[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int t = min(tq + 4 * i, T - 1);
          vs[i] = __hip_atomic_load(src + (size_t)t * C * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          vq[i] = __hip_atomic_load(src + (size_t)t * C * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        float s_ = 0.f, q_ = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const bool in = tq + 4 * i < T;
          s_ += in ? __uint_as_float(vs[i]) : 0.f; q_ += in ? __uint_as_float(vq[i]) : 0.f;
        }
        red[(tq * 64 + c) * 2] = s_; red[(tq * 64 + c) * 2 + 1] = q_;
      }
      lds_barrier();
      const int cpg = C >> 5;
      if (tid < 64) {
        const int cl = tid, c = n0 + cl;
        const float S1 = (red[cl * 2] + red[(64 + cl) * 2]) + (red[(128 + cl) * 2] + red[(192 + cl) * 2]);
        const float S2 = (red[cl * 2 + 1] + red[(64 + cl) * 2 + 1]) + (red[(128 + cl) * 2 + 1] + red[(192 + cl) * 2 + 1]);
        const float mu = gpar[0], r = gpar[1], ga = gpar[2], be = gpar[3], st = gpar[4], bt = gpar[5], sa = gpar[6];
        const float D1 = S1, D2 = r * (S2 - mu * S1);
        const float fm = (1.f + st) * (1.f + sa);
        if (pe.rs_tidx == 0 || prev_writer) {             // one workgroup per group stores the parameter / FiLM gradients
          const float Gf = ga * D2 + be * D1, Ge = D1;
          if (f.dfilm_t) { f.dfilm_t[(size_t)b * 2 * C + c] = Gf * (1.f + sa); f.dfilm_t[(size_t)b * 2 * C + C + c] = Ge * (1.f + sa); }
          if (f.dfilm_a) { f.dfilm_a[(size_t)b * 2 * C + c] = Gf * (1.f + st) + Ge * bt; f.dfilm_a[(size_t)b * 2 * C + C + c] = Ge; }
          if (f.dgb) { f.dgb[((size_t)b * 2 + 0) * C + c] = fm * D2; f.dgb[((size_t)b * 2 + 1) * C + c] = fm * D1; }
          if (f.dgam) atomicAdd(f.dgam + c, fm * D2);
          if (f.dbet) atomicAdd(f.dbet + c, fm * D1);
        }
        pc[2 * cl] = ga * fm * D1; pc[2 * cl + 1] = ga * fm * D2;
      }
      lds_barrier();
      if (tid < 64) {
        const int cl = tid, gl = cl / cpg;
        float P1 = 0.f, P2 = 0.f;
        for (int k = gl * cpg; k < (gl + 1) * cpg; ++k) { P1 += pc[2 * k]; P2 += pc[2 * k + 1]; }
        const float mu = gpar[0], r = gpar[1], invN = 1.f / ((float)f.HW * cpg);
        kk[2 * cl] = -r * r * P2 * invN;
        kk[2 * cl + 1] = (-r * P1 + r * r * mu * P2) * invN;
      }
      lds_barrier();
      // ---- dx = A du + K1 x + K0 (+ dres + dres2): gn_bwd_apply_loop's arithmetic on registers
      {
        const int cc = (tid & 7) * 8;
        float k1v[8], k0v[8];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const float4 t4 = *reinterpret_cast<const float4*>(kk + 2 * cc + 4 * q4);
          k1v[2 * q4] = t4.x; k0v[2 * q4] = t4.y; k1v[2 * q4 + 1] = t4.z; k0v[2 * q4 + 1] = t4.w;
        }
        bf16_t* dst = p.y;
        int opitch = C, oc = n0 + cc;
        if (p.due_x2) {
          if (n0 < p.due_C1) opitch = p.due_C1;
          else { dst = p.rs_dx2; opitch = C - p.due_C1; oc -= p.due_C1; }
        }
        auto apply_tile = [&](int oy, const uint4 (&xs)[DNI], const uint4 (&rs)[DNI], bool stashed) __attribute__((always_inline)) {
#pragma unroll
          for (int k = 0; k < DNI; ++k) {
            const int pl = (tid + k * NT) >> 3, pix = tile_pix<TWP>(pe, b, oy, pl);
            float o[8];
            const uint4 duk = stashed ? stash[k * NT + tid] : *reinterpret_cast<const uint4*>(reinterpret_cast<const float*>(Os) + pl * (BN + 4) + cc);
            const uint32_t xw[4] = {xs[k].x, xs[k].y, xs[k].z, xs[k].w}, dw[4] = {duk.x, duk.y, duk.z, duk.w};
            const uint32_t rw[4] = {rs[k].x, rs[k].y, rs[k].z, rs[k].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float x0 = __uint_as_float(xw[i] << 16), x1 = __uint_as_float(xw[i] & 0xffff0000u);
              const float d0 = __uint_as_float(dw[i] << 16), d1 = __uint_as_float(dw[i] & 0xffff0000u);
              o[2 * i] = dscv[2 * i] * d0 + k1v[2 * i] * x0 + k0v[2 * i] + __uint_as_float(rw[i] << 16);
              o[2 * i + 1] = dscv[2 * i + 1] * d1 + k1v[2 * i + 1] * x1 + k0v[2 * i + 1] + __uint_as_float(rw[i] & 0xffff0000u);
            }
            if (p.gnb_res2) {
              const uint4 r2 = *reinterpret_cast<const uint4*>(p.gnb_res2 + (unsigned)(pix * C + n0 + cc));
              const uint32_t r2w[4] = {r2.x, r2.y, r2.z, r2.w};
#pragma unroll
              for (int i = 0; i < 4; ++i) { o[2 * i] += __uint_as_float(r2w[i] << 16); o[2 * i + 1] += __uint_as_float(r2w[i] & 0xffff0000u); }
            }
            *reinterpret_cast<uint4*>(dst + (unsigned)(pix * opitch + oc)) =
                make_uint4(idf_pack_bf16(o[0], o[1]), idf_pack_bf16(o[2], o[3]), idf_pack_bf16(o[4], o[5]), idf_pack_bf16(o[6], o[7]));
          }
        };
        apply_tile(oy0, due_xr, drr, false);
        if constexpr (PAIR) { if (have_prev) apply_tile(prev_oy0, pxr, pdr, true); }
      }
    }
  };

  for (;;) {
    RS_STAMP(tt0);
    const bool has_next = item + istep < item_end;
    int nnt = nt, nb = b, nhalf = half, noy0 = oy0, nn0 = n0;
    if (has_next) decode(item + istep, nnt, nb, nhalf, noy0, nn0);
    pe.rs_x0 = half * TW; pe.rs_tidx = (oy0 / R) * halves + half;
    // ---- what the epilogue reads from memory, then the next tile's rows (vmcnt is in order: the epilogue's operands first)
    if constexpr (DUE) {
      due_fetch_x<BM, BN, NT, TWP>(pe, due_xr, b, oy0, n0, BM, tid);
      due_fetch_coef<BN>(p, b, n0, tid, dscv, dshv, dseed);
    }
    const bool refold = PRO && has_next && nb != b;       // the workgroup crosses into the next image (never at one or two tiles per CU)
    // (ALIAS && SYN: the rows would sit in registers across the tail, the wait and the apply -- 40 of them at 128 channels; that
    // form runs one round at the training shapes, so its next tile, when there is one, is fetched behind the apply instead)
    // the next tile right below this one in the same image half (and no a_out to write for its first row): rows 0, 1 come from LDS
    const bool keep_next = PRO && p.a_out != nullptr && nn0 == 0;
    const bool same_img = SH && has_next && (item >> 1) == ((item + 1) >> 1);     // SH: the other cout tile of the image in LDS
    const bool reuse = IDF_RS_REUSE && !SH && !ALIAS && has_next && nb == b && nhalf == half && noy0 == oy0 + R && !keep_next && !refold;
    if (has_next && !(ALIAS && SYN) && !same_img) issue_rows(nb, noy0, nhalf * TW, reuse);
    RS_STAMP(tt1);
    RS_ADD(4, tt0, tt1);

    // ---- the conv: per 64-channel pair 2 chunks x NCB column blocks x HR input rows; per step 3 fragment reads (one row,
    // kx = 0..2; read one step ahead) and up to 9 MFMAs (the output rows the input row feeds)
    f32x4_t acc[NCB * R];
#pragma unroll
    for (int i = 0; i < NCB * R; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kp = 0; kp < KP; ++kp) {
      const unsigned char* X0 = smem + kp * 2 * CHB;
      constexpr int NS = 2 * NCB * HR;
      bf16x8_t xa[3], xb[3];
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) xa[kx] = *reinterpret_cast<const bf16x8_t*>(X0 + lp[0][kx]);
      // the weights this wave needs next: the following pair of this tile, else the first pair of the next tile when it differs
      const bool wmore = kp + 1 < KP || (has_next && (KP > 1 || nn0 != n0));
      const int wkp = kp + 1 < KP ? kp + 1 : 0, wn16 = ((kp + 1 < KP ? n0 : nn0) >> 4) + cg;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int c = s / (NCB * HR), jj = (s / HR) % NCB, ly = s % HR;
        bf16x8_t (&cur)[3] = (s & 1) ? xb : xa;
        bf16x8_t (&nxt)[3] = (s & 1) ? xa : xb;
        if (s + 1 < NS) {
          const int nc = (s + 1) / (NCB * HR), njj = ((s + 1) / HR) % NCB, nly = (s + 1) % HR;
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) nxt[kx] = *reinterpret_cast<const bf16x8_t*>(X0 + nc * CHB + nly * WH * 64 + lp[njj][kx]);
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int orow = ly - ky;
          if (orow < 0 || orow >= R) continue;
#pragma unroll
          for (int kx = 0; kx < 3; ++kx)
            acc[jj * R + orow] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[c][ky * 3 + kx], cur[kx], acc[jj * R + orow], 0, 0, 0);
        }
        if (s == NCB * HR - 1 || s == NS - 1) {        // chunk c's fragments are through: their registers take the next pair's
          if (wmore) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) Wf[c][tap] = *wsrc(wkp, wn16, c, tap);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    // ---- epilogue: accumulators -> fp32 tile in LDS; the next tile's rows -> image (every wave is past its last fragment read)
    RS_STAMP(tt2);
    RS_ADD(5, tt1, tt2);
    lds_barrier();
    RS_STAMP(tt3);
    RS_ADD(6, tt2, tt3);
    {
      constexpr int PF = BN + 4;
      float* O = reinterpret_cast<float*>(Os);
#pragma unroll
      for (int jj = 0; jj < NCB; ++jj)
#pragma unroll
        for (int orow = 0; orow < R; ++orow) {
          const int pl = orow * TW + (NCB * ph + jj) * 16 + fr;
          const f32x4_t a = acc[jj * R + orow];
          *reinterpret_cast<float4*>(O + pl * PF + cg * 16 + fq * 4) = make_float4(a[0], a[1], a[2], a[3]);
        }
    }
    if (!ALIAS && has_next && !same_img) {
      if (refold) { FoldRegs f1; fold_finish(nb, noy0 == 0 && nn0 == 0 && nhalf == 0, false, f1); }
      if (reuse) copy_rows();
      write_rows(nb, noy0, nhalf * TW, keep_next, reuse);
    }
    RS_STAMP(tt4);
    RS_ADD(7, tt3, tt4);
    lds_barrier();
    RS_STAMP(tt5);
    RS_ADD(8, tt4, tt5);
    if constexpr (SYN) {
      due_epilogue_tail<BM, BN, NT, true, TWP, true>(pe, Os, b, oy0, n0, BM, tid, due_xr, dscv, dshv, dseed);
      if constexpr (LOCK) sync_apply(false);
      if constexpr (PAIR) {
        if (has_next) {
          // first tile of the pair: its partials are stored (published with the second tile's), its du moves out of the fp32 tile
#pragma unroll
          for (int k = 0; k < DNI; ++k)
            stash[k * NT + tid] = *reinterpret_cast<const uint4*>(reinterpret_cast<const float*>(Os) + ((tid + k * NT) >> 3) * (BN + 4) + (tid & 7) * 8);
          prev_oy0 = oy0; prev_writer = pe.rs_tidx == 0;
        }
      }
    } else if constexpr (DUE) {
      due_epilogue_tail<BM, BN, NT, true, TWP>(pe, Os, b, oy0, n0, BM, tid, due_xr, dscv, dshv, dseed);
    } else {
      uint4 none[(BM * (BN / 8) + NT - 1) / NT];
      lds_epilogue_tail<BM, BN, NT, true, TWP>(pe, Os, b, oy0, n0, BM, tid, none, false);
    }
    RS_STAMP(tt6);
    RS_ADD(9, tt5, tt6);
    RS_ADD(10, (first_item ? ts0 : tt0), tt6);
    first_item = false;
#ifdef IDF_RS_STAMP
    rs_sum[11] += 1;
#endif
    if (!has_next) break;
    if constexpr (ALIAS) {
      if constexpr (SYN) issue_rows(nb, noy0, nhalf * TW);
      lds_barrier();                                   // the fp32 tile has been read: the images' place is free again
      if (refold) { FoldRegs f1; fold_finish(nb, noy0 == 0 && nn0 == 0 && nhalf == 0, false, f1); }
      write_rows(nb, noy0, nhalf * TW, PRO && p.a_out != nullptr && nn0 == 0);
      lds_barrier();
    }
    item += istep; nt = nnt; b = nb; half = nhalf; oy0 = noy0; n0 = nn0;
  }
  if constexpr (PAIR) sync_apply(prev_oy0 >= 0);      // (behind the loop: nothing of it -- weights, the next rows -- is live across the wait)
  RS_FLUSH;
}

#ifdef IDF_RS_STAMP
}  // namespace
extern "C" int idf_debug_rs_stamps(void** dev_addr) {
  return hipGetSymbolAddress(dev_addr, HIP_SYMBOL(g_rs_stamps)) == hipSuccess ? 0 : 1;
}
namespace {
#endif

// ------------------------------------------------------------------------------------------------------------------- host
#define g_rs (idf_knobs().conv_rs)          // 0 off, 1 two 256-thread workgroups per CU, 2 one of 512
const int g_rs_min = 128;     // work items below which the launch leaves most CUs idle

// tiles per image (= T of the statistics partials) or 0 when the form does not cover the shape
int rs_tiles(int B, int H, int W, int Cin, int Cout) {
  if (!g_rs || B <= 0 || H != W || (W != 64 && W != 32) || (Cout % 64) || Cout <= 0) return 0;
  if (!((W == 64 && Cin == 64) || (W == 32 && (Cin == 128 || Cin == 64)))) return 0;
  if ((long)B * H * W * (Cin > Cout ? Cin : Cout) >= (1L << 31) || (long)Cout * 9 * Cin >= (1L << 31)) return 0;
  const int T = H / (256 / W);
  if ((long)B * T * (Cout / 64) < g_rs_min) return 0;
  return g_rs == 2 ? T : 2 * T;
}

// the forward GroupNorm-prologue conv takes the shared-image form (half-height tiles, both cout tiles per workgroup) where the map is
// 32x32 and Cout = 128: T doubles
bool rs_fwd_shared(int W, int Cout) { return g_rs == 1 && W == 32 && Cout == 128; }
int rs_fwd_tiles(int B, int H, int W, int Cin, int Cout) {
  const int T = rs_tiles(B, H, W, Cin, Cout);
  return (T && rs_fwd_shared(W, Cout)) ? 2 * T : T;
}

int rs_ncu() {
  // per device id (a process may drive more than one device; the count sizes the resident grid of the synchronised form)
  static int ncu[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (!ncu[dev]) {
    hipDeviceProp_t prop;
    ncu[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  return ncu[dev];
}

// the group-synchronised form's state, owned by the caller (the library allocates nothing): RS_SYNC_MAX counters -- one per
// (image, 64-channel slice) of a launch, zero when first handed in, only ever growing -- and behind them the error word; launches
// that share a state array must not overlap (the package keeps one per device and issues its compute on one stream)
constexpr int RS_SYNC_MAX = 8192;
unsigned g_rs_spin_max = 1u << 21;   // polls a workgroup spends on its group before it gives up (idf_conv_rs_set_spin_limit)

template <int W, int CIN, bool PRO, int EPI, int NPH, bool SH = false>
int launch_rs(C3P& p, hipStream_t st, bool probe = false) {
  using G = RsGeo<W, NPH, SH>;
  constexpr int halves = 3 - NPH;
  p.R = G::R; p.rs_halves = halves; p.tiles_per_img = (p.H / G::R) * halves; p.n_tiles = p.Cout / 64;
  p.rs_total = p.B * p.tiles_per_img * p.n_tiles;
  const int slots = rs_ncu() * halves;               // workgroups resident at once
  p.rs_per = idf_cdiv(p.rs_total, slots);
  if (SH) { if (p.n_tiles != 2) return 8; p.rs_per += p.rs_per & 1; }        // both cout tiles of a pixel tile in one workgroup
  int grid = idf_cdiv(p.rs_total, p.rs_per);
  if (EPI == 3) {
    if (W == 64) {
      // pairs: a workgroup's (at most two) consecutive items are vertically adjacent tiles of one image half; the grid is resident
      if (p.rs_per > 2 || (p.rs_per == 2 && ((p.H / G::R) & 1))) return 4;
    } else {
      // lock-step rounds: every workgroup of a group (tiles_per_img items) in the same round, the whole grid resident
      grid = p.rs_total < slots ? p.rs_total : slots;
      if (grid % p.tiles_per_img) return 4;
    }
    // ... and a 64-channel slice must hold whole GroupNorm groups (Cout / 32 channels each: not at 192 channels)
    if (64 % p.tiles_per_img || 64 % (p.Cout >> 5) || p.B * p.n_tiles > RS_SYNC_MAX) return 4;
    if (!probe && !p.rs_sync) return 5;
    p.rs_sync_err = p.rs_sync + RS_SYNC_MAX;
    p.rs_spin_max = g_rs_spin_max;
  }
  size_t lds = (size_t)(CIN / 32) * G::CHB;
  const size_t osz = (size_t)G::R * G::TW * (64 + 4) * sizeof(float);
  if (G::ALIAS_OS(CIN)) { p.rs_os_off = 0; if (osz > lds) lds = osz; }
  else { p.rs_os_off = (int)lds; lds += osz; }
  p.aux_off = (int)lds;
  lds += (size_t)4 * NPH * 64 * 8 + (EPI == 3 ? 1024 : 0);   // wave partials of the statistics (EPI 3: + the fold's scratch)
  p.rs_stash_off = (int)lds;
  if (EPI == 3 && W == 64) lds += (size_t)G::R * G::TW * 64 * 2;   // the first tile's du of a pair (packed)
  p.rs_cof_off = (int)lds;
  if (PRO) lds += (size_t)CIN * 16;
  if (lds > 160 * 1024 / halves) return 1;
  auto kern = conv_rs_bf16<W, CIN, PRO, EPI, NPH, SH>;
  static IdfLdsGrant grant;
  if (idf_ensure_lds((const void*)kern, lds, grant) != hipSuccess) return 2;
  if (EPI == 3) {
    static const int resident = [&] {
      int n = 0;
      return hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)kern, 256 * NPH, lds) == hipSuccess ? n : 0;
    }();
    if (resident < halves) return 6;                 // the spin needs the whole grid on the chip
  }
  if (probe) return 0;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256 * NPH), lds, st, p);
  return 0;
}

int dispatch_rs_sync(C3P& p, hipStream_t st, bool probe) {
  if (g_rs != 1) return 3;
  if (p.W == 64 && p.Cin == 64) return launch_rs<64, 64, false, 3, 1>(p, st, probe);
  if (p.W == 32 && p.Cin == 128) return launch_rs<32, 128, false, 3, 1>(p, st, probe);
  if (p.W == 32 && p.Cin == 64) return launch_rs<32, 64, false, 3, 1>(p, st, probe);
  return 3;
}

template <bool PRO, int EPI>
int dispatch_rs(C3P& p, hipStream_t st) {
  if (g_rs == 2) {
    if (p.W == 64 && p.Cin == 64) return launch_rs<64, 64, PRO, EPI, 2>(p, st);
    if (p.W == 32 && p.Cin == 128) return launch_rs<32, 128, PRO, EPI, 2>(p, st);
    if (p.W == 32 && p.Cin == 64) return launch_rs<32, 64, PRO, EPI, 2>(p, st);
    return 3;
  }
  if (p.W == 64 && p.Cin == 64) return launch_rs<64, 64, PRO, EPI, 1>(p, st);
  if (p.W == 32 && p.Cin == 128) return launch_rs<32, 128, PRO, EPI, 1>(p, st);
  if (p.W == 32 && p.Cin == 64) return launch_rs<32, 64, PRO, EPI, 1>(p, st);
  return 3;
}

}  // namespace

// Tiles per image (= T of the statistics / du partials the entry points below write: [B][T][Cout][2]) when the row-reuse form
// covers a stride-1 3x3 conv of this shape, else 0: square 64x64 maps with 64 input channels, 32x32 maps with 64 or 128,
// Cout % 64 == 0, at least IDF_CONV_RS_MIN (128) work items of 256 pixels x 64 couts.
extern "C" int idf_conv_rs_tiles(int B, int H, int W, int Cin, int Cout) { return rs_tiles(B, H, W, Cin, Cout); }
// ... and T of the statistics partials idf_conv_rs_gn_bf16 writes (the forward conv's tiles are half as high where it computes both cout
// tiles of a pixel tile from one halo image: 32x32 maps, Cout = 128)
extern "C" int idf_conv_rs_fwd_tiles(int B, int H, int W, int Cin, int Cout) { return rs_fwd_tiles(B, H, W, Cin, Cout); }

// y = conv3x3(dropout(SiLU(FiLM(GroupNorm(x))))) + bias (+ res): idf_conv_gn_bf16's contract (taps 9, one source) with the weights
// fragment-major (idf_pack_conv_weights_batched's w_frag: [Cin / 64][Cout / 16][tap][half][lane][8]).  T1 <= 32.
extern "C" int idf_conv_rs_gn_bf16(const void* x, const float* st1, int T1, const float* gamma, const float* beta, const float* film_t,
                                   const float* film_a, int ld_t, int ld_a, float eps, int act, const uint64_t* seed, uint32_t salt,
                                   float p_drop, const void* w_frag, const float* bias, const void* res, void* y, void* a_out,
                                   float* mean, float* rstd, float* sc, float* sh, float* st_out, int B, int H, int W, int Cin,
                                   int Cout, void* stream) {
  if (!rs_tiles(B, H, W, Cin, Cout) || T1 < 1 || T1 > 32)
    IDF_FAIL(IDF_ERR_UNSUPPORTED, "conv_rs_gn_bf16: B%d H%d W%d Cin%d Cout%d T1 %d not covered", B, H, W, Cin, Cout, T1);
  if (!x || !st1 || !w_frag || !y) IDF_FAIL(IDF_ERR_BADARG, "conv_rs_gn_bf16: null argument");
  if (act != 1 && act != 2) IDF_FAIL(IDF_ERR_BADARG, "conv_rs_gn_bf16: act must be 1 or 2");
  if ((sc != nullptr) != (sh != nullptr) || (sc != nullptr) != (mean != nullptr) || (sc != nullptr) != (rstd != nullptr))
    IDF_FAIL(IDF_ERR_BADARG, "conv_rs_gn_bf16: mean / rstd / sc / sh go together");
  C3P p;
  memset(&p, 0, sizeof(p));
  p.x = (const bf16_t*)x; p.C1 = Cin; p.w = (const bf16_t*)w_frag; p.bias = bias; p.res = (const bf16_t*)res; p.y = (bf16_t*)y;
  p.B = B; p.H = H; p.W = W; p.Hs = H; p.Ws = W; p.Cin = Cin; p.Cout = Cout;
  p.st_out = st_out; p.st1 = st1; p.T1 = T1;
  p.gamma = gamma; p.beta = beta; p.film_t = film_t; p.film_a = film_a;
  p.ld_t = ld_t ? ld_t : 2 * Cin; p.ld_a = ld_a ? ld_a : 2 * Cin; p.eps = eps;
  p.act = act; p.salt = salt; p.thr = idf_drop_thresh(p_drop);
  p.dscale = 1.0f / (1.0f - (float)p.thr / 65536.0f);
  p.seed = (act == 2 && p_drop > 0.f) ? seed : nullptr;
  p.a_out = (bf16_t*)a_out; p.mean_out = mean; p.rstd_out = rstd; p.sc_out = sc; p.sh_out = sh;
  int rc;
  if (rs_fwd_shared(W, Cout)) rc = Cin == 128 ? launch_rs<32, 128, true, 0, 1, true>(p, (hipStream_t)stream) : launch_rs<32, 64, true, 0, 1, true>(p, (hipStream_t)stream);
  else rc = dispatch_rs<true, 0>(p, (hipStream_t)stream);
  if (rc) IDF_FAIL(IDF_ERR_HIP, "conv_rs_gn_bf16: launch refused (%d)", rc);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}

// du = conv3x3(dy, w) * act'(x * sc + sh) * mask and its partials (sum du, sum du x): idf_conv_dgrad_chain_bf16's contract without a
// dy prologue (taps 9), w_frag = the data-gradient weights fragment-major; Cin = channels of dy, Cout = channels of x | x2 / du.
extern "C" int idf_conv_rs_dgrad_chain_bf16(const void* dy, const void* w_frag, const void* x, const void* x2, int C1, const float* sc,
                                            const float* sh, const uint64_t* seed, uint32_t salt, float p_drop, int act, void* out,
                                            float* part_out, int B, int H, int W, int Cin, int Cout, void* stream) {
  if (!rs_tiles(B, H, W, Cin, Cout))
    IDF_FAIL(IDF_ERR_UNSUPPORTED, "conv_rs_dgrad_chain_bf16: B%d H%d W%d Cin%d Cout%d not covered", B, H, W, Cin, Cout);
  if (!dy || !w_frag || !x || !sc || !sh || !out || !part_out || (act != 1 && act != 2))
    IDF_FAIL(IDF_ERR_BADARG, "conv_rs_dgrad_chain_bf16: null argument / act");
  if (!x2) C1 = Cout;
  if (x2 && (C1 <= 0 || C1 >= Cout || (C1 % 64))) IDF_FAIL(IDF_ERR_UNSUPPORTED, "conv_rs_dgrad_chain_bf16: C1 %d of %d", C1, Cout);
  C3P p;
  memset(&p, 0, sizeof(p));
  p.x = (const bf16_t*)dy; p.C1 = Cin; p.w = (const bf16_t*)w_frag; p.y = (bf16_t*)out;
  p.B = B; p.H = H; p.W = W; p.Hs = H; p.Ws = W; p.Cin = Cin; p.Cout = Cout;
  p.due_x = (const bf16_t*)x; p.due_x2 = (const bf16_t*)x2; p.due_C1 = C1; p.due_sc = sc; p.due_sh = sh;
  p.st_out = part_out;
  p.act = act; p.salt = salt; p.thr = idf_drop_thresh(p_drop);
  p.dscale = 1.0f / (1.0f - (float)p.thr / 65536.0f);
  p.seed = (act == 2 && p_drop > 0.f) ? seed : nullptr;
  if (int rc = dispatch_rs<false, 2>(p, (hipStream_t)stream)) IDF_FAIL(IDF_ERR_HIP, "conv_rs_dgrad_chain_bf16: launch refused (%d)", rc);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}

// Tiles per image (= T of the partials workspace [B][T][Cout][2]) when the group-synchronised form covers the shape on this device
// (the grid a whole number of groups, resident at once), else 0.
extern "C" int idf_conv_rs_dgrad_gn_tiles(int B, int H, int W, int Cin, int Cout) {
  const int T = rs_tiles(B, H, W, Cin, Cout);
  if (!T || !idf_knobs().conv_rs_sync) return 0;
  C3P p;
  memset(&p, 0, sizeof(p));
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
  return dispatch_rs_sync(p, nullptr, true) == 0 ? T : 0;
}

// The backward of conv3x3(dropout(act(FiLM(GroupNorm(x))))) w.r.t. x in ONE launch (modules.py:312-320, 283-288 backward):
// idf_conv_rs_dgrad_chain_bf16 followed by idf_gn_bwd_apply, with du kept in registers across a counter the tiles of an image
// meet at (conv_rs_bf16, EPI 3).  dy [B,H,W,Cin] = gradient of the conv's output; x (| x2, C1 channels in x) = the GroupNorm's
// input, Cout channels; dres / dres2 = gradients arriving over the residual / skip branches (dense [B,H,W,Cout]) or null;
// dx (| dx2) as x (| x2); part = workspace [B][T][Cout][2] floats (T = idf_conv_rs_dgrad_gn_tiles); the parameter / FiLM
// gradient outputs as idf_gn_bwd_apply's (dfilm_t / dfilm_a [B][2 Cout], dgb [B][2][Cout] or dgam / dbet accumulated).
extern "C" int idf_conv_rs_dgrad_gn_bf16(const void* dy, const void* w_frag, const void* x, const void* x2, int C1, const float* sc,
                                         const float* sh, const uint64_t* seed, uint32_t salt, float p_drop, int act, const void* dres,
                                         const void* dres2, void* dx, void* dx2, float* part, const float* gamma, const float* beta,
                                         const float* film_t, const float* film_a, int ld_t, int ld_a, const float* mean,
                                         const float* rstd, float* dfilm_t, float* dfilm_a, float* dgb, float* dgam, float* dbet,
                                         uint32_t* sync_state, int B, int H, int W, int Cin, int Cout, void* stream) {
  if (!sync_state) IDF_FAIL(IDF_ERR_BADARG, "conv_rs_dgrad_gn_bf16: null sync_state");
  if (!idf_conv_rs_dgrad_gn_tiles(B, H, W, Cin, Cout))
    IDF_FAIL(IDF_ERR_UNSUPPORTED, "conv_rs_dgrad_gn_bf16: B%d H%d W%d Cin%d Cout%d not covered", B, H, W, Cin, Cout);
  if (!dy || !w_frag || !x || !sc || !sh || !dx || !part || !mean || !rstd || (act != 1 && act != 2))
    IDF_FAIL(IDF_ERR_BADARG, "conv_rs_dgrad_gn_bf16: null argument / act");
  if (!x2) C1 = Cout;
  if (x2 && (C1 <= 0 || C1 >= Cout || (C1 % 64) || !dx2)) IDF_FAIL(IDF_ERR_UNSUPPORTED, "conv_rs_dgrad_gn_bf16: C1 %d of %d", C1, Cout);
  C3P p;
  memset(&p, 0, sizeof(p));
  p.x = (const bf16_t*)dy; p.C1 = Cin; p.w = (const bf16_t*)w_frag; p.y = (bf16_t*)dx; p.rs_dx2 = (bf16_t*)dx2;
  p.B = B; p.H = H; p.W = W; p.Hs = H; p.Ws = W; p.Cin = Cin; p.Cout = Cout;
  p.due_x = (const bf16_t*)x; p.due_x2 = (const bf16_t*)x2; p.due_C1 = C1; p.due_sc = sc; p.due_sh = sh;
  p.st_out = part; p.res = (const bf16_t*)dres; p.gnb_res2 = (const bf16_t*)dres2; p.rs_sync = sync_state;
  p.act = act; p.salt = salt; p.thr = idf_drop_thresh(p_drop);
  p.dscale = 1.0f / (1.0f - (float)p.thr / 65536.0f);
  p.seed = (act == 2 && p_drop > 0.f) ? seed : nullptr;
  GnFoldP& f = p.dyp_f;
  f.mean = mean; f.rstd = rstd; f.sc = sc; f.gamma = gamma; f.beta = beta; f.film_t = film_t; f.film_a = film_a;
  f.ld_t = ld_t ? ld_t : 2 * Cout; f.ld_a = ld_a ? ld_a : 2 * Cout;
  f.dfilm_t = dfilm_t; f.dfilm_a = dfilm_a; f.dgb = dgb; f.dgam = dgam; f.dbet = dbet; f.C = Cout; f.HW = H * W;
  if (int rc = dispatch_rs_sync(p, (hipStream_t)stream, false)) IDF_FAIL(IDF_ERR_HIP, "conv_rs_dgrad_gn_bf16: launch refused (%d)", rc);
  IDF_CHECK_LAUNCH();
  return IDF_OK;
}

// Words of the state array idf_conv_rs_dgrad_gn_bf16 takes (uint32): the counters, then the error word -- the number of workgroups that
// ever gave up waiting for their group (0 in a healthy process; anything else: a launch's grid was not resident at once and its
// results are garbage; zero the whole array to go on).
extern "C" int idf_conv_rs_sync_words(void) { return RS_SYNC_MAX + 16; }

// Diagnostic: the number of polls a workgroup of idf_conv_rs_dgrad_gn_bf16 spends waiting for its group before it gives up and bumps
// the error word (default 2^21, ~2 s; 0 restores the default).  Tests that provoke a time-out (a counter knocked off its multiple of 64)
// shorten it; returns the previous value.  Process-global, read at launch time.
extern "C" unsigned idf_conv_rs_set_spin_limit(unsigned polls) {
  const unsigned prev = g_rs_spin_max;
  g_rs_spin_max = polls ? polls : (1u << 21);
  return prev;
}
