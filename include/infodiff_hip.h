/* libinfodiff_hip -- C ABI of the MI355X (gfx950) kernels behind InfoDiffusion's
 * data-parallel hot path (AVDM UNet forward/backward, diffusion loss, DDPM/DDIM
 * sampler updates).
 *
 * The reference (isjakewong/InfoDiffusion) is pure PyTorch and has no FFI of its
 * own; this header is the boundary SURVEY.md section 8b proposes underneath the
 * reference's Python surface.  Each entry cites the reference code it replaces
 * (file:line in the reference tree).
 *
 * Conventions
 *  - Plain pointers and sizes only; the caller owns all memory (inputs, outputs,
 *    workspaces).  Nothing is allocated, freed or retained by the library.
 *  - Every call is asynchronous on `stream` (a hipStream_t passed as void*).
 *  - Activations are dense NHWC.  `dtype` selects the activation/weight storage
 *    type: IDF_F32 (0) or IDF_BF16 (1); accumulation, statistics, FiLM
 *    coefficients, losses and weight gradients are always fp32.
 *  - Return 0 on success, IDF_ERR_HIP when a HIP runtime call or launch failed, or
 *    IDF_ERR_UNSUPPORTED / IDF_ERR_BADARG; idf_last_error() (thread-local) holds
 *    the message (for IDF_ERR_HIP: the hipError string).  Never aborts, never throws.
 *  - Re-entrant and stateless: safe to call from PyTorch's autograd thread.
 */
#ifndef INFODIFF_HIP_H
#define INFODIFF_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IDF_F32 0
#define IDF_BF16 1
#define IDF_ERR_UNSUPPORTED 1001
#define IDF_ERR_BADARG 1002
#define IDF_ERR_HIP 1003

/* conv gather modes */
#define IDF_CONV_S1 0  /* stride 1, pad taps/2                                  */
#define IDF_CONV_S2 1  /* stride 2 (DownSample, modules.py:66)                   */
#define IDF_CONV_UP2 2 /* nearest x2 upsample fused into the read (modules.py:89-92) */
#define IDF_CONV_T2 3  /* transposed stride 2: data gradient of IDF_CONV_S2      */

int idf_version(void);
const char* idf_last_error(void);

/* ---- convolution (modules.py:63-93, 133-136, 264-293, 335-348; models.py:246, 280-284)
 * y[m,n] = sum_{tap,c} act(x[gather(m,tap),c]) * w[n][tap][c] + bias[n] (+ res[m,n])
 *   x [B,Hs,Ws,Cin], w [Cout][taps][Cin], y/res [B,Ho,Wo,Cout]; taps 1 or 9.
 *   act 0: none; 1: u = x*sc[b,c]+sh[b,c] (GroupNorm+FiLM fold, see idf_gn_coef_fwd);
 *   2: SiLU(u) then dropout(p_drop) when seed != NULL (counter-based, keyed by
 *   (*seed, salt, element index of x)).
 * The data gradient is the same call on dy with the flipped/transposed weight
 * shadow (idf_pack_conv_weight) and mode S1 (for S1/UP2 forward) or T2 (for S2). */
int idf_conv2d_fwd(const void* x, const void* w, const float* bias, const void* res, void* y,
                   const float* sc, const float* sh, const uint64_t* seed, uint32_t salt, float p_drop,
                   int B, int Hs, int Ws, int Cin, int Ho, int Wo, int Cout, int mode, int taps, int act,
                   int dtype, void* stream);

/* bf16 halo-tile form of the 3x3 conv (mode S1 / S2 / UP2 / T2, no prologue; H, W = output dims): one
 * input read per 32-channel chunk instead of one per tap.  IDF_ERR_UNSUPPORTED for shapes it does
 * not cover (Cin % 32, W not a power of two in 4..128): use idf_conv2d_fwd then.
 * st_out (optional, Cout % 8 == 0): per-channel GroupNorm statistics partials of the bf16 output y,
 * [B][T][Cout][2] fp32 = (sum, sum of squares) over each of the T = idf_conv_tiles(...) pixel tiles of an
 * image -- written by the epilogue (plain stores, fixed order), so the nn.GroupNorm that reads y
 * (modules.py:214-228, 264-288) needs no statistics pass of its own; consumed by idf_conv_gn_bf16. */
int idf_conv3x3_bf16(const void* x, const void* w, const float* bias, const void* res, void* y, int B, int H,
                     int W, int Cin, int Cout, int mode, float* st_out, void* stream);
/* 1x1 stride-1 convolution (AttnBlock q/k/v and proj, modules.py:136-139; ResBlock shortcuts, modules.py:228,
 * and their data gradients) through the same pipeline without the halo: w [Cout][Cin] bf16, optional
 * fp32 bias and bf16 residual.  IDF_ERR_UNSUPPORTED outside Cin % 32 == 0, Cout % 8 == 0, W a power
 * of two in 4..128 (use idf_bgemm then).  x2 != NULL: the input is the never-materialised channel
 * concatenation x [.., C1] | x2 [.., Cin - C1] of a skip connection (models.py:321 torch.cat), C1 % 32 == 0;
 * the GroupNorm one-launch kernels and the weight-gradient table take the same (x2, C1) pair.  st_out: as above. */
int idf_conv1x1_bf16(const void* x, const void* x2, int C1, const void* w, const float* bias, const void* res, void* y,
                     int B, int H, int W, int Cin, int Cout, float* st_out, void* stream);
/* Pixel tiles per image (= T of st_out) of the launch idf_conv3x3_bf16 / idf_conv1x1_bf16 (pro = 0) or
 * idf_conv_gn_bf16 (pro = 1) make for this shape (H, W = output dims; taps 9 or 1); -1 when the shape is not covered. */
int idf_conv_tiles(int B, int H, int W, int Cin, int Cout, int mode, int taps, int pro);
/* 1 when idf_conv_gn_bf16 is expected to beat idf_gn_fused_fwd (or idf_gn_coef_fwd + idf_gn_apply) followed by the plain
 * conv for this shape, 0 when the two-launch form wins (measured rules, see idf_conv3x3.hip); the host picks with it. */
int idf_conv_gn_advice(int B, int H, int W, int Cin, int Cout, int taps);

/* The conv / GroupNorm-SiLU / AdaGN fused block (modules.py:264-288 block1..3, 309-320 AuxResBlock.forward,
 * 145-150 AttnBlock GroupNorm + q/k/v, models.py:280-284 tail):
 *   y = conv( dropout( act( GroupNorm32(x) [*(1+s_t)+b_t] [*(1+s_a)+b_a] ) ) ) + bias (+ res)
 * as ONE launch: stride-1 3x3 (taps 9) or 1x1 (taps 1) over x [B,H,W,Cin] bf16 -- or over the never-materialised
 * concatenation x [..,C1] | x2 [..,Cin-C1] (models.py:321).  No statistics pass: st1 [B][T1][C1][2]
 * (st2 [B][T2][Cin-C1][2]) are the partials the producers of x (x2) left behind (st_out above, or
 * idf_gn_partials); every block folds them with gamma / beta [Cin] and the FiLM pairs film_t / film_a
 * ([B,2Cin] with row strides ld_t / ld_a, layout and fold as idf_gn_coef_fwd; NULL = absent) into the
 * per-(image, channel) affine u = x*sc+sh and applies  act 1: u;  act 2: SiLU(u), then dropout(p_drop) when
 * seed != NULL (keyed by (*seed, salt, element index), as idf_conv2d_fwd)  once per staged element.
 * Optional outputs (training; NULL otherwise): a_out [B,H,W,Cin] bf16 = the activated tensor (input of the
 * weight gradient), mean / rstd [B,32] and sc / sh [B,Cin] (all four or none; inputs of idf_gn_fused_bwd /
 * idf_gn_coef_bwd), st_out as above.  coef_ws (optional): B * Cin * 2 floats of scratch; with it, launches that cut an
 * image into many tiles fold the coefficients once per image in a small launch of their own instead of once per block.
 * IDF_ERR_UNSUPPORTED for shapes outside idf_conv3x3_bf16 / idf_conv1x1_bf16. */
int idf_conv_gn_bf16(const void* x, const void* x2, int C1, const float* st1, int T1, const float* st2, int T2,
                     const float* gamma, const float* beta, const float* film_t, const float* film_a, int ld_t,
                     int ld_a, float eps, int act, const uint64_t* seed, uint32_t salt, float p_drop,
                     const void* w, const float* bias, const void* res, void* y, void* a_out, float* mean,
                     float* rstd, float* sc, float* sh, float* st_out, float* coef_ws, int B, int H, int W, int Cin,
                     int Cout, int taps, void* stream);

/* Backward counterpart of the fused block on the small maps (H*W <= 256: the 16x16 / 8x8 / 4x4 levels, where one workgroup
 * tile is a whole image): the stride-1 data-gradient conv (taps 9 or 1; w = the data-gradient weights [Cout][taps][Cin]
 * with flipped taps, as idf_pack_conv_weight writes them) with the GroupNorm / FiLM / SiLU / dropout BACKWARD of
 * modules.py:264-288, 312-319 as its epilogue.  The conv result dA (gradient w.r.t. the activated tensor) never leaves the
 * chip: the block owns every pixel of its 64 channels, so it forms du = dA * act'(x*sc+sh) * mask, the per-(sample, channel)
 * sums, and dx = sc*du + k1*x + k0 (+ dres + dres2) itself -- arguments and side outputs (dfilm_t / dfilm_a [B,2C],
 * dgb [B,2,C] or the dgamma_acc / dbeta_acc accumulators) exactly as idf_gn_fused_bwd.  Cin = channels of dy, Cout = C =
 * channels of x and dx (a multiple of 64).  idf_conv_dgrad_gn_ok(): 0 not covered, 1 covered and measured faster than the
 * two launches it replaces, 2 covered only (the 1x1 q/k/v convs, maps above IDF_DGRAD_GN_MAXHW). */
int idf_conv_dgrad_gn_ok(int B, int H, int W, int Cin, int Cout, int taps);
int idf_conv_dgrad_gn_bf16(const void* dy, const void* w, const void* x, const void* dres, const void* dres2, void* dx,
                           const float* gamma, const float* beta, const float* film_t, const float* film_a, int ld_t,
                           int ld_a, const float* mean, const float* rstd, const float* sc, const float* sh,
                           float* dfilm_t, float* dfilm_a, float* dgb, float* dgamma_acc, float* dbeta_acc,
                           const uint64_t* seed, uint32_t salt, float p_drop, int act, int B, int H, int W, int Cin,
                           int Cout, int taps, void* stream);

/* The same backward at the BIG maps (64x64 / 32x32: a workgroup tile is a slice of an image, so the GroupNorm backward's
 * per-(sample, group) sums cannot close inside one block) -- the backward mirror of idf_conv_gn_bf16's producer-side
 * statistics (reference modules.py:264-288, 309-320 backward; run.py:197 loss.backward()).
 *
 *   out = conv(g, w),  w = data-gradient weights [Cout][taps][Cin] (flipped taps), stride 1, taps 9 or 1.
 *
 * du EPILOGUE (x != NULL): the forward conv read a = dropout(act(x*sc+sh)) (x [B,H,W,Cout], or the pair x [..,C1] |
 *   x2 [..,Cout-C1] with C1 % 64 == 0; sc / sh [B,Cout] as saved by idf_conv_gn_bf16; act / seed / salt / p_drop as
 *   there): out = du = dA * act'(x*sc+sh) * mask (bf16) and part_out [B][T][Cout][2] = per-pixel-tile partial sums
 *   (sum du, sum du*x) of the rounded du, T = idf_conv_dgrad_chain_tiles().  What is left of the GroupNorm backward is
 *   reduction-free: idf_gn_bwd_apply (a streaming pass), or --
 * dy PROLOGUE (in_x != NULL): -- the NEXT data-gradient conv (the one in front of that GroupNorm) takes (du, partials) in
 *   place of its input gradient: `dy` = du_in, in_x = the GroupNorm's input, in_part [B][in_T][Cin][2]; every block folds the
 *   partials with in_mean / in_rstd [B,32], in_sc [B,Cin], in_gamma / in_beta [Cin], in_film_t / in_film_a (layout and
 *   strides as idf_gn_coef_fwd) into g = A*du_in + K1*in_x + K0 while it stages its tile.  One block per image stores that
 *   GroupNorm's parameter / FiLM gradients (in_dfilm_t .. in_dbeta_acc: as idf_gn_fused_bwd's dfilm_t .. dbeta_acc), and g
 *   itself is written once to dy_out (optional; the weight gradient of the conv in front needs it).
 * With x == NULL the epilogue is the plain one (out = dA).  bf16; Cin % 32 == 0, Cout % 64 == 0, W a power of two in
 * [4, 128]; idf_conv_dgrad_chain_tiles() < 0: shape not covered. */
int idf_conv_dgrad_chain_tiles(int B, int H, int W, int Cin, int Cout, int taps);

/* ---- image-resident ResBlock forward for the 8x8 maps (round 4) ------------------------------------------------------
 * A whole AuxResBlock / ResBlock (modules.py:261-328, 206-258: three GroupNorm -> SiLU -> [Dropout] -> Conv3x3 stages, FiLM on
 * the second, + shortcut) or ResBlock_encoder (modules.py:331-366: two stages) in ONE launch, one 512-thread workgroup per
 * image: at 8x8 a workgroup owns all 64 pixels of every channel, so each GroupNorm's statistics close inside it and the next
 * stage's activated tensor is written straight into the LDS image its conv reads.  bf16, 128 couts, Cin 128 or 256 (possibly
 * the pair x | x2 of an up-path block, C1 % 32 == 0).  Arithmetic per element is that of idf_conv_gn_bf16 stage by stage.
 *   stage i:  a_i = dropout_i(SiLU(FiLM_i(GroupNorm_i(h_{i-1}))))   h_i = conv3x3(a_i, w_i) + bias_i     (h_{-1} = x | x2)
 *   y = h_last + (w_sc ? round_bf16(conv1x1(x | x2, w_sc) + b_sc) : x)
 * st1 / st2: statistics partials of x / x2 as their producers left them ([B][T][C][2], see idf_conv_tiles); st_out
 * [B][1][128][2]: those of y.  Training outputs per stage (all optional, NULL in inference): a_out [B,8,8,Cin_i] the activated
 * conv input (for the weight gradient), h_out [B,8,8,128] the stage's conv output (the next GroupNorm's input; the last
 * stage's h_out must be y), mean / rstd [B,32] and sc / sh [B,Cin_i] of the stage's GroupNorm (for idf_conv_dgrad_gn_bf16).
 * seed == NULL: no dropout; else stage i drops with probability p_drop where its `drop` is non-zero, keyed by (seed, salt_i,
 * element index) exactly as idf_conv_gn_bf16 does.  idf_resblock_small_ok: 1 when the shape is covered (C1 = 0: one source). */
typedef struct {
  const void* w;            /* [128][9][Cin_i] forward shadow (idf_pack_conv_weight) */
  const float* bias;        /* [128] */
  const float* gamma; const float* beta;                 /* [Cin_i] or NULL */
  const float* film_t; const float* film_a; int ld_t, ld_a;   /* [B][2*Cin_i] (row stride ld) or NULL */
  uint32_t salt; int drop;
  void* a_out; float* mean; float* rstd; float* sc; float* sh; void* h_out;
} IdfResblockStage;
typedef struct {
  const void* x; const void* x2; int C1, Cin;
  const float* st1; const float* st2; int T1, T2;
  int nstage;               /* 2 or 3 */
  IdfResblockStage s[3];
  const void* w_sc; const float* b_sc;                    /* 1x1 shortcut [128][Cin] + bias, or NULL: identity (Cin == 128) */
  void* y; float* st_out;
  const uint64_t* seed; float p_drop; float eps;
  int B;
  int w_layout;             /* 0: s[i].w is the forward shadow [128][9][Cin_i]; 1: the fragment-major shadow (idf_pack_conv_weights_batched) */
} IdfResblockArgs;
int idf_resblock_small_ok(int B, int H, int W, int Cin, int C1, int Cout, int nstage);

/* ---- per-op 3x3 convs of the 16x16 / 8x8 maps with fragment-major weights (round 4): a workgroup = 4 waves x 16 couts over
 * 64 pixels (4 rows of a 16x16 map, a whole 8x8 map) or a whole 16x16 map; weights straight into registers, the input of all
 * channel chunks in LDS, no workgroup barrier in the conv loop, the epilogue in the wave's registers.
 * idf_conv_wr_gn_bf16: idf_conv_gn_bf16's contract for act 2 (SiLU), taps 9: y = conv(dropout(SiLU(FiLM(GN(x | x2))))) + bias
 * (+ res), training outputs a_out / mean / rstd / sc / sh, st_out [B][idf_conv_wr_tiles(.., 0)][Cout][2].
 * idf_conv_wr_dgrad_gn_bf16: idf_conv_dgrad_gn_bf16's contract (act 2, taps 9) on a whole image per workgroup.
 * idf_conv_wr_tiles: pixel tiles per image of the forward form (whole = 0) / 1 when the whole-image form covers the shape
 * (whole = 1); 0: not covered (H = W in {8, 16}, Cin in {64, 128, 256}, Cout % 64 == 0, Cout <= 256). */
int idf_conv_wr_tiles(int B, int H, int W, int Cin, int Cout, int whole);

/* the head conv (models.py:258, 300: Conv2d(3 | 1, ch, 3, padding 1)): Cin <= 3, the whole contraction is one MFMA K-step;
 * w = the forward shadow [Cout][9][Cin]; st_out (optional) = statistics partials of y, [B][idf_conv_fewc_tiles()][Cout][2]
 * (0 tiles: shape not covered -- use idf_conv2d_fwd). */
int idf_conv_fewc_tiles(int B, int H, int W, int Cin, int Cout);
int idf_conv3x3_fewc_bf16(const void* x, const void* w, const float* bias, void* y, int B, int H, int W, int Cin, int Cout,
                          float* st_out, void* stream);
int idf_conv_wr_gn_bf16(const void* x, const void* x2, int C1, const float* st1, int T1, const float* st2, int T2,
                        const float* gamma, const float* beta, const float* film_t, const float* film_a, int ld_t, int ld_a,
                        float eps, const uint64_t* seed, uint32_t salt, float p_drop, const void* w_frag, const float* bias,
                        const void* res, void* y, void* a_out, float* mean, float* rstd, float* sc, float* sh, float* st_out,
                        int B, int H, int W, int Cin, int Cout, void* stream);
int idf_conv_wr_dgrad_gn_bf16(const void* dy, const void* w_frag, const void* x, const void* dres, const void* dres2, void* dx,
                              const float* gamma, const float* beta, const float* film_t, const float* film_a, int ld_t,
                              int ld_a, const float* mean, const float* rstd, const float* sc, const float* sh, float* dfilm_t,
                              float* dfilm_a, float* dgb, float* dgamma_acc, float* dbeta_acc, const uint64_t* seed,
                              uint32_t salt, float p_drop, int B, int H, int W, int Cin, int Cout, void* stream);
int idf_resblock_small_fwd(const IdfResblockArgs* args, void* stream);

/* ---- the ResBlock 3x3 convs of the 64x64 / 32x32 maps in the "register weights, row reuse" form (round 5, idf_conv_rs.hip):
 * a 512-thread workgroup per CU walks consecutive 256-pixel x 64-cout tiles; wave = 16 couts, ALL its weights of a 64-channel
 * pair in registers (fragment-major shadow, never through LDS), the halo image of every channel chunk resident in LDS, each
 * input row's three shifted fragments used for the three output rows they feed, the next tile's rows in flight during this
 * tile's MFMAs and epilogue.  Replaces the halo / direct-to-LDS kernels for modules.py:264-268, 283-288, 312-320 (forward) and
 * their data gradients where idf_conv_rs_tiles() != 0: H = W = 64 with 64 input channels, H = W = 32 with 64 or 128,
 * Cout % 64 == 0, >= 128 work items.  Same tiles and partial layouts as idf_conv_tiles / idf_conv_dgrad_chain_tiles give.
 * idf_conv_rs_gn_bf16:           idf_conv_gn_bf16's contract (taps 9, one source, T1 <= 16), w_frag fragment-major forward weights.
 * idf_conv_rs_dgrad_chain_bf16:  idf_conv_dgrad_chain_bf16's contract with the du epilogue and no dy prologue (taps 9): dy
 *                                [B,H,W,Cin], w_frag = fragment-major data-gradient weights, x | x2 = the GroupNorm input
 *                                [B,H,W,Cout] (C1 % 64 == 0), out = du, part_out [B][T][Cout][2]. */
int idf_conv_rs_tiles(int B, int H, int W, int Cin, int Cout);
int idf_conv_rs_fwd_tiles(int B, int H, int W, int Cin, int Cout);   /* T of idf_conv_rs_gn_bf16's st_out: twice idf_conv_rs_tiles where the forward conv
                                                                        computes both cout tiles of a pixel tile from one halo image (32x32, Cout = 128) */
int idf_conv_rs_gn_bf16(const void* x, const float* st1, int T1, const float* gamma, const float* beta, const float* film_t,
                        const float* film_a, int ld_t, int ld_a, float eps, int act, const uint64_t* seed, uint32_t salt,
                        float p_drop, const void* w_frag, const float* bias, const void* res, void* y, void* a_out, float* mean,
                        float* rstd, float* sc, float* sh, float* st_out, int B, int H, int W, int Cin, int Cout, void* stream);
int idf_conv_rs_dgrad_chain_bf16(const void* dy, const void* w_frag, const void* x, const void* x2, int C1, const float* sc,
                                 const float* sh, const uint64_t* seed, uint32_t salt, float p_drop, int act, void* out,
                                 float* part_out, int B, int H, int W, int Cin, int Cout, void* stream);

/* The same data-gradient conv with the WHOLE GroupNorm / FiLM / SiLU / dropout backward behind it in one launch
 * (/root/reference/modules.py:312-320, 283-288 backward: idf_conv_rs_dgrad_chain_bf16 + idf_gn_bwd_apply, with du held in
 * registers while the workgroups that own the tiles of one (image, 64-channel slice) meet at a counter and fold each other's
 * partial sums).  idf_conv_rs_dgrad_gn_tiles: T of the partials workspace when the form covers the shape on this device (the
 * grid must be a whole number of groups and resident at once), else 0.  dres / dres2: residual / skip gradients (dense
 * [B,H,W,Cout]) or null; dx | dx2 as x | x2; part: workspace [B][T][Cout][2]; dfilm_t / dfilm_a [B][2 Cout], dgb [B][2][Cout]
 * or dgam / dbet (accumulated) as idf_gn_bwd_apply's.  sync_state: idf_conv_rs_sync_words() uint32 words of device memory owned by
 * the caller, zero when first handed in and persistent across launches (counters that only grow, then an error word = the number of
 * workgroups that ever gave up waiting: 0 in a healthy process); launches that share a state array must not overlap. */
int idf_conv_rs_dgrad_gn_tiles(int B, int H, int W, int Cin, int Cout);
int idf_conv_rs_dgrad_gn_bf16(const void* dy, const void* w_frag, const void* x, const void* x2, int C1, const float* sc,
                              const float* sh, const uint64_t* seed, uint32_t salt, float p_drop, int act, const void* dres,
                              const void* dres2, void* dx, void* dx2, float* part, const float* gamma, const float* beta,
                              const float* film_t, const float* film_a, int ld_t, int ld_a, const float* mean, const float* rstd,
                              float* dfilm_t, float* dfilm_a, float* dgb, float* dgam, float* dbet, uint32_t* sync_state, int B,
                              int H, int W, int Cin, int Cout, void* stream);
int idf_conv_rs_sync_words(void);
/* Diagnostic: polls a workgroup spends waiting for its group before it gives up and bumps the error word (default 2^21, about 2 s;
 * 0 restores the default); returns the previous value.  Process-global, read at launch time.  For tests that provoke a time-out
 * (a counter knocked off its multiple of 64) and assert that the training loop reports it (infodiffusion_amd/trainer.py). */
unsigned idf_conv_rs_set_spin_limit(unsigned polls);

/* Deterministic mode (/root/reference/utils.py:64-71): the GroupNorm affine gradients of a whole backward pass in ONE launch.
 * Every GroupNorm-backward entry above writes per-image rows dgb [B][2][C] when it is given no accumulation slots; table = n
 * entries of {const float* rows; float* dgamma; float* dbeta; int B; int C;} (idf_gn_rows_desc_bytes() each); the launch adds
 * each entry's rows in image order into its slots: dgamma[c] += sum_b rows[b][0][c], dbeta[c] += sum_b rows[b][1][c]. */
int idf_gn_rows_desc_bytes(void);
int idf_gn_param_reduce_batched(const void* table, int n, int max_c, void* stream);

/* The backward of the same blocks: the data-gradient convs of stages nstage-1 .. first with the GroupNorm / FiLM / SiLU / dropout
 * backward behind each (what idf_conv_wr_dgrad_gn_bf16 computes per stage) in ONE launch, one workgroup per image; every stage
 * has 128 channels.  Stage i: x = its GroupNorm input (h_{i-1}; the block input for stage 0), w_frag = the data-gradient weights
 * of its conv fragment-major, mean .. sh as the forward pass saved them, dx = the gradient w.r.t. x (= the gradient of stage
 * i-1's conv output: read by that conv's weight gradient, and by this launch as the next stage's dy).  Side outputs per stage as
 * idf_conv_dgrad_gn_bf16 (dfilm_t / dfilm_a [B,256], dgb [B,2,128] or the dgamma_acc / dbeta_acc accumulators).
 * first = 0 (one-source block input, identity residual): the block's dy (+ dres2, the skip alias' gradient) joins stage 0's dx;
 * first >= 1: the launch stops above the block input (two-source inputs / 1x1 shortcuts keep their own launches). */
typedef struct {
  const void* w_frag; const void* x;
  const float* gamma; const float* beta; const float* film_t; const float* film_a; int ld_t, ld_a;
  const float* mean; const float* rstd; const float* sc; const float* sh;
  uint32_t salt; int drop;
  float* dfilm_t; float* dfilm_a; float* dgb; float* dgamma_acc; float* dbeta_acc;
  void* dx;
} IdfResblockBwdStage;
typedef struct {
  const void* dy; int nstage, first;
  IdfResblockBwdStage s[3];
  const void* dres2;
  const uint64_t* seed; float p_drop; int B;
} IdfResblockBwdArgs;
int idf_resblock_small_bwd(const IdfResblockBwdArgs* args, void* stream);
int idf_conv_dgrad_chain_bf16(const void* dy, const void* in_x, const float* in_part, int in_T, const float* in_mean,
                              const float* in_rstd, const float* in_sc, const float* in_gamma, const float* in_beta,
                              const float* in_film_t, const float* in_film_a, int in_ld_t, int in_ld_a,
                              float* in_dfilm_t, float* in_dfilm_a, float* in_dgb, float* in_dgamma_acc,
                              float* in_dbeta_acc, void* dy_out, const void* w, const void* x, const void* x2, int C1,
                              const float* sc, const float* sh, const uint64_t* seed, uint32_t salt, float p_drop,
                              int act, void* out, float* part_out, int B, int H, int W, int Cin, int Cout, int taps,
                              void* stream);
/* The fold of idf_conv_gn_bf16 as a launch of its own: mean / rstd [B][32] and (sc, sh) [B][C] of a GroupNorm stage
 * (modules.py:264-268, 283-288) from the statistics partials its input carries (st1 [B][T1][C1][2], st2 for the second
 * source of a skip pair).  With idf_gn_apply this is the streaming form of the GroupNorm pass for big tensors whose
 * convolution stays a launch of its own (idf_conv_gn_advice = 0).  ws: [B][2C] floats of scratch. */
int idf_gn_coef_from_stats(const float* st1, int T1, const float* st2, int T2, int C1, const float* gamma, const float* beta,
                           const float* film_t, const float* film_a, int ld_t, int ld_a, float eps, float* mean, float* rstd,
                           float* sc, float* sh, float* ws, int B, int HW, int C, void* stream);
/* ResBlock shortcuts riding in their neighbours' launches (round 3).  A block whose channel count changes computes
 * `self.shortcut(x)` (a 1x1 conv, modules.py:228, 248, 281) beside `self.block1(x)`: idf_conv_gn_sc_bf16 is idf_conv_gn_bf16
 * (3x3, Cout > 32) whose launch carries extra blocks computing  sc_y [B,H,W,sc_Cout] = conv1x1(x | x2, sc_w [sc_Cout][Cin]) +
 * sc_bias  over the RAW input (same pixel tiles; no prologue; centre tap only).  In backward the shortcut's data gradient
 * rides with the first conv's: idf_conv_dgrad_chain_sc_bf16 is idf_conv_dgrad_chain_bf16 (3x3, du epilogue, no dy
 * prologue) + extra blocks computing  sc_dx [B,H,W,Cout] = conv1x1(sc_dy [B,H,W,sc_Cin], sc_w [Cout][sc_Cin])  with sc_w
 * the shortcut's data-gradient weights (idf_pack_conv_weight); sc_dx then joins idf_gn_bwd_apply as `dres`.
 * sc_Cout % 8 == 0, sc_Cin % 32 == 0.  Two launches fewer per such block and step. */
int idf_conv_gn_sc_bf16(const void* x, const void* x2, int C1, const float* st1, int T1, const float* st2, int T2,
                        const float* gamma, const float* beta, const float* film_t, const float* film_a, int ld_t, int ld_a,
                        float eps, int act, const uint64_t* seed, uint32_t salt, float p_drop, const void* w, const float* bias,
                        const void* res, void* y, void* a_out, float* mean, float* rstd, float* sc, float* sh, float* st_out,
                        float* coef_ws, int B, int H, int W, int Cin, int Cout, int taps, void* stream, const void* sc_w,
                        const float* sc_bias, void* sc_y, int sc_Cout);
int idf_conv_dgrad_chain_sc_bf16(const void* dy, const void* w, const void* x, const void* x2, int C1, const float* sc,
                                 const float* sh, const uint64_t* seed, uint32_t salt, float p_drop, int act, void* out,
                                 float* part_out, int B, int H, int W, int Cin, int Cout, void* stream, const void* sc_dy,
                                 const void* sc_w, void* sc_dx, int sc_Cin);
/* dx = A*du + K1*x + K0 (+ dres + dres2) from the (du, part [B][T][C][2]) pair idf_conv_dgrad_chain_bf16 leaves behind: the
 * GroupNorm / FiLM backward of modules.py:312-318 without its reduction (one read of du, x and the branch gradients, one
 * write of dx; every block folds its image's partials first).  x may be the pair x [..,C1] | x2 (then dx2 is written too).
 * Side outputs (dfilm_t / dfilm_a [B,2C], dgb [B,2,C] or the dgamma_acc / dbeta_acc accumulators) as idf_gn_fused_bwd. */
int idf_gn_bwd_apply(const void* du, const float* part, int T, const void* x, const void* x2, int C1, const void* dres,
                     const void* dres2, void* dx, void* dx2, const float* gamma, const float* beta, const float* film_t,
                     const float* film_a, int ld_t, int ld_a, const float* mean, const float* rstd, const float* sc,
                     float* dfilm_t, float* dfilm_a, float* dgb, float* dgamma_acc, float* dbeta_acc, int B, int HW,
                     int C, void* stream);

/* dW[n][tap][c] (fp32, zeroed inside) = sum_m dy[m,n] * act(x[gather(m,tap),c]);
 * same prologue arguments as the forward so the activated input is recomputed. */
int idf_conv2d_wgrad(const void* x, const void* dy, float* dW, const float* sc, const float* sh,
                     const uint64_t* seed, uint32_t salt, float p_drop, int B, int Hs, int Ws, int Cin,
                     int Ho, int Wo, int Cout, int mode, int taps, int act, int dtype, void* stream);

/* bf16 fast path of the weight gradient (taps 9 or 1; mode S1, S2 or UP2; H, W = dy dims) on an
 * already-activated input `a`
 * (halo tile in LDS, transposed LDS reads); also accumulates db[n] = sum dy when db != NULL.
 * accumulate = 0: dW / db are zeroed inside first; 1: the kernel adds onto what they hold (the
 * host's gradient arena: every parameter gradient zeroed by ONE memset per step).
 * Cin / Cout are the operands' channel counts (multiples of 8: the image input and the epsilon / latent-head
 * outputs are zero-padded by the host); Cin_w / Cout_w (0 = Cin / Cout) are the gradient's: dW is
 * [Cout_w][taps][Cin_w] and db [Cout_w], so the padded convolutions write the parameter's own gradient layout
 * (and join the gradient arena and the batched launch like every other conv).
 * Returns IDF_ERR_UNSUPPORTED for shapes it does not cover (use idf_conv2d_wgrad then). */
int idf_conv_wgrad_bf16(const void* a, const void* dy, float* dW, float* db, int B, int H, int W, int Cin,
                        int Cout, int Cin_w, int Cout_w, int taps, int mode, int accumulate, void* stream);

/* Batched form of the same kernel: ALL weight gradients of a backward pass in one launch per
 * (taps, mode) class (the reference computes them one aten::convolution_backward at a time; they are
 * only read by the optimizer, so the host defers them to the end of backward).  The host fills one
 * table entry per convolution (idf_wgrad_desc_bytes() bytes each; arguments as idf_conv_wgrad_bf16,
 * always accumulating: dW / db pre-zeroed), chaining blk0 = sum of the previous blocks_out, copies
 * the table to device memory and launches it.  target_blocks = grid budget per problem (<= 0: blocks
 * are sized by work -- IDF_WGRAD_TPB pixel tiles per block, at least IDF_WGRAD_MINB blocks per problem --
 * because every problem shares the chip with the others); lds_bytes = max of the entries' lds_out.
 * Stride-1 3x3 problems run in a shared-tile form (three kernel rows on one staged tile) when their map fits it
 * (idf_wgrad_kr3_ok); the host keeps those that do not in a class of their own: mode | IDF_WGRAD_ROWSPLIT in both
 * calls selects the row-split kernel for that class. */
#define IDF_WGRAD_ROWSPLIT 16
/* mode 2 (UpSample) | IDF_WGRAD_UPSUB in both calls: the class in its sub-pixel form (16 tap products per low-resolution pixel
 * instead of 36: taps that read the same low-resolution pixel share one product, modules.py:78-93) where idf_wgrad_upsub_ok. */
#define IDF_WGRAD_UPSUB 32
/* mode 0, taps 9 | IDF_WGRAD_RING in both calls (round 6): the stride-1 3x3 class in its row-ring form where idf_wgrad_ring_ok --
 * 64 / 32 / 16 / 8-wide maps, tensors below 2 GB: a block's consecutive pixel tiles are vertically adjacent, so only a tile's R new
 * input rows are fetched (the rest wait in a ring of LDS rows), wave tiles of 64 couts x 16 cins x 9 taps, buffer loads whose
 * out-of-image lanes fall outside the descriptor's range (no branches) -- /root/reference/run.py:195-200's backward of every 3x3 conv. */
#define IDF_WGRAD_RING 64
int idf_wgrad_ring_ok(int B, int H, int W, int Cin, int Cout);
int idf_wgrad_upsub_ok(int H, int W);
int idf_wgrad_kr3_ok(int H, int W);
int idf_wgrad_desc_bytes(void);
int idf_wgrad_desc_fill(void* host_table, int index, const void* a, const void* a2, int C1, const void* dy, float* dW,
                        float* db, int B, int H, int W, int Cin, int Cout, int Cin_w, int Cout_w, int taps, int mode,
                        int target_blocks, int blk0, int* blocks_out, int* lds_out, float* ws, int red_blk0,
                        long* ws_floats_out, int* red_blocks_out);
/* Deterministic accumulation (round 5; the reference's convolution_backward is deterministic under --deterministic, utils.py:64-71):
 * with ws != NULL every pixel split of the entry writes its partial dW | db into its own slab of ws (plain stores, one writer per
 * element; *ws_floats_out floats, 16-byte aligned) instead of fp32 atomics, and idf_wgrad_reduce_batched -- one launch over the
 * WHOLE table of a flush, after its class launches -- adds the slabs to dW / db in slab order.  red_blk0 = running sum of the
 * *red_blocks_out values of the entries before this one; a first call with ws = NULL and host_table scratch sizes the workspace. */
int idf_wgrad_reduce_batched(const void* dev_table, int n, int total_red_blocks, void* stream);
int idf_conv_wgrad_bf16_batched(const void* dev_table, int n, int total_blocks, int lds_bytes, int taps, int mode,
                                void* stream);
/* UpSample's forward (modules.py:78-93: nearest x2, then conv3x3 pad 1) as four 2x2 convs on the LOW-resolution input -- the
 * sub-pixel form of the same sum, 16 tap products per four outputs instead of 36 (round 4; data gradient:
 * idf_upconv_dgrad_bf16 below; the weight gradient stays the UP2 class of idf_conv_wgrad_bf16_batched).  x [B, Hl, Wl, Cin] bf16; y [B, 2 Hl, 2 Wl, Cout];
 * w_sub_frag: the summed weights W'[py][px][ty][tx] = sum_{ky in S(py,ty), kx in S(px,tx)} W[ky][kx] (S(0,0) = {0}, S(0,1) = {1,2},
 * S(1,0) = {0,1}, S(1,1) = {2}; summed in fp32, rounded to bf16 once) as [Cout][16 taps = (py, px, ty, tx)][Cin] in the
 * fragment-major form [Cin/64][Cout/16][16][2][64 lanes = (k group, cout row)][8]; st_out [B][idf_upconv_tiles()][Cout][2]
 * (optional): statistics partials of y.  idf_upconv_tiles == 0: shape not covered (Wl in {8, 16, 32}; Cin, Cout % 64 == 0). */
int idf_upconv_tiles(int Hl, int Wl, int Cin, int Cout);
/* the summed weights of every UpSample conv of a network in one launch: table (device) = nrows x {const float* src; void* dst;
 * void* dstd; long so, si, st; int O, I} (56 bytes; src = the fp32 master weight, logical (o, i, tap) at o*so + i*si + tap*st;
 * dst = that conv's w_sub_frag; dstd (optional) = its w_sub_dgrad_frag: the same sums with rows = cins, k = couts),
 * max_pairs = the largest O * I of the rows. */
int idf_upconv_pack_batched(const void* table, int nrows, long max_pairs, void* stream);
/* The data gradient of the same layer in the same form: dx [B, Hl, Wl, Cin] from dy [B, 2 Hl, 2 Wl, Cout] -- 16 tap products per
 * low-resolution pixel instead of a 3x3 conv over the 4x larger dy and a 2x2 sum-pool pass.  idf_upconv_dgrad_ok: shape covered. */
/* DownSample's data gradient (modules.py:63-75: conv3x3 stride 2, pad 1) by output parity: per parity of the high-resolution pixel
 * only the taps that land on a dy pixel (1, 2, 2, 4 of the 9) instead of a 3x3 conv over the zero-stuffed dy.  dy [B, Hl, Wl, Cout];
 * w_dgrad_frag: the conv's data-gradient weights fragment-major (idf_pack_conv_weights_batched's `wdfrag`); res (optional): a
 * gradient arriving over another branch of the same input, added before the rounding; dx [B, 2 Hl, 2 Wl, Cin]. */
int idf_downconv_dgrad_ok(int Hl, int Wl, int Cin, int Cout);
int idf_downconv_dgrad_bf16(const void* dy, const void* w_dgrad_frag, const void* res, void* dx, int B, int Hl, int Wl, int Cin,
                            int Cout, void* stream);
int idf_upconv_dgrad_ok(int Hl, int Wl, int Cin, int Cout);
int idf_upconv_dgrad_bf16(const void* dy, const void* w_sub_dgrad_frag, void* dx, int B, int Hl, int Wl, int Cin, int Cout,
                          void* stream);
int idf_upconv_bf16(const void* x, const void* w_sub_frag, const float* bias, void* y, float* st_out, int B, int Hl, int Wl,
                    int Cin, int Cout, void* stream);

/* fp32 master weight (logical (o,i,tap) at o*so+i*si+tap*st) -> forward shadow
 * [O][taps][I] and/or data-gradient shadow [I][taps flipped][O], in `dtype`. */
int idf_pack_conv_weight(const float* src, long so, long si, long st, void* w_fwd, void* w_dgrad, int O, int I,
                         int taps, int dtype, void* stream);

/* the same for every conv of a network in one launch.  table (device): nrows x
 * {const float* src; void* w_fwd; void* w_dgrad; long so, si, st; int O, I, taps, Ototal, o0; long tile; void* w_frag;
 *  void* w_dgrad_frag}
 * -- one block per row = one (tap, 32-cout, 64-cin) tile, tile = tap | cout_tile << 8 | cin_tile << 32;
 * Ototal/o0 place a source tensor inside a concatenated (q|k|v) shadow.  w_frag (optional; 3x3, O % 16 == 0, I % 64 == 0):
 * a third shadow, the forward weights fragment-major for idf_resblock_small_fwd (IdfResblockArgs.w_layout = 1):
 * [I / 64][O / 16][tap][half][lane = fq * 16 + fr][8] with o = 16 * (O / 16 index) + fr, i = 64 * pair + 32 * half + 8 * fq + e;
 * w_dgrad_frag (optional; 3x3, I % 16 == 0, Ototal % 64 == 0): the data-gradient weights in the same form (for
 * idf_conv_wr_dgrad_gn_bf16). */
int idf_pack_conv_weights_batched(const void* table, int nrows, int dtype, void* stream);

/* ---- GroupNorm(32) + AdaGN/FiLM fold (modules.py:132, 214-228, 312-318; nn.GroupNorm eps 1e-5)
 * Writes mean/rstd [B,32] and the per-(b,c) affine sc/sh [B,C]:
 *   sc = rstd*gamma*(1+s_t)*(1+s_a),  sh = ((beta-mean*rstd*gamma)*(1+s_t)+b_t)*(1+s_a)+b_a
 * film_t / film_a: [B,2C] (scale = first half, shift = second half; torch.chunk order,
 * modules.py:314,317) or NULL; ld_t / ld_a = their row strides in floats (0 = dense 2C) so a slice
 * of one batched FiLM projection can be consumed in place.  workspace: idf_gn_workspace_floats(B,HW,C) floats. */
int idf_gn_workspace_floats(int B, int HW, int C);
int idf_gn_coef_fwd(const void* x, const float* gamma, const float* beta, const float* film_t,
                    const float* film_a, int ld_t, int ld_a, float eps, float* mean, float* rstd, float* sc, float* sh,
                    float* workspace, int B, int HW, int C, int dtype, void* stream);
/* Per-channel statistics partials of a tensor x [B,HW,C]: part [B][T][C][2] (sum, sum of squares over T pixel
 * chunks), T = idf_gn_partials_chunks(B, HW) -- the st_out of the conv entry points, for tensors that were not
 * produced by one of them (network inputs, fallback paths). */
int idf_gn_partials_chunks(int B, int HW);
int idf_gn_partials(const void* x, float* part, int B, int HW, int C, int dtype, void* stream);
/* a = act(x*sc+sh) materialised once (act 1 affine, 2 SiLU + dropout): GroupNorm-apply + FiLM +
 * SiLU + Dropout of modules.py:264-288, 312-319 as one read + one write */
int idf_gn_apply(const void* x, void* out, const float* sc, const float* sh, const uint64_t* seed, uint32_t salt,
                 float p_drop, int act, int B, int HW, int C, int dtype, void* stream);
/* idf_gn_apply over the never-materialised concatenation x [.., C1] | x2 [.., C - C1] of a skip pair (models.py:321);
 * out is dense over all C channels. */
int idf_gn_apply2(const void* x, const void* x2, int C1, void* out, const float* sc, const float* sh, const uint64_t* seed,
                  uint32_t salt, float p_drop, int act, int B, int HW, int C, int dtype, void* stream);
/* Backward through act(GN/FiLM(x)) given dA (gradient w.r.t. the activated tensor):
 * dx (+ dres), dfilm_t/dfilm_a [B,2C], dgb [B][2][C] (per-sample dgamma, dbeta;
 * sum over B with idf_colsum) when non-NULL, and/or atomic accumulation of the batch sums
 * straight into dgamma_acc[C] / dbeta_acc[C] (gradient arena) when non-NULL; k1/k0 [B,32] scratch. */
int idf_gn_coef_bwd(const void* dA, const void* x, const void* dres, void* dx, const float* gamma,
                    const float* beta, const float* film_t, const float* film_a, int ld_t, int ld_a,
                    const float* mean, const float* rstd, const float* sc, const float* sh, float* dfilm_t, float* dfilm_a,
                    float* dgb, float* dgamma_acc, float* dbeta_acc, float* k1, float* k0, float* workspace,
                    const uint64_t* seed, uint32_t salt, float p_drop, int act, int B, int HW, int C, int dtype,
                    void* stream);

/* One-launch forms for small samples (the 16x16 and 8x8 levels): statistics + fold + apply, and the
 * whole backward, one workgroup per (sample, slice of whole groups).  IDF_ERR_UNSUPPORTED for shapes
 * idf_gn_fused_ok() reports 0 for (use the three-launch forms above). */
int idf_gn_fused_ok(int B, int HW, int C, int C1, int dtype);   /* C1 > 0: two-source input x [..,C1] | x2 [..,C-C1] */
int idf_gn_fused_fwd(const void* x, const void* x2, int C1, void* out, const float* gamma, const float* beta, const float* film_t,
                     const float* film_a, int ld_t, int ld_a, float eps, float* mean, float* rstd, float* sc,
                     float* sh, const uint64_t* seed, uint32_t salt, float p_drop, int act, int B, int HW, int C,
                     int dtype, void* stream);
int idf_gn_fused_bwd(const void* dA, const void* x, const void* x2, int C1, const void* dres, const void* dres2,
                     void* dx, void* dx2,
                     const float* gamma, const float* beta,
                     const float* film_t, const float* film_a, int ld_t, int ld_a, const float* mean,
                     const float* rstd, const float* sc, const float* sh, float* dfilm_t, float* dfilm_a, float* dgb,
                     float* dgamma_acc, float* dbeta_acc, const uint64_t* seed, uint32_t salt, float p_drop, int act,
                     int B, int HW, int C, int dtype, void* stream);

/* ---- dense contractions: attention bmm's (modules.py:152-159), linears
 * (modules.py:22-27, 269-276; models.py:244, 470-472, LatentUNet 147-163) and gradients.
 * C[b][m][n] = alpha * sum_k opA[m][k]*opB[n][k] (+bias[n]) (+res[b][m][n]); ta/tb = 1 when the operand
 * is stored K-major ([K][M] / [K][N]).  splitk > 1 accumulates with fp32 atomics into a
 * pre-zeroed fp32 C (out_f32 = 1). */
int idf_bgemm(const void* A, const void* B, void* C, const float* bias, const void* res, int batch, long sA,
              long sB, long sC,
              int lda, int ldb, int ldc, int M, int N, int K, int ta, int tb, float alpha, int out_f32,
              int splitk, int dtype, void* stream);
/* The conditioning path of a UNet as one entry point per direction (three launches forward, four backward: products
 * of the same depth in the chain share a launch): TimeEmbedding (modules.py:9-38: table lookup -> Linear -> SiLU -> Linear), the latent embedding fc_a
 * (models.py:298-301; fc_silu = 1 for the Bottleneck's Sequential(SiLU, Linear), models.py:371), and every block's
 * FiLM projections Linear(SiLU(emb)) (modules.py:269-276) over the concatenated weights Wt [Nt][dim], Wa [Na][dim].
 * All fp32.  t [B] int64, table [T][d_model], W1 [dim][d_model], W2 [dim][dim], a [B][a_dim] or NULL (no latent
 * branch), Wfc [dim][a_dim].  Kept for the backward ([B][dim] each): h1, s1 = SiLU(h1), temb, st = SiLU(temb), aemb,
 * sa = SiLU(aemb).  Outputs film_t [B][Nt], film_a [B][Na]. */
int idf_temb_film_fwd(const long long* t, const float* table, int d_model, const float* W1, const float* b1,
                      const float* W2, const float* b2, int dim, const float* a, int a_dim, const float* Wfc,
                      const float* bfc, int fc_silu, const float* Wt, const float* bt, int Nt, const float* Wa,
                      const float* ba, int Na, float* h1, float* s1, float* temb, float* st, float* aemb, float* sa,
                      float* film_t, float* film_a, int B, void* stream);
/* Its backward: every gradient pointer is optional (NULL = not wanted).  Scratch: dS, (idf_temb_film_parts(Nt) +
 * idf_temb_film_parts(Na)) * B * dim floats -- the K slices of dfilm * W, summed in slice order (no atomics: the
 * latent's gradient is bit-reproducible) -- and g1 [B][dim]; B * dim % 4 == 0.  Gradients are written (not
 * accumulated) in the parameters' own layouts. */
int idf_temb_film_parts(int N);
int idf_temb_film_bwd(const float* dfilm_t, const float* dfilm_a, const long long* t, const float* table, int d_model,
                      const float* W2, int dim, const float* a, int a_dim, const float* Wfc, int fc_silu, const float* Wt,
                      int Nt, const float* Wa, int Na, const float* h1, const float* s1, const float* temb, const float* st,
                      const float* aemb, const float* sa, float* dS, float* g1, float* dWt, float* dbt, float* dWa,
                      float* dba, float* dW2, float* db2, float* dW1, float* db1, float* dWfc, float* dbfc, float* da, int B,
                      void* stream);
int idf_softmax_fwd(void* s, long R, int N, int dtype, void* stream);            /* modules.py:156 */
int idf_softmax_bwd(const void* P, void* dP, long R, int N, int dtype, void* stream);

/* Fused single-head attention of the AttnBlock (modules.py:129-164: bmm, softmax, bmm and their
 * backward) for the shapes idf_attn_fused_ok() accepts (N = 256 or 64 tokens -- the 16x16 levels and the 8x8 middle
 * block --, D = C in {64, 128}, bf16):
 * qkv [B, N, 3D] (q | k | v along channels), o [B, N, D], lse [B, N] row logsumexp kept for the
 * backward, dsum [B, N] scratch (row sums of P dP), dqkv [B, N, 3D].  scale = C^-1/2.
 * Scores / probabilities stay in registers; other shapes use idf_bgemm + idf_softmax_*. */
int idf_attn_fused_ok(int N, int D, int dtype);
int idf_attn_fwd(const void* qkv, void* o, float* lse, int B, int N, int D, float scale, int dtype, void* stream);
int idf_attn_bwd(const void* qkv, const void* dO, const float* lse, float* dsum, void* dqkv, int B, int N, int D,
                 float scale, int dtype, void* stream);
/* The same backward as ONE launch (round 3): given the forward's output o [B, N, D], the key-value half forms the row sums
 * sum_j P dP as dO . o (the same number up to the bf16 rounding of o), so the two halves depend on nothing of each other
 * and share a grid (query blocks | key-value blocks: 2 x 128 workgroups at B = 32). */
int idf_attn_bwd_o(const void* qkv, const void* dO, const float* lse, const void* o, void* dqkv, int B, int N, int D,
                   float scale, int dtype, void* stream);

/* The proj conv folded into V (round 4): (P V) Wp^T = P (V Wp^T), so with Wv' = Wp Wv and b' = Wp bv + bp (the rows of P sum to one:
 * a bias on V' passes through the product unchanged) the block is y = x + P V' -- no proj launch, no proj data gradient, no proj weight
 * gradient.  idf_attn_fold_batched: Wv' [C][C] and (bq | bk | b') [3C] of every attention block of a network in one launch -- table
 * (device) = nrows x {const float* wp, bp, wv, bv, bq, bk; float* wvf, bf; int C, pad} (72 bytes).  idf_attn_fwd_res: idf_attn_fwd with
 * y = x + o and the statistics partials of y (st_out [B][idf_attn_res_tiles()][D][2]) in its epilogue; o / lse optional, together.
 * idf_attn_fold_bwd_batched: the chain rule back to the parameters, in place in the gradient arena after the weight gradients ran
 * -- table = nrows x {float* g, gb; const float* wp, wv, bv; float* dwp, dbp, gs; int C, pad} (72 bytes): g = dL/dWv' and gb = dL/db'
 * arrive in proj_v's slots and leave as dWv = Wp^T g, dbv = Wp^T gb; dwp = g Wv^T + gb bv^T and dbp = gb are written; gs = scratch
 * of C * C + C floats per row (a copy of g / gb between the two passes). */
int idf_attn_fold_batched(const void* table, int nrows, int max_C, void* stream);
int idf_attn_fold_bwd_batched(const void* table, int nrows, int max_C, void* stream);
int idf_attn_res_tiles(int B, int N, int D, int dtype);
int idf_attn_fwd_res(const void* qkv, const void* xres, void* o, float* lse, void* y, float* st_out, int B, int N, int D,
                     float scale, void* stream);

/* The whole attention block (modules.py:145-164) of the N = 256-token level at C = 128 in ONE launch (round 4), the proj conv
 * folded into V (idf_attn_fold_batched above):
 *   y = x + softmax(q k^T scale) v',   q | k | v' = conv1x1(GroupNorm(x))
 * one workgroup per image; x [B, N, C] bf16 is the only activation read.  st [B][T][C][2]: the statistics partials x's producer
 * left behind (the GroupNorm's mean / rstd are folded from them in-kernel, as the GroupNorm-prologue convs do); gamma / beta:
 * the GroupNorm's affine; wqkv_frag: the q | k | v' weights [3C][C] in the fragment-major shadow form
 * (idf_pack_conv_weights_batched, `wfrag`, taps 1); bqkv [3C] = (bq | bk | b').
 * y [B, N, C]; st_out [B][1][C][2] (optional): statistics partials of y.  Training outputs, all or none: qkv [B, N, 3C], h =
 * GroupNorm(x) [B, N, C] (the q / k / v weight gradient's operand), o [B, N, C], lse [B, N], mean / rstd [B, 32], sc / sh [B, C] --
 * exactly what idf_conv_gn_bf16 + idf_attn_fwd_res leave behind, so the backward pass is the existing launches. */
int idf_attnblock_ok(int N, int C, int dtype);
int idf_attnblock_fwd(const void* x, const float* st, int T, const float* gamma, const float* beta, float eps,
                      const void* wqkv_frag, const float* bqkv, void* y, float* st_out, void* qkv, void* h, void* o, float* lse,
                      float* mean, float* rstd, float* sc, float* sh, float scale, int B, int N, int C, void* stream);

/* ---- elementwise / reductions */
/* Input pipeline on the device (reference data.py:149-171, ToTensor -> RandomHorizontalFlip -> Normalize):
 * uint8 NHWC image bytes -> fp32 NHWC-dense (x / 255 - 0.5) / 0.5, bit-identical to the torchvision chain;
 * flip [B] (or NULL): non-zero mirrors that sample horizontally.  The batch crosses PCIe as bytes. */
int idf_prep_u8(const uint8_t* src, const uint8_t* flip, float* dst, int B, int H, int W, int C, void* stream);

/* q_sample, models.py:702-704: xt = sqrt_ab[idx[b]]*x + sqrt_1mab[idx[b]]*eps; the two [T] tables
 * hold sqrt(alpha_bar), sqrt(1-alpha_bar) (host torch CPU ops => bit-exact gathers and fp32 result) */
int idf_qsample(const float* x, const float* eps, const long* idx, const float* sqrt_ab, const float* sqrt_1mab,
                float* xt32, void* xt, long per_sample, long n, int dtype, void* stream);
/* nn.Embedding lookup of the frozen sinusoid table, modules.py:23 */
int idf_gather_rows(const float* table, const long* idx, float* out, int B, int D, void* stream);
int idf_silu_fwd(const float* x, float* y, long n, void* stream);
int idf_silu_bwd(const float* x, const float* dy, float* dx, long n, void* stream);
/* models.py:640-646: res[0] = mean((out-eps)^2); res[1] = mean((c0*(x-c1*out)-x)^2)/T */
int idf_loss_fwd(const void* out, const float* eps, const float* x, float c0, float c1, float inv_T, float* res,
                 float* workspace, long n, int dtype, void* stream);
/* dout from the upstream gradients g[0] (denoise term) and g[g_stride] (recon term); g_stride 0: one scalar for both */
int idf_loss_bwd(const void* out, const float* eps, const float* x, float c0, float c1, float inv_T,
                 const float* g, int g_stride, void* dout, long n, int dtype, void* stream);
/* models.py:640-646 + 674-678 (the --mmd_weight branch) as one objective: res[4] = {denoise, recon, mmd(prior, lat),
 * (denoise + recon) + w_mmd * mmd}; prior [n,D], lat [m,D] fp32; workspace 2048 + 2n + m floats.  Backward: idf_loss_bwd
 * with g_stride 0 and idf_mmd_bwd with gscale = w_mmd, both on the gradient of res[3]. */
int idf_objective_fwd(const void* out, const float* eps, const float* x, float c0, float c1, float inv_T,
                      const float* prior, const float* lat, int n, int m, int D, float w_mmd, float* res,
                      float* workspace, long numel, int dtype, void* stream);
/* sampling.py:29-37 (mode 0 DDPM), 52-59 (1 DDIM as written, eta 0.01), 71-72 (2 reverse DDIM).
 * coef: [T][8] per-step scalars {c0,c1,c2,c3,d0,d1,sigma,-} built on the host with the reference's
 * fp32 expressions; *idx selects the row.  x/xo fp32 state, eps (and optional xo_t copy) in dtype. */
int idf_sampler_step(const float* x, const void* eps, const float* noise, float* xo, void* xo_t,
                     const long* idx, const float* coef, int mode, long n, int dtype, void* stream);
/* utils.py:74-90 RBF-kernel MMD (bandwidth dim^2); workspace 2n+m floats */
int idf_mmd_fwd(const float* x, const float* y, int n, int m, int D, float* out, float* workspace, void* stream);
int idf_mmd_bwd(const float* x, const float* y, int n, int m, int D, const float* g, float gscale, float* dy,
                void* stream);      /* dy = g[0] * gscale * d mmd / d y */
/* out[N] = sum over R rows; workspace idf_colsum_blocks(R)*N floats */
int idf_colsum_blocks(long R);
int idf_colsum(const void* in, float* out, float* workspace, long R, int N, int in_dtype, void* stream);
/* 2x2 sum pool [B,2Ho,2Wo,C] -> [B,Ho,Wo,C]: data gradient of the fused nearest upsample */
int idf_pool2_sum(const void* in, void* out, int B, int Ho, int Wo, int C, int dtype, void* stream);
/* latent denoiser row kernel (MLPLNAct, models.py:147-163): y = Dropout(SiLU(LayerNorm(lin*(1+cond))*g+b));
 * stats [R][2] = (mean, rstd); backward writes dlin, dcond and per-row dg|db contributions dgb [R][2W] */
int idf_ln_silu_fwd(const float* lin, const float* cond, const float* g, const float* b, float* y, float* stats,
                    int R, int Wd, float eps, const uint64_t* seed, uint32_t salt, float p_drop, void* stream);
int idf_ln_silu_bwd(const float* lin, const float* cond, const float* g, const float* b, const float* stats,
                    const float* dy, float* dlin, float* dcond, float* dgb, int R, int Wd, const uint64_t* seed,
                    uint32_t salt, float p_drop, void* stream);
/* fused optimizer tail (run.py:177,199-200): global grad-norm clip + AdamW (decoupled decay) over all
 * tensors.  table: nchunks x {float* p, float* g, float* m, float* v, long n} (device memory);
 * partial: nchunks floats; state: 8 floats, state[0] = step count (persistent), [4] = pre-clip norm;
 * lr: device float. */
int idf_clip_adamw(const void* table, int nchunks, float* partial, float* state, const float* lr, float max_norm,
                   float b1, float b2, float eps, float wd, int write_clipped_grads, void* stream);
/* test hook: the dropout keep-mask (scaled) a call site would apply */
int idf_dropout_mask(const uint64_t* seed, uint32_t salt, float p_drop, float* mask, long n, void* stream);

#ifdef __cplusplus
}
#endif
#endif
