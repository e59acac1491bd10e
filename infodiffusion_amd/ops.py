"""Host-side operators over the C ABI (include/infodiff_hip.h).

Raw launchers (`*_raw`) take/return torch CUDA tensors whose memory is dense NHWC
(`channels_last` with the reference's logical NCHW shape) and enqueue on the
current stream; `torch.autograd.Function`s on top provide backward through the
hand-written gradient kernels.  There is no non-HIP fallback.
"""
import ctypes
import functools
import os

import torch

from . import _lib, knobs
from ._lib import call, F32, BF16
from .grad_arena import slot_of, ParamGroup  # noqa: F401  (modules.py reaches ParamGroup through ops)

GN_EPS = 1e-5
S1, S2, UP2, T2 = 0, 1, 2, 3
CL = torch.channels_last


def _dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError('unsupported activation dtype %s' % t.dtype)


def _p(t, off=0):
    if t is None:
        return None
    return t.data_ptr() + off * t.element_size()


def _st():
    return torch.cuda.current_stream().cuda_stream


def _nhwc(t):
    """Dense NHWC memory for a logical NCHW tensor."""
    if not t.is_cuda:
        raise RuntimeError('infodiffusion_amd kernels run on the GPU only (got a %s tensor)' % t.device)
    return t.contiguous(memory_format=CL)


def _f32c(t):
    return None if t is None else t.contiguous().float()


def _ld(film):
    """Row stride (floats) of a [B, 2C] FiLM tensor that may be a column slice of a batched projection."""
    if film is None:
        return 0
    assert film.dtype == torch.float32 and film.stride(1) == 1
    return film.stride(0)


def empty_nhwc(B, C, H, W, dtype, device):
    return torch.empty((B, C, H, W), dtype=dtype, device=device, memory_format=CL)


# ------------------------------------------------------------------ raw calls
def pack_weight(weight, dtype, want_fwd, want_dgrad):
    """weight: fp32 [O, I, kh, kw] (any strides with stride[2] == kw*stride[3])."""
    O, I, kh, kw = weight.shape
    taps = kh * kw
    w = weight.detach()
    if taps > 1 and w.stride(2) != kw * w.stride(3):
        w = w.contiguous()
    st = w.stride(3) if taps > 1 else 0
    wf = torch.empty((O, taps, I), dtype=dtype, device=w.device) if want_fwd else None
    wd = torch.empty((I, taps, O), dtype=dtype, device=w.device) if want_dgrad else None
    call('idf_pack_conv_weight', _p(w), w.stride(0), w.stride(1), st, _p(wf), _p(wd), O, I, taps,
         F32 if dtype == torch.float32 else BF16, _st())
    return wf, wd


def out_hw(mode, H, W):
    if mode == S2:
        return (H + 1) // 2, (W + 1) // 2
    if mode == UP2:
        return 2 * H, 2 * W
    return H, W


def _halo_fits_s2(H, W):
    """Stride-2 forward through the halo kernel: 64-pixel output tiles, (2R+1) x (2W+1) staged pixels."""
    if W > 32:
        return False
    R = max(1, min(H, 64 // W))
    while H % R:
        R -= 1
    return (2 * R + 1) * (2 * W + 1) * 4 <= 1536


def _halo_fits(H, W, B, Cout):
    BM = 128 if (B * H * W // 128 * -(-Cout // 64) >= 256 and H * W >= 128) else 64
    if B * H * W // 256 * -(-Cout // 64) >= 256 and H * W >= 256 and Cout > 32:
        BM = 256
    while True:
        R = max(1, min(H, BM // W))
        while H % R:
            R -= 1
        if BM == 256 and (R + 2) * (W + 2) * 4 > 2048:
            BM = 128
            continue
        return (R + 2) * (W + 2) * 4 <= (2048 if BM == 256 else 1280)


def uses_halo_kernel(dtype, taps, act, mode, B, Cin, Cout, Ho, Wo):
    """True when conv_raw routes to conv3x3_halo_bf16 (idf_conv3x3.hip)."""
    if dtype != torch.bfloat16 or taps != 9 or act != 0 or Cin % 32 or not (4 <= Wo <= 128) or (Wo & (Wo - 1)):
        return False
    if mode == S2:
        return Cout > 32 and _halo_fits_s2(Ho, Wo)
    return _halo_fits(Ho, Wo, B, Cout)


_CONV1X1 = True


@functools.lru_cache(maxsize=None)
def conv_tiles(B, H, W, Cin, Cout, mode, taps, pro=0):
    """Pixel tiles per image of the conv launch for this shape (-1: not covered) = T of its statistics; pro: the
    launch is the GroupNorm-prologue conv."""
    return int(_lib.load().idf_conv_tiles(B, H, W, Cin, Cout, {S1: 0, S2: 1, UP2: 2, T2: 3}[mode], taps, int(pro)))


def _new_stats(B, Ho, Wo, Cin, Cout, mode, taps, device, pro=0):
    """Buffer for the per-channel GroupNorm statistics partials a conv launch writes for its output, or None."""
    if Cout % 8:
        return None
    T = conv_tiles(B, Ho, Wo, Cin, Cout, mode, taps, pro)
    return torch.empty((B, T, Cout, 2), dtype=torch.float32, device=device) if T > 0 else None


_FEWC = True       # the head conv (Cin <= 3) as one MFMA K-step (idf_conv3x3_fewc_bf16)


@functools.lru_cache(maxsize=None)
def fewc_tiles(B, H, W, Cin, Cout):
    return int(_lib.load().idf_conv_fewc_tiles(B, H, W, Cin, Cout))


def conv_raw(x, w_fwd, bias, residual, sc, sh, seed, salt, p_drop, mode, taps, act, Cout, out_hw_=None, want_stats=False):
    """x logical [B,Cin,Hs,Ws] NHWC-dense; w_fwd [Cout][taps][Cin] in x.dtype.  want_stats: returns (y, st) where
    st = the statistics partials of y (None when the kernel that ran does not produce them)."""
    B, Cin, Hs, Ws = x.shape
    Ho, Wo = out_hw_ if out_hw_ is not None else out_hw(mode, Hs, Ws)
    y = empty_nhwc(B, Cout, Ho, Wo, x.dtype, x.device)
    st = None

    def done():
        return (y, st) if want_stats else y
    if (taps == 1 and mode == S1 and act == 0 and x.dtype == torch.bfloat16 and Cin % 32 == 0 and Cout % 8 == 0
            and 4 <= Wo <= 128 and not (Wo & (Wo - 1)) and _CONV1X1):
        # 1x1 conv through the halo-conv pipeline without the halo (LDS-swizzled tiles, full-line epilogue)
        if want_stats:
            st = _new_stats(B, Ho, Wo, Cin, Cout, S1, 1, x.device)
        call('idf_conv1x1_bf16', _p(x), None, 0, _p(w_fwd), _p(bias), _p(residual), _p(y), B, Ho, Wo, Cin, Cout, _p(st),
             _st())
        return done()
    if taps == 1 and mode == S1 and act == 0 and Cin % 8 == 0 and Cout % 4 == 0:
        # 1x1 conv = [pixels, Cin] x [Cout, Cin]^T (+bias, +residual): the short-K GEMM
        M = B * Ho * Wo
        bgemm_raw(x, 0, w_fwd, 0, y, 0, bias, 1, 0, 0, 0, Cin, Cin, Cout, M, Cout, Cin, 0, 0, res=residual)
        return done()
    if (_FEWC and taps == 9 and mode == S1 and act == 0 and x.dtype == torch.bfloat16 and Cin <= 3 and residual is None
            and fewc_tiles(B, Ho, Wo, Cin, Cout) > 0):
        # the head conv: one MFMA K-step, statistics of y for the first GroupNorm on the way out
        if want_stats:
            st = torch.empty((B, fewc_tiles(B, Ho, Wo, Cin, Cout), Cout, 2), dtype=torch.float32, device=x.device)
        call('idf_conv3x3_fewc_bf16', _p(x), _p(w_fwd), _p(bias), _p(y), B, Ho, Wo, Cin, Cout, _p(st), _st())
        return done()
    if uses_halo_kernel(x.dtype, taps, act, mode, B, Cin, Cout, Ho, Wo):
        if want_stats:
            st = _new_stats(B, Ho, Wo, Cin, Cout, mode, 9, x.device)
        call('idf_conv3x3_bf16', _p(x), _p(w_fwd), _p(bias), _p(residual), _p(y), B, Ho, Wo, Cin, Cout,
             {S1: 0, S2: 1, UP2: 2, T2: 3}[mode], _p(st), _st())
        return done()
    call('idf_conv2d_fwd', _p(x), _p(w_fwd), _p(bias), _p(residual), _p(y), _p(sc), _p(sh), _p(seed),
         salt, float(p_drop), B, Hs, Ws, Cin, Ho, Wo, Cout, mode, taps, act, _dt(x), _st())
    return done()


_GN_FUSE = knobs.flag('IDF_GN_FUSE')


def gn_partials_raw(x):
    """Per-channel statistics partials [B, T, C, 2] of a tensor that did not come out of a conv launch."""
    B, C, H, W = x.shape
    T = int(_lib.load().idf_gn_partials_chunks(B, H * W))
    st = torch.empty((B, T, C, 2), dtype=torch.float32, device=x.device)
    call('idf_gn_partials', _p(x), _p(st), B, H * W, C, _dt(x), _st())
    return st


def stats_of(x):
    """The statistics partials riding on x (left by its producer), else one stand-alone pass over it."""
    st = getattr(x, '_gn', None)
    if st is not None and st.shape[0] == x.shape[0] and st.shape[2] == x.shape[1]:
        return st
    return gn_partials_raw(x)


def conv_gn_ok(x, x2, taps, Cout, advice=True):
    """The one-launch GroupNorm-prologue conv (idf_conv_gn_bf16) covers this input -- and, with `advice`, the library's
    measured policy (idf_conv_gn_advice) prefers it over the two-launch path at this shape."""
    if not (_GN_FUSE and x.is_cuda and x.dtype == torch.bfloat16):
        return False
    B, C1, H, W = x.shape
    Cin = C1 + (x2.shape[1] if x2 is not None else 0)
    if Cin % 32 or (x2 is not None and (C1 % 32 or x2.dtype != torch.bfloat16 or x2.shape[2:] != x.shape[2:])):
        return False
    if taps == 1 and not _CONV1X1:
        return False
    return conv_tiles(B, H, W, Cin, Cout, S1, taps, 1) > 0 and (not advice or _gn_advice(B, H, W, Cin, Cout, taps))


@functools.lru_cache(maxsize=None)
def _gn_advice(B, H, W, Cin, Cout, taps):
    return bool(_lib.load().idf_conv_gn_advice(B, H, W, Cin, Cout, taps))


_WR = knobs.flag('IDF_CONV_WR')          # the small-map convs with fragment-major weights (idf_conv_wr_*)
_WR_MAXB = 64   # launch-bound batches only (64 / 256 workgroups of 4 waves)


@functools.lru_cache(maxsize=None)
def wr_tiles(B, H, W, Cin, Cout, whole):
    return int(_lib.load().idf_conv_wr_tiles(B, H, W, Cin, Cout, int(whole)))


def _wr_frag(shadows, j, B, H, W, Cin, Cout, whole):
    """The fragment-major shadow j (2 forward, 3 data gradient) when idf_conv_wr_* covers the shape and the shadow exists; a
    conv met here for the first time is asked for one (it comes with the next re-pack)."""
    if not (_WR and shadows is not None and B <= _WR_MAXB and wr_tiles(B, H, W, Cin, Cout, whole) > 0):
        return None
    v = shadows.val[j]
    if v is None:
        shadows.request_frag()
    return v


_RS = knobs.flag('IDF_CONV_RS')          # the big-map ResBlock convs in the register-weights / row-reuse form (idf_conv_rs_*)
_RS_FWD_ALL = knobs.flag('IDF_CONV_RS_FWD')   # ... for every covered forward conv too (default: where it measured faster)


@functools.lru_cache(maxsize=None)
def rs_tiles(B, H, W, Cin, Cout):
    return int(_lib.load().idf_conv_rs_tiles(B, H, W, Cin, Cout))


@functools.lru_cache(maxsize=None)
def rs_fwd_tiles(B, H, W, Cin, Cout):
    """T of the statistics partials idf_conv_rs_gn_bf16 writes (half-height tiles where the forward conv computes both cout tiles of
    a pixel tile from one halo image: 32x32 maps with Cout = 128)."""
    return int(_lib.load().idf_conv_rs_fwd_tiles(B, H, W, Cin, Cout))


def _rs_frag(shadows, j, B, H, W, Cin, Cout):
    """The fragment-major shadow j (2 forward, 3 data gradient) when idf_conv_rs_* covers the shape and the shadow exists; a conv
    met here for the first time is asked for one (it comes with the re-pack at the end of this forward pass / the next one)."""
    if not (_RS and shadows is not None and rs_tiles(B, H, W, Cin, Cout) > 0):
        return None
    v = shadows.val[j]
    if v is None:
        shadows.request_frag()
    return v


def conv_gn_raw(x, x2, st1, st2, gn_w, gn_b, film_t, film_a, seed, salt, p_drop, act, w_fwd, bias, residual, Cout, taps,
                keep_a=False, keep_coef=False, want_stats=False, shortcut=None, shadows=None):
    """y = conv(act(GN/FiLM(x | x2))) + bias (+ residual) in one launch -> (y, a, mean, rstd, sc, sh, st_out);
    a / the coefficients are None unless asked for (training).  shortcut = (w_sc [Cs][Cin] kernel layout, bias_sc, Cs): the
    same launch also computes s = conv1x1(x | x2) + bias_sc (the block's shortcut over the raw input), returned last."""
    B, C1, H, W = x.shape
    Cin = C1 + (x2.shape[1] if x2 is not None else 0)
    dev = x.device
    y = empty_nhwc(B, Cout, H, W, x.dtype, dev)
    a = empty_nhwc(B, Cin, H, W, x.dtype, dev) if keep_a else None
    mean = rstd = sc = sh = None
    if keep_coef:
        mean = torch.empty((B, 32), dtype=torch.float32, device=dev)
        rstd = torch.empty((B, 32), dtype=torch.float32, device=dev)
        sc = torch.empty((B, Cin), dtype=torch.float32, device=dev)
        sh = torch.empty((B, Cin), dtype=torch.float32, device=dev)
    wfrag = _wr_frag(shadows, 2, B, H, W, Cin, Cout, 0) if (taps == 9 and act == 2 and shortcut is None) else None
    if wfrag is not None:
        # 16x16 / 8x8: weights fragment-major into registers, barrier-free conv loop, epilogue in the wave's registers
        st = torch.empty((B, wr_tiles(B, H, W, Cin, Cout, 0), Cout, 2), dtype=torch.float32, device=dev) if want_stats else None
        call('idf_conv_wr_gn_bf16', _p(x), _p(x2), C1, _p(st1), st1.shape[1], _p(st2), st2.shape[1] if st2 is not None else 0,
             _p(gn_w), _p(gn_b), _p(film_t), _p(film_a), _ld(film_t), _ld(film_a), GN_EPS, _p(seed), salt, float(p_drop),
             _p(wfrag), _p(bias), _p(residual), _p(y), _p(a), _p(mean), _p(rstd), _p(sc), _p(sh), _p(st), B, H, W, Cin, Cout, _st())
        return y, a, mean, rstd, sc, sh, st
    rfrag = None
    # forward: the row-reuse form where it measured faster than the halo / direct-to-LDS kernels -- the channel-changing convs of the
    # 32x32 maps (64->128: 17.3 vs 19.0 us, 128->64: 19.0 vs 25.2 us); at 64->64 @64x64 and 128->128 @32x32 the GroupNorm transform is
    # vector-bound either way (29.4 vs 28.2 us, 28.3 vs 27.8 us: profiles/r05_conv_rs.txt) and the older kernels stay
    # ... and every covered shape in inference (no activated tensor kept: DDIM-100 at B = 256 335 -> 349 img/s, the coefficient fold
    # runs once per workgroup and image half instead of in a launch of its own)
    # round 5, later: at Cout = 128 on the 32x32 maps a workgroup computes BOTH cout tiles from one halo image (the transform runs once
    # per tile instead of twice): that form takes 128->128 in training too
    if (taps == 9 and x2 is None and shortcut is None and x.dtype == torch.bfloat16 and st1.shape[1] <= 32
            and (_RS_FWD_ALL or Cin != Cout or not keep_a or (W == 32 and Cout == 128 and _RS_SHARED))):
        rfrag = _rs_frag(shadows, 2, B, H, W, Cin, Cout)
    if rfrag is not None:
        # 64x64 / 32x32: weights fragment-major into registers, whole-K halo image in LDS, row reuse, persistent over the CU's tiles
        st = torch.empty((B, rs_fwd_tiles(B, H, W, Cin, Cout), Cout, 2), dtype=torch.float32, device=dev) if (want_stats and Cout % 8 == 0) else None
        call('idf_conv_rs_gn_bf16', _p(x), _p(st1), st1.shape[1], _p(gn_w), _p(gn_b), _p(film_t), _p(film_a), _ld(film_t), _ld(film_a),
             GN_EPS, act, _p(seed), salt, float(p_drop), _p(rfrag), _p(bias), _p(residual), _p(y), _p(a), _p(mean), _p(rstd), _p(sc),
             _p(sh), _p(st), B, H, W, Cin, Cout, _st())
        return y, a, mean, rstd, sc, sh, st
    st = _new_stats(B, H, W, Cin, Cout, S1, taps, dev, 1) if want_stats else None
    ws = torch.empty((B, Cin, 2), dtype=torch.float32, device=dev) if B * H * W >= (1 << 18) else None
    args = (_p(x), _p(x2), C1, _p(st1), st1.shape[1], _p(st2), st2.shape[1] if st2 is not None else 0,
            _p(gn_w), _p(gn_b), _p(film_t), _p(film_a), _ld(film_t), _ld(film_a), GN_EPS, act, _p(seed), salt, float(p_drop),
            _p(w_fwd), _p(bias), _p(residual), _p(y), _p(a), _p(mean), _p(rstd), _p(sc), _p(sh), _p(st), _p(ws), B, H, W, Cin,
            Cout, taps, _st())
    if shortcut is not None:
        w_sc, b_sc, Cs = shortcut
        s = empty_nhwc(B, Cs, H, W, x.dtype, dev)
        call('idf_conv_gn_sc_bf16', *args, _p(w_sc), _p(b_sc), _p(s), Cs)
        return y, a, mean, rstd, sc, sh, st, s
    call('idf_conv_gn_bf16', *args)
    return y, a, mean, rstd, sc, sh, st


def conv_dgrad_raw(dy, w_dgrad, mode_fwd, taps, x_shape, residual=None, shadows=None):
    """Data gradient w.r.t. the (activated) conv input of logical shape x_shape (+ residual: a gradient
    arriving over another branch of the same input, added in the epilogue)."""
    B, Cin, Hs, Ws = x_shape
    subd = getattr(shadows, 'subd', None) if mode_fwd == UP2 else None
    if (subd is not None and _UPCONV and _UPCONV_DGRAD and dy.dtype == torch.bfloat16 and dy.is_cuda
            and _lib.load().idf_upconv_dgrad_ok(Hs, Ws, Cin, dy.shape[1])):
        # UpSample: 16 tap products per low-resolution pixel (summed sub-pixel weights) instead of the 3x3 conv over dy + pool pass
        out = empty_nhwc(B, Cin, Hs, Ws, dy.dtype, dy.device)
        call('idf_upconv_dgrad_bf16', _p(_nhwc(dy)), _p(subd), _p(out), B, Hs, Ws, Cin, dy.shape[1], _st())
        return out if residual is None else out + residual
    if mode_fwd == S2:
        if (shadows is not None and _DOWN_DGRAD and taps == 9 and dy.dtype == torch.bfloat16 and dy.is_cuda and Hs % 2 == 0 and Ws % 2 == 0
                and _lib.load().idf_downconv_dgrad_ok(Hs // 2, Ws // 2, Cin, dy.shape[1])):
            # DownSample: per parity of the high-resolution pixel only the taps that land on a dy pixel (9 tap products per four
            # outputs instead of the 36 of the 3x3 conv over the zero-stuffed dy)
            frag = shadows.val[3]
            if frag is None:
                shadows.request_frag()         # fragment-major data-gradient weights: from the next re-pack on
            else:
                out = empty_nhwc(B, Cin, Hs, Ws, dy.dtype, dy.device)
                call('idf_downconv_dgrad_bf16', _p(_nhwc(dy)), _p(frag), _p(_nhwc(residual) if residual is not None else None),
                     _p(out), B, Hs // 2, Ws // 2, Cin, dy.shape[1], _st())
                return out
        return conv_raw(dy, w_dgrad, None, residual, None, None, None, 0, 0.0, T2, taps, 0, Cin, (Hs, Ws))
    if mode_fwd == UP2:
        up = conv_raw(dy, w_dgrad, None, None, None, None, None, 0, 0.0, S1, taps, 0, Cin)
        out = empty_nhwc(B, Cin, Hs, Ws, dy.dtype, dy.device)
        call('idf_pool2_sum', _p(up), _p(out), B, Hs, Ws, Cin, _dt(dy), _st())
        return out if residual is None else out + residual
    return conv_raw(dy, w_dgrad, None, residual, None, None, None, 0, 0.0, S1, taps, 0, Cin)


def conv_wgrad_raw(x, dy, sc, sh, seed, salt, p_drop, mode, taps, act):
    """Returns fp32 dW with logical shape [O, I, kh, kw] and memory [O][taps][I]."""
    B, Cin, Hs, Ws = x.shape
    _, Cout, Ho, Wo = dy.shape
    k = 3 if taps == 9 else 1
    dW = torch.empty((Cout, k, k, Cin), dtype=torch.float32, device=x.device)
    call('idf_conv2d_wgrad', _p(x), _p(dy), _p(dW), _p(sc), _p(sh), _p(seed), salt, float(p_drop),
         B, Hs, Ws, Cin, Ho, Wo, Cout, mode, taps, act, _dt(x), _st())
    return dW.permute(0, 3, 1, 2)


def colsum_raw(t2d, out=None):
    """fp32 column sums of a [R, N] view (fp32 or bf16, dense); `out`: a dense fp32 [N] destination."""
    R, N = t2d.shape
    nb = _lib.load().idf_colsum_blocks(R)
    ws = torch.empty((nb * N,), dtype=torch.float32, device=t2d.device)
    if out is None:
        out = torch.empty((N,), dtype=torch.float32, device=t2d.device)
    call('idf_colsum', _p(t2d), _p(out), _p(ws), R, N, _dt(t2d), _st())
    return out


def gn_coef_fwd_raw(x, gamma, beta, film_t, film_a):
    B, C, H, W = x.shape
    dev = x.device
    mean = torch.empty((B, 32), dtype=torch.float32, device=dev)
    rstd = torch.empty((B, 32), dtype=torch.float32, device=dev)
    sc = torch.empty((B, C), dtype=torch.float32, device=dev)
    sh = torch.empty((B, C), dtype=torch.float32, device=dev)
    ws = torch.empty((_lib.load().idf_gn_workspace_floats(B, H * W, C),), dtype=torch.float32, device=dev)
    call('idf_gn_coef_fwd', _p(x), _p(gamma), _p(beta), _p(film_t), _p(film_a), _ld(film_t), _ld(film_a), GN_EPS,
         _p(mean), _p(rstd),
         _p(sc), _p(sh), _p(ws), B, H * W, C, _dt(x), _st())
    return mean, rstd, sc, sh


def gn_coef_from_stats_raw(st, C, HW, gamma, beta, film_t, film_a, st2=None):
    """(mean, rstd, sc, sh) of a GroupNorm stage from the statistics partials st [B][T][C][2] its input carries (st2: the
    partials of the second source of a skip pair; st then covers the first st.shape[2] of the C channels)."""
    B, dev = st.shape[0], st.device
    mean = torch.empty((B, 32), dtype=torch.float32, device=dev)
    rstd = torch.empty((B, 32), dtype=torch.float32, device=dev)
    sc = torch.empty((B, C), dtype=torch.float32, device=dev)
    sh = torch.empty((B, C), dtype=torch.float32, device=dev)
    ws = torch.empty((B, 2 * C), dtype=torch.float32, device=dev)
    call('idf_gn_coef_from_stats', _p(st), st.shape[1], _p(st2), st2.shape[1] if st2 is not None else 0,
         st.shape[2] if st2 is not None else C, _p(gamma), _p(beta), _p(film_t), _p(film_a), _ld(film_t), _ld(film_a), GN_EPS,
         _p(mean), _p(rstd), _p(sc), _p(sh), _p(ws), B, HW, C, _st())
    return mean, rstd, sc, sh


def gn_apply2_raw(x, x2, sc, sh, seed, salt, p_drop, act):
    """a = act((x | x2) * sc + sh), dense over all channels: the streaming pass over a skip pair read in place."""
    B, C1, H, W = x.shape
    C = C1 + x2.shape[1]
    a = empty_nhwc(B, C, H, W, x.dtype, x.device)
    call('idf_gn_apply2', _p(x), _p(x2), C1, _p(a), _p(sc), _p(sh), _p(seed), salt, float(p_drop), act, B, H * W, C, _dt(x), _st())
    return a


_GN_STREAM_MINPIX = 1 << 18     # B * H * W from which the unfused GroupNorm streams


def _gn_acc(acc):
    """acc = (gamma slot, beta slot) of the gradient arena -> their views when both are free."""
    if _WGRAD_DET:         # deterministic mode: no atomics into the arena -- per-image rows (the callers' dgb path, _gn_out)
        return None
    if acc is None or acc[0] is None or acc[1] is None or not (acc[0].available() and acc[1].available()):
        return None
    return acc[0].take(), acc[1].take()


class GnRowsBatch:
    """Deterministic mode: the GroupNorm affine gradients of a backward pass.  Every GroupNorm-backward launch writes per-image
    rows dgb [B][2][C] into a slice of ONE workspace and returns its arena slot views (which AccumulateGrad merely adopts); ONE
    launch at the end of the pass (`idf_gn_param_reduce_batched`) adds every stage's rows in image order into its slots -- in
    place of a column-sum launch per stage (124 launches / 0.6 ms of a CelebA step).  A backward pass visits its stages in the same
    order every step, so the slices and with them the launch's table repeat: the table is uploaded when a new sequence shows up
    (eagerly: the warm-up steps of a capture leave the one the captured pass finds)."""
    ws = None          # fp32 workspace, handed out front to back during a pass
    cursor = 0
    want = 0           # floats the last pass asked for (the workspace grows between passes)
    pending = []       # (rows address, gamma slot address, beta slot address, B, C)
    keep = []          # tensors the pending launch reads (overflow rows)
    tables = {}        # sequence -> device table
    _graph_bufs = []   # workspaces / tables whose addresses a captured graph holds: never freed, never evicted (as WgradBatch's)
    _cb_queued = False
    _task = -1

    @classmethod
    def _pin(cls, t):
        if t is not None and not any(t is b for b in cls._graph_bufs):
            cls._graph_bufs.append(t)

    @classmethod
    def rows(cls, views, B, C, dev):
        task = torch._C._current_graph_task_id()
        if cls._cb_queued and task != cls._task:
            cls._drop()                                     # the pass that queued the flush never ended (a node raised)
        n = B * 2 * C
        if cls.ws is not None and cls.ws.device == dev and cls.cursor + n <= cls.ws.numel():
            dgb = cls.ws[cls.cursor:cls.cursor + n].view(B, 2 * C)
            if torch.cuda.is_current_stream_capturing():
                cls._pin(cls.ws)          # the graph's launches write these rows at every replay: a later, larger pass must not free them
        else:
            dgb = torch.empty((B, 2 * C), dtype=torch.float32, device=dev)     # first pass / overflow: a tensor of its own
            cls.keep.append(dgb)
        cls.cursor += n
        # (addresses, not the views: AccumulateGrad adopts a gradient only while nobody else holds the tensor -- see LazyGrad)
        cls.pending.append((dgb.data_ptr(), views[0].data_ptr(), views[1].data_ptr(), B, C))
        if not cls._cb_queued:
            try:
                torch.autograd.Variable._execution_engine.queue_callback(cls.flush)
                cls._cb_queued, cls._task = True, task
            except RuntimeError:                            # not inside a backward pass: nothing to defer to (flushed by the caller)
                pass
        return dgb

    @classmethod
    def _drop(cls):
        cls.pending, cls.keep, cls.cursor, cls._cb_queued = [], [], 0, False

    @classmethod
    def flush(cls):
        items, cls.pending = cls.pending, []
        cls._cb_queued = False
        cls.want, cls.cursor = max(cls.want, cls.cursor), 0
        if not items:
            cls.keep = []
            return
        import numpy as np
        key = tuple(items)
        tab = cls.tables.get(key)
        if tab is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError('GnRowsBatch: run one eager step before graph capture (the reduce launch\'s table)')
            nb = int(_lib.load().idf_gn_rows_desc_bytes())
            assert nb == 32
            host = np.zeros((len(items),), dtype=np.dtype([('rows', '<i8'), ('dg', '<i8'), ('db', '<i8'), ('B', '<i4'), ('C', '<i4')]))
            for i, it in enumerate(items):
                host[i] = it
            # a small LRU over the tables no captured graph reads (pinned ones stay and do not count: a process that has captured
            # eight steps must still keep BOTH tables of a data-parallel step's two backward passes between its eager warm-up steps)
            loose = [k for k, v in cls.tables.items() if not any(v is b for b in cls._graph_bufs)]
            if len(loose) >= 8:
                cls.tables.pop(loose[0])
            tab = torch.from_numpy(host.view(np.uint8).copy()).to(torch.device('cuda', torch.cuda.current_device()))
            cls.tables[key] = tab
        if torch.cuda.is_current_stream_capturing():
            cls._pin(tab)
        call('idf_gn_param_reduce_batched', _p(tab), len(items), max(it[4] for it in items), _st())
        cls.keep = []
        if not torch.cuda.is_current_stream_capturing() and (cls.ws is None or cls.ws.numel() < cls.want):
            # (between passes: nothing of the finished pass's rows is read after the launch above -- stream order)
            cls.ws = torch.empty((2 * cls.want,), dtype=torch.float32, device=tab.device)


def _gn_out(acc, B, C, dev):
    """Where a GroupNorm backward leaves dgamma / dbeta -> (acc, dgb, views): acc = the arena slot views the kernel accumulates into
    (atomics: the default), or dgb = per-image rows [B][2][C] -- with views = the slot views a deferred launch fills
    (deterministic mode, GnRowsBatch) or None (the caller sums the rows itself: no free slots)."""
    slots = acc
    acc = _gn_acc(acc)
    if acc is not None:
        return acc, None, None
    if _WGRAD_DET and slots is not None and slots[0] is not None and slots[1] is not None and \
            slots[0].available() and slots[1].available():
        views = (slots[0].take(), slots[1].take())
        return None, GnRowsBatch.rows(views, B, C, dev), views
    return None, torch.empty((B, 2 * C), dtype=torch.float32, device=dev), None


def _gn_done():
    """Behind a launch that wrote rows for GnRowsBatch: outside a backward pass there is no end-of-pass callback -- reduce now."""
    if GnRowsBatch.pending and not GnRowsBatch._cb_queued:
        GnRowsBatch.flush()


def gn_coef_bwd_raw(dA, x, dres, gamma, beta, film_t, film_a, mean, rstd, sc, sh, seed, salt, p_drop, act, acc=None):
    B, C, H, W = x.shape
    dev = x.device
    dx = torch.empty_like(x, memory_format=CL)
    dft = torch.empty(film_t.shape, dtype=torch.float32, device=dev) if film_t is not None else None
    dfa = torch.empty(film_a.shape, dtype=torch.float32, device=dev) if film_a is not None else None
    acc, dgb, later = _gn_out(acc, B, C, dev)
    k1 = torch.empty((B, 32), dtype=torch.float32, device=dev)
    k0 = torch.empty((B, 32), dtype=torch.float32, device=dev)
    ws = torch.empty((_lib.load().idf_gn_workspace_floats(B, H * W, C),), dtype=torch.float32, device=dev)
    call('idf_gn_coef_bwd', _p(dA), _p(x), _p(dres), _p(dx), _p(gamma), _p(beta), _p(film_t), _p(film_a),
         _ld(film_t), _ld(film_a), _p(mean), _p(rstd), _p(sc), _p(sh), _p(dft), _p(dfa), _p(dgb),
         _p(acc[0]) if acc else None, _p(acc[1]) if acc else None, _p(k1), _p(k0), _p(ws), _p(seed),
         salt, float(p_drop), act, B, H * W, C, _dt(x), _st())
    if acc or later:
        g_ = acc or later
        _gn_done()
        return dx, g_[0], g_[1], dft, dfa
    dgam = colsum_raw(dgb)
    return dx, dgam[:C], dgam[C:], dft, dfa


def bgemm_raw(A, offA, B_, offB, Cout, offC, bias, batch, sA, sB, sC, lda, ldb, ldc, M, N, K, ta, tb,
              alpha=1.0, out_f32=False, splitk=1, dtype=None, res=None):
    call('idf_bgemm', _p(A, offA), _p(B_, offB), _p(Cout, offC), _p(bias), _p(res), batch, sA, sB, sC, lda, ldb, ldc,
         M, N, K, ta, tb, float(alpha), int(out_f32), splitk, _dt(A) if dtype is None else dtype, _st())


# ------------------------------------------------------------- fused conv op
@functools.lru_cache(maxsize=None)
def _gn_fused_ok(B, HW, C, dt, C1=0):
    return bool(_lib.load().idf_gn_fused_ok(B, HW, C, C1, dt))


def gn_small_ok(x, x2=None):
    """The one-launch GroupNorm kernels cover this tensor (idf_gn_fused_ok); x2: second source of a
    never-materialised channel concatenation x | x2."""
    B, C, H, W = x.shape
    if x2 is not None:
        return _gn_fused_ok(B, H * W, C + x2.shape[1], _dt(x), C)
    return _gn_fused_ok(B, H * W, C, _dt(x))


def gn_fused_fwd_raw(x, gamma, beta, film_t, film_a, seed, salt, p_drop, act, x2=None):
    """statistics + FiLM fold + apply in one launch -> (a, mean, rstd, sc, sh).  x2: the input is the
    channel concatenation x | x2 (read in place, never materialised); `a` is dense over all channels."""
    B, C1, H, W = x.shape
    C = C1 + (x2.shape[1] if x2 is not None else 0)
    dev = x.device
    mean = torch.empty((B, 32), dtype=torch.float32, device=dev)
    rstd = torch.empty((B, 32), dtype=torch.float32, device=dev)
    sc = torch.empty((B, C), dtype=torch.float32, device=dev)
    sh = torch.empty((B, C), dtype=torch.float32, device=dev)
    a = empty_nhwc(B, C, H, W, x.dtype, dev)
    call('idf_gn_fused_fwd', _p(x), _p(x2), C1, _p(a), _p(gamma), _p(beta), _p(film_t), _p(film_a), _ld(film_t), _ld(film_a),
         GN_EPS, _p(mean), _p(rstd), _p(sc), _p(sh), _p(seed), salt, float(p_drop), act, B, H * W, C, _dt(x), _st())
    return a, mean, rstd, sc, sh


def gn_fused_bwd_raw(dA, x, gamma, beta, film_t, film_a, mean, rstd, sc, sh, seed, salt, p_drop, act, acc=None,
                     dres=None, x2=None, dres2=None):
    """x2: two-source input (see gn_fused_fwd_raw); then the first result is the pair (dx, dx2)."""
    B, C1, H, W = x.shape
    C = C1 + (x2.shape[1] if x2 is not None else 0)
    dev = x.device
    dx = torch.empty_like(x, memory_format=CL)
    dx2 = torch.empty_like(x2, memory_format=CL) if x2 is not None else None
    dft = torch.empty(film_t.shape, dtype=torch.float32, device=dev) if film_t is not None else None
    dfa = torch.empty(film_a.shape, dtype=torch.float32, device=dev) if film_a is not None else None
    acc, dgb, later = _gn_out(acc, B, C, dev)
    call('idf_gn_fused_bwd', _p(dA), _p(x), _p(x2), C1, _p(dres), _p(dres2), _p(dx), _p(dx2), _p(gamma), _p(beta), _p(film_t), _p(film_a), _ld(film_t),
         _ld(film_a), _p(mean), _p(rstd), _p(sc), _p(sh), _p(dft), _p(dfa), _p(dgb),
         _p(acc[0]) if acc else None, _p(acc[1]) if acc else None, _p(seed), salt, float(p_drop),
         act, B, H * W, C, _dt(x), _st())
    if x2 is not None:
        dx = (dx, dx2)
    if acc or later:
        g_ = acc or later
        _gn_done()
        return dx, g_[0], g_[1], dft, dfa
    dgam = colsum_raw(dgb)
    return dx, dgam[:C], dgam[C:], dft, dfa


_DGRAD_GN = knobs.flag('IDF_DGRAD_GN')


@functools.lru_cache(maxsize=None)
def _dgrad_gn_shape_ok(B, H, W, Cin, Cout, taps):
    return int(_lib.load().idf_conv_dgrad_gn_ok(B, H, W, Cin, Cout, taps))


def conv_dgrad_gn_ok(dy, x, mode, taps, advice=True):
    """The data-gradient conv can carry the GroupNorm backward as its epilogue (idf_conv_dgrad_gn_bf16: small maps, bf16)
    -- and, with `advice`, the library's measured policy prefers that to the two launches."""
    if not (_DGRAD_GN and mode == S1 and x.is_cuda and x.dtype == torch.bfloat16 and dy.dtype == torch.bfloat16):
        return False
    B, C, H, W = x.shape
    ok = _dgrad_gn_shape_ok(B, H, W, dy.shape[1], C, taps)
    return ok == 1 or (ok == 2 and not advice)


def conv_dgrad_gn_raw(dy, w_dgrad, x, gamma, beta, film_t, film_a, mean, rstd, sc, sh, seed, salt, p_drop, act, taps,
                      acc=None, dres=None, dres2=None, shadows=None):
    """dx and the GroupNorm's parameter / FiLM gradients from dy in ONE launch: the stride-1 data-gradient conv with the
    GroupNorm backward as its epilogue.  Returns what gn_fused_bwd_raw returns."""
    B, C, H, W = x.shape
    dev = x.device
    dx = torch.empty_like(x, memory_format=CL)
    dft = torch.empty(film_t.shape, dtype=torch.float32, device=dev) if film_t is not None else None
    dfa = torch.empty(film_a.shape, dtype=torch.float32, device=dev) if film_a is not None else None
    acc, dgb, later = _gn_out(acc, B, C, dev)
    # (whole 16x16 images in this form -- 4 waves x 256 pixels -- measured SLOWER than the 512-thread register-staged kernel:
    # 25.8 vs 19.7 us at B = 32, profiles/r04_conv_wr.txt: 8x8 only)
    wfrag = _wr_frag(shadows, 3, B, H, W, dy.shape[1], C, 1) if (taps == 9 and act == 2 and C in (128, 256) and W == 8) else None
    if wfrag is not None:
        call('idf_conv_wr_dgrad_gn_bf16', _p(dy), _p(wfrag), _p(x), _p(dres), _p(dres2), _p(dx), _p(gamma), _p(beta), _p(film_t),
             _p(film_a), _ld(film_t), _ld(film_a), _p(mean), _p(rstd), _p(sc), _p(sh), _p(dft), _p(dfa), _p(dgb),
             _p(acc[0]) if acc else None, _p(acc[1]) if acc else None, _p(seed), salt, float(p_drop), B, H, W, dy.shape[1], C,
             _st())
        if acc or later:
            g_ = acc or later
            _gn_done()
            return dx, g_[0], g_[1], dft, dfa
        dgam = colsum_raw(dgb)
        return dx, dgam[:C], dgam[C:], dft, dfa
    call('idf_conv_dgrad_gn_bf16', _p(dy), _p(w_dgrad), _p(x), _p(dres), _p(dres2), _p(dx), _p(gamma), _p(beta), _p(film_t),
         _p(film_a), _ld(film_t), _ld(film_a), _p(mean), _p(rstd), _p(sc), _p(sh), _p(dft), _p(dfa), _p(dgb),
         _p(acc[0]) if acc else None, _p(acc[1]) if acc else None, _p(seed), salt, float(p_drop), act, B, H, W,
         dy.shape[1], C, taps, _st())
    if acc or later:
        g_ = acc or later
        _gn_done()
        return dx, g_[0], g_[1], dft, dfa
    dgam = colsum_raw(dgb)
    return dx, dgam[:C], dgam[C:], dft, dfa


# ------------------------------------------------------------ backward chain at the big maps
_BWD_CHAIN = knobs.flag('IDF_BWD_CHAIN')     # du epilogue + streaming apply instead of the one-launch GroupNorm backward
# ... and the apply pass folded into the NEXT data-gradient conv's prologue (LazyGrad).  Built, tested against fp32 autograd, measured, and
# OFF by default: the prologue (in-block coefficient fold + two tensors staged per vector + the side write of dy) costs the
# conv +25 us at 64->64 @64^2, B = 32, against the 12 us streaming pass it removes -- 10.89 vs 10.43 ms per step on one box
# (profiles/r03_c_ab_chain_lazy.txt, r03_d_step_inventory_chain_lazy.txt)
_SC_FUSE = knobs.flag('IDF_SC_FUSE')       # a block's 1x1 shortcut (and its data gradient) inside its first conv's launches
_SC_FUSE_MAXPIX = 8192   # ... where those launches leave CUs idle: B * H * W up to 32 x 16 x 16
_BWD_LAZY = knobs.flag('IDF_BWD_LAZY')
_WGRAD_DET = knobs.raw('IDF_DETERMINISTIC') != '0'    # (default) batched weight gradients: slab partials + ordered reduce instead of fp32 atomics


def set_deterministic(on):
    """Bit-reproducible training steps (the reference's --deterministic: utils.py:64-71 sets cudnn.deterministic): every
    accumulation that used fp32 atomics takes its ordered form -- weight gradients write per-split slabs that ONE reduce launch adds in
    slab order (idf_wgrad_reduce_batched), GroupNorm parameter gradients go through per-image rows that ONE launch at the end of
    the pass adds in image order (GnRowsBatch) instead of atomics into the arena.  Round 5: the DEFAULT (0.01-0.05 ms per CelebA
    step against the atomic forms, profiles/r05_deterministic.txt); set_deterministic(False) / IDF_DETERMINISTIC=0 select atomics."""
    global _WGRAD_DET
    on = bool(on) and knobs.raw('IDF_DETERMINISTIC') != '0'          # IDF_DETERMINISTIC=0 pins the atomic forms (A/B runs)
    if on != _WGRAD_DET:
        _WGRAD_DET = on
        WgradBatch._bufs.clear()          # launch plans were built for the other mode


@functools.lru_cache(maxsize=None)
def chain_tiles(B, H, W, Cin, Cout, taps):
    """Pixel tiles per image of idf_conv_dgrad_chain_bf16 for this data-gradient conv (Cin = channels of dy, Cout = channels
    of dx); < 0: not covered."""
    return int(_lib.load().idf_conv_dgrad_chain_tiles(B, H, W, Cin, Cout, taps))


class LazyGrad:
    """The gradient w.r.t. a GroupNorm stage's input that exists only as (du, per-tile partial sums): the data-gradient conv
    in FRONT of that GroupNorm forms dx = A*du + K1*x + K0 while staging its tile (idf_conv_dgrad_chain_bf16's dy prologue)
    and stores the GroupNorm's parameter / FiLM gradients where the autograd node that owns those parameters has already
    promised them: dft / dfa (returned as the FiLM pairs' gradients; read by the batched projection's backward, after every
    block) and acc = the ADDRESSES of the gamma / beta arena slots (addresses, not tensors: AccumulateGrad adopts a
    gradient only while nobody else holds it -- a second reference makes it clone the still-empty view)."""
    __slots__ = ('du', 'part', 'x', 'gn_w', 'gn_b', 'film_t', 'film_a', 'mean', 'rstd', 'sc', 'dft', 'dfa', 'acc')

    def __init__(self, **kw):
        for k in self.__slots__:
            setattr(self, k, kw.get(k))

    def materialize(self):
        """dx as a tensor (the streaming apply pass): for a consumer that cannot take the pair."""
        acc = self.acc
        call('idf_gn_bwd_apply', _p(self.du), _p(self.part), self.part.shape[1], _p(self.x), None, 0, None, None,
             _p(self.du), None, _p(self.gn_w), _p(self.gn_b), _p(self.film_t), _p(self.film_a), _ld(self.film_t),
             _ld(self.film_a), _p(self.mean), _p(self.rstd), _p(self.sc), _p(self.dft), _p(self.dfa), None,
             acc[0] if acc else None, acc[1] if acc else None, self.du.shape[0], self.du.shape[2] * self.du.shape[3],
             self.du.shape[1], _st())
        return self.du        # in place: du -> dx (each element is read and written by the same thread)


_LAZY_PENDING = {}     # du.data_ptr() -> LazyGrad on its way to the producing conv's backward


def _lazy_check_consumed():
    """End of a backward pass: a pair nobody consumed means a gradient tensor holding du was taken for dx."""
    if _LAZY_PENDING:
        n = len(_LAZY_PENDING)
        _LAZY_PENDING.clear()
        raise RuntimeError('infodiffusion_amd: %d lazy GroupNorm gradient(s) reached no data-gradient conv (IDF_BWD_LAZY=0 '
                           'disables the hand-off)' % n)


def conv_dgrad_chain_raw(dy, w_dgrad, taps, Cout, lazy=None, want_dy=False, x=None, x2=None, sc=None, sh=None, seed=None,
                         salt=0, p_drop=0.0, act=0, shortcut=None, shadows=None):
    """Stride-1 data-gradient conv of the backward chain -> (out, part, dy_mat).  lazy: `dy` is lazy.du and the real input
    gradient is formed in the prologue (written to dy_mat when want_dy); x (+ x2): du epilogue, out = du and part = its
    per-tile partial sums -- else out = dA.  shortcut = (ds, w_sc_dgrad): the launch also computes the block shortcut's
    data gradient dxs = conv1x1(ds) (3x3, du epilogue, no lazy input) -> (out, part, dxs)."""
    B, Cin, H, W = dy.shape
    out = empty_nhwc(B, Cout, H, W, dy.dtype, dy.device)
    part = None
    C1 = x.shape[1] if (x is not None and x2 is not None) else 0
    if (x is not None and lazy is None and shortcut is None and taps == 9 and dy.dtype == torch.bfloat16 and act
            and (x2 is None or C1 % 64 == 0)):
        rfrag = _rs_frag(shadows, 3, B, H, W, Cin, Cout)
        if rfrag is not None:
            part = torch.empty((B, rs_tiles(B, H, W, Cin, Cout), Cout, 2), dtype=torch.float32, device=dy.device)
            call('idf_conv_rs_dgrad_chain_bf16', _p(dy), _p(rfrag), _p(x), _p(x2), C1, _p(sc), _p(sh), _p(seed), salt, float(p_drop),
                 act, _p(out), _p(part), B, H, W, Cin, Cout, _st())
            return out, part, None
    if x is not None:
        part = torch.empty((B, chain_tiles(B, H, W, Cin, Cout, taps), Cout, 2), dtype=torch.float32, device=dy.device)
    if shortcut is not None:
        ds, w_sc = shortcut
        dxs = empty_nhwc(B, Cout, H, W, dy.dtype, dy.device)
        call('idf_conv_dgrad_chain_sc_bf16', _p(dy), _p(w_dgrad), _p(x), _p(x2), C1, _p(sc), _p(sh), _p(seed), salt,
             float(p_drop), act, _p(out), _p(part), B, H, W, Cin, Cout, _st(), _p(ds), _p(w_sc), _p(dxs), ds.shape[1])
        return out, part, dxs
    dy_mat = torch.empty_like(dy, memory_format=CL) if (lazy is not None and want_dy) else None
    L = lazy
    acc = L.acc if L is not None else None
    call('idf_conv_dgrad_chain_bf16', _p(dy), _p(L.x) if L else None, _p(L.part) if L else None, L.part.shape[1] if L else 0,
         _p(L.mean) if L else None, _p(L.rstd) if L else None, _p(L.sc) if L else None, _p(L.gn_w) if L else None,
         _p(L.gn_b) if L else None, _p(L.film_t) if L else None, _p(L.film_a) if L else None,
         _ld(L.film_t) if L else 0, _ld(L.film_a) if L else 0, _p(L.dft) if L else None, _p(L.dfa) if L else None, None,
         acc[0] if acc else None, acc[1] if acc else None, _p(dy_mat), _p(w_dgrad), _p(x), _p(x2), C1, _p(sc), _p(sh),
         _p(seed), salt, float(p_drop), act, _p(out), _p(part), B, H, W, Cin, Cout, taps, _st())
    return out, part, dy_mat


_RS_SHARED = True      # (A/B: the 128->128 @32x32 forward conv of a training step on the shared-image form; False: halo kernel)
_RS_SYNC = [True]     # the trainer turns the form off around launches that share the chip with a collective (sync_convs)


class sync_convs:
    """with ops.sync_convs(False): the data-gradient convs issued inside do not take the group-synchronised form
    (idf_conv_rs_dgrad_gn_bf16): its workgroups wait for each other inside the launch and need the whole grid on the chip at
    once -- true when nothing else runs, not beside an RCCL kernel that holds CUs for the length of an all-reduce."""

    def __init__(self, on):
        self.on = bool(on)

    def __enter__(self):
        self.prev = _RS_SYNC[0]
        _RS_SYNC[0] = self.on

    def __exit__(self, *exc):
        _RS_SYNC[0] = self.prev
        return False


_RS_SYNC_STATE = {}     # device index -> the synchronised form's counters + error word (uint32, zero-initialised, persistent)
_RS_SYNC_DEAD = [False]  # set for the rest of the process once a time-out was seen (retire_sync_convs): the form is not taken again


def switch_state():
    """The kernel-selection switches a captured graph bakes in (tests / A-B harnesses toggle them at run time): part of the key of
    every graph that is kept across calls (sampling._graphed)."""
    g = globals()
    return tuple(g[k] if not isinstance(g[k], list) else tuple(g[k]) for k in (
        '_FEWC', '_GN_FUSE', '_WR', '_WR_MAXB', '_RS', '_RS_FWD_ALL', '_RS_SHARED', '_SC_FUSE', '_RB_SMALL', '_RB_SMALL_MAXB',
        '_RB_WFRAG', '_UPCONV', '_ATTN_FOLD', '_ATTN_BLOCK', '_ATTN_BLOCK_MINB', '_TEMB_FUSED'))


def retire_sync_convs():
    """After a time-out of the group-synchronised conv: never take that form again in this process (the chain + apply pair
    replaces it -- workgroups that do not wait for each other) and clear the counters the failed launches left mid-count."""
    _RS_SYNC_DEAD[0] = True
    for st in _RS_SYNC_STATE.values():
        st.zero_()


def sync_convs_retired():
    return _RS_SYNC_DEAD[0]


def _rs_sync_state(dev):
    st = _RS_SYNC_STATE.get(dev.index)
    if st is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError('conv_dgrad_gn_sync_raw: run one eager step before graph capture (the counters)')
        st = _RS_SYNC_STATE[dev.index] = torch.zeros((int(_lib.load().idf_conv_rs_sync_words()),), dtype=torch.int32, device=dev)
    return st


def rs_sync_timeouts(reset=True):
    """Workgroups of the synchronised form that ever gave up waiting for their group, over all devices (0 in a healthy process;
    reads device memory: synchronises).  reset: zero the state of a device that reported any."""
    n = 0
    for st in _RS_SYNC_STATE.values():
        k = int(st[-16].item())
        if k and reset:
            st.zero_()
        n += k
    return n


def conv_dgrad_gn_sync_raw(dy, Cout, x, gamma, beta, film_t, film_a, mean, rstd, sc, sh, seed, salt, p_drop, act, acc=None,
                           dres=None, dres2=None, x2=None, shadows=None):
    """The backward of conv3x3(dropout(act(FiLM(GroupNorm(x | x2))))) w.r.t. its input in one launch on the 64x64 / 32x32 maps
    (conv_dgrad_chain_raw + gn_bwd_apply_raw, du never written) -> what gn_bwd_apply_raw returns, or None when the form does
    not cover the call."""
    B, Cin, H, W = dy.shape
    if not (_RS_SYNC[0] and not _RS_SYNC_DEAD[0] and act and dy.dtype == torch.bfloat16 and shadows is not None):
        return None
    C1 = x.shape[1] if x2 is not None else 0
    if x2 is not None and C1 % 64:
        return None
    T = int(_lib.load().idf_conv_rs_dgrad_gn_tiles(B, H, W, Cin, Cout))
    if not T:
        return None
    rfrag = _rs_frag(shadows, 3, B, H, W, Cin, Cout)
    if rfrag is None:
        return None
    dev = x.device
    part = torch.empty((B, T, Cout, 2), dtype=torch.float32, device=dev)
    dx = torch.empty_like(x, memory_format=CL)
    dx2 = torch.empty_like(x2, memory_format=CL) if x2 is not None else None
    dft = torch.empty(film_t.shape, dtype=torch.float32, device=dev) if film_t is not None else None
    dfa = torch.empty(film_a.shape, dtype=torch.float32, device=dev) if film_a is not None else None
    acc, dgb, later = _gn_out(acc, B, Cout, dev)
    call('idf_conv_rs_dgrad_gn_bf16', _p(dy), _p(rfrag), _p(x), _p(x2), C1, _p(sc), _p(sh), _p(seed), salt, float(p_drop), act,
         _p(dres), _p(dres2), _p(dx), _p(dx2), _p(part), _p(gamma), _p(beta), _p(film_t), _p(film_a), _ld(film_t), _ld(film_a),
         _p(mean), _p(rstd), _p(dft), _p(dfa), _p(dgb), _p(acc[0]) if acc else None, _p(acc[1]) if acc else None,
         _p(_rs_sync_state(dev)), B, H, W, Cin, Cout, _st())
    if x2 is not None:
        dx = (dx, dx2)
    if acc or later:
        g_ = acc or later
        _gn_done()
        return dx, g_[0], g_[1], dft, dfa
    dgam = colsum_raw(dgb)
    return dx, dgam[:Cout], dgam[Cout:], dft, dfa


def gn_bwd_apply_raw(du, part, x, gamma, beta, film_t, film_a, mean, rstd, sc, acc=None, dres=None, dres2=None, x2=None):
    """dx (+ parameter / FiLM gradients) from the (du, part) pair a du-epilogue conv left behind: what gn_fused_bwd_raw
    returns, by a streaming pass (no reduction left in it)."""
    B, C1, H, W = x.shape
    C = C1 + (x2.shape[1] if x2 is not None else 0)
    dev = x.device
    dx = torch.empty_like(x, memory_format=CL)
    dx2 = torch.empty_like(x2, memory_format=CL) if x2 is not None else None
    dft = torch.empty(film_t.shape, dtype=torch.float32, device=dev) if film_t is not None else None
    dfa = torch.empty(film_a.shape, dtype=torch.float32, device=dev) if film_a is not None else None
    acc, dgb, later = _gn_out(acc, B, C, dev)
    call('idf_gn_bwd_apply', _p(du), _p(part), part.shape[1], _p(x), _p(x2), C1 if x2 is not None else 0, _p(dres), _p(dres2),
         _p(dx), _p(dx2), _p(gamma), _p(beta), _p(film_t), _p(film_a), _ld(film_t), _ld(film_a), _p(mean), _p(rstd), _p(sc),
         _p(dft), _p(dfa), _p(dgb), _p(acc[0]) if acc else None, _p(acc[1]) if acc else None, B, H * W, C, _st())
    if x2 is not None:
        dx = (dx, dx2)
    if acc or later:
        g_ = acc or later
        _gn_done()
        return dx, g_[0], g_[1], dft, dfa
    dgam = colsum_raw(dgb)
    return dx, dgam[:C], dgam[C:], dft, dfa


def gn_apply_raw(x, sc, sh, seed, salt, p_drop, act):
    """a = act(x*sc + sh) (+ dropout): one read + one write."""
    B, C, H, W = x.shape
    a = torch.empty_like(x, memory_format=CL)
    call('idf_gn_apply', _p(x), _p(a), _p(sc), _p(sh), _p(seed), salt, float(p_drop), act, B, H * W, C, _dt(x), _st())
    return a


def _fast_wgrad_ok(Cin, Cout, Ho, Wo, dtype, mode, taps):
    if dtype != torch.bfloat16 or mode not in (S1, S2, UP2) or (mode != S1 and taps != 9):
        return False
    if Cin % 8 or Cout % 8 or Wo < 4 or (Wo & (Wo - 1)):
        return False
    R = min(Ho, (64 if mode == S2 else 128) // Wo)
    return R >= 1 and Ho % R == 0 and (R * Wo) % 32 == 0 and R * ((2 if mode == S2 else 1) * Wo + 2) <= 160


def _pad_channels(t, mult=8):
    """Zero-pad the channel dim of an NHWC-dense tensor to a multiple of `mult`."""
    C = t.shape[1]
    Cp = -(-C // mult) * mult
    if Cp == C:
        return t
    # one launch (cat with a cached block of zeros) instead of a fill and a strided copy
    key = (t.shape[0], Cp - C, t.shape[2], t.shape[3], t.dtype, t.device)
    z = _PAD_ZEROS.get(key)
    if z is None:
        z = _PAD_ZEROS[key] = torch.zeros(key[:4], dtype=t.dtype, device=t.device).contiguous(memory_format=CL)
    out = torch.cat([t, z], dim=1)
    return out if out.is_contiguous(memory_format=CL) else out.contiguous(memory_format=CL)


_PAD_ZEROS = {}


def _slot_views(w_slot, b_slot, want_bias, shape):
    """(dW view, db view) from the gradient arena when both are free this step, else None."""
    if w_slot is None or not w_slot.available() or tuple(w_slot.view.shape) != tuple(shape):
        return None
    if not w_slot.view.permute(0, 2, 3, 1).is_contiguous():      # kernel layout [O][kh][kw][I]
        return None
    if want_bias and (b_slot is None or not b_slot.available()):
        return None
    return w_slot.take(), (b_slot.take() if want_bias else None)


@functools.lru_cache(maxsize=None)
def _upsub_ok(H, W):
    return bool(_lib.load().idf_wgrad_upsub_ok(H, W))


@functools.lru_cache(maxsize=None)
def _kr3_ok(H, W):
    return bool(_lib.load().idf_wgrad_kr3_ok(H, W))


def _ring_ok(B, H, W, Cin, Cout):
    return bool(_lib.load().idf_wgrad_ring_ok(B, H, W, Cin, Cout))


class WgradBatch:
    """Deferred weight gradients.  Only the optimizer reads them, so backward just queues each
    convolution's (a, dy, arena slots) and ONE launch per (taps, mode) class at the end of the
    backward pass computes them all (`idf_conv_wgrad_bf16_batched`): no per-conv launch, small
    problems share the chip, one problem's atomic tail overlaps its neighbours' loads."""
    enabled = knobs.flag('IDF_WGRAD_BATCH')
    pending = []       # (a, dy, dW address, db address, B, H, W, Cin, Cout, taps, mode, a2, C1, Cin_w, Cout_w)
    _bufs = {}         # key of a flush -> [pinned host table, device table, key, launch plan, copy event]   (eager: LRU, insertion-ordered)
    _LRU = 6
    _graph_bufs = []   # tables referenced by captured graphs: never touched again

    # (Flushing part of the queue mid-backward on a side stream -- uncapped in round 3, as a capped persistent grid in round 4 --
    # never gained on one GPU: profiles/r03_v_ab_wgrad_side.txt, r04_wgrad_side_capped.txt.  Removed in round 5.)
    _cb_queued = False
    _task = -1         # autograd graph task that queued the flush

    @classmethod
    def add(cls, item):
        task = torch._C._current_graph_task_id()
        if cls._cb_queued and task != cls._task:
            # the pass (autograd graph task) that queued the flush never ended -- the engine skips its final callbacks when a
            # node raises: its items belong to no launch, and this pass queues a flush of its own
            cls._drop()
        cls.pending.append(item)
        if not cls._cb_queued:
            try:
                torch.autograd.Variable._execution_engine.queue_callback(cls.flush)
                cls._cb_queued = True
                cls._task = task
            except RuntimeError:          # not inside a backward pass: nothing to defer to
                cls.flush()
                return

    @classmethod
    def reset(cls):
        """Forget everything queued by a backward pass that did not end (the autograd engine skips its final callbacks
        when a node raises, so `flush` never ran): stale (a, dy, slot address) items must not be launched by a later
        step, and `_cb_queued` must not keep later passes from queueing their own end-of-backward flush (`add` also notices a
        new graph task by itself).  Safe to call whenever no backward pass is running (trainer: before every forward, and
        after a failed capture)."""
        cls._drop()
        GnRowsBatch._drop()
        _LAZY_PENDING.clear()
        del _FOLD_BWD_ROWS[:]

    @classmethod
    def _drop(cls):
        cls.pending = []
        cls._cb_queued = False

    @classmethod
    def flush(cls):
        """End of a backward pass (or an explicit barrier): launch what is queued on the current stream."""
        cls._cb_queued = False
        cls._flush_pending()

    @classmethod
    def _flush_pending(cls):
        items, cls.pending = cls.pending, []
        if not items:
            return
        groups = {}
        lib = _lib.load()
        for it in items:
            mode = it[10]
            if it[9] == 9 and mode == S1 and _ring_ok(it[4], it[5], it[6], it[7], it[8]):
                mode |= 64          # IDF_WGRAD_RING: the row-ring form (round 6)
            elif it[9] == 9 and mode == S1 and not _kr3_ok(it[5], it[6]):
                mode |= 16          # IDF_WGRAD_ROWSPLIT: a map the shared-tile kernel does not take (W = 128) is a class of its own
            if it[9] == 9 and mode == UP2 and _upsub_ok(it[5], it[6]):
                mode |= 32          # IDF_WGRAD_UPSUB: the UpSample class in its sub-pixel form
            groups.setdefault((it[9], mode), []).append(it)
        nb = lib.idf_wgrad_desc_bytes()
        capturing = torch.cuda.is_current_stream_capturing()
        # longest blocks first: a launch's workgroups start in table order, and the queue arrives in backward order -- the 64x64 / 32x32
        # problems of the networks' first levels (64 pixel tiles per workgroup) LAST, behind the 16x16 / 8x8 ones (16 / 8 tiles): their
        # workgroups then start in the launch's last round and run alone (simulated on the CelebA step's 114 problems: 321 -> 295 tile
        # times against a balanced 289; measured: profiles/r06_wgrad.txt).  Stable sort: the order is a function of the shapes only.
        for grp in groups.values():
            grp.sort(key=lambda it: -(it[5] * it[6]))
        classes = list(groups.items())
        total = len(items)
        # ONE table (and one host-to-device copy) for all the (taps, mode) classes of this flush; each class launches on
        # its slice.  Tables are kept per KEY (every operand address of the flush): a step's flushes -- under data
        # parallelism the early backbone flush and the final one carry the same classes -- each find their own table
        # again at the next step and neither synchronises nor rebuilds; a small LRU bounds the set.
        key = tuple((it[0].data_ptr(), it[1].data_ptr()) + it[2:11] + (_p(it[11]), it[12], it[13], it[14])
                    for _, grp in classes for it in grp)
        buf = cls._bufs.pop(key, None)
        ws_off, ws_tot = [], 0
        if _WGRAD_DET and (buf is None or buf[2] != key or capturing):
            # deterministic accumulation: every pixel split writes its partial into a slab of its own, one reduce launch adds the
            # slabs in a fixed order.  Sizing pass: slab floats per entry (the plan does not depend on the workspace address).
            scratch = ctypes.create_string_buffer(nb)
            nblk, nlds, nws, nred = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_long(0), ctypes.c_int(0)
            for (taps, mode), grp in classes:
                for (a, dy, dW, db, B, H, W, Cin, Cout, _, _, a2, C1, Cin_w, Cout_w) in grp:
                    _lib.check(lib.idf_wgrad_desc_fill(ctypes.addressof(scratch), 0, _p(a), _p(a2), C1, _p(dy), dW, db, B, H, W, Cin, Cout,
                                                       Cin_w, Cout_w, taps, mode, 0, 0, ctypes.byref(nblk), ctypes.byref(nlds), None,
                                                       0, ctypes.byref(nws), ctypes.byref(nred)), 'idf_wgrad_desc_fill')
                    ws_off.append(ws_tot)
                    ws_tot += nws.value
        if buf is None:
            reuse = None
            if capturing:
                # no pinned allocation inside a capture: take over a table an eager step made (the capture's own
                # warm-up pass left one of the right size; the capture follows a device-wide synchronisation)
                fit = [k for k, v in cls._bufs.items() if v[0].numel() >= total * nb and
                       (not _WGRAD_DET or (len(v) > 5 and v[5] is not None and v[5].numel() >= ws_tot))]
                if fit:
                    reuse = cls._bufs.pop(fit[-1])
            elif len(cls._bufs) >= cls._LRU:
                reuse = cls._bufs.pop(next(iter(cls._bufs)))          # least recently used
            if reuse is not None and reuse[0].numel() >= total * nb:
                buf = reuse
                buf[2] = None
            else:
                dev = items[0][0].device
                buf = [torch.empty((max(total, 16) * nb,), dtype=torch.uint8).pin_memory(),
                       torch.empty((max(total, 16) * nb,), dtype=torch.uint8, device=dev), None, [], None]
        if capturing:
            # the captured copy node reads this pinned table at every replay: retire the pair from
            # eager use (the warm-up step allocated it; later eager steps get a fresh one)
            cls._graph_bufs.append(buf)
        else:
            cls._bufs[key] = buf                                       # most recently used
        if buf[2] != key or capturing:
            if not capturing and buf[4] is not None:
                buf[4].synchronize()        # the previous copy out of this pinned table has been issued AND done
                # (under capture nothing may be synchronised -- and nothing needs to be: the capture follows a
                # device-wide synchronisation, and the pair is retired to the graph)
            plan, off = [], 0
            nblk, nlds = ctypes.c_int(0), ctypes.c_int(0)
            nws, nred = ctypes.c_long(0), ctypes.c_int(0)
            if _WGRAD_DET and (len(buf) < 6 or buf[5] is None or buf[5].numel() < ws_tot):
                if capturing:
                    raise RuntimeError('WgradBatch: run one eager step before capture (weight-gradient workspace)')
                while len(buf) < 6:
                    buf.append(None)
                buf[5] = torch.empty((ws_tot,), dtype=torch.float32, device=items[0][0].device)
            k, red = 0, 0
            for (taps, mode), grp in classes:
                host = buf[0].data_ptr() + off * nb
                blk, lds = 0, 0
                for i, (a, dy, dW, db, B, H, W, Cin, Cout, _, _, a2, C1, Cin_w, Cout_w) in enumerate(grp):
                    ws = buf[5].data_ptr() + 4 * ws_off[k] if _WGRAD_DET else None
                    _lib.check(lib.idf_wgrad_desc_fill(host, i, _p(a), _p(a2), C1, _p(dy), dW, db, B, H, W, Cin, Cout, Cin_w,
                                                       Cout_w, taps, mode, 0, blk, ctypes.byref(nblk), ctypes.byref(nlds), ws, red,
                                                       ctypes.byref(nws), ctypes.byref(nred)),
                               'idf_wgrad_desc_fill')
                    blk += nblk.value
                    lds = max(lds, nlds.value)
                    red += nred.value
                    k += 1
                plan.append((off, len(grp), blk, lds, taps, mode))
                off += len(grp)
            plan.append(('reduce', total, red if _WGRAD_DET else 0))
            buf[1][:total * nb].copy_(buf[0][:total * nb], non_blocking=True)
            buf[2], buf[3] = key, plan
            if not capturing:               # eager runs can be a step ahead of the GPU: guard the table's next rewrite
                buf[4] = torch.cuda.Event()
                buf[4].record()
        base = buf[1].data_ptr()
        plan = buf[3]
        for ent in plan:
            if ent[0] == 'reduce':
                if ent[2]:
                    call('idf_wgrad_reduce_batched', base, ent[1], ent[2], _st())
                continue
            off, n, blk, lds, taps, mode = ent
            call('idf_conv_wgrad_bf16_batched', base + off * nb, n, blk, lds, taps, mode, _st())


def conv_wgrad_bias_raw(a, dy, mode, taps, want_bias, w_slot=None, b_slot=None, defer=False):
    """dW (fp32, logical [O,I,kh,kw], memory [O][taps][I]) and db for a conv whose
    (already activated) input is `a`.  Channel counts that are not multiples of 8
    (image input, epsilon output) are zero-padded so the MFMA kernel covers them.
    With gradient-arena slots the kernel accumulates straight into them (no memset)."""
    B, Cin, Hs, Ws = a.shape
    _, Cout, Ho, Wo = dy.shape
    k = 3 if taps == 9 else 1
    if a.dtype == torch.bfloat16:
        ap, dyp = _pad_channels(a), _pad_channels(dy)
        Cip, Cop = ap.shape[1], dyp.shape[1]
        if _fast_wgrad_ok(Cip, Cop, Ho, Wo, a.dtype, mode, taps):
            # the kernel writes the parameter's own [Cout][taps][Cin] layout whatever the operands were padded to
            views = _slot_views(w_slot, b_slot, want_bias, (Cout, Cin, k, k))
            if views is not None:
                dW, db = views
                if defer and WgradBatch.enabled:
                    # slot ADDRESSES, not the tensors: AccumulateGrad adopts a gradient only if nobody else holds it
                    WgradBatch.add((ap, dyp, _p(dW), _p(db), B, Ho, Wo, Cip, Cop, taps, mode, None, 0, Cin, Cout))
                else:
                    call('idf_conv_wgrad_bf16', _p(ap), _p(dyp), _p(dW), _p(db), B, Ho, Wo, Cip, Cop, Cin, Cout, taps, mode, 1,
                         _st())
                return dW, db
            nW = Cout * k * k * Cin
            buf = torch.empty((nW + Cout,), dtype=torch.float32, device=a.device)   # dW | db: one memset
            dW = buf[:nW].view(Cout, k, k, Cin)
            db = buf[nW:] if want_bias else None
            call('idf_conv_wgrad_bf16', _p(ap), _p(dyp), _p(dW), _p(db), B, Ho, Wo, Cip, Cop, Cin, Cout, taps, mode, 0, _st())
            return dW.permute(0, 3, 1, 2), db
    dW = conv_wgrad_raw(a, dy, None, None, None, 0, 0.0, mode, taps, 0)
    db = colsum_raw(dy.permute(0, 2, 3, 1).reshape(B * Ho * Wo, Cout)) if want_bias else None
    return dW, db


def _grad_free(slot):
    """No gradient accumulated yet on the slot's parameter (on every member of a parameter group)."""
    if slot is None:
        return False
    members = getattr(slot, 'slots', None) or [slot]
    for m in members:
        p = m.ref()
        if p is None or p.grad is not None:
            return False
    return True


class _FusedConv(torch.autograd.Function):
    """y = conv(act(GN/FiLM(x))) + bias (+ residual);  act per `cfg`.

    cfg = dict(mode, taps, act, p_drop, salt, shadows) where shadows() returns the
    (forward, data-gradient) weight shadows in the activation dtype.

    bf16: ONE launch (`idf_conv_gn_bf16`) -- the GroupNorm statistics of x were left behind by the launch
    that produced x (`xst`; or one stand-alone pass), each block folds them into the per-(sample, channel)
    affine and applies it, SiLU and dropout while staging its tile; in training that launch also writes
    the activated tensor `a` (kept for the weight gradient) and the coefficients the GroupNorm backward
    needs.  fp32 / shapes the halo kernels do not cover: `a` is materialised by the GroupNorm kernels
    first (one launch, or three for big samples) and the conv is a plain implicit GEMM on it.

    Outputs: y [, statistics partials of y (want_stats; not differentiable)] [, 1-2 aliases of x].
    """

    @staticmethod
    def forward(ctx, x, weight, bias, gn_w, gn_b, film_t, film_a, residual, seed, cfg, train=False, slots=None,
                passthrough=False, xst=None, want_stats=False, lazy_out=False, pre=None):
        ctx.set_materialize_grads(False)      # no zero-filled gradients for the statistics output / unused aliases
        x = _nhwc(x)
        residual = _nhwc(residual) if residual is not None else None
        act, mode, taps = cfg['act'], cfg['mode'], cfg['taps']
        p_drop = cfg['p_drop'] if seed is not None else 0.0
        mean = rstd = sc = sh = None
        a = x
        st = None
        w_fwd = cfg['shadows'](x.dtype, train)[0]
        Cout = weight.shape[0]
        # needs_input_grad follows the inputs' requires_grad even under no_grad(): without the mode check every inference conv kept
        # the activated tensor and the GroupNorm coefficients "for the backward pass" (one extra tensor write per conv: sampling)
        need = ctx.needs_input_grad if slots is not None else (False,) * len(ctx.needs_input_grad)
        bias_k = cfg.get('bias_values')          # the values the kernel adds (AttnBlock: the folded (bq | bk | b')); `bias` routes gradients
        if bias_k is None:
            bias_k = bias
        if pre is not None:
            # the launch that computed this conv already ran (the one-launch attention block, attn_block_fwd_raw): `pre` is
            # what conv_gn_raw / conv_raw would have returned; this node only records the backward pass
            y, a, mean, rstd, sc, sh, st = pre
            if not act:
                a = x
        elif act and mode == S1 and xst is not None and conv_gn_ok(x, None, taps, Cout):
            y, a, mean, rstd, sc, sh, st = conv_gn_raw(
                x, None, xst, None, gn_w, gn_b, film_t, film_a, seed, cfg['salt'], p_drop, act, w_fwd, bias_k, residual,
                Cout, taps, keep_a=need[1], keep_coef=any(need[i] for i in (0, 3, 4, 5, 6)), want_stats=want_stats,
                shadows=cfg['shadows'])
        else:
            ast = getattr(x, '_gn', None)
            if (act and ast is not None and x.dtype == torch.bfloat16 and x.shape[1] % 32 == 0 and ast.shape[0] == x.shape[0]
                    and ast.shape[2] == x.shape[1] and x.shape[0] * x.shape[2] * x.shape[3] >= _GN_STREAM_MINPIX):
                # big tensor whose conv stays a launch of its own: coefficients from the producer's partials + the
                # streaming apply (5.4 TB/s) instead of the one-launch GroupNorm (serial phases per CU, 3.8 TB/s)
                mean, rstd, sc, sh = gn_coef_from_stats_raw(ast, x.shape[1], x.shape[2] * x.shape[3], gn_w, gn_b, film_t, film_a)
                a = gn_apply_raw(x, sc, sh, seed, cfg['salt'], p_drop, act)
            elif act and gn_small_ok(x):
                a, mean, rstd, sc, sh = gn_fused_fwd_raw(x, gn_w, gn_b, film_t, film_a, seed, cfg['salt'], p_drop, act)
            elif act:
                mean, rstd, sc, sh = gn_coef_fwd_raw(x, gn_w, gn_b, film_t, film_a)
                a = gn_apply_raw(x, sc, sh, seed, cfg['salt'], p_drop, act)
            y = conv_raw(a, w_fwd, bias_k, residual, None, None, None, 0, 0.0, mode, taps, 0, Cout, want_stats=want_stats)
            if want_stats:
                y, st = y
        ctx.cfg, ctx.p_drop, ctx.slots = cfg, p_drop, slots or (None, None, None, None)
        ctx.has_res = residual is not None
        ctx.has_st = st is not None
        # the backward chain of the big maps (idf_conv_dgrad_chain_bf16) covers this conv's data gradient: its GroupNorm
        # backward then leaves the one-launch kernel, and the conv may take the gradient of its OUTPUT as a LazyGrad pair
        ctx.chain_ok = (_BWD_CHAIN and mode == S1 and x.dtype == torch.bfloat16 and x.is_cuda and
                        chain_tiles(x.shape[0], x.shape[2], x.shape[3], Cout, x.shape[1], taps) > 0)
        ctx.lazy_in_ok = ctx.chain_ok and _BWD_LAZY
        ctx.lazy_out = bool(lazy_out)
        ctx.save_for_backward(x, a if act else None, weight, bias, gn_w, gn_b, film_t, film_a, mean, rstd, sc, sh,
                              seed)
        outs = (y,)
        if st is not None:
            ctx.mark_non_differentiable(st)
            outs += (st,)
        if passthrough:
            # extra outputs = x itself (1 or 2 aliases): whatever gradient the block's residual / shortcut
            # branch, or a skip connection branching off x, sends back arrives HERE and is added inside the
            # GroupNorm backward kernel / the data-gradient epilogue (no autograd add pass)
            outs += tuple(x.detach() for _ in range(int(passthrough)))
        return outs if len(outs) > 1 else y

    @staticmethod
    def backward(ctx, dy, *extra):
        if ctx.has_st:
            extra = extra[1:]
        dxp = extra[0] if len(extra) > 0 else None
        dxp2 = extra[1] if len(extra) > 1 else None
        x, a, weight, bias, gn_w, gn_b, film_t, film_a, mean, rstd, sc, sh, seed = ctx.saved_tensors
        cfg, p_drop = ctx.cfg, ctx.p_drop
        act, mode, taps, salt = cfg['act'], cfg['mode'], cfg['taps'], cfg['salt']
        if a is None:
            a = x
        if dy is None:        # y itself unused: only what arrived over the aliases flows on
            dy = torch.zeros((x.shape[0], weight.shape[0]) + out_hw(mode, x.shape[2], x.shape[3]), dtype=x.dtype,
                             device=x.device).contiguous(memory_format=CL)
        need = ctx.needs_input_grad
        want_dgrad = need[0] or (act and (need[3] or need[5] or need[6]))
        # the gradient of y may arrive as a (du, partials) pair from the GroupNorm stage that read y (LazyGrad): this
        # conv's data-gradient launch then forms it while staging its tile; anything else needs the tensor
        lazy_in = _LAZY_PENDING.pop(dy.data_ptr(), None)
        if lazy_in is not None and not (ctx.chain_ok and want_dgrad):
            dy, lazy_in = lazy_in.materialize(), None
        dy = _nhwc(dy.to(x.dtype))
        dW = db = dx = dgw = dgb = dft = dfa = dres = None
        want_b = bias is not None and need[2]
        if want_dgrad:
            w_dgrad = cfg['shadows'](x.dtype, True)[1]
            dres_in = _nhwc(dxp.to(x.dtype)) if dxp is not None else None
            dres2_in = _nhwc(dxp2.to(x.dtype)) if dxp2 is not None else None
            if dres_in is None:
                dres_in, dres2_in = dres2_in, None
            gslots = (ctx.slots[2], ctx.slots[3]) if (need[3] and need[4]) else None
            fused_bwd = bool(act) and lazy_in is None and conv_dgrad_gn_ok(dy, x, mode, taps)
            chain = not fused_bwd and ctx.chain_ok and (bool(act) or lazy_in is not None)
            if chain:
                # big maps: the conv's epilogue emits du and the per-tile sums; what is left of the GroupNorm backward
                # is a streaming pass -- or nothing, when the conv in front of this GroupNorm takes the pair
                want_dy = lazy_in is not None and (need[1] or want_b or (ctx.has_res and need[7]))
                got = None
                if act and lazy_in is None and taps == 9 and not ctx.lazy_out:
                    got = conv_dgrad_gn_sync_raw(dy, x.shape[1], x, gn_w, gn_b, film_t, film_a, mean, rstd, sc, sh, seed, salt,
                                                 p_drop, act, gslots, dres_in, dres2_in, shadows=cfg['shadows'])
                if got is not None:
                    dx, dgw, dgb, dft, dfa = got
                elif act:
                    du, part, dy_mat = conv_dgrad_chain_raw(dy, w_dgrad, taps, x.shape[1], lazy_in, want_dy, x=x, sc=sc, sh=sh,
                                                            seed=seed, salt=salt, p_drop=p_drop, act=act, shadows=cfg['shadows'])
                    gacc = _gn_acc(gslots) if (ctx.lazy_out and dres_in is None) else None
                    if ctx.lazy_out and dres_in is None and (gacc is not None or not (need[3] or need[4])):
                        dft = torch.empty(film_t.shape, dtype=torch.float32, device=x.device) if film_t is not None else None
                        dfa = torch.empty(film_a.shape, dtype=torch.float32, device=x.device) if film_a is not None else None
                        _LAZY_PENDING[du.data_ptr()] = LazyGrad(du=du, part=part, x=x, gn_w=gn_w, gn_b=gn_b, film_t=film_t,
                                                                film_a=film_a, mean=mean, rstd=rstd, sc=sc, dft=dft, dfa=dfa,
                                                                acc=(_p(gacc[0]), _p(gacc[1])) if gacc is not None else None)
                        if len(_LAZY_PENDING) == 1:
                            try:
                                torch.autograd.Variable._execution_engine.queue_callback(_lazy_check_consumed)
                            except RuntimeError:
                                pass
                        dx = du
                        if gacc is not None:
                            dgw, dgb = gacc
                    else:
                        dx, dgw, dgb, dft, dfa = gn_bwd_apply_raw(du, part, x, gn_w, gn_b, film_t, film_a, mean, rstd, sc,
                                                                  gslots, dres_in, dres2_in)
                else:
                    dx, _, dy_mat = conv_dgrad_chain_raw(dy, w_dgrad, taps, x.shape[1], lazy_in, want_dy)
                    for ex in (dres_in, dres2_in):
                        if ex is not None:
                            dx = dx + ex
                if lazy_in is not None:
                    dy = dy_mat           # None when nothing below reads it
            elif fused_bwd:
                # small maps: the data-gradient conv's tile is a whole image, its epilogue IS the GroupNorm backward
                dx, dgw, dgb, dft, dfa = conv_dgrad_gn_raw(dy, w_dgrad, x, gn_w, gn_b, film_t, film_a, mean, rstd, sc, sh,
                                                           seed, salt, p_drop, act, taps, gslots, dres_in, dres2_in,
                                                           shadows=cfg['shadows'])
            else:
                if not act and dres_in is not None and dres2_in is None and x.dtype == torch.bfloat16:
                    dA = conv_dgrad_raw(dy, w_dgrad, mode, taps, x.shape, residual=dres_in, shadows=cfg['shadows'])   # joined in the epilogue
                    dres_in = None
                else:
                    dA = conv_dgrad_raw(dy, w_dgrad, mode, taps, x.shape, shadows=cfg['shadows'])
                if act and gn_small_ok(x):
                    dx, dgw, dgb, dft, dfa = gn_fused_bwd_raw(dA, x, gn_w, gn_b, film_t, film_a, mean, rstd, sc, sh,
                                                              seed, salt, p_drop, act, gslots, dres_in, dres2=dres2_in)
                elif act:
                    if dres2_in is not None:
                        dres_in = dres_in + dres2_in
                    dx, dgw, dgb, dft, dfa = gn_coef_bwd_raw(dA, x, dres_in, gn_w, gn_b, film_t, film_a, mean, rstd,
                                                             sc, sh, seed, salt, p_drop, act, gslots)
                else:
                    dx = dA
                    for ex in (dres_in, dres2_in):
                        if ex is not None:
                            dx = dx + ex
        elif need[0] and (dxp is not None or dxp2 is not None):
            dx = dxp if dxp2 is None else (dxp2 if dxp is None else dxp + dxp2)
        # (after the data gradient: with a lazy input the gradient tensor itself is that launch's side output)
        if need[1]:
            # deferring is safe only when AccumulateGrad merely adopts the slot views (no kernel reads them
            # before the end-of-backward launch)
            defer = _grad_free(ctx.slots[0]) and (bias is None or _grad_free(ctx.slots[1]))
            dW, db = conv_wgrad_bias_raw(a, dy, mode, taps, want_b, ctx.slots[0], ctx.slots[1], defer)
        elif want_b:
            B, Co, Ho, Wo = dy.shape
            db = colsum_raw(dy.permute(0, 2, 3, 1).reshape(B * Ho * Wo, Co))
        if ctx.has_res and need[7]:
            dres = dy
        return dx, dW, db, dgw, dgb, dft, dfa, dres, None, None, None, None, None, None, None, None, None


def _tag(t, st):
    """Attach the statistics partials of t (what the GroupNorm reading t would compute) to it."""
    if st is not None:
        t._gn = st
    return t


def fused_conv(x, weight, bias, cfg, gn_w=None, gn_b=None, film_t=None, film_a=None, residual=None, seed=None,
               passthrough=False, want_stats=False, x_single_use=False, pre=None):
    """passthrough = 1 / 2 returns (y, x') / (y, x', x''): the extra outputs alias x, and gradients sent to
    them (the residual / shortcut branch of a ResBlock, a skip connection) are added to dx inside this
    op's GroupNorm backward kernel (or its data-gradient epilogue).
    want_stats: the conv's epilogue leaves the GroupNorm statistics of y behind (y._gn), so the GroupNorm-fused
    conv that consumes y needs no statistics pass."""
    train = torch.is_grad_enabled() and x.requires_grad   # the data-gradient shadow will be needed
    # x_single_use: the caller promises that this op is the ONLY reader of x.  When x came out of a conv whose data-gradient
    # launch can form its own input gradient (lazy_in_ok), this op's GroupNorm backward may hand (du, partials) back
    # instead of dx (LazyGrad) -- one streaming pass less per stage on the 64x64 / 32x32 maps
    lazy_out = bool(x_single_use and train and cfg['act'] and not passthrough and
                    getattr(x.grad_fn, 'lazy_in_ok', False))
    slots = None
    if torch.is_grad_enabled():
        slots = (slot_of(weight), slot_of(bias), slot_of(gn_w), slot_of(gn_b))
    xst = None
    in_st = getattr(x, '_gn', None)
    want_stats = bool(want_stats) and x.is_cuda and x.dtype == torch.bfloat16
    if cfg['act'] and cfg['mode'] == S1 and conv_gn_ok(x, None, cfg['taps'], weight.shape[0]):
        xst = stats_of(x)
        in_st = xst
    out = _FusedConv.apply(x, weight, bias, gn_w, gn_b, film_t, film_a, residual, seed, cfg, train, slots, int(passthrough),
                           xst, want_stats, lazy_out, pre)
    if not isinstance(out, tuple):
        return out
    y, rest = out[0], list(out[1:])
    if rest and rest[0].dtype == torch.float32 and rest[0].dim() == 4 and rest[0].shape[-1] == 2 and (
            len(rest) > int(passthrough)):
        _tag(y, rest.pop(0))
    for alias in rest:
        _tag(alias, in_st)
    return (y,) + tuple(rest) if rest else y


# ------------------------------------------- ResBlock entry on a skip concatenation
def _defer_or_launch_wgrad(a, dy, w_slot, b_slot, taps, a2=None):
    """Arena-accumulating weight (+bias) gradient of a stride-1 conv, deferred to the end of backward when
    nothing but AccumulateGrad will touch the slots; returns (dW, db) slot views or None (no free slots)."""
    B, Cin1, H, W = a.shape
    Cin = Cin1 + (a2.shape[1] if a2 is not None else 0)
    Cout = dy.shape[1]
    k = 3 if taps == 9 else 1
    if not (_grad_free(w_slot) and _grad_free(b_slot) and WgradBatch.enabled
            and _fast_wgrad_ok(Cin, Cout, H, W, a.dtype, S1, taps)):
        return None
    views = _slot_views(w_slot, b_slot, True, (Cout, Cin, k, k))
    if views is None:
        return None
    WgradBatch.add((a, dy, _p(views[0]), _p(views[1]), B, H, W, Cin, Cout, taps, S1, a2, Cin1, Cin, Cout))
    return views


class _BlockEntryCat(torch.autograd.Function):
    """First stage of an up-path ResBlock whose input is the skip concatenation x = cat(x1, x2)
    (models.py:321) WITHOUT materialising it:   h = conv3x3(SiLU(GN(x))) + b,   s = shortcut1x1(x) + bs.
    GroupNorm, the 1x1 shortcut and the shortcut's weight gradient read the two tensors in place; the
    backward returns dx1 and dx2 as separate dense tensors straight from the GroupNorm backward kernel
    (which also adds the shortcut's data gradient), so neither the concat nor its split copies exist."""

    @staticmethod
    def forward(ctx, x1, x2, w, b, gn_w, gn_b, sw, sb, cfg, cfg_sc, train, slots, st1=None, st2=None):
        ctx.set_materialize_grads(False)
        x1, x2 = _nhwc(x1), _nhwc(x2)
        B, C1, H, W = x1.shape
        C = C1 + x2.shape[1]
        need = ctx.needs_input_grad if slots is not None else (False,) * len(ctx.needs_input_grad)     # (no_grad(): see _FusedConv)
        w_fwd = cfg['shadows'](x1.dtype, train)[0]
        st = None
        if st1 is not None and st2 is not None and conv_gn_ok(x1, x2, 9, w.shape[0]):
            # GroupNorm + SiLU applied while the conv stages the two sources; statistics from their producers
            # (the 1x1 shortcut over the same raw input rides in the launch: IDF_SC_FUSE)
            ride = _SC_FUSE and B * H * W <= _SC_FUSE_MAXPIX and w.shape[0] > 32 and sw.shape[0] % 8 == 0
            if cfg['act'] == 2 and _wr_frag(cfg['shadows'], 2, B, H, W, C, w.shape[0], 0) is not None:
                ride = False       # the fragment-major form carries no rider: the 1x1 shortcut is its own launch below
            out = conv_gn_raw(x1, x2, st1, st2, gn_w, gn_b, None, None, None, cfg['salt'], 0.0, cfg['act'], w_fwd, b, None,
                              w.shape[0], 9, keep_a=need[2], keep_coef=any(need[i] for i in (0, 1, 4, 5)), want_stats=True,
                              shortcut=(cfg_sc['shadows'](x1.dtype, train)[0], sb, sw.shape[0]) if ride else None,
                              shadows=cfg['shadows'])
            h, a, mean, rstd, sc, sh, st = out[:7]
            s = out[7] if ride else None
        else:
            a1, a2 = getattr(x1, '_gn', None), getattr(x2, '_gn', None)
            if (a1 is not None and a2 is not None and x1.dtype == torch.bfloat16 and C % 32 == 0 and C1 % 8 == 0
                    and a1.shape[0] == B and a2.shape[0] == B and a1.shape[2] == C1 and a2.shape[2] == C - C1
                    and B * H * W >= _GN_STREAM_MINPIX):
                # big pair whose conv stays a launch of its own: coefficients from the producers' partials + streaming apply
                mean, rstd, sc, sh = gn_coef_from_stats_raw(a1, C, H * W, gn_w, gn_b, None, None, st2=a2)
                a = gn_apply2_raw(x1, x2, sc, sh, None, cfg['salt'], 0.0, cfg['act'])
            else:
                a, mean, rstd, sc, sh = gn_fused_fwd_raw(x1, gn_w, gn_b, None, None, None, cfg['salt'], 0.0, cfg['act'], x2=x2)
            h, st = conv_raw(a, w_fwd, b, None, None, None, None, 0, 0.0, S1, 9, 0, w.shape[0], want_stats=True)
            s = None
        if s is None:
            s = empty_nhwc(B, sw.shape[0], H, W, x1.dtype, x1.device)
            call('idf_conv1x1_bf16', _p(x1), _p(x2), C1, _p(cfg_sc['shadows'](x1.dtype, train)[0]), _p(sb), None, _p(s),
                 B, H, W, C, sw.shape[0], None, _st())
        ctx.cfg, ctx.cfg_sc, ctx.slots = cfg, cfg_sc, slots or (None,) * 6
        ctx.has_st = st is not None
        ctx.chain_ok = _BWD_CHAIN and C1 % 64 == 0 and chain_tiles(B, H, W, w.shape[0], C, 9) > 0     # see _FusedConv
        ctx.lazy_in_ok = ctx.chain_ok and _BWD_LAZY
        ctx.save_for_backward(x1, x2, a, w, b, gn_w, gn_b, sw, sb, mean, rstd, sc, sh)
        if st is not None:
            ctx.mark_non_differentiable(st)
            return h, s, st
        return h, s

    @staticmethod
    def backward(ctx, dh, ds, *_):
        x1, x2, a, w, b, gn_w, gn_b, sw, sb, mean, rstd, sc, sh = ctx.saved_tensors
        cfg, cfg_sc = ctx.cfg, ctx.cfg_sc
        ws, bs, gws, gbs, sws, sbs = ctx.slots
        B, C1, H, W = x1.shape
        C = C1 + x2.shape[1]
        if dh is None:
            dh = torch.zeros((B, w.shape[0], H, W), dtype=x1.dtype, device=x1.device).contiguous(memory_format=CL)
        if ds is None:
            ds = torch.zeros((B, sw.shape[0], H, W), dtype=x1.dtype, device=x1.device).contiguous(memory_format=CL)
        lazy_in = _LAZY_PENDING.pop(dh.data_ptr(), None)       # the gradient of h as a (du, partials) pair: see _FusedConv
        if lazy_in is not None and not ctx.chain_ok:
            dh, lazy_in = lazy_in.materialize(), None
        dh, ds = _nhwc(dh.to(x1.dtype)), _nhwc(ds.to(x1.dtype))
        # shortcut: weight gradient over the two-source input, data gradient dense (joins in the GN backward)
        got = _defer_or_launch_wgrad(x1, ds, sws, sbs, 1, a2=x2)
        if got is None:
            xc = torch.cat([x1, x2], dim=1).contiguous(memory_format=CL)
            got = conv_wgrad_bias_raw(xc, ds, S1, 1, True)
        dsW, dsb = got
        w_sc_dgrad = cfg_sc['shadows'](x1.dtype, True)[1]
        w_dgrad = cfg['shadows'](x1.dtype, True)[1]
        ride = (ctx.chain_ok and _SC_FUSE and B * H * W <= _SC_FUSE_MAXPIX and lazy_in is None and
                ds.shape[1] % 32 == 0)                      # shortcut dgrad in the chain launch
        dxs = None if ride else conv_dgrad_raw(ds, w_sc_dgrad, S1, 1, (B, C, H, W))
        if ctx.chain_ok:
            # the conv's epilogue emits du and the per-tile sums over the two-source input; dx1 / dx2 by a streaming pass
            got = None
            if ride:
                du, part, dxs = conv_dgrad_chain_raw(dh, w_dgrad, 9, C, x=x1, x2=x2, sc=sc, sh=sh, act=cfg['act'],
                                                     shortcut=(ds, w_sc_dgrad))
            else:
                if lazy_in is None:
                    got = conv_dgrad_gn_sync_raw(dh, C, x1, gn_w, gn_b, None, None, mean, rstd, sc, sh, None, cfg['salt'], 0.0,
                                                 cfg['act'], (gws, gbs), dxs, None, x2=x2, shadows=cfg['shadows'])
                if got is None:
                    du, part, dh_mat = conv_dgrad_chain_raw(dh, w_dgrad, 9, C, lazy_in, lazy_in is not None, x=x1, x2=x2, sc=sc,
                                                            sh=sh, act=cfg['act'], shadows=cfg['shadows'])
                    if lazy_in is not None:
                        dh = dh_mat
            if got is not None:
                (dx1, dx2), dgw, dgb, _, _ = got
            else:
                (dx1, dx2), dgw, dgb, _, _ = gn_bwd_apply_raw(du, part, x1, gn_w, gn_b, None, None, mean, rstd, sc, (gws, gbs),
                                                               dres=dxs, x2=x2)
        else:
            dA = conv_dgrad_raw(dh, w_dgrad, S1, 9, a.shape)
            (dx1, dx2), dgw, dgb, _, _ = gn_fused_bwd_raw(dA, x1, gn_w, gn_b, None, None, mean, rstd, sc, sh, None,
                                                           cfg['salt'], 0.0, cfg['act'], (gws, gbs), dres=dxs, x2=x2)
        got = _defer_or_launch_wgrad(a, dh, ws, bs, 9)
        if got is None:
            got = conv_wgrad_bias_raw(a, dh, S1, 9, True)
        dW, db = got
        return dx1, dx2, dW, db, dgw, dgb, dsW, dsb, None, None, None, None, None, None


def block_entry_cat_ok(x1, x2, conv_w, sc_w):
    """Shapes the two-source kernels cover (else the caller concatenates)."""
    if not (x1.is_cuda and x1.dtype == torch.bfloat16 and x2.dtype == torch.bfloat16 and x1.shape[2:] == x2.shape[2:]):
        return False
    B, C1, H, W = x1.shape
    C = C1 + x2.shape[1]
    if not (C1 % 32 == 0 and x2.shape[1] % 32 == 0 and 4 <= W <= 128 and not (W & (W - 1)) and sc_w.shape[0] % 8 == 0
            and conv_w.shape[1] == C and _CONV1X1 and uses_halo_kernel(x1.dtype, 9, 0, S1, B, C, conv_w.shape[0], H, W)):
        return False
    # the GroupNorm backward (and the forward where the one-launch conv does not apply) runs in the one-launch
    # two-source GroupNorm kernels; an inference pass needs only the GroupNorm-prologue conv
    if gn_small_ok(x1, x2) and C1 % 64 == 0:
        return True
    return not torch.is_grad_enabled() and conv_gn_ok(x1, x2, 9, conv_w.shape[0])


def block_entry_cat(x1, x2, conv, gn, shortcut, cfg, cfg_sc):
    train = torch.is_grad_enabled() and (x1.requires_grad or x2.requires_grad)
    slots = None
    if torch.is_grad_enabled():
        slots = tuple(slot_of(p) for p in (conv.weight, conv.bias, gn.weight, gn.bias, shortcut.weight, shortcut.bias))
    st1 = st2 = None
    if conv_gn_ok(x1, x2, 9, conv.weight.shape[0]):
        st1, st2 = stats_of(x1), stats_of(x2)
    out = _BlockEntryCat.apply(x1, x2, conv.weight, conv.bias, gn.weight, gn.bias, shortcut.weight, shortcut.bias,
                               cfg, cfg_sc, train, slots, st1, st2)
    if len(out) == 3:
        _tag(out[0], out[2])
    return out[0], out[1]


# ------------------------------------------- image-resident ResBlock at the 8x8 maps
_RB_SMALL = knobs.flag('IDF_RB_SMALL')
# one workgroup per image.  Measured up to B = 256 (profiles/r04_resblock_small.txt): DDIM-100 at B = 256 291 -> 303 img/s, the
# B = 128 train step 4958 -> 4994 img/s against a limit of 64; the per-conv fragment-major form (_WR_MAXB) LOSES beyond 64 (276 img/s)
_RB_SMALL_MAXB = knobs.num('IDF_RB_SMALL_MAXB')
_RB_SMALL_BWD = True      # ... and its backward (idf_resblock_small_bwd)
_RB_WFRAG = True              # fragment-major weights (modules._Shadows.request_frag): 35.5 -> 23.2 us per block


@functools.lru_cache(maxsize=None)
def _rb_small_shape_ok(B, H, W, Cin, C1, Cout, nstage):
    return bool(_lib.load().idf_resblock_small_ok(B, H, W, Cin, C1, Cout, nstage))


def resblock_small_ok(x, x2, Cout, nstage):
    """idf_resblock_small_fwd covers this block input (bf16, 8x8, 128 couts, batch small enough to be launch-bound)."""
    if not (_RB_SMALL and x.is_cuda and x.dtype == torch.bfloat16 and (x2 is None or x2.dtype == torch.bfloat16)):
        return False
    B, C1, H, W = x.shape
    if B > _RB_SMALL_MAXB or (x2 is not None and x2.shape[2:] != x.shape[2:]):
        return False
    Cin = C1 + (x2.shape[1] if x2 is not None else 0)
    return _rb_small_shape_ok(B, H, W, Cin, C1 if x2 is not None else 0, Cout, nstage)


class _ResBlockSmall(torch.autograd.Function):
    """A whole ResBlock / AuxResBlock / ResBlock_encoder at 8x8 as ONE forward launch (idf_resblock_small_fwd: one
    workgroup per image, every GroupNorm closes in-block, the stages chained through LDS).  The backward pass is the
    per-stage one of `_FusedConv` / `_BlockEntryCat` -- the data-gradient convs that carry the GroupNorm backward as their
    epilogue, deferred weight gradients -- issued stage by stage from the tensors the forward launch left behind.

    tensors = [w, b, gn_w, gn_b] per stage (+ [sc_w, sc_b] when meta['has_sc']).  Outputs: y, statistics partials of y
    (non-differentiable), [an alias of x for the skip connection that branches off the block input]."""

    @staticmethod
    def forward(ctx, x, x2, film_t, film_a, seed, st1, st2, meta, *tensors):
        ctx.set_materialize_grads(False)
        x = _nhwc(x)
        x2 = _nhwc(x2) if x2 is not None else None
        n, has_sc, cfgs = meta['nstage'], meta['has_sc'], meta['cfgs']
        B, C1, H, W = x.shape
        Cin = C1 + (x2.shape[1] if x2 is not None else 0)
        Cout = tensors[0].shape[0]
        dev, dt = x.device, x.dtype
        train = meta['train']
        need = ctx.needs_input_grad
        A = _lib.ResblockArgs()
        A.x, A.x2, A.C1, A.Cin = _p(x), _p(x2), C1 if x2 is not None else Cin, Cin
        A.st1, A.T1 = _p(st1), st1.shape[1]
        A.st2, A.T2 = (_p(st2), st2.shape[1]) if st2 is not None else (None, 0)
        A.nstage = n
        y = empty_nhwc(B, Cout, H, W, dt, dev)
        st_out = torch.empty((B, 1, Cout, 2), dtype=torch.float32, device=dev)
        keep = []           # per stage: a, h (None for the last), mean, rstd, sc, sh
        p_drop = meta['p_drop'] if seed is not None else 0.0
        # weights: the fragment-major shadows once every stage has one (requested at a conv's first pass through here, packed
        # with the next re-pack of the network's shadows), the [cout][tap][cin] forward shadows until then
        shadows = [cfgs[i]['shadows'](dt, train) for i in range(n)]
        frag = _RB_WFRAG and all(v[2] is not None for v in shadows)
        if _RB_WFRAG and not frag:
            for i in range(n):
                cfgs[i]['shadows'].request_frag()
        for i in range(n):
            w, b, gw, gb = tensors[4 * i:4 * i + 4]
            S = A.s[i]
            ci = Cin if i == 0 else Cout
            S.w, S.bias = _p(shadows[i][2 if frag else 0]), _p(b)
            S.gamma, S.beta = _p(gw), _p(gb)
            ft, fa = (film_t, film_a) if i == meta['film_stage'] else (None, None)
            S.film_t, S.film_a, S.ld_t, S.ld_a = _p(ft), _p(fa), _ld(ft), _ld(fa)
            S.salt, S.drop = cfgs[i]['salt'], int(meta['drop'][i] and seed is not None)
            a = h = mean = rstd = sc = sh = None
            if train:
                a = empty_nhwc(B, ci, H, W, dt, dev)
                mean = torch.empty((B, 32), dtype=torch.float32, device=dev)
                rstd = torch.empty((B, 32), dtype=torch.float32, device=dev)
                sc = torch.empty((B, ci), dtype=torch.float32, device=dev)
                sh = torch.empty((B, ci), dtype=torch.float32, device=dev)
                if i + 1 < n:
                    h = empty_nhwc(B, Cout, H, W, dt, dev)
            S.a_out, S.mean, S.rstd, S.sc, S.sh = _p(a), _p(mean), _p(rstd), _p(sc), _p(sh)
            S.h_out = _p(y) if i + 1 == n else _p(h)
            keep += [a, h, mean, rstd, sc, sh]
        if has_sc:
            A.w_sc, A.b_sc = _p(meta['cfg_sc']['shadows'](dt, train)[0]), _p(tensors[4 * n + 1])
        A.y, A.st_out = _p(y), _p(st_out)
        A.seed, A.p_drop, A.eps, A.B = _p(seed), float(p_drop), GN_EPS, B
        A.w_layout = 1 if frag else 0
        call('idf_resblock_small_fwd', ctypes.byref(A), _st())
        ctx.meta, ctx.p_drop = meta, p_drop
        ctx.save_for_backward(x, x2, film_t, film_a, seed, *keep, *tensors)
        ctx.mark_non_differentiable(st_out)
        outs = (y, st_out)
        if meta['passthrough']:
            outs += (x.detach(),)
        return outs

    @staticmethod
    def _fused_backward(ctx, dy, dskip, grads):
        """Stages n-1 .. first of the backward pass as ONE launch (idf_resblock_small_bwd) where the data-gradient weights
        exist fragment-major and the GroupNorm affine gradients can go straight into their arena slots: fills `grads` for those
        stages (weight gradients deferred as ever) -> (gradient handed to the stage below / the block's dx, dFiLM_t, dFiLM_a,
        first), or None (the per-stage launches run)."""
        meta = ctx.meta
        n, has_sc, cfgs, slots = meta['nstage'], meta['has_sc'], meta['cfgs'], meta['slots']
        sv = ctx.saved_tensors
        x, x2, film_t, film_a, seed = sv[:5]
        keep = sv[5:5 + 6 * n]
        tensors = sv[5 + 6 * n:]
        B = x.shape[0]
        first = 0 if (x2 is None and not has_sc) else 1
        if not (_RB_SMALL_BWD and n - first >= 2 and B <= _RB_SMALL_MAXB):
            return None
        shadows = [cfgs[i]['shadows'](x.dtype, True) for i in range(n)]
        if any(shadows[i][3] is None for i in range(first, n)):
            for i in range(first, n):
                cfgs[i]['shadows'].request_frag()
            return None
        accs = []
        if any(slots[4 * i + 2] is None or slots[4 * i + 3] is None or not (slots[4 * i + 2].available() and slots[4 * i + 3].available())
               for i in range(first, n)):
            return None             # (a slot taken twice: gradient accumulation without zero_grad -- the stand-alone path adds)
        for i in range(first, n):
            accs.append(_gn_out((slots[4 * i + 2], slots[4 * i + 3]), B, 128, x.device))     # (acc, None, None) | (None, rows, views)
        dev, dt = x.device, x.dtype
        A = _lib.ResblockBwdArgs()
        A.dy, A.nstage, A.first = _p(dy), n, first
        A.dres2 = _p(dskip) if first == 0 else None
        A.seed, A.p_drop, A.B = _p(seed), float(ctx.p_drop), B
        dft = dfa = None
        dxs = {}
        for i in range(first, n):
            S = A.s[i]
            gx = x if i == 0 else keep[6 * (i - 1) + 1]
            mean, rstd, sc, sh = keep[6 * i + 2:6 * i + 6]
            gw, gb = tensors[4 * i + 2], tensors[4 * i + 3]
            ft, fa = (film_t, film_a) if i == meta['film_stage'] else (None, None)
            S.w_frag, S.x = _p(shadows[i][3]), _p(gx)
            S.gamma, S.beta, S.film_t, S.film_a, S.ld_t, S.ld_a = _p(gw), _p(gb), _p(ft), _p(fa), _ld(ft), _ld(fa)
            S.mean, S.rstd, S.sc, S.sh = _p(mean), _p(rstd), _p(sc), _p(sh)
            S.salt, S.drop = cfgs[i]['salt'], int(bool(meta['drop'][i]) and seed is not None)
            if ft is not None:
                dft = torch.empty(ft.shape, dtype=torch.float32, device=dev)
            if fa is not None:
                dfa = torch.empty(fa.shape, dtype=torch.float32, device=dev)
            S.dfilm_t, S.dfilm_a = (_p(dft), _p(dfa)) if (ft is not None or fa is not None) else (None, None)
            acc, rows, _ = accs[i - first]
            S.dgb, S.dgamma_acc, S.dbeta_acc = _p(rows), _p(acc[0]) if acc else None, _p(acc[1]) if acc else None
            dxs[i] = torch.empty_like(gx, memory_format=CL)
            S.dx = _p(dxs[i])
        call('idf_resblock_small_bwd', ctypes.byref(A), _st())
        for i in range(n - 1, first - 1, -1):
            gi = dy if i == n - 1 else dxs[i + 1]          # the gradient of stage i's conv output
            ws, bs = slots[4 * i], slots[4 * i + 1]
            dW, db = conv_wgrad_bias_raw(keep[6 * i], gi, S1, 9, True, ws, bs, _grad_free(ws) and _grad_free(bs))
            acc, _, later = accs[i - first]
            g_ = acc or later
            grads[4 * i:4 * i + 4] = [dW, db, g_[0], g_[1]]
        _gn_done()
        return dxs[first], dft, dfa, first

    @staticmethod
    def backward(ctx, dy, _dst, *dalias):
        meta = ctx.meta
        n, has_sc, cfgs, slots = meta['nstage'], meta['has_sc'], meta['cfgs'], meta['slots']
        sv = ctx.saved_tensors
        x, x2, film_t, film_a, seed = sv[:5]
        keep = sv[5:5 + 6 * n]
        tensors = sv[5 + 6 * n:]
        p_drop = ctx.p_drop
        dt = x.dtype
        B, C1, H, W = x.shape
        Cout = tensors[0].shape[0]
        if dy is None:
            dy = torch.zeros((B, Cout, H, W), dtype=dt, device=x.device).contiguous(memory_format=CL)
        g = _nhwc(dy.to(dt))
        dskip = _nhwc(dalias[0].to(dt)) if (dalias and dalias[0] is not None) else None
        grads = [None] * (4 * n + (2 if has_sc else 0))
        dft = dfa = None
        # the residual branch: an identity joins the first stage's GroupNorm backward as it stands, a 1x1 shortcut first
        # takes its weight gradient (over the raw input, read in place) and its data gradient
        ds = g
        fused = _ResBlockSmall._fused_backward(ctx, g, dskip, grads)
        if fused is not None:
            g, dft, dfa, first = fused
            if first == 0:           # the launch went all the way down to the block input
                return (g, None, dft, dfa, None, None, None, None) + tuple(grads)
        for i in (range(n - 1, 0, -1) if fused is None else ()):
            a, h_prev = keep[6 * i], keep[6 * (i - 1) + 1]
            mean, rstd, sc, sh = keep[6 * i + 2:6 * i + 6]
            w, b, gw, gb = tensors[4 * i:4 * i + 4]
            ws, bs, gws, gbs = slots[4 * i:4 * i + 4]
            ft, fa = (film_t, film_a) if i == meta['film_stage'] else (None, None)
            sd = seed if meta['drop'][i] else None
            pd = p_drop if meta['drop'][i] else 0.0
            w_dgrad = cfgs[i]['shadows'](dt, True)[1]
            dx, dgw, dgb, dft_i, dfa_i = conv_dgrad_gn_raw(g, w_dgrad, h_prev, gw, gb, ft, fa, mean, rstd, sc, sh, sd,
                                                           cfgs[i]['salt'], pd, 2, 9, (gws, gbs), shadows=cfgs[i]['shadows'])
            if ft is not None or fa is not None:
                dft, dfa = dft_i, dfa_i
            defer = _grad_free(ws) and _grad_free(bs)
            dW, db = conv_wgrad_bias_raw(a, g, S1, 9, True, ws, bs, defer)
            grads[4 * i:4 * i + 4] = [dW, db, dgw, dgb]
            g = dx
        # first stage: its input is the block input
        a0 = keep[0]
        mean, rstd, sc, sh = keep[2:6]
        w, b, gw, gb = tensors[0:4]
        ws, bs, gws, gbs = slots[0:4]
        w_dgrad = cfgs[0]['shadows'](dt, True)[1]
        dx2 = None
        if x2 is None:
            dres = ds
            if has_sc:
                sw, sb = tensors[4 * n], tensors[4 * n + 1]
                sws, sbs = slots[4 * n], slots[4 * n + 1]
                grads[4 * n], grads[4 * n + 1] = conv_wgrad_bias_raw(x, ds, S1, 1, True, sws, sbs, _grad_free(sws) and _grad_free(sbs))
                dres = conv_dgrad_raw(ds, meta['cfg_sc']['shadows'](dt, True)[1], S1, 1, x.shape)
            dx, dgw, dgb, _, _ = conv_dgrad_gn_raw(g, w_dgrad, x, gw, gb, None, None, mean, rstd, sc, sh, None,
                                                   cfgs[0]['salt'], 0.0, 2, 9, (gws, gbs), dres, dskip, shadows=cfgs[0]['shadows'])
        else:
            # a skip pair never hands out a passthrough alias (models._UNetSkeleton._run): a gradient arriving on one would be lost
            assert dskip is None, 'resblock_small: a two-source block entry cannot carry a passthrough alias'
            sw, sb = tensors[4 * n], tensors[4 * n + 1]
            sws, sbs = slots[4 * n], slots[4 * n + 1]
            dx, dx2, dgw, dgb, dsW, dsb = _entry_cat_bwd(x, x2, g, ds, gw, gb, mean, rstd, sc, sh, cfgs[0], meta['cfg_sc'],
                                                          (gws, gbs), (sws, sbs))
            grads[4 * n], grads[4 * n + 1] = dsW, dsb
        defer = _grad_free(ws) and _grad_free(bs)
        dW, db = conv_wgrad_bias_raw(a0, g, S1, 9, True, ws, bs, defer)
        grads[0:4] = [dW, db, dgw, dgb]
        return (dx, dx2, dft, dfa, None, None, None, None) + tuple(grads)


def _entry_cat_bwd(x1, x2, dh, ds, gn_w, gn_b, mean, rstd, sc, sh, cfg, cfg_sc, gslots, sslots):
    """Backward of a block entry over the skip pair x1 | x2 (what `_BlockEntryCat.backward` does for its own node):
    dh = gradient of the first conv's output, ds = gradient of the 1x1 shortcut's output -> (dx1, dx2, dgamma, dbeta,
    shortcut dW, shortcut db); the first conv's own weight gradient is the caller's."""
    B, C1, H, W = x1.shape
    C = C1 + x2.shape[1]
    got = _defer_or_launch_wgrad(x1, ds, sslots[0], sslots[1], 1, a2=x2)
    if got is None:
        xc = torch.cat([x1, x2], dim=1).contiguous(memory_format=CL)
        got = conv_wgrad_bias_raw(xc, ds, S1, 1, True)
    dsW, dsb = got
    w_sc_dgrad = cfg_sc['shadows'](x1.dtype, True)[1]
    w_dgrad = cfg['shadows'](x1.dtype, True)[1]
    chain_ok = _BWD_CHAIN and C1 % 64 == 0 and chain_tiles(B, H, W, dh.shape[1], C, 9) > 0
    if chain_ok:
        ride = _SC_FUSE and B * H * W <= _SC_FUSE_MAXPIX and ds.shape[1] % 32 == 0
        if ride:
            du, part, dxs = conv_dgrad_chain_raw(dh, w_dgrad, 9, C, x=x1, x2=x2, sc=sc, sh=sh, act=cfg['act'],
                                                 shortcut=(ds, w_sc_dgrad))
        else:
            dxs = conv_dgrad_raw(ds, w_sc_dgrad, S1, 1, (B, C, H, W))
            got = conv_dgrad_gn_sync_raw(dh, C, x1, gn_w, gn_b, None, None, mean, rstd, sc, sh, None, cfg['salt'], 0.0, cfg['act'],
                                         gslots, dxs, None, x2=x2, shadows=cfg['shadows'])
            if got is not None:
                (dx1, dx2), dgw, dgb, _, _ = got
                return dx1, dx2, dgw, dgb, dsW, dsb
            du, part, _ = conv_dgrad_chain_raw(dh, w_dgrad, 9, C, x=x1, x2=x2, sc=sc, sh=sh, act=cfg['act'], shadows=cfg['shadows'])
        (dx1, dx2), dgw, dgb, _, _ = gn_bwd_apply_raw(du, part, x1, gn_w, gn_b, None, None, mean, rstd, sc, gslots, dres=dxs,
                                                       x2=x2)
    else:
        dxs = conv_dgrad_raw(ds, w_sc_dgrad, S1, 1, (B, C, H, W))
        dA = conv_dgrad_raw(dh, w_dgrad, S1, 9, (B, C, H, W))
        (dx1, dx2), dgw, dgb, _, _ = gn_fused_bwd_raw(dA, x1, gn_w, gn_b, None, None, mean, rstd, sc, sh, None, cfg['salt'], 0.0,
                                                       cfg['act'], gslots, dres=dxs, x2=x2)
    return dx1, dx2, dgw, dgb, dsW, dsb


def resblock_small(x, x2, stages, shortcut, film_t, film_a, film_stage, seed, p_drop, drop, want_alias):
    """stages = [(conv, gn, cfg)] (2 or 3), shortcut = (conv, cfg) or None.  -> y (statistics attached) [, alias of x]."""
    tensors, slots = [], []
    for conv, gn, _ in stages:
        tensors += [conv.weight, conv.bias, gn.weight, gn.bias]
    if shortcut is not None:
        tensors += [shortcut[0].weight, shortcut[0].bias]
    # anything that will ask this node for a gradient -- the block input OR only its weights (frozen prefix, fine-tuning the
    # deepest blocks) OR the conditioning -- needs the forward launch to keep a / mean / rstd / sc / sh
    train = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (x, x2, film_t, film_a, *tensors))
    if torch.is_grad_enabled():
        slots = [slot_of(t) for t in tensors]
    else:
        slots = [None] * len(tensors)
    meta = dict(nstage=len(stages), has_sc=shortcut is not None, cfgs=[c for _, _, c in stages],
                cfg_sc=shortcut[1] if shortcut is not None else None, slots=slots, passthrough=int(bool(want_alias)),
                train=train, film_stage=film_stage, drop=list(drop), p_drop=p_drop)
    st1 = stats_of(x)
    st2 = stats_of(x2) if x2 is not None else None
    out = _ResBlockSmall.apply(x, x2, film_t, film_a, seed, st1, st2, meta, *tensors)
    y = _tag(out[0], out[1])
    if want_alias:
        return y, _tag(out[2], st1)
    return y


# ------------------------------------------------------------------ attention
_ATTN_BWD_ONE = True      # query and key-value halves of the backward in one launch


class _Attention(torch.autograd.Function):
    """qkv [B, 3C, H, W] (NHWC-dense: [B, N, 3C]) -> softmax(q k^T C^-1/2) v as [B, C, H, W]."""

    @staticmethod
    def forward(ctx, qkv, pre=None):
        qkv = _nhwc(qkv)
        B, C3, H, W = qkv.shape
        C, N = C3 // 3, H * W
        dev, dt = qkv.device, qkv.dtype
        ctx.fused = bool(_lib.load().idf_attn_fused_ok(N, C, _dt(qkv)))
        if pre is not None:         # (o, lse) of the one-launch attention block: record the backward pass only
            o, lse = pre
            ctx.save_for_backward(qkv, lse, o)
            return o
        if ctx.fused:       # one launch, scores and probabilities never leave the registers
            o = empty_nhwc(B, C, H, W, dt, dev)
            lse = torch.empty((B, N), dtype=torch.float32, device=dev)
            call('idf_attn_fwd', _p(qkv), _p(o), _p(lse), B, N, C, float(int(C) ** (-0.5)), _dt(qkv), _st())
            ctx.save_for_backward(qkv, lse, o)          # o: lets the backward run as one launch (idf_attn_bwd_o)
            return o
        S = torch.empty((B, N, N), dtype=dt, device=dev)
        bgemm_raw(qkv, 0, qkv, C, S, 0, None, B, N * C3, N * C3, N * N, C3, C3, N, N, N, C, 0, 0,
                  alpha=float(int(C) ** (-0.5)))
        call('idf_softmax_fwd', _p(S), B * N, N, _dt(S), _st())
        o = empty_nhwc(B, C, H, W, dt, dev)
        bgemm_raw(S, 0, qkv, 2 * C, o, 0, None, B, N * N, N * C3, N * C, N, C3, C, N, C, N, 0, 1)
        ctx.save_for_backward(qkv, S)
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, P = ctx.saved_tensors[:2]
        B, C3, H, W = qkv.shape
        C, N = C3 // 3, H * W
        do = _nhwc(do.to(qkv.dtype))
        scale = float(int(C) ** (-0.5))
        dqkv = torch.empty_like(qkv, memory_format=CL)
        if ctx.fused:                        # P is the saved row logsumexp here
            if _ATTN_BWD_ONE:
                call('idf_attn_bwd_o', _p(qkv), _p(do), _p(P), _p(ctx.saved_tensors[2]), _p(dqkv), B, N, C, scale, _dt(qkv), _st())
            else:
                dsum = torch.empty_like(P)
                call('idf_attn_bwd', _p(qkv), _p(do), _p(P), _p(dsum), _p(dqkv), B, N, C, scale, _dt(qkv), _st())
            return dqkv, None
        # dV = P^T dO
        bgemm_raw(P, 0, do, 0, dqkv, 2 * C, None, B, N * N, N * C, N * C3, N, C, C3, N, C, N, 1, 1)
        # dP = dO V^T
        dP = torch.empty_like(P)
        bgemm_raw(do, 0, qkv, 2 * C, dP, 0, None, B, N * C, N * C3, N * N, C, C3, N, N, N, C, 0, 0)
        call('idf_softmax_bwd', _p(P), _p(dP), B * N, N, _dt(P), _st())
        # dQ = scale * dS K ; dK = scale * dS^T Q
        bgemm_raw(dP, 0, qkv, C, dqkv, 0, None, B, N * N, N * C3, N * C3, N, C3, C3, N, C, N, 0, 1, alpha=scale)
        bgemm_raw(dP, 0, qkv, 0, dqkv, C, None, B, N * N, N * C3, N * C3, N, C3, C3, N, C, N, 1, 1, alpha=scale)
        return dqkv, None


def attention(qkv, pre=None):
    return _Attention.apply(qkv, pre)


# ------------------------------------------------- UpSample at inference: four 2x2 convs on the low-resolution input
_UPCONV = knobs.flag('IDF_UPCONV')
_UPCONV_DGRAD = True
_DOWN_DGRAD = True
_UP_SETS = (((0,), (1, 2)), ((0, 1), (2,)))          # S(parity, tap): the 3x3 kernel rows / columns a low-resolution tap stands for


def upconv_tiles(x, Cout):
    """Statistics tiles per image of idf_upconv_bf16 for this (low-resolution) input; 0: shape not covered."""
    if not (_UPCONV and x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4):
        return 0
    return int(_lib.load().idf_upconv_tiles(x.shape[2], x.shape[3], x.shape[1], Cout))


def upconv_pack(weight, dgrad=False):
    """[O, I, 3, 3] master weights -> the summed sub-pixel weights W'[py][px][ty][tx] (fp32 sums, one bf16 rounding) as
    [O][16][I] (dgrad: [I][16][O], the data gradient's rows = cins), fragment-major (the layout of
    idf_pack_conv_weights_batched's `wfrag`)."""
    with torch.no_grad():
        w = weight.detach().float()
        O, I = w.shape[:2]
        taps = []
        for py in range(2):
            for px in range(2):
                for ty in range(2):
                    for tx in range(2):
                        # kernel rows first, then columns, left to right: the order of upconv_pack_kernel (same bits)
                        cols = []
                        for kx in _UP_SETS[px][tx]:
                            rows = [w[:, :, ky, kx] for ky in _UP_SETS[py][ty]]
                            cols.append(rows[0] if len(rows) == 1 else rows[0] + rows[1])
                        taps.append(cols[0] if len(cols) == 1 else cols[0] + cols[1])
        m = torch.stack(taps, dim=1).to(torch.bfloat16)                  # [O][16][I]
        if dgrad:
            m = m.permute(2, 1, 0).contiguous()                          # [I][16][O]
        N, _, K = m.shape
        return m.view(N // 16, 16, 16, K // 64, 2, 4, 8).permute(3, 0, 2, 4, 5, 1, 6).reshape(-1).contiguous()


def upconv_raw(x, w_sub_frag, bias, Cout, tiles):
    """y = conv3x3(nearest_x2(x)) + bias from the low-resolution x -> (y, statistics partials of y)."""
    x = _nhwc(x)
    B, C, H, W = x.shape
    y = empty_nhwc(B, Cout, 2 * H, 2 * W, x.dtype, x.device)
    st = torch.empty((B, tiles, Cout, 2), dtype=torch.float32, device=x.device)
    call('idf_upconv_bf16', _p(x), _p(w_sub_frag), _p(bias), _p(y), _p(st), B, H, W, C, Cout, _st())
    return y, st


# ------------------------------------------------- the attention block's proj conv folded into V
_ATTN_FOLD = knobs.flag('IDF_ATTN_FOLD')
# 1 = autograd hands the attention / objective nodes zero-filled gradients for their statistics / terms outputs (13 fill launches
# per step; the A/B switch of profiles/r04_conv_wr.txt)
_MATERIALIZE = False


def attn_res_tiles(qkv):
    """Statistics tiles per image of idf_attn_fwd_res for this q | k | v tensor; 0: not covered (or the fold is switched off)."""
    if not (_ATTN_FOLD and qkv.is_cuda and qkv.dtype == torch.bfloat16 and qkv.dim() == 4):
        return 0
    return int(_lib.load().idf_attn_res_tiles(qkv.shape[0], qkv.shape[2] * qkv.shape[3], qkv.shape[1] // 3, BF16))


class _AttentionRes(torch.autograd.Function):
    """y = x + softmax(q k^T C^-1/2) v' with the statistics partials of y (idf_attn_fwd_res): the attention block behind a
    q | k | v conv whose V weights already carry the proj conv (Wv' = Wp Wv).  The gradient of y goes to the attention backward
    as dO and, unchanged, to x (the alias the GroupNorm-prologue conv handed out: it joins that GroupNorm's backward)."""

    @staticmethod
    def forward(ctx, qkv, x, tiles, pre=None):
        qkv, x = _nhwc(qkv), _nhwc(x)
        ctx.set_materialize_grads(_MATERIALIZE)      # no zero-filled gradient for the statistics output
        if pre is not None:          # (y, st, o, lse) of the one-launch attention block: record the backward pass only
            y, st, o, lse = pre
            ctx.save_for_backward(qkv, lse, o)
            ctx.mark_non_differentiable(st)
            return y, st
        B, C, H, W = x.shape
        dev = x.device
        train = torch.is_grad_enabled() or qkv.requires_grad
        y = empty_nhwc(B, C, H, W, x.dtype, dev)
        st = torch.empty((B, tiles, C, 2), dtype=torch.float32, device=dev)
        o = empty_nhwc(B, C, H, W, x.dtype, dev) if train else None
        lse = torch.empty((B, H * W), dtype=torch.float32, device=dev) if train else None
        call('idf_attn_fwd_res', _p(qkv), _p(x), _p(o), _p(lse), _p(y), _p(st), B, H * W, C, float(int(C) ** (-0.5)), _st())
        if train:
            ctx.save_for_backward(qkv, lse, o)
        ctx.mark_non_differentiable(st)
        return y, st

    @staticmethod
    def backward(ctx, dy, _dst):
        if dy is None:
            return None, None, None, None
        qkv, lse, o = ctx.saved_tensors
        B, C3, H, W = qkv.shape
        C, N = C3 // 3, H * W
        dy = _nhwc(dy.to(qkv.dtype))
        dqkv = torch.empty_like(qkv, memory_format=CL)
        call('idf_attn_bwd_o', _p(qkv), _p(dy), _p(lse), _p(o), _p(dqkv), B, N, C, float(int(C) ** (-0.5)), _dt(qkv), _st())
        return dqkv, dy, None, None


def attention_res(qkv, x, tiles, pre=None):
    y, st = _AttentionRes.apply(qkv, x, tiles, pre)
    return _tag(y, st)


_FOLD_BWD_ROWS = []       # fix-ups of the backward pass in flight: (g, gb, wp, wv, bv, dwp, dbp, C) device addresses
_FOLD_BWD_TABLES = {}     # tuple of rows -> device table (addresses are fixed: gradient arena, parameters)
_FOLD_BWD_TASK = [-1, None]     # autograd graph task the rows belong to, their device


def _fold_bwd_run():
    """End of the backward pass (queued behind WgradBatch.flush): the gradients of the folded weights, which the q | k | v weight
    gradient left in proj_v's arena slots, become the gradients of proj and proj_v -- one launch for all blocks of the pass."""
    rows = tuple(_FOLD_BWD_ROWS)
    del _FOLD_BWD_ROWS[:]
    if not rows:
        return
    tab = _FOLD_BWD_TABLES.get(rows)
    if tab is None:
        import numpy as np
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError('attention fold: run one eager training step before graph capture')
        dt = np.dtype([('g', '<i8'), ('gb', '<i8'), ('wp', '<i8'), ('wv', '<i8'), ('bv', '<i8'), ('dwp', '<i8'), ('dbp', '<i8'),
                       ('gs', '<i8'), ('C', '<i4'), ('pad', '<i4')])
        scratch = torch.empty((sum(r[7] * r[7] + r[7] for r in rows),), dtype=torch.float32, device=_FOLD_BWD_TASK[1])
        offs, off = [], 0
        for r in rows:
            offs.append(scratch.data_ptr() + 4 * off)
            off += r[7] * r[7] + r[7]
        host = np.array([r[:7] + (o, r[7], 0) for r, o in zip(rows, offs)], dtype=dt)
        tab = (torch.from_numpy(host.view(np.uint8).reshape(len(rows), -1).copy()).to(scratch.device), scratch)
        _FOLD_BWD_TABLES[rows] = tab
    call('idf_attn_fold_bwd_batched', tab[0].data_ptr(), len(rows), max(r[7] for r in rows), _st())


class _FoldProjV(torch.autograd.Function):
    """Marks the q | k | v weight / bias concatenations as carrying the proj conv in their V third (the VALUES the kernels read
    are the folded buffers the block keeps; these tensors only route gradients).  Backward: the q | k | v weight gradient
    arrives for (Wq, Wk, Wv' = Wp Wv) and (bq, bk, b' = Wp bv + bp); the chain rule to proj and proj_v is one batched launch at
    the end of the backward pass when the gradients live in the arena (deferred weight gradients), else torch products."""

    @staticmethod
    def forward(ctx, wcat, bcat, wp, bp, wv, bv):
        ctx.save_for_backward(wp, wv, bv)
        ctx.slots = (slot_of(wp), slot_of(bp))
        return wcat.view_as(wcat), bcat.view_as(bcat)

    @staticmethod
    def backward(ctx, dw, db):
        wp, wv, bv = ctx.saved_tensors
        C = wp.shape[0]
        sw, sb = ctx.slots
        arena = sw.arena if sw is not None else None
        if (arena is not None and sb is not None and dw is not None and db is not None and arena.holds(dw) and arena.holds(db)
                and sw.available() and sb.available() and dw.is_contiguous() and db.is_contiguous()):
            gw, gbp = sw.take(), sb.take()
            g, gb = dw[2 * C:], db[2 * C:]
            # rows of a backward pass that did not end (the engine skips the final callbacks when a node raises) belong to no
            # launch: the first fix-up of every pass (graph task) starts the list and queues that pass's callback
            task = torch._C._current_graph_task_id()
            if task != _FOLD_BWD_TASK[0] or not _FOLD_BWD_ROWS:
                del _FOLD_BWD_ROWS[:]
                _FOLD_BWD_TASK[0], _FOLD_BWD_TASK[1] = task, wp.device
                torch.autograd.Variable._execution_engine.queue_callback(_fold_bwd_run)
            _FOLD_BWD_ROWS.append((g.data_ptr(), gb.data_ptr(), wp.data_ptr(), wv.data_ptr(), bv.data_ptr(), gw.data_ptr(),
                                   gbp.data_ptr(), C))
            return dw, db, gw, gbp, None, None
        # gradients that are ordinary tensors (no arena, or the weight gradient ran eagerly): the same kernel on a one-row table, now
        # (dW' / db' of the V third are rewritten in place into dWv / dbv -- on private copies: autograd may hold `dw` elsewhere)
        WgradBatch.flush()
        import numpy as np
        dw2, db2 = dw.float().contiguous().clone(), db.float().contiguous().clone()
        dwp = torch.empty(wp.shape, dtype=torch.float32, device=wp.device).contiguous(memory_format=torch.contiguous_format)
        dbp = torch.empty((C,), dtype=torch.float32, device=wp.device)
        dwp.zero_()
        dbp.zero_()
        scratch = torch.empty((C * C + C,), dtype=torch.float32, device=wp.device)
        dt = np.dtype([('g', '<i8'), ('gb', '<i8'), ('wp', '<i8'), ('wv', '<i8'), ('bv', '<i8'), ('dwp', '<i8'), ('dbp', '<i8'),
                       ('gs', '<i8'), ('C', '<i4'), ('pad', '<i4')])
        row = np.array([(dw2[2 * C:].data_ptr(), db2[2 * C:].data_ptr(), wp.data_ptr(), wv.data_ptr(), bv.data_ptr(), dwp.data_ptr(),
                         dbp.data_ptr(), scratch.data_ptr(), C, 0)], dtype=dt)
        tab = torch.from_numpy(row.view(np.uint8).reshape(1, -1).copy()).to(wp.device)
        call('idf_attn_fold_bwd_batched', tab.data_ptr(), 1, C, _st())
        ctx._keep = (tab, scratch)          # alive until the launch has run (stream order; freed with the node)
        return dw2.to(dw.dtype), db2.to(db.dtype), dwp.view_as(wp), dbp, None, None


def fold_proj_v(wcat, bcat, wp, bp, wv, bv):
    """(wcat, bcat) with the gradient routing of the folded V third (see _FoldProjV); a no-op without gradients."""
    if not (torch.is_grad_enabled() and wcat.requires_grad):
        return wcat, bcat
    grp_w, grp_b = getattr(wcat, '_idf_cat_group', None), getattr(bcat, '_idf_cat_group', None)
    w2, b2 = _FoldProjV.apply(wcat, bcat, wp, bp, wv, bv)
    if grp_w is not None:
        w2._idf_cat_group = grp_w
    if grp_b is not None:
        b2._idf_cat_group = grp_b
    return w2, b2


# ------------------------------------------------- the attention block in one launch
_ATTN_BLOCK = True
# One 4-wave workgroup per image: a win only once the batch alone fills the chip (DDIM-100 at B = 256: 306 -> 310.5 img/s).  At
# B = 32 the launch is 32 workgroups of one wave per SIMD -- every LDS and memory latency exposed -- and the block costs ~70 us
# against 30 us for the three per-op launches (train step 9.35 -> 9.74 ms with it at every batch; profiles/r04_attn_block.txt)
_ATTN_BLOCK_MINB = knobs.num('IDF_ATTN_BLOCK_MINB')


def attn_block_ok(x, policy=True):
    """The one-launch attention block (idf_attnblock_fwd) covers this input: bf16, 256 tokens, 128 channels -- and, with
    `policy`, the batch is one the launch pays at."""
    return bool(_ATTN_BLOCK and _ATTN_FOLD and x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4
                and (not policy or x.shape[0] >= _ATTN_BLOCK_MINB)
                and _lib.load().idf_attnblock_ok(x.shape[2] * x.shape[3], x.shape[1], BF16))


def attn_block_fwd_raw(x, xst, gn_w, gn_b, wqkv_frag, bqkv, train):
    """y = x + attention(q | k | v'), q | k | v' = conv1x1(GroupNorm(x)) (the proj conv folded into V) in one launch
    -> (y, st_y, qkv, h, o, lse, mean, rstd, sc, sh); everything after st_y is None unless `train` (what the backward reads)."""
    x = _nhwc(x)
    B, C, H, W = x.shape
    dev = x.device
    y = empty_nhwc(B, C, H, W, x.dtype, dev)
    st = torch.empty((B, 1, C, 2), dtype=torch.float32, device=dev)
    qkv = h = o = lse = mean = rstd = sc = sh = None
    if train:
        qkv = empty_nhwc(B, 3 * C, H, W, x.dtype, dev)
        h, o = empty_nhwc(B, C, H, W, x.dtype, dev), empty_nhwc(B, C, H, W, x.dtype, dev)
        lse = torch.empty((B, H * W), dtype=torch.float32, device=dev)
        mean, rstd = (torch.empty((B, 32), dtype=torch.float32, device=dev) for _ in range(2))
        sc, sh = (torch.empty((B, C), dtype=torch.float32, device=dev) for _ in range(2))
    call('idf_attnblock_fwd', _p(x), _p(xst), xst.shape[1], _p(gn_w), _p(gn_b), GN_EPS, _p(wqkv_frag), _p(bqkv), _p(y), _p(st),
         _p(qkv), _p(h), _p(o), _p(lse), _p(mean), _p(rstd), _p(sc), _p(sh), float(int(C) ** (-0.5)), B, H * W, C, _st())
    return y, st, qkv, h, o, lse, mean, rstd, sc, sh


# ------------------------------------------------- concatenated parameter views
class _CatParams(torch.autograd.Function):
    """cat(params, dim 0) for a ParamGroup whose storage is adjacent: the forward is a view, the
    backward hands each parameter its arena slot when the consumer wrote the gradient of the
    concatenation straight into the group's arena region (else plain row slices)."""

    @staticmethod
    def forward(ctx, group, *params):
        ctx.group = group
        return group.cat()

    @staticmethod
    def backward(ctx, dcat):
        grp = ctx.group
        slots = [slot_of(q) for q in grp.params]
        if all(s is not None for s in slots):
            from .grad_arena import _GroupSlot
            parts = _GroupSlot(grp, slots).split(dcat)
            if parts is not None:
                return (None,) + tuple(parts)
        return (None,) + tuple(dcat.split(grp.rows, dim=0))


def cat_params(group):
    """The concatenation of a ParamGroup's parameters as an autograd-tracked view."""
    group.ensure()
    if not (torch.is_grad_enabled() and any(p.requires_grad for p in group.params)):
        return group.cat()
    out = _CatParams.apply(group, *group.params)
    out._idf_cat_group = group
    return out


# --------------------------------------------------------------------- linear
class _Linear(torch.autograd.Function):
    """y = silu?(x) @ W^T + b, fp32 [B, K] -> [B, N]."""

    @staticmethod
    def forward(ctx, x, weight, bias, silu_in, slots=None):
        x = _f32c(x)
        if not x.is_cuda:
            raise RuntimeError('infodiffusion_amd kernels run on the GPU only')
        w = _f32c(weight)
        xs = x
        if silu_in:
            xs = torch.empty_like(x)
            call('idf_silu_fwd', _p(x), _p(xs), x.numel(), _st())
        Bn, K = x.shape
        N = w.shape[0]
        # a handful of output tiles over a long contraction (encoder fc_a: 32 x 32 over K = 4096): the K range
        # is cut into `sk` batch entries whose partial products are summed in a fixed order -- no atomics,
        # the forward pass stays bit-reproducible
        tiles = -(-Bn // 64) * -(-N // 64)
        sk = 32 if (tiles < 8 and K >= 2048 and K % (32 * 64) == 0) else 1
        if sk > 1:
            part = torch.empty((sk, Bn * N), dtype=torch.float32, device=x.device)
            bgemm_raw(xs, 0, w, 0, part, 0, None, sk, K // sk, K // sk, Bn * N, K, K, N, Bn, N, K // sk, 0, 0, dtype=F32)
            y = colsum_raw(part).view(Bn, N)
            if bias is not None:
                y += _f32c(bias)
        else:
            y = torch.empty((Bn, N), dtype=torch.float32, device=x.device)
            bgemm_raw(xs, 0, w, 0, y, 0, _f32c(bias), 1, 0, 0, 0, K, K, N, Bn, N, K, 0, 0, dtype=F32)
        ctx.silu_in, ctx.slots = silu_in, slots or (None, None)
        ctx.save_for_backward(x, xs, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, xs, w = ctx.saved_tensors
        dy = _f32c(dy)
        Bn, K = x.shape
        N = w.shape[0]
        need = ctx.needs_input_grad
        dx = dW = db = None
        if need[0]:
            # few output tiles but a long contraction (batched FiLM: N ~ 5k): the contraction is cut into `sk`
            # batch entries and the partial products summed in a fixed order (bit-reproducible, no atomics)
            tiles = -(-Bn // 64) * -(-K // 64)
            sk = max(1, min(32, N // 256)) if tiles < 64 else 1
            while sk > 1 and N % sk:
                sk -= 1
            if sk > 1:
                part = torch.empty((sk, Bn * K), dtype=torch.float32, device=x.device)
                bgemm_raw(dy, 0, w, 0, part, 0, None, sk, N // sk, (N // sk) * K, Bn * K, N, K, K, Bn, K, N // sk, 0, 1,
                          out_f32=True, dtype=F32)
                dxs = colsum_raw(part).view(Bn, K)
            else:
                dxs = torch.empty_like(x)
                bgemm_raw(dy, 0, w, 0, dxs, 0, None, 1, 0, 0, 0, N, K, K, Bn, K, N, 0, 1, out_f32=True, dtype=F32)
            if ctx.silu_in:
                dx = torch.empty_like(x)
                call('idf_silu_bwd', _p(x), _p(dxs), _p(dx), x.numel(), _st())
            else:
                dx = dxs
        ws, bs = ctx.slots
        if need[1]:
            # straight into the parameter's (or the parameter group's) gradient-arena slot when it is free
            if ws is not None and ws.available() and tuple(ws.view.shape) == tuple(w.shape) and ws.view.is_contiguous():
                dW = ws.take()
            else:
                dW = torch.empty_like(w)
            bgemm_raw(dy, 0, xs, 0, dW, 0, None, 1, 0, 0, 0, N, K, K, N, K, Bn, 1, 1, dtype=F32)
        if need[2]:
            dbv = bs.take() if (bs is not None and bs.available() and tuple(bs.view.shape) == (N,)) else None
            db = colsum_raw(dy, out=dbv)
        return dx, dW, db, None, None


def linear(x, weight, bias=None, silu_in=False):
    slots = (slot_of(weight), slot_of(bias)) if torch.is_grad_enabled() else None
    return _Linear.apply(x, weight, bias, silu_in, slots)


# ------------------------------------------------- conditioning path (time / latent embedding -> FiLM)
_TEMB_FUSED = knobs.flag('IDF_TEMB_FUSED')


def _grad_dst(slot, like):
    """Where a parameter gradient is written: the parameter's gradient-arena slot when it is free, else a new tensor."""
    if slot is not None and slot.available() and tuple(slot.view.shape) == tuple(like.shape) and slot.view.is_contiguous():
        return slot.take()
    return torch.empty(like.shape, dtype=torch.float32, device=like.device)


class _TembFilm(torch.autograd.Function):
    """film_t = Linear_t(SiLU(temb)), film_a = Linear_a(SiLU(aemb)) with temb = TimeEmbedding(t) (modules.py:9-38) and
    aemb = fc_a(a) (models.py:298-301) -- the whole conditioning path through idf_temb_film_fwd / _bwd (three launches
    forward, four backward, instead of one per product, SiLU, bias gradient and split-K reduction)."""

    @staticmethod
    def forward(ctx, t, a, table, W1, b1, W2, b2, Wfc, bfc, Wt, bt, Wa, ba, fc_silu, slots):
        B = t.shape[0]
        dev = table.device
        t = t.contiguous()
        has_a = a is not None
        par = [_f32c(v) if v is not None else None for v in (table, W1, b1, W2, b2, Wfc, bfc, Wt, bt, Wa, ba)]
        table, W1, b1, W2, b2, Wfc, bfc, Wt, bt, Wa, ba = par
        a = _f32c(a) if has_a else None
        dim, d_model = W1.shape
        Nt = Wt.shape[0]
        Na = Wa.shape[0] if (has_a and Wa is not None) else 0
        a_dim = a.shape[1] if has_a else 0
        keep = torch.empty((6, B, dim), dtype=torch.float32, device=dev)        # h1, s1, temb, st, aemb, sa
        film_t = torch.empty((B, Nt), dtype=torch.float32, device=dev)
        film_a = torch.empty((B, Na), dtype=torch.float32, device=dev) if Na else None
        call('idf_temb_film_fwd', _p(t), _p(table), d_model, _p(W1), _p(b1), _p(W2), _p(b2), dim, _p(a), a_dim, _p(Wfc),
             _p(bfc), int(bool(fc_silu)), _p(Wt), _p(bt), Nt, _p(Wa), _p(ba), Na, _p(keep[0]), _p(keep[1]), _p(keep[2]),
             _p(keep[3]), _p(keep[4]), _p(keep[5]), _p(film_t), _p(film_a), B, _st())
        ctx.fc_silu, ctx.slots, ctx.has_a, ctx.Na = bool(fc_silu), slots or (None,) * 10, has_a, Na
        ctx.save_for_backward(t, a, table, W1, W2, Wfc, Wt, Wa, keep)
        return film_t, film_a

    @staticmethod
    def backward(ctx, dft, dfa):
        t, a, table, W1, W2, Wfc, Wt, Wa, keep = ctx.saved_tensors
        need = ctx.needs_input_grad         # t, a, table, W1, b1, W2, b2, Wfc, bfc, Wt, bt, Wa, ba, fc_silu, slots
        B, dev = t.shape[0], table.device
        dim, d_model = W1.shape
        Nt = Wt.shape[0]
        has_a = ctx.has_a and ctx.Na > 0
        dft = _f32c(dft) if dft is not None else torch.zeros((B, Nt), dtype=torch.float32, device=dev)
        if has_a:
            dfa = _f32c(dfa) if dfa is not None else torch.zeros((B, ctx.Na), dtype=torch.float32, device=dev)
        sl = ctx.slots            # W1, b1, W2, b2, Wfc, bfc, Wt, bt, Wa, ba
        shapes = (W1, W1[:, 0], W2, W2[:, 0], Wfc, Wfc[:, 0] if Wfc is not None else None, Wt, Wt[:, 0], Wa,
                  Wa[:, 0] if Wa is not None else None)
        g = [None] * 10
        for i in range(10):
            live = i < 4 or (i in (6, 7)) or has_a
            if need[3 + i] and live and shapes[i] is not None:
                g[i] = _grad_dst(sl[i], shapes[i])
        da = torch.empty_like(a) if (has_a and need[1]) else None
        lib = _lib.load()
        parts = lib.idf_temb_film_parts(Nt) + (lib.idf_temb_film_parts(ctx.Na) if has_a else 0)
        scratch = torch.empty((parts + 1, B, dim), dtype=torch.float32, device=dev)      # K slices of dS | dh1s
        dS, dh1s = scratch[:parts], scratch[parts]
        call('idf_temb_film_bwd', _p(dft), _p(dfa) if has_a else None, _p(t), _p(table), d_model, _p(W2), dim, _p(a),
             a.shape[1] if a is not None else 0, _p(Wfc), int(ctx.fc_silu), _p(Wt), Nt, _p(Wa), ctx.Na, _p(keep[0]),
             _p(keep[1]), _p(keep[2]), _p(keep[3]), _p(keep[4]), _p(keep[5]), _p(dS), _p(dh1s), _p(g[6]), _p(g[7]), _p(g[8]),
             _p(g[9]), _p(g[2]), _p(g[3]), _p(g[0]), _p(g[1]), _p(g[4]), _p(g[5]), _p(da), B, _st())
        if ctx.has_a and not has_a and need[1]:
            da = torch.zeros_like(a)
        return (None, da, None, g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7], g[8], g[9], None, None)


def temb_film_ok(table, *tensors):
    return _TEMB_FUSED and table.is_cuda and all(v is None or v.dtype == torch.float32 for v in tensors)


def temb_film(t, table, l1, l2, film_t, a=None, fc=None, fc_silu=False, film_a=None):
    """(film_t [B, Nt], film_a [B, Na] or None).  l1, l2, fc: nn.Linear modules; film_t / film_a: (weight, bias) of the
    concatenated FiLM projections (cat_params views of the blocks' ParamGroups)."""
    ps = [l1.weight, l1.bias, l2.weight, l2.bias, fc.weight if fc is not None else None,
          fc.bias if fc is not None else None, film_t[0], film_t[1], film_a[0] if film_a is not None else None,
          film_a[1] if film_a is not None else None]
    slots = tuple(slot_of(v) for v in ps) if torch.is_grad_enabled() else None
    return _TembFilm.apply(t, a, table, *ps, fc_silu, slots)


# ------------------------------------------------------------ input pipeline
def prep_u8(img_u8, flip=None):
    """uint8 NHWC image batch [B, H, W, C] on the GPU -> fp32 activations (x / 255 - 0.5) / 0.5, logical
    [B, C, H, W] in NHWC-dense memory (what q_sample and the encoder read without a layout copy);
    flip: optional uint8 [B], non-zero = mirror horizontally (reference data.py:149-171)."""
    if not img_u8.is_cuda or img_u8.dtype != torch.uint8 or img_u8.dim() != 4:
        raise RuntimeError('prep_u8 wants a uint8 [B, H, W, C] tensor on the GPU')
    img_u8 = img_u8.contiguous()
    B, H, W, Cc = img_u8.shape
    out = torch.empty((B, Cc, H, W), dtype=torch.float32, device=img_u8.device).contiguous(memory_format=CL)
    if flip is not None:
        flip = flip.to(device=img_u8.device, dtype=torch.uint8).contiguous()
    call('idf_prep_u8', _p(img_u8), _p(flip), _p(out), B, H, W, Cc, _st())
    return out


# --------------------------------------------------------- gather / q_sample
def gather_rows(table, idx):
    idx = idx.contiguous()
    out = torch.empty((idx.numel(), table.shape[1]), dtype=torch.float32, device=table.device)
    call('idf_gather_rows', _p(table), _p(idx), _p(out), idx.numel(), table.shape[1], _st())
    return out


def qsample_tables(alpha_bars):
    """sqrt(ab), sqrt(1-ab) by the reference's torch CPU ops (models.py:704), on ab's device."""
    ab = alpha_bars.detach().cpu()
    return torch.sqrt(ab).to(alpha_bars.device), torch.sqrt(1 - ab).to(alpha_bars.device)


def q_sample(x, eps, idx, tables, act_dtype):
    """models.py:702-704.  x, eps fp32; tables = qsample_tables(alpha_bars).  Returns x_tilde
    in act_dtype (the fp32 result is bit-identical to the reference's CPU path)."""
    x, eps = _nhwc(x.float()), _nhwc(eps.float())
    xt = torch.empty_like(x, dtype=act_dtype, memory_format=CL)
    per = x.numel() // x.shape[0]
    call('idf_qsample', _p(x), _p(eps), _p(idx.contiguous()), _p(tables[0]), _p(tables[1]), None, _p(xt), per,
         x.numel(), F32 if act_dtype == torch.float32 else BF16, _st())
    return xt


# ----------------------------------------------------------------------- loss
class _DiffLoss(torch.autograd.Function):
    """(mean((out-eps)^2), mean((x0-x)^2)/T) with x0 from the t=0 constants (models.py:640-646)."""

    @staticmethod
    def forward(ctx, out, eps, x, c0, c1, inv_T):
        out, eps, x = _nhwc(out), _nhwc(eps), _nhwc(x)
        res = torch.empty((2,), dtype=torch.float32, device=out.device)
        ws = torch.empty((2048,), dtype=torch.float32, device=out.device)
        call('idf_loss_fwd', _p(out), _p(eps), _p(x), c0, c1, inv_T, _p(res), _p(ws), out.numel(), _dt(out), _st())
        ctx.k = (c0, c1, inv_T)
        ctx.save_for_backward(out, eps, x)
        return res

    @staticmethod
    def backward(ctx, g):
        out, eps, x = ctx.saved_tensors
        c0, c1, inv_T = ctx.k
        dout = torch.empty_like(out, memory_format=CL)
        call('idf_loss_bwd', _p(out), _p(eps), _p(x), c0, c1, inv_T, _p(_f32c(g)), 1, _p(dout), out.numel(),
             _dt(out), _st())
        return dout, None, None, None, None, None


def diff_loss(out, eps, x, c0, c1, inv_T):
    return _DiffLoss.apply(out, eps, x, float(c0), float(c1), float(inv_T))


class _MMD(torch.autograd.Function):
    """utils.py:85-90; gradient w.r.t. y only (x = prior samples)."""

    @staticmethod
    def forward(ctx, x, y):
        x, y = _f32c(x), _f32c(y)
        n, D = x.shape
        m = y.shape[0]
        out = torch.empty((1,), dtype=torch.float32, device=y.device)
        ws = torch.empty((2 * n + m,), dtype=torch.float32, device=y.device)
        call('idf_mmd_fwd', _p(x), _p(y), n, m, D, _p(out), _p(ws), _st())
        ctx.save_for_backward(x, y)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        n, D = x.shape
        m = y.shape[0]
        dy = torch.empty_like(y)
        call('idf_mmd_bwd', _p(x), _p(y), n, m, D, _p(_f32c(g).reshape(1)), 1.0, _p(dy), _st())
        return None, dy


def mmd(x, y):
    return _MMD.apply(x, y)


class _Objective(torch.autograd.Function):
    """InfoDiff's --mmd_weight objective as ONE scalar (models.py:640-646, 674-678):
        total = (mean((out-eps)^2) + mean((x0-x)^2)/T) + w * MMD(prior, lat)
    -- loss partials, MMD row sums and one tail block forward (3 launches instead of the 7 of diff_loss + the scalar adds +
    compute_mmd + the weighted add); backward one node: d out (idf_loss_bwd) and d lat (idf_mmd_bwd) from the upstream scalar.
    Outputs: total, terms [3] = (denoise, recon, mmd) (not differentiable: for --verbose_loss)."""

    @staticmethod
    def forward(ctx, out, eps, x, prior, lat, c0, c1, inv_T, w):
        out, eps, x = _nhwc(out), _nhwc(eps), _nhwc(x)
        prior, lat = _f32c(prior), _f32c(lat)
        n, D = prior.shape
        m = lat.shape[0]
        res = torch.empty((4,), dtype=torch.float32, device=out.device)
        ws = torch.empty((2048 + 2 * n + m,), dtype=torch.float32, device=out.device)
        ctx.set_materialize_grads(_MATERIALIZE)      # no zero-filled gradient for the (not differentiable) terms
        call('idf_objective_fwd', _p(out), _p(eps), _p(x), c0, c1, inv_T, _p(prior), _p(lat), n, m, D, w, _p(res), _p(ws),
             out.numel(), _dt(out), _st())
        ctx.k = (c0, c1, inv_T, w)
        ctx.save_for_backward(out, eps, x, prior, lat)
        terms = res[:3]
        ctx.mark_non_differentiable(terms)
        return res[3], terms

    @staticmethod
    def backward(ctx, g, _):
        if g is None:
            return (None,) * 9
        out, eps, x, prior, lat = ctx.saved_tensors
        c0, c1, inv_T, w = ctx.k
        g = _f32c(g).reshape(1)
        dout = dlat = None
        if ctx.needs_input_grad[0]:
            dout = torch.empty_like(out, memory_format=CL)
            call('idf_loss_bwd', _p(out), _p(eps), _p(x), c0, c1, inv_T, _p(g), 0, _p(dout), out.numel(), _dt(out), _st())
        if ctx.needs_input_grad[4]:
            n, D = prior.shape
            dlat = torch.empty_like(lat)
            call('idf_mmd_bwd', _p(prior), _p(lat), n, lat.shape[0], D, _p(g), w, _p(dlat), _st())
        return dout, None, None, None, dlat, None, None, None, None


def objective_mmd(out, eps, x, prior, lat, c0, c1, inv_T, w):
    return _Objective.apply(out, eps, x, prior, lat, float(c0), float(c1), float(inv_T), float(w))


# ------------------------------------------------------------------- sampler
def sampler_coef_table(betas, alphas, alpha_bars, alpha_prev_bars, eta=0.01):
    """[3][T][8] per-step scalars, evaluated with the reference's own 0-dim fp32 torch
    expressions on the CPU (sampling.py:30,35 / 52,57-58 / 71-72) => bitwise-equal scalars."""
    b, al, ab, apb = [t.detach().cpu() for t in (betas, alphas, alpha_bars, alpha_prev_bars)]
    T = len(ab)
    tab = torch.zeros((3, T, 8), dtype=torch.float32)
    for idx in range(T):
        tab[0, idx, 0] = torch.sqrt(1 / al[idx])
        tab[0, idx, 1] = b[idx] / torch.sqrt(1 - ab[idx])
        tab[0, idx, 2] = torch.sqrt((1 - apb[idx]) / (1 - ab[idx]) * b[idx])
        tab[1, idx, 0] = tab[2, idx, 0] = torch.sqrt(1 - apb[idx])
        tab[1, idx, 1] = tab[2, idx, 1] = torch.sqrt(apb[idx])
        if idx > 0:
            sigma = eta * torch.sqrt((1 - apb[idx - 1]) / (1 - ab[idx - 1])) * torch.sqrt(b[idx - 1])
            tab[1, idx, 4] = torch.sqrt(apb[idx - 1])
            tab[1, idx, 5] = torch.sqrt(1 - apb[idx - 1] - sigma ** 2)
            tab[1, idx, 6] = sigma
        if idx + 1 < T:
            tab[2, idx, 2] = torch.sqrt(apb[idx + 1])
            tab[2, idx, 3] = torch.sqrt(1 - apb[idx + 1])
    return tab.to(alpha_bars.device)


def sampler_step(x, eps_hat, noise, idx_t, coef, mode, out_t_dtype=None):
    """x fp32 state; eps_hat in its own dtype; idx_t: 1-element int64 CUDA tensor;
    coef = sampler_coef_table(...)[mode].  Returns (x_next fp32, copy in out_t_dtype or None)."""
    xo = torch.empty_like(x)
    xo_t = torch.empty_like(x, dtype=out_t_dtype) if (out_t_dtype is not None and out_t_dtype != torch.float32) else None
    call('idf_sampler_step', _p(x), _p(eps_hat), _p(noise), _p(xo), _p(xo_t), _p(idx_t), _p(coef), mode,
         x.numel(), _dt(eps_hat), _st())
    return xo, xo_t


def dropout_mask(seed, salt, p, numel):
    m = torch.empty((numel,), dtype=torch.float32, device=seed.device)
    call('idf_dropout_mask', _p(seed), salt, float(p), _p(m), numel, _st())
    return m
