"""GPU parity tests of the raw HIP kernels (through the C ABI) against plain
PyTorch fp32 references computed on the host."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from infodiffusion_amd import ops  # noqa: E402

CL = torch.channels_last
DEV = 'cuda'


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def rnd(seed, *shape):
    g = torch.Generator(device='cpu')
    g.manual_seed(seed)
    return torch.randn(*shape, generator=g)


TOL = {torch.float32: 2e-5, torch.bfloat16: 2e-2}


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('case', [
    # (B, Cin, H, W, Cout, taps, mode)
    (2, 64, 16, 16, 64, 9, ops.S1), (3, 128, 8, 8, 128, 9, ops.S1), (2, 192, 8, 8, 64, 9, ops.S1),
    (2, 64, 16, 16, 64, 9, ops.S2), (2, 64, 8, 8, 64, 9, ops.UP2), (2, 128, 8, 8, 384, 1, ops.S1),
    (2, 3, 16, 16, 64, 9, ops.S1), (2, 64, 16, 16, 3, 9, ops.S1), (2, 1, 32, 32, 32, 9, ops.S1),
    (2, 32, 8, 8, 1, 9, ops.S1), (5, 96, 4, 4, 32, 9, ops.S1), (2, 256, 32, 32, 128, 9, ops.S1),
    # 1x1 through the halo pipeline (bf16): 64 / 128 / 256-pixel tiles, ragged cout tile, tiny images
    (3, 128, 16, 16, 128, 1, ops.S1), (33, 64, 64, 64, 128, 1, ops.S1), (5, 192, 32, 32, 64, 1, ops.S1),
    (2, 64, 4, 4, 72, 1, ops.S1), (40, 128, 16, 16, 384, 1, ops.S1),
    # >= 1536 tiles: the direct-to-LDS variant (global_load_lds, two blocks per CU), 3x3 and 1x1
    (96, 64, 64, 64, 64, 9, ops.S1), (48, 128, 64, 64, 128, 1, ops.S1),
    # stride 2 through the halo kernel: 32 / 16 / 4-wide outputs
    (3, 64, 64, 64, 64, 9, ops.S2), (2, 128, 32, 32, 128, 9, ops.S2), (2, 128, 8, 8, 128, 9, ops.S2),
])
def test_conv_fwd_dgrad_wgrad(case, dtype):
    B, Cin, H, W, Cout, taps, mode = case
    k = 3 if taps == 9 else 1
    x = rnd(1, B, Cin, H, W)
    w = rnd(2, Cout, Cin, k, k) / (Cin * taps) ** 0.5
    b = rnd(3, Cout)
    xq = x.to(dtype).float()
    wq = w.to(dtype).float()
    xin = F.interpolate(xq, scale_factor=2.0, mode='nearest') if mode == ops.UP2 else xq
    xin = xin.clone().requires_grad_(True)
    wr = wq.clone().requires_grad_(True)
    ref = F.conv2d(xin, wr, b, stride=2 if mode == ops.S2 else 1, padding=k // 2)
    res = rnd(4, *ref.shape).to(dtype).float()
    xd = x.to(DEV).to(dtype).contiguous(memory_format=CL)
    wf, wd = ops.pack_weight(w.to(DEV), dtype, True, True)
    y = ops.conv_raw(xd, wf, b.to(DEV), res.to(DEV).to(dtype).contiguous(memory_format=CL), None, None, None, 0,
                     0.0, mode, taps, 0, Cout)
    assert y.shape == ref.shape
    assert rel(y, ref + res) < TOL[dtype]
    gy = rnd(5, *ref.shape).to(dtype).float()
    ref.backward(gy)
    gyd = gy.to(DEV).to(dtype).contiguous(memory_format=CL)
    dx = ops.conv_dgrad_raw(gyd, wd, mode, taps, x.shape)
    gx_ref = xin.grad
    if mode == ops.UP2:
        gx_ref = gx_ref.view(B, Cin, H, 2, W, 2).sum(dim=(3, 5))
    assert rel(dx, gx_ref) < TOL[dtype] * (2 if dtype == torch.bfloat16 else 1)
    dW = ops.conv_wgrad_raw(xd, gyd, None, None, None, 0, 0.0, mode, taps, 0)
    assert dW.shape == w.shape
    assert rel(dW, wr.grad) < 5e-5


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('C,H', [(64, 16), (192, 8), (128, 4), (32, 32), (256, 8)])
def test_gn_film_silu_conv_prologue(C, H, dtype):
    """GroupNorm + FiLM(t) + FiLM(a) + SiLU folded into the conv staging, fwd + bwd."""
    B, Cout = 3, 64
    x = (rnd(1, B, C, H, H) * 1.5 + 0.3).to(dtype).float()
    gam, bet = 1 + 0.1 * rnd(2, C), 0.1 * rnd(3, C)
    ft, fa = 0.3 * rnd(4, B, 2 * C), 0.3 * rnd(5, B, 2 * C)
    w = (rnd(6, Cout, C, 3, 3) / (9 * C) ** 0.5).to(dtype).float()
    leaves = [t.clone().requires_grad_(True) for t in (x, gam, bet, ft, fa, w)]
    xr, gr, br, ftr, far, wr = leaves
    h = F.group_norm(xr, 32, gr, br, 1e-5)
    st, bt = torch.chunk(ftr[:, :, None, None], 2, dim=1)
    h = h * (1 + st) + bt
    sa, ba = torch.chunk(far[:, :, None, None], 2, dim=1)
    h = h * (1 + sa) + ba
    ref = F.conv2d(F.silu(h), wr, None, padding=1)
    gy = rnd(7, *ref.shape).to(dtype).float()
    ref.backward(gy)

    xd = x.to(DEV).to(dtype).contiguous(memory_format=CL)
    g = lambda t: t.to(DEV).contiguous()
    mean, rstd, sc, sh = ops.gn_coef_fwd_raw(xd, g(gam), g(bet), g(ft), g(fa))
    wf, wd = ops.pack_weight(w.to(DEV), dtype, True, True)
    y = ops.conv_raw(xd, wf, None, None, sc, sh, None, 0, 0.0, ops.S1, 9, 2, Cout)
    tol = TOL[dtype]
    assert rel(y, ref) < tol
    gyd = gy.to(DEV).to(dtype).contiguous(memory_format=CL)
    dA = ops.conv_dgrad_raw(gyd, wd, ops.S1, 9, x.shape)
    dx, dgam, dbet, dft, dfa = ops.gn_coef_bwd_raw(dA, xd, None, g(gam), g(bet), g(ft), g(fa), mean, rstd, sc, sh,
                                                  None, 0, 0.0, 2)
    btol = 3e-4 if dtype == torch.float32 else 4e-2
    assert rel(dx, xr.grad) < btol
    assert rel(dgam, gr.grad) < btol
    assert rel(dbet, br.grad) < btol
    assert rel(dft, ftr.grad) < btol
    assert rel(dfa, far.grad) < btol
    dW = ops.conv_wgrad_raw(xd, gyd, sc, sh, None, 0, 0.0, ops.S1, 9, 2)
    assert rel(dW, wr.grad) < (1e-4 if dtype == torch.float32 else 2e-2)


def test_dropout_mask_consistency():
    """The conv prologue, the wgrad recompute and the GN backward all apply the
    mask `idf_dropout_mask` exports; keep-rate matches p."""
    B, C, H, Cout = 2, 64, 8, 64
    dtype = torch.float32
    x = rnd(1, B, C, H, H)
    w = rnd(6, Cout, C, 3, 3) / (9 * C) ** 0.5
    seed = torch.tensor([123456789], dtype=torch.int64, device=DEV)
    xd = x.to(DEV).contiguous(memory_format=CL)
    ones = torch.ones(C, device=DEV)
    zeros = torch.zeros(C, device=DEV)
    mean, rstd, sc, sh = ops.gn_coef_fwd_raw(xd, ones, zeros, None, None)
    mask = ops.dropout_mask(seed, 7, 0.1, x.numel()).view(B, H, H, C).permute(0, 3, 1, 2).cpu()
    keep = float((mask > 0).float().mean())
    assert abs(keep - 0.9) < 0.02
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    ref = F.conv2d(F.silu(F.group_norm(xr, 32, None, None, 1e-5)) * mask, wr, None, padding=1)
    wf, wd = ops.pack_weight(w.to(DEV), dtype, True, True)
    y = ops.conv_raw(xd, wf, None, None, sc, sh, seed, 7, 0.1, ops.S1, 9, 2, Cout)
    assert rel(y, ref) < 2e-5
    gy = rnd(7, *ref.shape)
    ref.backward(gy)
    gyd = gy.to(DEV).contiguous(memory_format=CL)
    dA = ops.conv_dgrad_raw(gyd, wd, ops.S1, 9, x.shape)
    dx = ops.gn_coef_bwd_raw(dA, xd, None, ones, zeros, None, None, mean, rstd, sc, sh, seed, 7, 0.1, 2)[0]
    assert rel(dx, xr.grad) < 3e-4
    dW = ops.conv_wgrad_raw(xd, gyd, sc, sh, seed, 7, 0.1, ops.S1, 9, 2)
    assert rel(dW, wr.grad) < 1e-4


def _gn_act_conv_reference(x, gam, bet, ft, fa, act, mask, w, dy, extra=()):
    """fp32 PyTorch autograd of  y = conv(dropout(act(FiLM_a(FiLM_t(GroupNorm(x))))))  (modules.py:312-320, 283-288),
    backward from dy: -> (dx + sum(extra), dgamma, dbeta, dFiLM_t, dFiLM_a).  x / dy / w hold bf16-representable values."""
    C = x.shape[1]
    leaves = [t.detach().clone().float().requires_grad_(True) if t is not None else None for t in (x, gam, bet, ft, fa)]
    xr, gr, br, ftr, far = leaves
    u = F.group_norm(xr, 32, gr, br, eps=1e-5)
    if ftr is not None:
        u = u * (1 + ftr[:, :C, None, None]) + ftr[:, C:, None, None]
    if far is not None:
        u = u * (1 + far[:, :C, None, None]) + far[:, C:, None, None]
    if act == 2:
        u = F.silu(u)
        if mask is not None:
            u = u * mask
    y = F.conv2d(u, w.float(), None, padding=w.shape[-1] // 2)
    y.backward(dy.float())
    dx = xr.grad
    for e in extra:
        dx = dx + e.float()
    return [dx, gr.grad, br.grad, ftr.grad if ftr is not None else None, far.grad if far is not None else None]


@pytest.mark.parametrize('case', [
    # (B, Cin of dy, C of x, H, W, taps, act, film, p_drop, n_res)
    (3, 128, 128, 16, 16, 9, 2, True, 0.1, 1), (2, 256, 128, 8, 8, 9, 2, True, 0.1, 2), (33, 128, 128, 16, 16, 9, 2, False, 0.0, 0),
    (2, 384, 128, 16, 16, 1, 1, False, 0.0, 1), (3, 128, 256, 8, 8, 9, 2, True, 0.0, 1), (2, 64, 64, 4, 4, 9, 2, False, 0.1, 0),
    (2, 128, 64, 8, 16, 9, 2, True, 0.1, 1),
])
def test_dgrad_conv_with_groupnorm_backward_epilogue(case):
    """idf_conv_dgrad_gn_bf16 (data-gradient conv whose epilogue is the GroupNorm / FiLM / SiLU / dropout backward) against
    fp32 PyTorch autograd of conv(dropout(act(FiLM(GroupNorm(x))))) with the product's dropout mask: dx (+ branch
    gradients), dgamma, dbeta, dFiLM_t, dFiLM_a <= 4e-2 (bf16 operands).  The two-launch path it replaces is held to
    the same reference."""
    B, Cin, C, H, W, taps, act, film, p_drop, n_res = case
    k = 3 if taps == 9 else 1
    x = (0.3 + rnd(1, B, C, H, W)).to(DEV).bfloat16().contiguous(memory_format=CL)
    dy = rnd(2, B, Cin, H, W).to(DEV).bfloat16().contiguous(memory_format=CL)
    wgt = (rnd(3, Cin, C, k, k) / (C * taps) ** 0.5).to(DEV).bfloat16().float()   # forward conv C -> Cin; its data gradient maps dy -> dA
    _, wd = ops.pack_weight(wgt, torch.bfloat16, True, True)
    gam, bet = (1 + 0.1 * rnd(4, C)).to(DEV), (0.1 * rnd(5, C)).to(DEV)
    ft = (0.2 * rnd(6, B, 2 * C)).to(DEV) if film else None
    fa = (0.2 * rnd(7, B, 2 * C)).to(DEV) if film else None
    seed = torch.tensor([987654321], dtype=torch.int64, device=DEV) if p_drop else None
    res = [rnd(8 + i, B, C, H, W).to(DEV).bfloat16().contiguous(memory_format=CL) for i in range(n_res)]
    dres, dres2 = (res + [None, None])[:2]
    mask = ops.dropout_mask(seed, 5, p_drop, x.numel()).view(B, H, W, C).permute(0, 3, 1, 2) if p_drop else None
    want = _gn_act_conv_reference(x, gam, bet, ft, fa, act, mask, wgt, dy, res)
    _, mean, rstd, sc, sh = ops.gn_fused_fwd_raw(x, gam, bet, ft, fa, seed, 5, p_drop, act)
    assert ops.conv_dgrad_gn_ok(dy, x, ops.S1, taps, advice=False)          # coverage, not the policy
    got = ops.conv_dgrad_gn_raw(dy, wd, x, gam, bet, ft, fa, mean, rstd, sc, sh, seed, 5, p_drop, act, taps, dres=dres,
                                dres2=dres2)
    dA = ops.conv_dgrad_raw(dy, wd, ops.S1, taps, x.shape)
    two = ops.gn_fused_bwd_raw(dA, x, gam, bet, ft, fa, mean, rstd, sc, sh, seed, 5, p_drop, act, dres=dres, dres2=dres2)
    names = ('dx', 'dgamma', 'dbeta', 'dfilm_t', 'dfilm_a')
    for path, outs in (('fused', got), ('two-launch', two)):
        for nm, g, r in zip(names, outs, want):
            assert (g is None) == (r is None), (path, nm)
            if g is not None:
                # the FiLM gradients are [B, 2C]: compared half by half (scale | shift have different magnitudes)
                parts = [(g, r)] if g.dim() != 2 else [(g[:, :C], r[:, :C]), (g[:, C:], r[:, C:])]
                for gg, rr in parts:
                    assert rel(gg, rr) < 4e-2, (path, nm, rel(gg, rr))
    # exactness where rounding plays no part: the accumulate-into-slots form gives the same sums as the per-sample form
    class Slot:
        def __init__(self, n):
            self.t = torch.zeros(n, device=DEV)

        def available(self):
            return True

        def take(self):
            return self.t
    acc = (Slot(C), Slot(C))
    got2 = ops.conv_dgrad_gn_raw(dy, wd, x, gam, bet, ft, fa, mean, rstd, sc, sh, seed, 5, p_drop, act, taps, acc=acc,
                                 dres=dres, dres2=dres2)
    assert torch.equal(got2[0], got[0])
    assert rel(got2[1], got[1]) < 1e-5 and rel(got2[2], got[2]) < 1e-5


@pytest.mark.parametrize('case', [
    # (B, Cin of dy, C1, C2 of x, H, W, taps, act, film, p_drop, n_res)
    (33, 64, 64, 0, 64, 64, 9, 2, True, 0.1, 1),      # 256-pixel tiles, 16 per image
    (16, 128, 128, 0, 32, 32, 9, 2, True, 0.1, 2), (2, 64, 64, 0, 64, 64, 9, 2, False, 0.0, 0),      # 64-pixel tiles
    (33, 64, 128, 64, 64, 64, 9, 2, False, 0.0, 1),   # two-source GroupNorm input (192 = 128 | 64 channels)
    (4, 128, 128, 128, 16, 16, 9, 2, False, 0.0, 1), (3, 384, 128, 0, 16, 16, 1, 1, False, 0.0, 1),   # small maps, 1x1 + affine
    (5, 128, 64, 0, 32, 32, 9, 2, True, 0.1, 0),
])
def test_backward_chain_du_epilogue_and_streaming_apply(case):
    """The big-map backward of conv(dropout(act(FiLM(GroupNorm(x))))): idf_conv_dgrad_chain_bf16 with the du epilogue
    (du + per-tile partial sums) followed by idf_gn_bwd_apply (the reduction-free rest) against fp32 PyTorch autograd:
    dx (+ branch gradients; two-source inputs: dx1 | dx2), dgamma, dbeta, dFiLM_t, dFiLM_a <= 4e-2."""
    B, Cin, C1, C2, H, W, taps, act, film, p_drop, n_res = case
    C, k = C1 + C2, 3 if taps == 9 else 1
    x1 = (0.3 + rnd(1, B, C1, H, W)).to(DEV).bfloat16().contiguous(memory_format=CL)
    x2 = (rnd(11, B, C2, H, W) - 0.2).to(DEV).bfloat16().contiguous(memory_format=CL) if C2 else None
    xc = torch.cat([x1, x2], dim=1).contiguous(memory_format=CL) if C2 else x1
    dy = rnd(2, B, Cin, H, W).to(DEV).bfloat16().contiguous(memory_format=CL)
    wgt = (rnd(3, Cin, C, k, k) / (C * taps) ** 0.5).to(DEV).bfloat16().float()
    _, wd = ops.pack_weight(wgt, torch.bfloat16, True, True)
    gam, bet = (1 + 0.1 * rnd(4, C)).to(DEV), (0.1 * rnd(5, C)).to(DEV)
    ft = (0.2 * rnd(6, B, 2 * C)).to(DEV) if film else None
    fa = (0.2 * rnd(7, B, 2 * C)).to(DEV) if film else None
    seed = torch.tensor([987654321], dtype=torch.int64, device=DEV) if p_drop else None
    res = [rnd(8 + i, B, C, H, W).to(DEV).bfloat16().contiguous(memory_format=CL) for i in range(n_res)]
    dres, dres2 = (res + [None, None])[:2]
    mask = ops.dropout_mask(seed, 5, p_drop, xc.numel()).view(B, H, W, C).permute(0, 3, 1, 2) if p_drop else None
    want = _gn_act_conv_reference(xc, gam, bet, ft, fa, act, mask, wgt, dy, res)
    _, mean, rstd, sc, sh = ops.gn_fused_fwd_raw(x1, gam, bet, ft, fa, seed, 5, p_drop, act, x2=x2) if ops.gn_small_ok(x1, x2) \
        else (None,) + ops.gn_coef_fwd_raw(xc, gam, bet, ft, fa)
    T = ops.chain_tiles(B, H, W, Cin, C, taps)
    assert T > 0
    du, part, _ = ops.conv_dgrad_chain_raw(dy, wd, taps, C, x=x1, x2=x2, sc=sc, sh=sh, seed=seed, salt=5, p_drop=p_drop, act=act)
    assert part.shape == (B, T, C, 2)
    # the partial sums are those of the du the launch wrote
    s1 = du.double().sum(dim=(2, 3))
    s2 = (du.double() * xc.double()).sum(dim=(2, 3))
    got = part.double().sum(dim=1)
    assert float((got[..., 0] - s1).abs().max()) < 1e-4 * (1 + float(s1.abs().max()))
    assert float((got[..., 1] - s2).abs().max()) < 1e-4 * (1 + float(s2.abs().max()))
    out = ops.gn_bwd_apply_raw(du, part, x1, gam, bet, ft, fa, mean, rstd, sc, dres=dres, dres2=dres2, x2=x2)
    dx = torch.cat(out[0], dim=1) if C2 else out[0]
    for nm, g, r in zip(('dx', 'dgamma', 'dbeta', 'dfilm_t', 'dfilm_a'), (dx,) + tuple(out[1:]), want):
        assert (g is None) == (r is None), nm
        if g is not None:
            parts = [(g, r)] if g.dim() != 2 else [(g[:, :C], r[:, :C]), (g[:, C:], r[:, C:])]
            for gg, rr in parts:
                assert rel(gg, rr) < 4e-2, (nm, rel(gg, rr))


@pytest.mark.parametrize('case', [
    # (B, C0, C1, C2, H, film, p_drop, plain_tail): x [C0] -GN1-conv1-> y1 [C1] -GN2-conv2-> y2 [C2]
    (33, 64, 64, 64, 64, True, 0.1, False), (16, 128, 128, 128, 32, True, 0.1, False), (2, 64, 64, 128, 64, False, 0.0, False),
    (5, 64, 128, 64, 32, True, 0.0, True), (3, 128, 128, 128, 16, False, 0.0, False),
])
def test_backward_chain_dy_prologue(case):
    """Two stacked stages y2 = conv2(act(GN2(conv1(act(GN1(x)))))): the gradient of y1 never exists as a tensor of its own
    -- conv2's data-gradient launch leaves (du2, partials), conv1's data-gradient launch forms dy1 = A*du2 + K1*y1 + K0 in
    its prologue (idf_conv_dgrad_chain_bf16), writes it once for the weight gradient and stores GN2's parameter / FiLM
    gradients.  Against fp32 PyTorch autograd: dy1, dx, both GroupNorms' dgamma / dbeta / dFiLM.  plain_tail: conv1 has no
    GroupNorm in front (plain epilogue behind the prologue)."""
    B, C0, C1, C2, H, film, p_drop, plain = case
    W = H
    x = (0.3 + rnd(1, B, C0, H, W)).bfloat16().float()
    w1 = (rnd(2, C1, C0, 3, 3) / (9 * C0) ** 0.5).bfloat16().float()
    w2 = (rnd(3, C2, C1, 3, 3) / (9 * C1) ** 0.5).bfloat16().float()
    g1, b1, g2, b2 = 1 + 0.1 * rnd(4, C0), 0.1 * rnd(5, C0), 1 + 0.1 * rnd(6, C1), 0.1 * rnd(7, C1)
    ft2 = 0.2 * rnd(8, B, 2 * C1) if film else None
    fa2 = 0.2 * rnd(9, B, 2 * C1) if film else None
    dy2 = rnd(10, B, C2, H, W).bfloat16().float()
    d = lambda t: None if t is None else t.to(DEV)
    xd = x.to(DEV).bfloat16().contiguous(memory_format=CL)
    seed = torch.tensor([424243], dtype=torch.int64, device=DEV) if p_drop else None
    # ---- forward through the product kernels (the tensors the backward reads are the bf16 ones the forward stored)
    wf1, wd1 = ops.pack_weight(d(w1), torch.bfloat16, True, True)
    wf2, wd2 = ops.pack_weight(d(w2), torch.bfloat16, True, True)
    if plain:
        y1 = ops.conv_raw(xd, wf1, None, None, None, None, None, 0, 0.0, ops.S1, 9, 0, C1)
        sc1 = sh1 = m1 = r1 = None
    else:
        y1, _, m1, r1, sc1, sh1, _ = ops.conv_gn_raw(xd, None, ops.gn_partials_raw(xd), None, d(g1), d(b1), None, None, None, 3,
                                                     0.0, 2, wf1, None, None, C1, 9, keep_coef=True)
    _, _, m2, r2, sc2, sh2, _ = ops.conv_gn_raw(y1, None, ops.gn_partials_raw(y1), None, d(g2), d(b2), d(ft2), d(fa2), seed, 7,
                                                p_drop, 2, wf2, None, None, C2, 9, keep_coef=True)
    # ---- reference: fp32 autograd over the same graph
    mask2 = ops.dropout_mask(seed, 7, p_drop, y1.numel()).view(B, H, W, C1).permute(0, 3, 1, 2).cpu() if p_drop else None
    xr = x.clone().requires_grad_(True)
    leaves = [t.clone().requires_grad_(True) if t is not None else None for t in (g1, b1, g2, b2, ft2, fa2)]
    g1r, b1r, g2r, b2r, ft2r, fa2r = leaves
    u1 = xr if plain else F.silu(F.group_norm(xr, 32, g1r, b1r, eps=1e-5))
    y1r = F.conv2d(u1, w1, None, padding=1)
    y1r.retain_grad()
    u2 = F.group_norm(y1r, 32, g2r, b2r, eps=1e-5)
    if film:
        u2 = u2 * (1 + ft2r[:, :C1, None, None]) + ft2r[:, C1:, None, None]
        u2 = u2 * (1 + fa2r[:, :C1, None, None]) + fa2r[:, C1:, None, None]
    u2 = F.silu(u2)
    if mask2 is not None:
        u2 = u2 * mask2
    F.conv2d(u2, w2, None, padding=1).backward(dy2)
    assert rel(y1, y1r) < 2e-2
    # ---- the chain
    dy2d = dy2.to(DEV).bfloat16().contiguous(memory_format=CL)
    du2, part2, _ = ops.conv_dgrad_chain_raw(dy2d, wd2, 9, C1, x=y1, sc=sc2, sh=sh2, seed=seed, salt=7, p_drop=p_drop, act=2)
    dft = torch.empty(B, 2 * C1, device=DEV) if film else None
    dfa = torch.empty(B, 2 * C1, device=DEV) if film else None
    acc = (torch.zeros(C1, device=DEV), torch.zeros(C1, device=DEV))
    lazy = ops.LazyGrad(du=du2, part=part2, x=y1, gn_w=d(g2), gn_b=d(b2), film_t=d(ft2), film_a=d(fa2), mean=m2, rstd=r2,
                        sc=sc2, dft=dft, dfa=dfa, acc=(acc[0].data_ptr(), acc[1].data_ptr()))
    if plain:
        dx, _, dy1 = ops.conv_dgrad_chain_raw(du2, wd1, 9, C0, lazy=lazy, want_dy=True)
    else:
        du1, part1, dy1 = ops.conv_dgrad_chain_raw(du2, wd1, 9, C0, lazy=lazy, want_dy=True, x=xd, sc=sc1, sh=sh1, act=2)
        dx, dg1, db1, _, _ = ops.gn_bwd_apply_raw(du1, part1, xd, d(g1), d(b1), None, None, m1, r1, sc1)
        assert rel(dg1, g1r.grad) < 4e-2 and rel(db1, b1r.grad) < 4e-2
    assert rel(dy1, y1r.grad) < 4e-2, rel(dy1, y1r.grad)
    assert rel(dx, xr.grad) < 4e-2, rel(dx, xr.grad)
    assert rel(acc[0], g2r.grad) < 4e-2 and rel(acc[1], b2r.grad) < 4e-2
    if film:
        for got, want in ((dft, ft2r.grad), (dfa, fa2r.grad)):
            assert rel(got[:, :C1], want[:, :C1]) < 4e-2 and rel(got[:, C1:], want[:, C1:]) < 4e-2
    # the same gradient through the streaming pass (LazyGrad.materialize: what a consumer that cannot take the pair gets)
    acc2 = (torch.zeros(C1, device=DEV), torch.zeros(C1, device=DEV))
    du2b, part2b, _ = ops.conv_dgrad_chain_raw(dy2d, wd2, 9, C1, x=y1, sc=sc2, sh=sh2, seed=seed, salt=7, p_drop=p_drop, act=2)
    lazy2 = ops.LazyGrad(du=du2b, part=part2b, x=y1, gn_w=d(g2), gn_b=d(b2), film_t=d(ft2), film_a=d(fa2), mean=m2, rstd=r2,
                         sc=sc2, dft=dft, dfa=dfa, acc=(acc2[0].data_ptr(), acc2[1].data_ptr()))
    assert rel(lazy2.materialize(), y1r.grad) < 4e-2


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('B,C,H', [(2, 128, 16), (3, 128, 8), (33, 64, 8), (2, 64, 4), (5, 64, 16), (33, 128, 16),
                                   (130, 128, 16), (129, 64, 16)])      # B >= 128: one workgroup walks all four row blocks
def test_attention(B, C, H, dtype):
    """16x16 and 8x8 bf16 cases (N = 256 / 64 tokens) run the fused kernels (idf_attn_fwd / idf_attn_bwd), the rest
    bmm + softmax."""
    qkv = rnd(1, B, 3 * C, H, H).to(dtype).float()
    qr = qkv.clone().requires_grad_(True)
    q, k, v = [t.permute(0, 2, 3, 1).reshape(B, H * H, C) for t in torch.chunk(qr, 3, dim=1)]
    w = F.softmax(torch.bmm(q, k.transpose(1, 2)) * (int(C) ** (-0.5)), dim=-1)
    ref = torch.bmm(w, v).view(B, H, H, C).permute(0, 3, 1, 2)
    go = rnd(2, *ref.shape).to(dtype).float()
    ref.backward(go)
    qd = qkv.to(DEV).to(dtype).contiguous(memory_format=CL).requires_grad_(True)
    o = ops.attention(qd)
    assert rel(o, ref) < TOL[dtype]
    o.backward(go.to(DEV).to(dtype).contiguous(memory_format=CL))
    assert rel(qd.grad, qr.grad) < (1e-4 if dtype == torch.float32 else 4e-2)


@pytest.mark.parametrize('B,K,N,silu', [(32, 256, 256, True), (3, 64, 256, False), (5, 4096, 32, False),
                                        (7, 32, 128, True), (32, 256, 4992, True)])
def test_linear(B, K, N, silu):
    x, w, b = rnd(1, B, K), rnd(2, N, K) / K ** 0.5, rnd(3, N)
    xr, wr, br = [t.clone().requires_grad_(True) for t in (x, w, b)]
    ref = F.linear(F.silu(xr) if silu else xr, wr, br)
    gy = rnd(4, B, N)
    ref.backward(gy)
    xd, wd_, bd = [t.to(DEV).requires_grad_(True) for t in (x, w, b)]
    y = ops.linear(xd, wd_, bd, silu)
    assert rel(y, ref) < 2e-5
    y.backward(gy.to(DEV))
    assert rel(xd.grad, xr.grad) < 5e-5
    assert rel(wd_.grad, wr.grad) < 5e-5
    assert rel(bd.grad, br.grad) < 5e-5


def test_qsample_bit_exact_and_gather():
    T = 1000
    ab = torch.cumprod(1 - torch.linspace(start=1e-5, end=1e-2, steps=T), dim=0)
    x, eps = rnd(1, 4, 3, 16, 16), rnd(2, 4, 3, 16, 16)
    idx = torch.tensor([0, 17, 500, 999])
    used = ab[idx][:, None, None, None]
    ref = torch.sqrt(used) * x + torch.sqrt(1 - used) * eps
    xt = ops.q_sample(x.to(DEV), eps.to(DEV), idx.to(DEV), ops.qsample_tables(ab.to(DEV)), torch.float32)
    assert torch.equal(xt.cpu(), ref)
    table = rnd(3, T, 64)
    out = ops.gather_rows(table.to(DEV), idx.to(DEV))
    assert torch.equal(out.cpu(), table[idx])




@pytest.mark.parametrize('k,mode', [(3, ops.S1), (1, ops.S1), (3, ops.S2), (3, ops.UP2)])
@pytest.mark.parametrize('case', [(2, 64, 64, 64, 64), (2, 128, 32, 32, 128), (3, 192, 16, 16, 64), (4, 128, 8, 8, 128),
                                  (2, 256, 32, 32, 128), (2, 96, 16, 16, 32), (1, 64, 64, 64, 8), (8, 64, 8, 4, 128),
                                  (3, 128, 16, 16, 384), (2, 3, 32, 32, 64), (2, 64, 32, 32, 3), (2, 64, 16, 16, 1)])
def test_wgrad_bf16_fast(case, k, mode):
    """Transposed-LDS-read weight-gradient kernel (+ fused bias gradient) vs PyTorch.
    (B, Cin, Ho, Wo, Cout): Ho, Wo are the OUTPUT dims."""
    B, Cin, H, W, Cout = case
    Hs, Ws = (2 * H, 2 * W) if mode == ops.S2 else ((H // 2, W // 2) if mode == ops.UP2 else (H, W))
    a = rnd(1, B, Cin, Hs, Ws).bfloat16().float()
    gy = rnd(2, B, Cout, H, W).bfloat16().float()
    w = torch.zeros(Cout, Cin, k, k, requires_grad=True)
    bb = torch.zeros(Cout, requires_grad=True)
    ain = F.interpolate(a, scale_factor=2.0, mode='nearest') if mode == ops.UP2 else a
    F.conv2d(ain, w, bb, stride=2 if mode == ops.S2 else 1, padding=k // 2).backward(gy)
    ad = a.to(DEV).bfloat16().contiguous(memory_format=CL)
    gd = gy.to(DEV).bfloat16().contiguous(memory_format=CL)
    pad8 = lambda c: -(-c // 8) * 8
    if mode == ops.S2 and (H * W) % 32 and W * min(H, 64 // W) % 32:
        pytest.skip('not tileable')
    assert ops._fast_wgrad_ok(pad8(Cin), pad8(Cout), H, W, torch.bfloat16, mode, k * k)
    dW, db = ops.conv_wgrad_bias_raw(ad, gd, mode, k * k, True)
    assert dW.shape == w.shape and db.shape == bb.shape
    assert rel(dW, w.grad) < 1e-4
    assert rel(db, bb.grad) < 1e-4


def test_gn_apply():
    B, C, H = 2, 64, 8
    x = rnd(1, B, C, H, H)
    sc, sh = rnd(2, B, C), rnd(3, B, C)
    xd = x.to(DEV).contiguous(memory_format=CL)
    a = ops.gn_apply_raw(xd, sc.to(DEV), sh.to(DEV), None, 0, 0.0, 2)
    assert rel(a, F.silu(x * sc[:, :, None, None] + sh[:, :, None, None])) < 1e-5
    a = ops.gn_apply_raw(xd.bfloat16(), sc.to(DEV), sh.to(DEV), None, 0, 0.0, 1)
    assert rel(a, x.bfloat16().float() * sc[:, :, None, None] + sh[:, :, None, None]) < 1e-2


def test_fused_clip_adamw_matches_torch():
    """Fused clip + AdamW (3 launches) vs clip_grad_norm_ + torch.optim.AdamW, 4 steps, mixed layouts."""
    from infodiffusion_amd.optim import FusedClipAdamW
    shapes = [(64, 64, 3, 3), (128,), (256, 64), (70000,), (3, 5)]
    ps_a = [torch.nn.Parameter(rnd(i, *s).to(DEV)) for i, s in enumerate(shapes)]
    ps_a[0].data = ps_a[0].data.contiguous(memory_format=CL)
    ps_b = [torch.nn.Parameter(p.detach().clone(memory_format=torch.preserve_format)) for p in ps_a]
    dead = torch.nn.Parameter(torch.zeros(4, device=DEV))        # never gets a grad
    oa = FusedClipAdamW(ps_a + [dead], lr=1e-2, weight_decay=1e-2, max_norm=1.0)
    ob = torch.optim.AdamW(ps_b, lr=1e-2, weight_decay=1e-2)
    for step in range(4):
        for i, (pa, pb) in enumerate(zip(ps_a, ps_b)):
            g = rnd(100 * step + i, *pa.shape).to(DEV) * (3.0 if step % 2 else 0.01)
            if i == 0:
                g = g.contiguous(memory_format=CL)
            pa.grad = g.clone(memory_format=torch.preserve_format)
            pb.grad = g.clone(memory_format=torch.preserve_format)
        tn = torch.nn.utils.clip_grad_norm_(ps_b, 1.0)
        ob.step()
        oa.step()
        assert abs(float(oa.total_norm()) - float(tn)) / float(tn) < 1e-5
        for pa, pb in zip(ps_a, ps_b):
            assert rel(pa, pb) < 2e-6
            assert rel(pa.grad, pb.grad) < 2e-6      # clipped grads written back like clip_grad_norm_
    assert dead.grad is None


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('C,H', [(128, 16), (256, 16), (128, 8), (256, 8), (32, 8), (64, 4)])
def test_gn_one_launch_small(C, H, dtype):
    """One-launch GroupNorm(+FiLM+SiLU) forward and backward vs the three-launch path and PyTorch."""
    B = 3
    x = (rnd(1, B, C, H, H) * 1.5 + 0.3).to(dtype).float()
    gam, bet = 1 + 0.1 * rnd(2, C), 0.1 * rnd(3, C)
    ft, fa = 0.3 * rnd(4, B, 2 * C), 0.3 * rnd(5, B, 2 * C)
    xr, gr, br, ftr, far = [t.clone().requires_grad_(True) for t in (x, gam, bet, ft, fa)]
    h = F.group_norm(xr, 32, gr, br, 1e-5)
    st, bt = torch.chunk(ftr[:, :, None, None], 2, dim=1)
    sa, ba = torch.chunk(far[:, :, None, None], 2, dim=1)
    ref = F.silu((h * (1 + st) + bt) * (1 + sa) + ba)
    gy = rnd(7, *ref.shape).to(dtype).float()
    ref.backward(gy)
    g = lambda t: t.to(DEV).contiguous()
    xd = x.to(DEV).to(dtype).contiguous(memory_format=CL)
    if not ops.gn_small_ok(xd):
        pytest.skip('sample does not fit one workgroup in this dtype')
    a, mean, rstd, sc, sh = ops.gn_fused_fwd_raw(xd, g(gam), g(bet), g(ft), g(fa), None, 0, 0.0, 2)
    tol = TOL[dtype]
    assert rel(a, ref) < tol
    m2, r2, sc2, sh2 = ops.gn_coef_fwd_raw(xd, g(gam), g(bet), g(ft), g(fa))
    assert rel(mean, m2) < 1e-5 and rel(rstd, r2) < 1e-5 and rel(sc, sc2) < 1e-5 and rel(sh, sh2) < 1e-5
    dA = gy.to(DEV).to(dtype).contiguous(memory_format=CL)
    dx, dgam, dbet, dft, dfa = ops.gn_fused_bwd_raw(dA, xd, g(gam), g(bet), g(ft), g(fa), mean, rstd, sc, sh, None, 0,
                                                   0.0, 2)
    btol = 3e-4 if dtype == torch.float32 else 4e-2
    for got, want in ((dx, xr.grad), (dgam, gr.grad), (dbet, br.grad), (dft, ftr.grad), (dfa, far.grad)):
        assert rel(got, want) < btol
    # the residual-branch gradient joins inside the kernel (dx + dres in one pass)
    dres = rnd(9, *x.shape).to(dtype)
    dx2 = ops.gn_fused_bwd_raw(dA, xd, g(gam), g(bet), g(ft), g(fa), mean, rstd, sc, sh, None, 0, 0.0, 2,
                               dres=dres.to(DEV).contiguous(memory_format=CL))[0]
    assert rel(dx2, xr.grad + dres.float()) < btol


@pytest.mark.parametrize('B,Cin,H,Cout', [(3, 3, 64, 64), (2, 1, 32, 64), (5, 3, 16, 128), (33, 3, 64, 64)])
def test_head_conv_few_input_channels_vs_pytorch(B, Cin, H, Cout):
    """idf_conv3x3_fewc_bf16 (the network's head conv: Cin <= 3, one MFMA K-step, statistics of y in the epilogue) through
    ops.conv_raw against fp32 PyTorch on the same bf16-valued operands; the statistics partials against the channel sums of y."""
    x = rnd(1, B, Cin, H, H).to(DEV).bfloat16().contiguous(memory_format=CL)
    w = (rnd(2, Cout, Cin, 3, 3) / (9 * Cin) ** 0.5).to(DEV)
    bias = rnd(3, Cout).to(DEV)
    wf, _ = ops.pack_weight(w, torch.bfloat16, True, False)
    names = []
    orig = ops.call
    ops.call = lambda n, *a: (names.append(n), orig(n, *a))[1]
    try:
        y, st = ops.conv_raw(x, wf, bias, None, None, None, None, 0, 0.0, ops.S1, 9, 0, Cout, want_stats=True)
    finally:
        ops.call = orig
    assert names == ['idf_conv3x3_fewc_bf16'], names
    ref = F.conv2d(x.float(), wf.float().view(Cout, 3, 3, Cin).permute(0, 3, 1, 2), bias, padding=1)
    assert rel(y, ref) < 1e-2, rel(y, ref)
    assert st is not None and st.shape == (B, H * H // 256, Cout, 2)
    s1, s2 = _chan_sums(y)
    got = st.double().sum(dim=1)
    assert float((got[..., 1] - s2).abs().max() / s2.abs().max()) < 1e-5
    assert float((got[..., 0] - s1).abs().max()) < 2e-3 * (1 + float(s1.abs().max()))


class _FragShadows:
    """What ops._wr_frag wants of a modules._Shadows: val = [forward, data-gradient, fragment-major forward, fragment-major
    data-gradient] shadows of one 3x3 conv, packed here by torch ops ([N][taps][K] -> [K/64][N/16][tap][half][fq][fr][8])."""

    def __init__(self, w):
        wf, wd = ops.pack_weight(w, torch.bfloat16, True, True)

        def frag(m):
            N, taps, K = m.shape
            if N % 16 or K % 64:
                return None
            return m.view(N // 16, 16, taps, K // 64, 2, 4, 8).permute(3, 0, 2, 4, 5, 1, 6).contiguous().view(-1)
        self.val = [wf, wd, frag(wf), frag(wd)]

    def request_frag(self):
        raise AssertionError('fragment-major shadow missing')


@pytest.mark.parametrize('case', [
    # (B, C1, C2, H, Cout, film, p_drop)
    (3, 128, 0, 16, 128, True, 0.1), (2, 128, 128, 16, 128, False, 0.0), (3, 128, 0, 8, 128, True, 0.1),
    (2, 128, 128, 8, 128, False, 0.1), (33, 64, 0, 16, 64, True, 0.0), (2, 128, 128, 16, 256, False, 0.0),
])
def test_conv_wr_groupnorm_prologue_conv_vs_pytorch(case):
    """idf_conv_wr_gn_bf16 (the 16x16 / 8x8 GroupNorm-prologue conv with fragment-major weights in registers, barrier-free conv
    loop, epilogue in the wave's registers) against fp32 PyTorch -- y, the activated tensor, mean / rstd, the statistics partials
    of y -- and against idf_conv_gn_bf16 on the same inputs."""
    B, C1, C2, H, Cout, film, p_drop = case
    C = C1 + C2
    x1 = (0.5 + 1.5 * rnd(1, B, C1, H, H)).to(DEV).bfloat16().contiguous(memory_format=CL)
    x2 = (rnd(2, B, C2, H, H) - 0.3).to(DEV).bfloat16().contiguous(memory_format=CL) if C2 else None
    xc = torch.cat([x1, x2], dim=1).contiguous(memory_format=CL) if C2 else x1
    gam, bet = (1 + 0.1 * rnd(3, C)).to(DEV), (0.1 * rnd(4, C)).to(DEV)
    ft = (0.2 * rnd(5, B, 2 * C)).to(DEV) if film else None
    fa = (0.2 * rnd(6, B, 2 * C)).to(DEV) if film else None
    w = (rnd(7, Cout, C, 3, 3) / (C * 9) ** 0.5).to(DEV)
    bias = rnd(8, Cout).to(DEV)
    res = rnd(9, B, Cout, H, H).to(DEV).bfloat16().contiguous(memory_format=CL)
    seed = torch.tensor([123456789], dtype=torch.int64, device=DEV) if p_drop else None
    sh_ = _FragShadows(w)
    assert ops.wr_tiles(B, H, H, C, Cout, 0) == (1 if H == 8 else 4)
    st1, st2 = ops.gn_partials_raw(x1), (ops.gn_partials_raw(x2) if C2 else None)
    names = []
    orig = ops.call
    ops.call = lambda n, *a: (names.append(n), orig(n, *a))[1]
    try:
        y, a, mean, rstd, sc, sh, st = ops.conv_gn_raw(x1, x2, st1, st2, gam, bet, ft, fa, seed, 7, p_drop, 2, sh_.val[0], bias, res,
                                                       Cout, 9, keep_a=True, keep_coef=True, want_stats=True, shadows=sh_)
    finally:
        ops.call = orig
    assert names == ['idf_conv_wr_gn_bf16'], names
    u = F.group_norm(xc.float(), 32, gam, bet, eps=1e-5)
    if film:
        u = u * (1 + ft[:, :C, None, None]) + ft[:, C:, None, None]
        u = u * (1 + fa[:, :C, None, None]) + fa[:, C:, None, None]
    u = F.silu(u)
    if p_drop:
        u = u * ops.dropout_mask(seed, 7, p_drop, xc.numel()).view(B, H, H, C).permute(0, 3, 1, 2)
    ref = F.conv2d(u, w, bias, padding=1) + res.float()
    assert rel(y, ref) < 2e-2, rel(y, ref)
    assert rel(a, u) < 1e-2
    mu = xc.float().reshape(B, 32, -1).mean(dim=2)
    var = xc.float().reshape(B, 32, -1).var(dim=2, unbiased=False)
    assert rel(mean, mu) < 1e-5 and rel(rstd, (var + 1e-5).rsqrt()) < 1e-5
    s1, s2 = _chan_sums(y)
    got = st.double().sum(dim=1)
    assert st.shape == (B, 1 if H == 8 else 4, Cout, 2)
    assert float((got[..., 1] - s2).abs().max() / s2.abs().max()) < 1e-5
    assert float((got[..., 0] - s1).abs().max()) < 2e-3 * (1 + float(s1.abs().max()))
    # the register-staged kernel on the same inputs: same coefficients, outputs within a bf16 ulp or two
    y0, a0, m0, r0, sc0, sh0, _ = ops.conv_gn_raw(x1, x2, st1, st2, gam, bet, ft, fa, seed, 7, p_drop, 2, sh_.val[0], bias, res,
                                                  Cout, 9, keep_a=True, keep_coef=True, want_stats=True)
    assert rel(sc, sc0) < 1e-5 and rel(sh, sh0) < 1e-5
    assert float((a.float() - a0.float()).abs().max()) <= 2 ** -7 * float(a0.float().abs().max())
    assert rel(y, y0) < 1e-2


@pytest.mark.parametrize('case', [
    # (B, Cin, H, Cout, film, p_drop, with_res): 64x64 with 64 channels, 32x32 with 128 / 64; one and several tiles per workgroup
    (32, 64, 64, 64, True, 0.1, True), (9, 64, 64, 64, False, 0.0, False), (70, 64, 64, 128, True, 0.1, False),
    (32, 128, 32, 128, True, 0.1, True), (16, 128, 32, 128, False, 0.0, False), (150, 128, 32, 128, True, 0.0, True),
    (32, 64, 32, 128, True, 0.1, False), (64, 128, 32, 64, False, 0.1, True),
])
def test_conv_rs_groupnorm_prologue_conv_vs_pytorch(case, monkeypatch):
    """idf_conv_rs_gn_bf16 (round 5: the GroupNorm-prologue conv of the 64x64 / 32x32 maps with the weights fragment-major in
    registers, the whole-K halo image in LDS, row reuse, persistent over the CU's tiles) against fp32 PyTorch -- y, the
    activated tensor, mean / rstd / sc / sh, the statistics partials of y -- and against idf_conv_gn_bf16 on the same inputs."""
    B, C, H, Cout, film, p_drop, with_res = case
    monkeypatch.setattr(ops, '_RS_FWD_ALL', True)      # every covered forward shape (the product routes only those it measured faster)
    x1 = (0.5 + 1.5 * rnd(1, B, C, H, H)).to(DEV).bfloat16().contiguous(memory_format=CL)
    gam, bet = (1 + 0.1 * rnd(3, C)).to(DEV), (0.1 * rnd(4, C)).to(DEV)
    ft = (0.2 * rnd(5, B, 2 * C)).to(DEV) if film else None
    fa = (0.2 * rnd(6, B, 2 * C)).to(DEV) if film else None
    w = (rnd(7, Cout, C, 3, 3) / (C * 9) ** 0.5).to(DEV)
    bias = rnd(8, Cout).to(DEV)
    res = rnd(9, B, Cout, H, H).to(DEV).bfloat16().contiguous(memory_format=CL) if with_res else None
    seed = torch.tensor([123456789], dtype=torch.int64, device=DEV) if p_drop else None
    sh_ = _FragShadows(w)
    T = ops.rs_fwd_tiles(B, H, H, C, Cout)
    base = H // (256 // H)          # whole-row tiles; two half-width tiles per strip; half-height tiles where both cout tiles share an image
    assert T in (base, 2 * base, 4 * base) and (T == 4 * base) == (H == 32 and Cout == 128 and ops.rs_tiles(B, H, H, C, Cout) == 2 * base), (case, T)
    st1 = ops.gn_partials_raw(x1)
    if st1.shape[1] > 16:         # the form takes what a conv producer leaves: <= 16 partials per image (regrouped sums are partials too)
        st1 = st1.view(B, 16, st1.shape[1] // 16, C, 2).sum(dim=2).contiguous()
    names = []
    orig = ops.call
    ops.call = lambda n, *a: (names.append(n), orig(n, *a))[1]
    try:
        y, a, mean, rstd, sc, sh, st = ops.conv_gn_raw(x1, None, st1, None, gam, bet, ft, fa, seed, 7, p_drop, 2, sh_.val[0], bias, res,
                                                       Cout, 9, keep_a=True, keep_coef=True, want_stats=True, shadows=sh_)
    finally:
        ops.call = orig
    assert names == ['idf_conv_rs_gn_bf16'], names
    # inference (no activated-tensor output): a workgroup that walks down an image half keeps the two halo rows it shares with the
    # tile above in LDS instead of fetching and transforming them again -- the same bits
    y_inf = ops.conv_gn_raw(x1, None, st1, None, gam, bet, ft, fa, seed, 7, p_drop, 2, sh_.val[0], bias, res, Cout, 9, keep_a=False,
                            keep_coef=False, want_stats=False, shadows=sh_)[0]
    assert torch.equal(y_inf, y)
    u = F.group_norm(x1.float(), 32, gam, bet, eps=1e-5)
    if film:
        u = u * (1 + ft[:, :C, None, None]) + ft[:, C:, None, None]
        u = u * (1 + fa[:, :C, None, None]) + fa[:, C:, None, None]
    u = F.silu(u)
    if p_drop:
        u = u * ops.dropout_mask(seed, 7, p_drop, x1.numel()).view(B, H, H, C).permute(0, 3, 1, 2)
    ref = F.conv2d(u, w, bias, padding=1)
    if with_res:
        ref = ref + res.float()
    assert rel(y, ref) < 2e-2, rel(y, ref)
    assert rel(a, u) < 1e-2
    mu = x1.float().reshape(B, 32, -1).mean(dim=2)
    var = x1.float().reshape(B, 32, -1).var(dim=2, unbiased=False)
    assert rel(mean, mu) < 1e-5 and rel(rstd, (var + 1e-5).rsqrt()) < 1e-5
    s1, s2 = _chan_sums(y)
    got = st.double().sum(dim=1)
    assert st.shape == (B, T, Cout, 2)
    assert float((got[..., 1] - s2).abs().max() / s2.abs().max()) < 1e-5
    assert float((got[..., 0] - s1).abs().max()) < 2e-3 * (1 + float(s1.abs().max()))
    # the halo / direct-to-LDS kernels on the same inputs: same coefficients, outputs within a bf16 ulp or two
    names.clear()
    ops.call = lambda n, *a: (names.append(n), orig(n, *a))[1]
    try:
        y0, a0, m0, r0, sc0, sh0, _ = ops.conv_gn_raw(x1, None, st1, None, gam, bet, ft, fa, seed, 7, p_drop, 2, sh_.val[0], bias, res,
                                                      Cout, 9, keep_a=True, keep_coef=True, want_stats=True)
    finally:
        ops.call = orig
    assert 'idf_conv_rs_gn_bf16' not in names
    assert rel(sc, sc0) < 1e-5 and rel(sh, sh0) < 1e-5
    assert float((a.float() - a0.float()).abs().max()) <= 2 ** -7 * float(a0.float().abs().max())
    assert rel(y, y0) < 1e-2


@pytest.mark.parametrize('case', [
    # (B, channels of dy, C1, C2 of x | x2, H, act, film, p_drop, n_res)
    (32, 64, 64, 0, 64, 2, True, 0.1, 1), (9, 64, 64, 0, 64, 2, False, 0.0, 0), (33, 64, 128, 64, 64, 2, False, 0.0, 1),
    (32, 128, 128, 0, 32, 2, True, 0.1, 2), (40, 128, 128, 128, 32, 2, False, 0.0, 1), (64, 64, 128, 0, 32, 2, True, 0.1, 0),
    (64, 128, 64, 0, 32, 1, False, 0.0, 0), (32, 64, 64, 64, 64, 2, False, 0.0, 1),
])
def test_conv_rs_backward_chain_du_epilogue_vs_pytorch_autograd(case):
    """idf_conv_rs_dgrad_chain_bf16 (round 5: the data-gradient conv of the 64x64 / 32x32 maps in the register-weights form with
    the du epilogue) + idf_gn_bwd_apply against fp32 PyTorch autograd of conv(dropout(act(FiLM(GroupNorm(x | x2))))) with the
    product's dropout mask: dx (two-source inputs: dx1 | dx2), dgamma, dbeta, dFiLM_t, dFiLM_a <= 4e-2; du and its partials
    against idf_conv_dgrad_chain_bf16 on the same inputs."""
    B, Cin, C1, C2, H, act, film, p_drop, n_res = case
    W, C = H, C1 + C2
    x1 = (0.3 + rnd(1, B, C1, H, W)).to(DEV).bfloat16().contiguous(memory_format=CL)
    x2 = (rnd(11, B, C2, H, W) - 0.2).to(DEV).bfloat16().contiguous(memory_format=CL) if C2 else None
    xc = torch.cat([x1, x2], dim=1).contiguous(memory_format=CL) if C2 else x1
    dy = rnd(2, B, Cin, H, W).to(DEV).bfloat16().contiguous(memory_format=CL)
    wgt = (rnd(3, Cin, C, 3, 3) / (C * 9) ** 0.5).to(DEV).bfloat16().float()
    sh_ = _FragShadows(wgt)
    gam, bet = (1 + 0.1 * rnd(4, C)).to(DEV), (0.1 * rnd(5, C)).to(DEV)
    ft = (0.2 * rnd(6, B, 2 * C)).to(DEV) if film else None
    fa = (0.2 * rnd(7, B, 2 * C)).to(DEV) if film else None
    seed = torch.tensor([987654321], dtype=torch.int64, device=DEV) if p_drop else None
    res = [rnd(8 + i, B, C, H, W).to(DEV).bfloat16().contiguous(memory_format=CL) for i in range(n_res)]
    dres, dres2 = (res + [None, None])[:2]
    mask = ops.dropout_mask(seed, 5, p_drop, xc.numel()).view(B, H, W, C).permute(0, 3, 1, 2) if p_drop else None
    want = _gn_act_conv_reference(xc, gam, bet, ft, fa, act, mask, wgt, dy, res)
    mean, rstd, sc, sh = ops.gn_coef_fwd_raw(xc, gam, bet, ft, fa)
    T = ops.rs_tiles(B, H, W, Cin, C)
    assert T in (H // (256 // H), 2 * (H // (256 // H)))
    names = []
    orig = ops.call
    ops.call = lambda n, *a: (names.append(n), orig(n, *a))[1]
    try:
        du, part, _ = ops.conv_dgrad_chain_raw(dy, sh_.val[1], 9, C, x=x1, x2=x2, sc=sc, sh=sh, seed=seed, salt=5, p_drop=p_drop, act=act,
                                               shadows=sh_)
    finally:
        ops.call = orig
    assert names == ['idf_conv_rs_dgrad_chain_bf16'], names
    assert part.shape == (B, T, C, 2)
    s1 = du.double().sum(dim=(2, 3))
    s2 = (du.double() * xc.double()).sum(dim=(2, 3))
    got = part.double().sum(dim=1)
    assert float((got[..., 0] - s1).abs().max()) < 1e-4 * (1 + float(s1.abs().max()))
    assert float((got[..., 1] - s2).abs().max()) < 1e-4 * (1 + float(s2.abs().max()))
    du0, part0, _ = ops.conv_dgrad_chain_raw(dy, sh_.val[1], 9, C, x=x1, x2=x2, sc=sc, sh=sh, seed=seed, salt=5, p_drop=p_drop, act=act)
    assert rel(du, du0) < 1e-2, rel(du, du0)
    out = ops.gn_bwd_apply_raw(du, part, x1, gam, bet, ft, fa, mean, rstd, sc, dres=dres, dres2=dres2, x2=x2)
    dx = torch.cat(out[0], dim=1) if C2 else out[0]
    for nm, g, r in zip(('dx', 'dgamma', 'dbeta', 'dfilm_t', 'dfilm_a'), (dx,) + tuple(out[1:]), want):
        assert (g is None) == (r is None), nm
        if g is not None:
            parts = [(g, r)] if g.dim() != 2 else [(g[:, :C], r[:, :C]), (g[:, C:], r[:, C:])]
            for gg, rr in parts:
                assert rel(gg, rr) < 4e-2, (nm, rel(gg, rr))
    # ---- the same backward in ONE launch (idf_conv_rs_dgrad_gn_bf16: du held on chip, the image's workgroups meet at a counter):
    # against the two launches above (same du rounding; the sums differ only in their order) and against fp32 autograd; every
    # repetition bit-identical (a stale read of another workgroup's partials would show here) and nobody gave up waiting
    ops.rs_sync_timeouts(True)
    names = []
    ops.call = lambda n, *a: (names.append(n), orig(n, *a))[1]
    try:
        one = ops.conv_dgrad_gn_sync_raw(dy, C, x1, gam, bet, ft, fa, mean, rstd, sc, sh, seed, 5, p_drop, act, None, dres, dres2, x2=x2,
                                         shadows=sh_)
    finally:
        ops.call = orig
    # (a 64-channel slice must hold whole GroupNorm groups: 192 channels stay on the two launches; at 64x64 a workgroup holds at
    # most two tiles across the wait: B = 32 with 64 channels -- the ResBlocks of the first level -- but not B = 33 or 128 channels)
    covered = int(ops._lib.load().idf_conv_rs_dgrad_gn_tiles(B, H, W, Cin, C)) > 0
    assert covered == (64 % (C // 32) == 0 and (H == 32 or B * (C // 64) <= 32)), (covered, case)
    if not covered:
        assert one is None
        return
    assert one is not None and 'idf_conv_rs_dgrad_gn_bf16' in names, names
    dx1 = torch.cat(one[0], dim=1) if C2 else one[0]
    assert rel(dx1, dx) < 1e-2, rel(dx1, dx)          # bf16 outputs: one ulp of the largest element is 2^-7
    for nm, g, g0, r in zip(('dx', 'dgamma', 'dbeta', 'dfilm_t', 'dfilm_a'), (dx1,) + tuple(one[1:]), (dx,) + tuple(out[1:]), want):
        assert (g is None) == (r is None), nm
        if g is not None:
            parts = [(g, g0, r)] if g.dim() != 2 else [(g[:, :C], g0[:, :C], r[:, :C]), (g[:, C:], g0[:, C:], r[:, C:])]
            for gg, g00, rr in parts:
                assert rel(gg, rr) < 4e-2, (nm, rel(gg, rr))
                assert rel(gg, g00) < (1e-2 if nm == 'dx' else 2e-3), (nm, rel(gg, g00))
    for _ in range(12):
        again = ops.conv_dgrad_gn_sync_raw(dy, C, x1, gam, bet, ft, fa, mean, rstd, sc, sh, seed, 5, p_drop, act, None, dres, dres2, x2=x2,
                                           shadows=sh_)
        dxa = torch.cat(again[0], dim=1) if C2 else again[0]
        assert torch.equal(dxa, dx1)
        for g, g0 in zip(again[1:], one[1:]):
            assert (g is None and g0 is None) or torch.equal(g, g0)
    assert ops.rs_sync_timeouts(True) == 0


@pytest.mark.parametrize('case', [
    # (B, channels of dy, C of x, H, film, p_drop, n_res)
    (3, 128, 128, 8, True, 0.1, 1), (2, 128, 128, 8, True, 0.1, 2), (2, 128, 256, 8, True, 0.0, 1), (33, 128, 128, 8, False, 0.0, 0),
    (2, 64, 128, 8, False, 0.1, 1),
])
def test_conv_wr_dgrad_with_groupnorm_backward_vs_pytorch_autograd(case):
    """idf_conv_wr_dgrad_gn_bf16 (whole-image data-gradient conv with the GroupNorm / FiLM / SiLU / dropout backward in the
    wave's registers) against fp32 PyTorch autograd of conv(dropout(SiLU(FiLM(GroupNorm(x))))) with the product's dropout
    mask: dx (+ branch gradients), dgamma, dbeta, dFiLM_t, dFiLM_a <= 4e-2 -- the bound idf_conv_dgrad_gn_bf16 is held to."""
    B, Cin, C, H, film, p_drop, n_res = case
    x = (0.3 + rnd(1, B, C, H, H)).to(DEV).bfloat16().contiguous(memory_format=CL)
    dy = rnd(2, B, Cin, H, H).to(DEV).bfloat16().contiguous(memory_format=CL)
    wgt = (rnd(3, Cin, C, 3, 3) / (C * 9) ** 0.5).to(DEV).bfloat16().float()
    sh_ = _FragShadows(wgt)
    gam, bet = (1 + 0.1 * rnd(4, C)).to(DEV), (0.1 * rnd(5, C)).to(DEV)
    ft = (0.2 * rnd(6, B, 2 * C)).to(DEV) if film else None
    fa = (0.2 * rnd(7, B, 2 * C)).to(DEV) if film else None
    seed = torch.tensor([987654321], dtype=torch.int64, device=DEV) if p_drop else None
    res = [rnd(8 + i, B, C, H, H).to(DEV).bfloat16().contiguous(memory_format=CL) for i in range(n_res)]
    dres, dres2 = (res + [None, None])[:2]
    mask = ops.dropout_mask(seed, 5, p_drop, x.numel()).view(B, H, H, C).permute(0, 3, 1, 2) if p_drop else None
    want = _gn_act_conv_reference(x, gam, bet, ft, fa, 2, mask, wgt, dy, res)
    _, mean, rstd, sc, sh = ops.gn_fused_fwd_raw(x, gam, bet, ft, fa, seed, 5, p_drop, 2)
    names = []
    orig = ops.call
    ops.call = lambda n, *a: (names.append(n), orig(n, *a))[1]
    try:
        got = ops.conv_dgrad_gn_raw(dy, sh_.val[1], x, gam, bet, ft, fa, mean, rstd, sc, sh, seed, 5, p_drop, 2, 9, dres=dres,
                                    dres2=dres2, shadows=sh_)
    finally:
        ops.call = orig
    assert names[0] == 'idf_conv_wr_dgrad_gn_bf16', names
    for nm, g, r in zip(('dx', 'dgamma', 'dbeta', 'dfilm_t', 'dfilm_a'), got, want):
        assert (g is None) == (r is None), nm
        if g is not None:
            parts = [(g, r)] if g.dim() != 2 else [(g[:, :C], r[:, :C]), (g[:, C:], r[:, C:])]
            for gg, rr in parts:
                assert rel(gg, rr) < 4e-2, (nm, rel(gg, rr))


@pytest.mark.parametrize('kind,cin,dual,train', [('aux', 128, False, True), ('aux', 256, True, True), ('enc', 128, False, True),
                                                 ('enc', 256, True, False), ('aux', 128, False, False), ('res', 128, False, True)])
def test_resblock_small_one_launch_matches_the_per_op_path(kind, cin, dual, train, monkeypatch):
    """The image-resident ResBlock at 8x8 (idf_resblock_small_fwd: the whole block forward as ONE launch, one workgroup
    per image, GroupNorms closed in-block) against the per-op path of the same module on the same inputs, parameters and
    dropout seed: output, input gradients (both sources of a skip pair), FiLM gradients and every parameter gradient.  The
    two paths do the same arithmetic per element; only the summation order of the GroupNorm statistics differs, so the
    bound is a few bf16 ulps, not the fp32-autograd bound of the kernel tests (the per-op kernels carry that one)."""
    from infodiffusion_amd import modules
    from infodiffusion_amd.optim import FusedClipAdamW
    torch.manual_seed(11)
    if kind == 'aux':
        blk = modules.AuxResBlock(cin, 128, tdim=256, dropout=0.1)
    elif kind == 'res':
        blk = modules.ResBlock(cin, 128, tdim=256, dropout=0.1)
    else:
        blk = modules.ResBlock_encoder(cin, 128, dropout=0.1)
    blk = blk.to(DEV)
    with torch.no_grad():
        for name, prm in blk.named_parameters():      # GroupNorm affines and biases away from their (1, 0) initial values
            if prm.dim() == 1:
                prm.add_(0.3 * rnd(hash(name) % 1000, *prm.shape).to(DEV))
    for m in blk.modules():
        if isinstance(m, torch.nn.Conv2d) and m.kernel_size != (1, 1):
            m.weight.data = m.weight.data.contiguous(memory_format=CL)
    blk.ctx.act_dtype = torch.bfloat16
    blk.train(train)
    blk.ctx.seed = torch.tensor([424242], dtype=torch.int64, device=DEV) if train else None
    opt = FusedClipAdamW(blk.parameters(), lr=0.0, weight_decay=0.0)
    B = 5
    x1 = rnd(1, B, 128, 8, 8).to(DEV).bfloat16().contiguous(memory_format=CL)
    x2 = rnd(2, B, cin - 128, 8, 8).to(DEV).bfloat16().contiguous(memory_format=CL) if dual else None
    ft = (0.3 * rnd(3, B, 256)).to(DEV)
    fa = (0.3 * rnd(4, B, 256)).to(DEV)
    dyw = rnd(5, B, 128, 8, 8).to(DEV)
    names = []
    orig_call = ops.call

    def counted(name, *a):
        names.append(name)
        return orig_call(name, *a)
    monkeypatch.setattr(ops, 'call', counted)

    def run(fused):
        monkeypatch.setattr(ops, '_RB_SMALL', fused)
        del names[:]
        ins = [x1.clone().requires_grad_(True)] + ([x2.clone().requires_grad_(True)] if dual else [])
        f_t, f_a = ft.clone().requires_grad_(True), fa.clone().requires_grad_(True)
        opt.zero_grad()
        xin = tuple(ins) if dual else ins[0]
        if kind == 'aux':
            blk._film = {'t': f_t, 'a': f_a}
            y = blk(xin, None, None)
        elif kind == 'res':
            blk._film = {'t': f_t}
            y = blk(xin, None)
        else:
            y = blk(xin)
        fwd = list(names)
        if train:
            (y.float() * dyw).sum().backward()
            torch.cuda.synchronize()
        g = {'x%d' % i: v.grad for i, v in enumerate(ins)}
        if kind != 'enc':
            g['film_t'] = f_t.grad
        if kind == 'aux':
            g['film_a'] = f_a.grad
        g.update({k: prm.grad.detach().float().clone() for k, prm in blk.named_parameters() if prm.grad is not None})
        return y.detach().float().clone(), {k: v.detach().float().clone() for k, v in g.items() if v is not None}, fwd

    y_ref, g_ref, fwd_ref = run(False)
    assert 'idf_resblock_small_fwd' not in fwd_ref and any(n.startswith('idf_conv_gn') for n in fwd_ref)
    # three times: the first fused pass reads the [cout][tap][cin] shadows and asks for fragment-major ones; they exist from
    # the next re-pack on (the backward pass of a training step, else the next forward pass) and the later passes read those
    for attempt in range(3):
        y_got, g_got, fwd_got = run(True)
        assert fwd_got.count('idf_resblock_small_fwd') == 1 and not any(n.startswith('idf_conv') for n in fwd_got), fwd_got
        assert rel(y_got, y_ref) < 1e-2, (attempt, rel(y_got, y_ref))
        if train:
            assert set(g_got) == set(g_ref) and len(g_ref) >= (8 if kind == 'enc' else 12)
            for k in g_ref:
                assert rel(g_got[k], g_ref[k]) < 2e-2, (attempt, k, rel(g_got[k], g_ref[k]))
    assert blk._sh_block1.val[2] is not None and blk._sh_block2.val[2] is not None


def test_wgrad_batch_survives_a_backward_pass_that_raised():
    """The autograd engine runs no end-of-backward callbacks when a node raises, so the deferred weight gradients queued by
    such a pass are never launched and `_cb_queued` stays set: a LATER backward pass that trusted the flag would queue no
    flush of its own and silently train with zero conv weight gradients.  `WgradBatch.add` tells a new pass by its autograd
    graph task and drops the stale items; the trainer also resets before every forward and after a failed capture.  Either
    way the next pass must deliver the gradients of a clean run and launch nothing of the dead one."""
    from infodiffusion_amd import modules
    from infodiffusion_amd.optim import FusedClipAdamW
    torch.manual_seed(0)
    blk = modules.ResBlock_encoder(32, 32, dropout=0.0).to(DEV)
    blk.ctx.act_dtype = torch.bfloat16
    for m in blk.modules():         # as _UNetSkeleton._post: 3x3 master weights in [O][kh][kw][I] memory, the layout the kernel writes
        if isinstance(m, torch.nn.Conv2d) and m.kernel_size != (1, 1):
            m.weight.data = m.weight.data.contiguous(memory_format=CL)
    opt = FusedClipAdamW(blk.parameters(), lr=0.0, weight_decay=0.0)        # the gradient arena the deferred launches write
    x = rnd(3, 2, 32, 16, 16).to(DEV).bfloat16().contiguous(memory_format=CL)
    dy = rnd(4, 2, 32, 16, 16).to(DEV).bfloat16().contiguous(memory_format=CL)

    def run(fail):
        xin = x.clone().requires_grad_(True)
        h = xin * 1.0
        if fail:
            def boom(g):
                raise RuntimeError('boom')
            h.register_hook(boom)              # the LAST node of the pass: every conv has queued its weight gradient by then
        opt.zero_grad()
        blk(h).backward(dy)
        torch.cuda.synchronize()
        return {k: p.grad.detach().float().clone() for k, p in blk.named_parameters() if p.grad is not None}

    assert ops.WgradBatch.enabled
    clean = run(False)
    wkeys = [k for k in clean if k.endswith('3.weight') or k.endswith('2.weight')]
    assert any(float(clean[k].abs().max()) > 0 for k in wkeys)
    with pytest.raises(RuntimeError, match='boom'):
        run(True)
    # whether the engine still ran the end-of-backward callbacks depends on where the pass died (it skips them when
    # nodes are left with half-accumulated inputs); put the class in the state a skipped flush leaves behind
    canary_w = torch.zeros(32 * 9 * 32, dtype=torch.float32, device=DEV)
    canary_b = torch.zeros(32, dtype=torch.float32, device=DEV)
    # an item of the dead pass: nothing may accumulate into its slots any more
    stale = (x, dy, canary_w.data_ptr(), canary_b.data_ptr(), 2, 16, 16, 32, 32, 9, ops.S1, None, 0, 32, 32)
    for how in ('task', 'reset'):
        ops.WgradBatch._cb_queued = True
        ops.WgradBatch._task = -7
        ops.WgradBatch.pending = [stale]
        if how == 'reset':
            ops.WgradBatch.reset()
            assert not ops.WgradBatch.pending and not ops.WgradBatch._cb_queued
        # 'task': the next pass is another autograd graph task -- `add` drops the stale items and queues its own flush
        again = run(False)
        assert not ops.WgradBatch.pending and not ops.WgradBatch._cb_queued
        for k in clean:
            assert float((again[k] - clean[k]).abs().max()) <= 1e-5 * max(1.0, float(clean[k].abs().max())), (how, k)
        assert float(canary_w.abs().max()) == 0.0 and float(canary_b.abs().max()) == 0.0, how


@pytest.mark.parametrize('split', [0, 5, 11])
def test_wgrad_bf16_batched_matches_single_launches(split, monkeypatch):
    """All weight gradients of a backward pass from ONE table-driven launch per (taps, mode) class
    (idf_conv_wgrad_bf16_batched, accumulating into gradient-arena slots) == the per-conv launches.  `split`: the first
    `split` convs are flushed by an early barrier (the data-parallel step flushes the backbone's gradients before the encoder's
    backward pass), the rest at the end."""
    from infodiffusion_amd.grad_arena import GradArena, slot_of
    cases = [(4, 32, 32, 32, 32, 9, ops.S1), (4, 64, 16, 16, 128, 9, ops.S1), (2, 128, 8, 8, 128, 1, ops.S1),
             (4, 32, 32, 32, 64, 9, ops.S1), (4, 128, 8, 8, 128, 9, ops.S1), (2, 64, 16, 16, 64, 9, ops.S2),
             (2, 64, 16, 16, 64, 9, ops.UP2), (2, 64, 64, 64, 64, 9, ops.S1),
             # operands zero-padded to 8 channels, gradient in the parameter's own extents (image input, epsilon / latent heads)
             (3, 3, 32, 32, 64, 9, ops.S1), (3, 64, 32, 32, 3, 9, ops.S1), (2, 64, 16, 16, 1, 9, ops.S1),
             # round 6, the row-ring form (idf_wgrad_ring_ok): pixel splits that start and end in the middle of an image (5 x 32 tiles
             # over 16 workgroups), several cin / cout tiles incl. a partial one, every map width it takes, odd batches
             (5, 64, 64, 64, 64, 9, ops.S1), (3, 128, 32, 32, 192, 9, ops.S1), (7, 64, 16, 16, 72, 9, ops.S1), (9, 128, 8, 8, 128, 9, ops.S1),
             (3, 3, 64, 64, 64, 9, ops.S1), (1, 192, 64, 64, 64, 9, ops.S1)]
    ws, bs, data, refs = [], [], [], []
    for i, (B, Cin, H, W, Cout, taps, mode) in enumerate(cases):
        k = 3 if taps == 9 else 1
        ws.append(torch.nn.Parameter(torch.zeros(Cout, Cin, k, k, device=DEV).contiguous(memory_format=CL)))
        bs.append(torch.nn.Parameter(torch.zeros(Cout, device=DEV)))
        Hs, Ws = (2 * H, 2 * W) if mode == ops.S2 else ((H // 2, W // 2) if mode == ops.UP2 else (H, W))
        a = rnd(10 + i, B, Cin, Hs, Ws).to(DEV).bfloat16().contiguous(memory_format=CL)
        dy = rnd(40 + i, B, Cout, H, W).to(DEV).bfloat16().contiguous(memory_format=CL)
        data.append((a, dy))
        refs.append(ops.conv_wgrad_bias_raw(a, dy, mode, taps, True))
    arena = GradArena(ws + bs)
    assert ops.WgradBatch.enabled and not ops.WgradBatch.pending
    ops.WgradBatch._cb_queued = True                 # hold the queue open as a running backward pass would
    ops.WgradBatch._task = torch._C._current_graph_task_id()        # (-1 out here: `add` sees the pass that queued the flush)
    outs = []
    for i, ((a, dy), w, b, c) in enumerate(zip(data, ws, bs, cases)):
        outs.append(ops.conv_wgrad_bias_raw(a, dy, c[6], c[5], True, slot_of(w), slot_of(b), True))
        if split and i + 1 == split:
            assert all(float(o[0].abs().max()) == 0.0 for o in outs)      # nothing launched yet
            ops.WgradBatch.flush()
            assert not ops.WgradBatch.pending
            ops.WgradBatch._cb_queued = True
    assert len(ops.WgradBatch.pending) == len(cases) - split
    assert all(float(o[0].abs().max()) == 0.0 for o in outs[split:])      # nothing launched yet
    ops.WgradBatch.flush()
    assert not ops.WgradBatch.pending
    for (rW, rb), (oW, ob) in zip(refs, outs):
        assert arena.holds(oW) and arena.holds(ob)
        assert rel(oW.cpu(), rW.cpu()) < 1e-5 and rel(ob.cpu(), rb.cpu()) < 1e-5
    # ... and == PyTorch's own fp32 autograd of the same convolution on the same (bf16-valued) operands: the batched launch
    # (the largest kernel of the step) is pinned against the framework directly, not only against its per-conv sibling.
    # fp32 accumulation over <= 2^15 products of bf16 values in a different order: 1e-4 of the gradient's largest entry.
    import torch.nn.functional as F
    for (a, dy), (oW, ob), (B, Cin, H, W, Cout, taps, mode) in zip(data, outs, cases):
        k = 3 if taps == 9 else 1
        w0 = torch.zeros(Cout, Cin, k, k, device=DEV, requires_grad=True)
        b0 = torch.zeros(Cout, device=DEV, requires_grad=True)
        af = a.float()
        if mode == ops.UP2:
            af = F.interpolate(af, scale_factor=2, mode='nearest')
        y = F.conv2d(af, w0, b0, stride=2 if mode == ops.S2 else 1, padding=k // 2)
        y.backward(dy.float())
        assert float((oW.float() - w0.grad).abs().max()) <= 1e-4 * float(w0.grad.abs().max()), (B, Cin, H, W, Cout, taps, mode)
        assert float((ob.float() - b0.grad).abs().max()) <= 1e-4 * float(b0.grad.abs().max()), (B, Cin, H, W, Cout, taps, mode)


@pytest.mark.parametrize('H,C1,C2,Cout,B', [(64, 64, 64, 64, 3), (32, 128, 64, 128, 5), (16, 64, 128, 64, 4)])
def test_wgrad_ring_two_source_input_matches_the_concatenation(H, C1, C2, Cout, B):
    """The batched 3x3 weight gradient on an (a, a2) pair that is never concatenated (up-path skip pairs) == the same launch on
    torch.cat([a, a2]), bit for bit (the ring kernel takes each 64-channel cin tile from the tensor it lies in), and == fp32 PyTorch
    autograd at 1e-4."""
    import torch.nn.functional as F
    from infodiffusion_amd.grad_arena import GradArena, slot_of
    a1 = rnd(1, B, C1, H, H).to(DEV).bfloat16().contiguous(memory_format=CL)
    a2 = rnd(2, B, C2, H, H).to(DEV).bfloat16().contiguous(memory_format=CL)
    dy = rnd(3, B, Cout, H, H).to(DEV).bfloat16().contiguous(memory_format=CL)
    cat = torch.cat([a1, a2], 1).contiguous(memory_format=CL)
    outs = []
    for pair in (True, False):
        w = torch.nn.Parameter(torch.zeros(Cout, C1 + C2, 3, 3, device=DEV).contiguous(memory_format=CL))
        b = torch.nn.Parameter(torch.zeros(Cout, device=DEV))
        arena = GradArena([w, b])
        got = ops._defer_or_launch_wgrad(a1 if pair else cat, dy, slot_of(w), slot_of(b), 9, a2 if pair else None)
        assert got is not None
        ops.WgradBatch.flush()
        assert arena.holds(got[0])
        outs.append((got[0].clone(), got[1].clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    w0 = torch.zeros(Cout, C1 + C2, 3, 3, device=DEV, requires_grad=True)
    b0 = torch.zeros(Cout, device=DEV, requires_grad=True)
    F.conv2d(cat.float(), w0, b0, padding=1).backward(dy.float())
    assert float((outs[0][0].float() - w0.grad).abs().max()) <= 1e-4 * float(w0.grad.abs().max())
    assert float((outs[0][1].float() - b0.grad).abs().max()) <= 1e-4 * float(b0.grad.abs().max())


@pytest.mark.parametrize('path,H,B,arena', [('block', 16, 3, True), ('fold', 16, 3, True), ('fold', 8, 5, True), ('fold', 16, 130, True),
                                            ('fold', 16, 3, False)])
@pytest.mark.parametrize('train', [False, True])
def test_attention_block_with_proj_folded_into_v(path, H, B, arena, train, monkeypatch):
    """The AttnBlock (modules.py:145-164) at 128 channels with the proj conv folded into V (Wv' = Wp Wv, b' = Wp bv + bp; the
    rows of the softmax sum to one): y = x + P V' --
    'fold':  the GroupNorm-prologue q | k | v' conv, then attention with the residual and the statistics of y in its epilogue
             (idf_attn_fwd_res; 16x16 with four workgroups per image or -- B >= 128 -- one, and the 8x8 middle block): two launches;
    'block': the whole block as ONE launch at 16x16 (idf_attnblock_fwd);
    (a) against fp32 PyTorch of the reference's block (q, k, v, softmax, bmm, proj, + x) on the same bf16-valued input and
        parameters (2e-2 of the output's range, as the unfolded per-op path), the statistics partials against the output's sums;
    (b) against the unfolded per-op path of the same module; and -- training -- input gradient and EVERY parameter gradient,
        proj's and proj_v's through the chain rule of the fold (idf_attn_fold_bwd_batched in the gradient arena; torch products
        without one), against fp32 PyTorch autograd (4e-2, the bound of the kernel tests) and the unfolded path."""
    import torch.nn.functional as F
    from infodiffusion_amd import modules
    from infodiffusion_amd.optim import FusedClipAdamW
    torch.manual_seed(5)
    blk = modules.AttnBlock(128).to(DEV)
    with torch.no_grad():
        blk.proj.weight.mul_(1e5 * 0.5)               # (the reference initialises proj with gain 1e-5: give the branch a voice)
        for name, prm in blk.named_parameters():
            if prm.dim() == 1:
                prm.add_(0.2 * rnd(hash(name) % 1000, *prm.shape).to(DEV))
    blk.train(train)
    opt = FusedClipAdamW(blk.parameters(), lr=0.0, weight_decay=0.0) if arena else None
    N = H * H
    x0 = (0.2 + rnd(1, B, 128, H, H)).to(DEV).bfloat16().contiguous(memory_format=CL)
    dyw = rnd(2, B, 128, H, H).to(DEV)
    names = []
    orig_call = ops.call

    def counted(name, *a):
        names.append(name)
        return orig_call(name, *a)
    monkeypatch.setattr(ops, 'call', counted)
    monkeypatch.setattr(ops, '_ATTN_BLOCK_MINB', 1)          # coverage, not the policy (B >= 256)

    def run(which):
        monkeypatch.setattr(ops, '_ATTN_FOLD', which != 'ops')
        monkeypatch.setattr(ops, '_ATTN_BLOCK', which == 'block')
        del names[:]
        x = x0.clone().requires_grad_(train)
        if opt is not None:
            opt.zero_grad()
        else:
            blk.zero_grad()
        with torch.set_grad_enabled(train):
            y = blk(x)
        fwd = list(names)
        st = getattr(y, '_gn', None)
        g = {}
        if train:
            (y.float() * dyw).sum().backward()
            torch.cuda.synchronize()
            g = {'x': x.grad.detach().float().clone()}
            g.update({k: prm.grad.detach().float().clone() for k, prm in blk.named_parameters() if prm.grad is not None})
        return y.detach().float().clone(), st, g, fwd

    y_ref, st_ref, g_ref, fwd_ref = run('ops')
    assert 'idf_attnblock_fwd' not in fwd_ref and 'idf_attn_fwd_res' not in fwd_ref and 'idf_attn_fwd' in fwd_ref
    run(path)                                         # ('block': asks for the fragment-major q | k | v' weights -- next re-pack)
    canary = None
    if arena and train:
        # a fix-up row left behind by a backward pass that never ended (the engine skips the end-of-pass callbacks when a node
        # raises): the next pass is another graph task -- it must start its own list, queue its own callback and launch nothing stale
        ones = torch.ones(128 * 128 + 128, dtype=torch.float32, device=DEV)
        canary = torch.zeros(128 * 128 + 128, dtype=torch.float32, device=DEV)
        ops._FOLD_BWD_ROWS[:] = [(ones.data_ptr(), ones.data_ptr(), ones.data_ptr(), ones.data_ptr(), ones.data_ptr(),
                                  canary.data_ptr(), canary.data_ptr() + 4 * 128 * 128, 128)]
        ops._FOLD_BWD_TASK[0] = -7
    y, st, g, fwd = run(path)
    if canary is not None:
        assert not ops._FOLD_BWD_ROWS and float(canary.abs().max()) == 0.0
    if path == 'block':
        assert fwd.count('idf_attnblock_fwd') == 1 and not any(n.startswith('idf_attn_fwd') or n.startswith('idf_conv') for n in fwd), fwd
    else:
        assert fwd.count('idf_attn_fwd_res') == 1 and 'idf_attn_fwd' not in fwd and sum(n.startswith('idf_conv') for n in fwd) == 1, fwd
    # (a) fp32 PyTorch of the reference's block -- on the CPU: PyTorch-ROCm's own GroupNorm backward returns wrong dgamma / dbeta
    # from B * 32 groups > 4096 on (B = 129: 0.76 / 0.62 off its CPU result, dx exact; this image's torch 2.10.0+rocm7.0)
    prm = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in blk.named_parameters()}
    xr = x0.float().cpu().clone().requires_grad_(True)
    hn = F.group_norm(xr, 32, prm['group_norm.weight'], prm['group_norm.bias'], eps=1e-5)
    q, k, v = (F.conv2d(hn, prm[n + '.weight'], prm[n + '.bias']) for n in ('proj_q', 'proj_k', 'proj_v'))
    q, k, v = (t.permute(0, 2, 3, 1).reshape(B, N, 128) for t in (q, k, v))
    w = torch.softmax(torch.bmm(q, k.transpose(1, 2)) * 128 ** -0.5, dim=-1)
    hh = torch.bmm(w, v).view(B, H, H, 128).permute(0, 3, 1, 2)
    want = xr + F.conv2d(hh, prm['proj.weight'], prm['proj.bias'])
    if train:
        (want * dyw.bfloat16().float().cpu()).sum().backward()
    want = want.detach().to(DEV)
    assert rel(y, want) < 2e-2, rel(y, want)
    assert rel(y_ref, want) < 2e-2
    assert st is not None and st.shape[0] == B and st.shape[2:] == (128, 2)
    T = st.shape[1]
    assert T == (1 if (path == 'block' or H == 8 or B >= 128) else 4)
    yb = y.permute(0, 2, 3, 1).reshape(B, T, N // T, 128)
    assert rel(st[..., 0], yb.sum(2)) < 1e-5 and rel(st[..., 1], (yb * yb).sum(2)) < 1e-5
    # (b) the unfolded per-op path; gradients
    assert rel(y, y_ref) < 1e-2, rel(y, y_ref)
    if train:
        assert set(g) == set(g_ref) == set(list(prm) + ['x'])
        top = float(prm['proj_v.bias'].grad.abs().max())
        for kname in g_ref:
            if kname == 'proj_k.bias':
                # mathematically zero (a bias on k shifts every score of a row alike: softmax does not see it) -- rounding noise on
                # every path, held against the scale of a live gradient
                assert float(g[kname].abs().max()) < 0.1 * top and float(g_ref[kname].abs().max()) < 0.1 * top
                continue
            ref32 = (xr.grad if kname == 'x' else prm[kname].grad).to(DEV)
            assert rel(g[kname], ref32) < 4e-2, (kname, rel(g[kname], ref32))
            assert rel(g[kname], g_ref[kname]) < 4e-2, (kname, rel(g[kname], g_ref[kname]))


@pytest.mark.parametrize('B,C,Co,Hl', [(3, 128, 128, 8), (2, 128, 128, 16), (2, 128, 128, 32), (2, 256, 256, 16), (5, 64, 192, 8)])
def test_upsample_conv_as_four_subpixel_convs(B, C, Co, Hl, monkeypatch):
    """UpSample's forward (idf_upconv_bf16: nearest x2 + conv3x3 as four 2x2 convs on the low-resolution input with summed
    weights) against fp32 PyTorch of interpolate + conv2d on the same bf16-valued input and fp32 weights (1e-2 of the output's
    range: bf16 operands, one rounding of each summed weight), against the kernel it replaces (the up-sampling read fused into
    the 3x3 conv), its statistics partials against the output's own sums, borders included (the zero padding of the
    low-resolution tile must be the zero padding of the up-sampled image); the summed weights packed by the batched kernel
    (idf_upconv_pack_batched, through a ShadowSet) against the torch restatement bit for bit; and -- training -- the backward
    pass (the 3x3 conv's own kernels) against fp32 PyTorch autograd."""
    import torch.nn.functional as F
    from infodiffusion_amd import modules
    torch.manual_seed(3)
    up = modules.UpSample(C).to(DEV)
    if Co != C:
        up.main = torch.nn.Conv2d(C, Co, 3, stride=1, padding=1).to(DEV)
        up._cfg = modules._cfg(modules._Shadows(up.main), ops.UP2, 9, modules._ACT_NONE)
        up._cfg['shadows'].want_sub = True
    up.main.weight.data = up.main.weight.data.contiguous(memory_format=CL)
    with torch.no_grad():
        up.main.bias.add_(0.3 * rnd(1, Co).to(DEV))
    x = rnd(2, B, C, Hl, Hl).to(DEV).bfloat16().contiguous(memory_format=CL)
    dyw = rnd(4, B, Co, 2 * Hl, 2 * Hl).to(DEV)
    names = []
    orig_call = ops.call

    def counted(name, *a):
        names.append(name)
        return orig_call(name, *a)
    monkeypatch.setattr(ops, 'call', counted)
    with torch.no_grad():
        y = up(x)
    assert names.count('idf_upconv_bf16') == 1 and not any(n.startswith('idf_conv') for n in names), names
    # the batched pack kernel == the torch restatement of the sums
    sset = modules.ShadowSet(up)
    ref_sub = ops.upconv_pack(up.main.weight)
    up._cfg['shadows'].sub.zero_()
    up._cfg['shadows'].key = None
    sset.refresh(torch.bfloat16, True)
    assert torch.equal(up._cfg['shadows'].sub.view(torch.int16), ref_sub.view(torch.int16))
    if Co % 64 == 0:
        assert torch.equal(up._cfg['shadows'].subd.view(torch.int16), ops.upconv_pack(up.main.weight, dgrad=True).view(torch.int16))
    # training: same forward launch, the 3x3 conv's backward
    del names[:]
    xg = x.clone().requires_grad_(True)
    up.zero_grad()
    yt = up(xg)
    assert names.count('idf_upconv_bf16') == 1
    (yt.float() * dyw).sum().backward()
    assert torch.equal(yt.detach(), y)
    # the data gradient in the same sub-pixel form (no 3x3 conv over dy, no pool pass) where the shape is covered
    if Co % 64 == 0:
        assert names.count('idf_upconv_dgrad_bf16') == 1 and 'idf_pool2_sum' not in names, names
    monkeypatch.setattr(ops, '_UPCONV', False)
    del names[:]
    with torch.no_grad():
        y_old = up(x)
    assert 'idf_upconv_bf16' not in names
    xr = x.float().clone().requires_grad_(True)
    wr, br = up.main.weight.detach().clone().requires_grad_(True), up.main.bias.detach().clone().requires_grad_(True)
    want = F.conv2d(F.interpolate(xr, scale_factor=2, mode='nearest'), wr, br, padding=1)
    (want * dyw.bfloat16().float()).sum().backward()
    want = want.detach()
    assert y.shape == want.shape
    assert rel(y, want) < 1e-2, rel(y, want)
    assert rel(y, y_old) < 1e-2, rel(y, y_old)
    # the border rows / columns separately (a wrong halo shows there first)
    for sl in ((slice(None), slice(None), 0), (slice(None), slice(None), -1), (slice(None), slice(None), slice(None), 0),
               (slice(None), slice(None), slice(None), -1)):
        assert rel(y[sl], want[sl]) < 1e-2, sl
    st = getattr(y, '_gn', None)
    assert st is not None and st.shape[0] == B and st.shape[2:] == (Co, 2)
    T = st.shape[1]
    yb = y.float().permute(0, 2, 3, 1).reshape(B, T, (4 * Hl * Hl) // T, Co)
    assert rel(st[..., 0], yb.sum(2)) < 1e-5 and rel(st[..., 1], (yb * yb).sum(2)) < 1e-5
    assert rel(xg.grad, xr.grad) < 2e-2, rel(xg.grad, xr.grad)
    assert rel(up.main.weight.grad, wr.grad) < 2e-2 and rel(up.main.bias.grad, br.grad) < 2e-2


@pytest.mark.parametrize('B,C,H', [(3, 64, 64), (2, 128, 32), (2, 128, 16), (2, 256, 32)])
@pytest.mark.parametrize('alias', [False, True])
def test_downsample_data_gradient_by_output_parity(B, C, H, alias, monkeypatch):
    """DownSample's data gradient (idf_downconv_dgrad_bf16: per parity of the high-resolution pixel only the taps that land on a dy
    pixel -- 1, 2, 2, 4 of the 9 -- with the conv's fragment-major data-gradient weights) against fp32 PyTorch autograd of
    conv2d(stride 2, pad 1) on the same bf16-valued operands (2e-2: bf16 operands), borders separately, and against the kernel
    it replaces (the 3x3 conv over the zero-stuffed dy); `alias`: the skip connection's gradient joins in the epilogue."""
    import torch.nn.functional as F
    from infodiffusion_amd import modules
    torch.manual_seed(9)
    dn = modules.DownSample(C).to(DEV)
    dn.main.weight.data = dn.main.weight.data.contiguous(memory_format=CL)
    x = rnd(1, B, C, H, H).to(DEV).bfloat16().contiguous(memory_format=CL)
    dyw = rnd(2, B, C, H // 2, H // 2).to(DEV)
    dsk = rnd(3, B, C, H, H).to(DEV)
    names = []
    orig_call = ops.call

    def counted(name, *a):
        names.append(name)
        return orig_call(name, *a)
    monkeypatch.setattr(ops, 'call', counted)

    def run():
        del names[:]
        xg = x.clone().requires_grad_(True)
        dn.zero_grad()
        if alias:
            y, xa = dn(xg, want_alias=True)
            ((y.float() * dyw).sum() + (xa.float() * dsk).sum()).backward()
        else:
            (dn(xg).float() * dyw).sum().backward()
        return xg.grad.detach().float().clone(), list(names)

    run()                                   # asks for the fragment-major data-gradient weights (next re-pack)
    g, used = run()
    assert used.count('idf_downconv_dgrad_bf16') == 1, used
    monkeypatch.setattr(ops, '_DOWN_DGRAD', False)
    g_old, used_old = run()
    assert 'idf_downconv_dgrad_bf16' not in used_old
    xr = x.float().clone().requires_grad_(True)
    yr = F.conv2d(xr, dn.main.weight.detach(), dn.main.bias.detach(), stride=2, padding=1)
    ((yr * dyw.bfloat16().float()).sum() + ((xr * dsk.bfloat16().float()).sum() if alias else 0.0)).backward()
    want = xr.grad
    assert rel(g, want) < 2e-2, rel(g, want)
    assert rel(g, g_old) < 2e-2, rel(g, g_old)
    for sl in ((slice(None), slice(None), 0), (slice(None), slice(None), -1), (slice(None), slice(None), slice(None), 0),
               (slice(None), slice(None), slice(None), -1)):
        assert rel(g[sl], want[sl]) < 2e-2, sl


@pytest.mark.parametrize('C1,C2,H', [(128, 128, 16), (128, 64, 32), (64, 64, 64), (128, 64, 64), (256, 256, 8)])
def test_two_source_groupnorm_conv1x1_wgrad(C1, C2, H):
    """The (x, x2) pair read in place == the same kernels on the materialised concatenation: one-launch
    GroupNorm forward / backward (dx split into dx1, dx2), 1x1 conv, and the 1x1 weight gradient."""
    from infodiffusion_amd.grad_arena import GradArena, slot_of
    B, C = 3, C1 + C2
    x1 = rnd(1, B, C1, H, H).to(DEV).bfloat16().contiguous(memory_format=CL)
    x2 = rnd(2, B, C2, H, H).to(DEV).bfloat16().contiguous(memory_format=CL)
    xc = torch.cat([x1, x2], dim=1).contiguous(memory_format=CL)
    gam, bet = (1 + 0.1 * rnd(3, C)).to(DEV), (0.1 * rnd(4, C)).to(DEV)
    assert ops.gn_small_ok(x1, x2) and ops.gn_small_ok(xc)
    a2, m2, r2, sc2, sh2 = ops.gn_fused_fwd_raw(x1, gam, bet, None, None, None, 0, 0.0, 2, x2=x2)
    a1, m1, r1, sc1, sh1 = ops.gn_fused_fwd_raw(xc, gam, bet, None, None, None, 0, 0.0, 2)
    assert torch.equal(a1, a2) and torch.equal(m1, m2) and torch.equal(sc1, sc2)
    dA = rnd(5, B, C, H, H).to(DEV).bfloat16().contiguous(memory_format=CL)
    dres = rnd(6, B, C, H, H).to(DEV).bfloat16().contiguous(memory_format=CL)
    (dx1, dx2), dg2, db2, _, _ = ops.gn_fused_bwd_raw(dA, x1, gam, bet, None, None, m1, r1, sc1, sh1, None, 0, 0.0, 2,
                                                     dres=dres, x2=x2)
    dxc, dg1, db1, _, _ = ops.gn_fused_bwd_raw(dA, xc, gam, bet, None, None, m1, r1, sc1, sh1, None, 0, 0.0, 2,
                                               dres=dres)
    assert torch.equal(dx1, dxc[:, :C1]) and torch.equal(dx2, dxc[:, C1:])
    assert rel(dg2, dg1) < 1e-6 and rel(db2, db1) < 1e-6
    # 1x1 conv over the pair
    Cout = 128
    w = (rnd(7, Cout, C, 1, 1) / C ** 0.5).to(DEV)
    bias = rnd(8, Cout).to(DEV)
    wf, _ = ops.pack_weight(w, torch.bfloat16, True, False)
    y1 = ops.conv_raw(xc, wf, bias, None, None, None, None, 0, 0.0, ops.S1, 1, 0, Cout)
    y2 = ops.empty_nhwc(B, Cout, H, H, torch.bfloat16, x1.device)
    ops.call('idf_conv1x1_bf16', x1.data_ptr(), x2.data_ptr(), C1, wf.data_ptr(), bias.data_ptr(), None, y2.data_ptr(),
             B, H, H, C, Cout, None, torch.cuda.current_stream().cuda_stream)
    assert torch.equal(y1, y2)
    # weight gradient over the pair (table-driven launch, arena slots)
    dy = rnd(9, B, Cout, H, H).to(DEV).bfloat16().contiguous(memory_format=CL)
    ref_dW, ref_db = ops.conv_wgrad_bias_raw(xc, dy, ops.S1, 1, True)
    wp = torch.nn.Parameter(torch.zeros(Cout, C, 1, 1, device=DEV))
    bp = torch.nn.Parameter(torch.zeros(Cout, device=DEV))
    arena = GradArena([wp, bp])
    ops.WgradBatch._cb_queued = True
    ops.WgradBatch._task = torch._C._current_graph_task_id()
    got = ops._defer_or_launch_wgrad(x1, dy, slot_of(wp), slot_of(bp), 1, a2=x2)
    assert got is not None
    ops.WgradBatch.flush()
    assert rel(got[0], ref_dW) < 1e-5 and rel(got[1], ref_db) < 1e-5


def test_prep_u8_matches_torchvision_chain_bitwise():
    """idf_prep_u8 == ToTensor -> RandomHorizontalFlip -> Normalize(0.5, 0.5) of reference data.py:149-171,
    bit for bit, landing directly in the NHWC-dense layout the model reads."""
    g = torch.Generator().manual_seed(3)
    img = torch.randint(0, 256, (5, 16, 12, 3), generator=g, dtype=torch.uint8)       # [B, H, W, C]
    flip = torch.tensor([0, 1, 1, 0, 1], dtype=torch.uint8)
    ref = img.permute(0, 3, 1, 2).float() / 255.0
    ref = torch.where(flip.bool()[:, None, None, None], ref.flip(-1), ref)
    ref = (ref - 0.5) / 0.5
    out = ops.prep_u8(img.to(DEV), flip)
    assert out.shape == ref.shape and out.is_contiguous(memory_format=CL)
    assert torch.equal(out.cpu(), ref)
    assert torch.equal(ops.prep_u8(img.to(DEV)).cpu(), (img.permute(0, 3, 1, 2).float() / 255.0 - 0.5) / 0.5)


def _chan_sums(y):
    """(sum, sum of squares) per (image, channel) of a bf16 NHWC tensor, in double."""
    yd = y.double()
    return yd.sum(dim=(2, 3)), (yd * yd).sum(dim=(2, 3))


@pytest.mark.parametrize('case', [
    # (B, Cin, H, W, Cout, taps, mode): every tile shape of the halo family + the direct-to-LDS variant
    (2, 64, 16, 16, 64, 9, ops.S1), (3, 128, 8, 8, 128, 9, ops.S1), (33, 64, 64, 64, 64, 9, ops.S1),
    (16, 128, 32, 32, 128, 9, ops.S1), (96, 64, 64, 64, 64, 9, ops.S1), (3, 64, 64, 64, 64, 9, ops.S2),
    (2, 64, 8, 8, 64, 9, ops.UP2), (33, 128, 32, 32, 128, 9, ops.UP2), (3, 128, 16, 16, 128, 1, ops.S1),
    (33, 64, 64, 64, 128, 1, ops.S1), (2, 64, 4, 4, 72, 1, ops.S1), (48, 128, 64, 64, 128, 1, ops.S1),
    (2, 32, 32, 32, 32, 9, ops.S1),
])
def test_conv_epilogue_statistics(case):
    """st_out of the conv entry points == per-(image, channel) sums of the bf16 output, in T partials."""
    B, Cin, H, W, Cout, taps, mode = case
    k = 3 if taps == 9 else 1
    x = rnd(1, B, Cin, H, W).to(DEV).bfloat16().contiguous(memory_format=CL)
    w = (rnd(2, Cout, Cin, k, k) / (Cin * taps) ** 0.5).to(DEV)
    bias = rnd(3, Cout).to(DEV)
    Ho, Wo = ops.out_hw(mode, H, W)
    res = rnd(4, B, Cout, Ho, Wo).to(DEV).bfloat16().contiguous(memory_format=CL)
    wf, _ = ops.pack_weight(w, torch.bfloat16, True, False)
    y0 = ops.conv_raw(x, wf, bias, res, None, None, None, 0, 0.0, mode, taps, 0, Cout)
    y, st = ops.conv_raw(x, wf, bias, res, None, None, None, 0, 0.0, mode, taps, 0, Cout, want_stats=True)
    assert st is not None and st.shape == (B, ops.conv_tiles(B, Ho, Wo, Cin, Cout, mode, taps), Cout, 2)
    assert torch.equal(y, y0)
    s1, s2 = _chan_sums(y)
    got = st.double().sum(dim=1)
    assert float((got[..., 0] - s1).abs().max()) < 2e-3 * (1 + float(s1.abs().max()))
    assert float((got[..., 1] - s2).abs().max() / s2.abs().max()) < 1e-5
    # the stand-alone pass gives the same sums
    st2 = ops.gn_partials_raw(y).double().sum(dim=1)
    assert float((st2[..., 1] - s2).abs().max() / s2.abs().max()) < 1e-5
    assert float((st2[..., 0] - s1).abs().max()) < 2e-3 * (1 + float(s1.abs().max()))


@pytest.mark.parametrize('case', [
    # (B, C1, C2, H, Cout, taps, act, film, p_drop, keep)
    (2, 64, 0, 16, 64, 9, 2, True, 0.0, True), (3, 128, 0, 8, 128, 9, 2, True, 0.1, True),
    (33, 64, 0, 64, 64, 9, 2, True, 0.1, True), (16, 128, 0, 32, 128, 9, 2, False, 0.0, False),
    (96, 64, 0, 64, 64, 9, 2, True, 0.1, True),      # direct-to-LDS variant, in-LDS prologue
    (40, 128, 0, 64, 64, 9, 2, False, 0.0, False),
    (3, 128, 0, 16, 384, 1, 1, False, 0.0, True),    # AttnBlock: affine prologue + q|k|v 1x1
    (48, 128, 0, 64, 128, 1, 1, False, 0.0, True),
    (2, 128, 64, 16, 64, 9, 2, False, 0.0, True), (33, 128, 64, 64, 64, 9, 2, False, 0.0, True),   # 192-channel pairs
    (4, 128, 128, 32, 128, 9, 2, False, 0.0, True), (2, 256, 256, 8, 256, 9, 2, False, 0.0, False),
    (2, 64, 0, 32, 3, 9, 2, False, 0.0, True), (2, 32, 0, 32, 1, 9, 2, False, 0.0, False),        # tails (ragged couts)
    (5, 96, 0, 4, 32, 9, 2, True, 0.0, True),
])
def test_groupnorm_prologue_conv_one_launch(case):
    """idf_conv_gn_bf16 (statistics from the producer, GroupNorm / FiLM / SiLU / dropout applied while the conv
    stages its tile) against fp32 PyTorch, and against the two-launch path's activated tensor / coefficients."""
    B, C1, C2, H, Cout, taps, act, film, p_drop, keep = case
    C, k = C1 + C2, 3 if taps == 9 else 1
    x1 = (0.5 + 1.5 * rnd(1, B, C1, H, H)).to(DEV).bfloat16().contiguous(memory_format=CL)
    x2 = (rnd(2, B, C2, H, H) - 0.3).to(DEV).bfloat16().contiguous(memory_format=CL) if C2 else None
    xc = torch.cat([x1, x2], dim=1).contiguous(memory_format=CL) if C2 else x1
    gam, bet = (1 + 0.1 * rnd(3, C)).to(DEV), (0.1 * rnd(4, C)).to(DEV)
    ft = (0.2 * rnd(5, B, 2 * C)).to(DEV) if film else None
    fa = (0.2 * rnd(6, B, 2 * C)).to(DEV) if film else None
    w = (rnd(7, Cout, C, k, k) / (C * taps) ** 0.5).to(DEV)
    bias = rnd(8, Cout).to(DEV)
    res = rnd(9, B, Cout, H, H).to(DEV).bfloat16().contiguous(memory_format=CL) if Cout % 8 == 0 else None
    wf, _ = ops.pack_weight(w, torch.bfloat16, True, False)
    seed = torch.tensor([123456789], dtype=torch.int64, device=DEV) if p_drop else None
    assert ops.conv_gn_ok(x1, x2, taps, Cout, advice=False)      # coverage, not the policy
    st1, st2 = ops.gn_partials_raw(x1), (ops.gn_partials_raw(x2) if C2 else None)
    y, a, mean, rstd, sc, sh, st = ops.conv_gn_raw(x1, x2, st1, st2, gam, bet, ft, fa, seed, 7, p_drop, act, wf, bias, res,
                                                   Cout, taps, keep_a=keep, keep_coef=keep, want_stats=Cout % 8 == 0)
    # fp32 reference of the same op (dropout mask exported from the product's counter-based generator)
    u = F.group_norm(xc.float(), 32, gam, bet, eps=1e-5)
    if film:
        u = u * (1 + ft[:, :C, None, None]) + ft[:, C:, None, None]
        u = u * (1 + fa[:, :C, None, None]) + fa[:, C:, None, None]
    if act == 2:
        u = F.silu(u)
        if p_drop:
            m = ops.dropout_mask(seed, 7, p_drop, xc.numel()).view(B, H, H, C).permute(0, 3, 1, 2)
            u = u * m
    ref = F.conv2d(u, w, bias, padding=k // 2) + (res.float() if res is not None else 0)
    assert rel(y, ref) < 2e-2, rel(y, ref)
    if keep:
        assert rel(a, u) < 1e-2
        mu = xc.float().reshape(B, 32, -1).mean(dim=2)
        var = xc.float().reshape(B, 32, -1).var(dim=2, unbiased=False)
        assert rel(mean, mu) < 1e-5 and rel(rstd, (var + 1e-5).rsqrt()) < 1e-5
        # coefficients / activated tensor of the stand-alone GroupNorm kernels
        if ops.gn_small_ok(x1, x2):
            a0, m0, r0, sc0, sh0 = ops.gn_fused_fwd_raw(x1, gam, bet, ft, fa, seed, 7, p_drop, act, x2=x2)
            assert rel(sc, sc0) < 1e-5 and rel(sh, sh0) < 1e-5
            assert float((a.float() - a0.float()).abs().max()) <= 2 ** -7 * float(a0.float().abs().max())   # one bf16 ulp
    if st is not None:
        s1, s2 = _chan_sums(y)
        got = st.double().sum(dim=1)
        assert float((got[..., 1] - s2).abs().max() / s2.abs().max()) < 1e-5
    # statistics handed over by a producing conv instead of the stand-alone pass: same result
    if not C2 and C1 % 32 == 0 and taps == 9 and H >= 4:
        wp = (rnd(10, C1, C1, 3, 3) / (C1 * 9) ** 0.5).to(DEV)
        wpf, _ = ops.pack_weight(wp, torch.bfloat16, True, False)
        xp, stp = ops.conv_raw(x1, wpf, None, None, None, None, None, 0, 0.0, ops.S1, 9, 0, C1, want_stats=True)
        y1 = ops.conv_gn_raw(xp, None, stp, None, gam, bet, ft, fa, seed, 7, p_drop, act, wf, bias, res, Cout, taps)[0]
        y2 = ops.conv_gn_raw(xp, None, ops.gn_partials_raw(xp), None, gam, bet, ft, fa, seed, 7, p_drop, act, wf, bias, res,
                             Cout, taps)[0]
        assert rel(y1, y2) < 1e-2


@pytest.mark.parametrize('B,with_a,fc_silu', [(2, True, False), (32, True, False), (5, False, False), (32, True, True),
                                               (70, True, False)])
def test_conditioning_path_one_entry_matches_pytorch(B, with_a, fc_silu):
    """idf_temb_film_fwd / _bwd (TimeEmbedding + fc_a + every block's FiLM projections; three launches forward, four backward:
    modules.py:9-38, 269-276; models.py:298-301, 371) against fp32 PyTorch autograd of the same chain: both outputs and
    every parameter gradient + the latent's.  Ragged widths (not multiples of the 64-wide tile) on purpose."""
    import torch.nn.functional as F
    T, d_model, dim, a_dim, Nt, Na = 50, 64, 256, 32, 2 * (64 + 100), 200
    g = torch.Generator(device='cpu')
    g.manual_seed(B)
    mk = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(DEV)
    table = mk(T, d_model)
    t = torch.randint(0, T, (B,), generator=g).to(DEV)
    P = dict(W1=mk(dim, d_model, sc=d_model ** -0.5), b1=mk(dim, sc=0.1), W2=mk(dim, dim, sc=dim ** -0.5), b2=mk(dim, sc=0.1),
             Wfc=mk(dim, a_dim, sc=a_dim ** -0.5), bfc=mk(dim, sc=0.1), Wt=mk(Nt, dim, sc=dim ** -0.5), bt=mk(Nt, sc=0.1),
             Wa=mk(Na, dim, sc=dim ** -0.5), ba=mk(Na, sc=0.1))
    a = mk(B, a_dim)
    gt, ga = mk(B, Nt), mk(B, Na)

    def leaves():
        return {k: v.clone().requires_grad_(True) for k, v in P.items()}, a.clone().requires_grad_(True)

    R, ar = leaves()
    temb = F.linear(F.silu(F.linear(table[t], R['W1'], R['b1'])), R['W2'], R['b2'])
    ft_ref = F.linear(F.silu(temb), R['Wt'], R['bt'])
    loss = (ft_ref * gt).sum()
    if with_a:
        aemb = F.linear(F.silu(ar) if fc_silu else ar, R['Wfc'], R['bfc'])
        fa_ref = F.linear(F.silu(aemb), R['Wa'], R['ba'])
        loss = loss + (fa_ref * ga).sum()
    loss.backward()

    O_, ao = leaves()
    n = lambda k: O_[k] if with_a else None
    ft, fa = ops._TembFilm.apply(t, ao if with_a else None, table, O_['W1'], O_['b1'], O_['W2'], O_['b2'], n('Wfc'), n('bfc'),
                                 O_['Wt'], O_['bt'], n('Wa'), n('ba'), fc_silu, None)
    assert rel(ft, ft_ref) < 1e-5
    lo = (ft * gt).sum()
    if with_a:
        assert rel(fa, fa_ref) < 1e-5
        lo = lo + (fa * ga).sum()
    else:
        assert fa is None
    lo.backward()
    for k in P:
        if not with_a and k in ('Wfc', 'bfc', 'Wa', 'ba'):
            continue
        assert rel(O_[k].grad, R[k].grad) < 2e-5, k
    if with_a:
        assert rel(ao.grad, ar.grad) < 2e-5


@pytest.mark.parametrize('case', [
    # (B, C1, C2, H, Cout, Cs)
    (2, 128, 64, 16, 64, 64), (3, 128, 128, 8, 128, 128), (33, 128, 64, 64, 64, 64), (4, 128, 128, 32, 128, 128),
    (2, 64, 0, 16, 128, 128), (5, 256, 128, 16, 128, 72),
])
def test_shortcut_rides_in_the_first_convs_launches(case):
    """idf_conv_gn_sc_bf16 / idf_conv_dgrad_chain_sc_bf16: a ResBlock's 1x1 shortcut (modules.py:228, 248, 281) and its
    data gradient as extra blocks of the block's first 3x3 launch and of that conv's data-gradient launch -- against fp32
    PyTorch, and the main job's outputs unchanged (bitwise) by the rider."""
    B, C1, C2, H, Cout, Cs = case
    C = C1 + C2
    x1 = (0.5 + 1.5 * rnd(1, B, C1, H, H)).to(DEV).bfloat16().contiguous(memory_format=CL)
    x2 = (rnd(2, B, C2, H, H) - 0.3).to(DEV).bfloat16().contiguous(memory_format=CL) if C2 else None
    xc = torch.cat([x1, x2], dim=1).contiguous(memory_format=CL) if C2 else x1
    gam, bet = (1 + 0.1 * rnd(3, C)).to(DEV), (0.1 * rnd(4, C)).to(DEV)
    w = (rnd(7, Cout, C, 3, 3) / (C * 9) ** 0.5).to(DEV)
    bias = rnd(8, Cout).to(DEV)
    wsc = (rnd(11, Cs, C, 1, 1) / C ** 0.5).to(DEV)
    bsc = rnd(12, Cs).to(DEV)
    wf, wd = ops.pack_weight(w, torch.bfloat16, True, True)
    wsf, wsd = ops.pack_weight(wsc, torch.bfloat16, True, True)
    st1, st2 = ops.gn_partials_raw(x1), (ops.gn_partials_raw(x2) if C2 else None)
    plain = ops.conv_gn_raw(x1, x2, st1, st2, gam, bet, None, None, None, 7, 0.0, 2, wf, bias, None, Cout, 9, keep_a=True,
                            keep_coef=True, want_stats=True)
    ride = ops.conv_gn_raw(x1, x2, st1, st2, gam, bet, None, None, None, 7, 0.0, 2, wf, bias, None, Cout, 9, keep_a=True,
                           keep_coef=True, want_stats=True, shortcut=(wsf, bsc, Cs))
    y, a, mean, rstd, sc, sh, st, s = ride
    u = F.silu(F.group_norm(xc.float(), 32, gam, bet, eps=1e-5))
    assert rel(y, F.conv2d(u, w, bias, padding=1)) < 2e-2
    assert rel(s, F.conv2d(xc.float(), wsc, bsc)) < 1e-2
    assert rel(a, u) < 1e-2 and rel(sc, plain[4]) < 1e-6 and rel(sh, plain[5]) < 1e-6
    assert rel(y, plain[0]) < 1e-2          # (another kernel variant may have computed `plain`: not bitwise)
    s1, s2 = _chan_sums(y)
    assert float((st.double().sum(dim=1)[..., 1] - s2).abs().max() / s2.abs().max()) < 1e-5

    # backward: du / partials of the main job + the shortcut's data gradient
    if not ops.chain_tiles(B, H, H, Cout, C, 9) > 0 or Cs % 32:
        return
    dh = rnd(20, B, Cout, H, H).to(DEV).bfloat16().contiguous(memory_format=CL)
    ds = rnd(21, B, Cs, H, H).to(DEV).bfloat16().contiguous(memory_format=CL)
    du0, part0, _ = ops.conv_dgrad_chain_raw(dh, wd, 9, C, x=x1, x2=x2, sc=sc, sh=sh, act=2)
    du, part, dxs = ops.conv_dgrad_chain_raw(dh, wd, 9, C, x=x1, x2=x2, sc=sc, sh=sh, act=2, shortcut=(ds, wsd))
    assert torch.equal(du, du0) and torch.equal(part, part0)
    xr = xc.float().clone().requires_grad_(True)
    F.conv2d(xr, wsc, bsc).backward(ds.float())
    assert rel(dxs, xr.grad) < 1e-2


@pytest.mark.parametrize('B,C,H,film', [(3, 128, 32, True), (2, 64, 64, False), (5, 256, 8, True)])
def test_groupnorm_coefficients_from_producer_statistics(B, C, H, film):
    """idf_gn_coef_from_stats (+ idf_gn_apply): the streaming form of the GroupNorm pass for big tensors whose conv stays a
    launch of its own -- coefficients folded from the partials a producing conv left behind -- against fp32 PyTorch and the
    one-launch GroupNorm."""
    Cp = 64
    xin = rnd(1, B, Cp, H, H).to(DEV).bfloat16().contiguous(memory_format=CL)
    wp = (rnd(2, C, Cp, 3, 3) / (Cp * 9) ** 0.5).to(DEV)
    wpf, _ = ops.pack_weight(wp, torch.bfloat16, True, False)
    x, st = ops.conv_raw(xin, wpf, None, None, None, None, None, 0, 0.0, ops.S1, 9, 0, C, want_stats=True)
    gam, bet = (1 + 0.1 * rnd(3, C)).to(DEV), (0.1 * rnd(4, C)).to(DEV)
    ft = (0.2 * rnd(5, B, 2 * C)).to(DEV) if film else None
    fa = (0.2 * rnd(6, B, 2 * C)).to(DEV) if film else None
    mean, rstd, sc, sh = ops.gn_coef_from_stats_raw(st, C, H * H, gam, bet, ft, fa)
    a = ops.gn_apply_raw(x, sc, sh, None, 0, 0.0, 2)
    u = F.group_norm(x.float(), 32, gam, bet, eps=1e-5)
    if film:
        u = u * (1 + ft[:, :C, None, None]) + ft[:, C:, None, None]
        u = u * (1 + fa[:, :C, None, None]) + fa[:, C:, None, None]
    assert rel(a, F.silu(u)) < 1e-2
    mu = x.float().reshape(B, 32, -1).mean(dim=2)
    var = x.float().reshape(B, 32, -1).var(dim=2, unbiased=False)
    assert rel(mean, mu) < 1e-5 and rel(rstd, (var + 1e-5).rsqrt()) < 1e-5
    a0, m0, r0, sc0, sh0 = ops.gn_fused_fwd_raw(x, gam, bet, ft, fa, None, 0, 0.0, 2)
    assert rel(sc, sc0) < 1e-5 and rel(sh, sh0) < 1e-5
    # a skip pair (x | x2) read in place: partials of both producers, idf_gn_apply2
    x2, st2 = ops.conv_raw(xin, wpf, None, None, None, None, None, 0, 0.0, ops.S1, 9, 0, C, want_stats=True)
    x2 = (x2.float() * 0.5 + 0.25).bfloat16().contiguous(memory_format=CL)
    st2 = ops.gn_partials_raw(x2)
    g2, b2 = torch.cat([gam, gam * 0.5]), torch.cat([bet, bet + 0.1])
    m2, r2, sc2, sh2 = ops.gn_coef_from_stats_raw(st, 2 * C, H * H, g2, b2, None, None, st2=st2)
    a2 = ops.gn_apply2_raw(x, x2, sc2, sh2, None, 0, 0.0, 2)
    ref2 = F.silu(F.group_norm(torch.cat([x, x2], dim=1).float(), 32, g2, b2, eps=1e-5))
    assert rel(a2, ref2) < 1e-2
