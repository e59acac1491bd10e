"""CPU oracle for the InfoDiffusion hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

A functional, state-dict driven fp32/NCHW restatement (stock ATen ops on CPU) of
the algorithm the reference defines for the path BASELINE.json names.  Only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module; the product package `infodiffusion_amd` never does.

Parity status: PINNED.  Every function here is checked against the reference
itself (imported from /root/reference in the build container by
`tools/gen_golden.py`) and against the committed fixtures in `tests/golden/`
(`tests/test_oracle_golden.py`).  The reference ships no tests or golden
vectors of its own.

Each function cites the reference file:line it follows.  The parameter naming is
the reference's state_dict naming, so one synthetic state dict drives the
reference, this oracle and the HIP product.
"""
import math
import zlib

import torch
import torch.nn.functional as F

GN_GROUPS = 32
GN_EPS = 1e-5
DDIM_ETA = 0.01  # sampling.py:45


# --------------------------------------------------------------------------
# configuration / layout helpers
# --------------------------------------------------------------------------
class Cfg:
    """Attribute bag mirroring the `args` fields the path reads."""

    def __init__(self, **kw):
        d = dict(beta1=1e-5, betaT=1e-2, diffusion_steps=1000, input_size=64,
                 input_channels=3, is_bottleneck=False, unets_channels=64,
                 encoder_channels=64, a_dim=32, mmd_weight=0.1, kld_weight=0.0,
                 prior='regular', batch_size=32, use_C=False, C_max=25.0,
                 epochs=50, deterministic=True, model='diff', split_step=500,
                 mode='train', is_latent=False)
        d.update(kw)
        self.__dict__.update(d)

    @property
    def shape(self):
        return (self.input_channels, self.input_size, self.input_size)


def dataset_cfg(dataset, **kw):
    """data.py:63-102 -- per-dataset (C, H, W) and channel widths."""
    table = {
        'fmnist': (1, 32, 32), 'mnist': (1, 32, 32), 'dsprites': (1, 32, 32),
        'celeba': (3, 64, 64), 'cifar10': (3, 64, 32), 'chairs': (3, 32, 64),
        'ffhq': (3, 64, 64),
    }
    c, ch, size = table[dataset]
    return Cfg(input_channels=c, unets_channels=ch, encoder_channels=ch,
               input_size=size, **kw)


def ch_mult_for(cfg, vanilla=False):
    """models.py:619-622 (InfoDiff) / 743-746 (Diff)."""
    if cfg.input_size == 28:
        return [1, 2, 4]
    return [1, 2, 4, 8] if vanilla else [1, 2, 2, 2]


def unet_layout(ch, ch_mult, attn=(2,), num_res_blocks=2):
    """Block lists of the UNet skeleton shared by AuxiliaryUNet / Encoder / UNet
    (models.py:248-278, 432-462, 16-46).  Entries: ('res', cin, cout, attn) |
    ('down', c) | ('up', c)."""
    down, chs, now = [], [ch], ch
    for i, mult in enumerate(ch_mult):
        out = ch * mult
        for _ in range(num_res_blocks):
            down.append(('res', now, out, i in attn))
            now = out
            chs.append(now)
        if i != len(ch_mult) - 1:
            down.append(('down', now))
            chs.append(now)
    mid = [('res', now, now, True), ('res', now, now, False)]
    up = []
    for i, mult in reversed(list(enumerate(ch_mult))):
        out = ch * mult
        for _ in range(num_res_blocks + 1):
            up.append(('res', chs.pop() + now, out, i in attn))
            now = out
        if i != 0:
            up.append(('up', now))
    assert not chs
    return down, mid, up, now


# --------------------------------------------------------------------------
# synthetic weights protocol (so the GPU box can rebuild identical weights
# without the reference or a 90 MB checkpoint)
# --------------------------------------------------------------------------
def synth_tensor(key, shape, table_T=None):
    """Deterministic value for state-dict entry `key` of `shape`."""
    # MLPLNAct registers linear_emb twice (models.py:114-115): one value for both keys
    key = key.replace('.cond_layers.1.', '.linear_emb.')
    g = torch.Generator(device='cpu')
    g.manual_seed(zlib.crc32(key.encode()) & 0x7FFFFFFF)
    shape = tuple(shape)
    if key.endswith('timembedding.0.weight'):
        return sinusoid_table(shape[0], shape[1])
    r = torch.randn(shape, generator=g, dtype=torch.float32)
    if len(shape) == 1:
        if key.endswith('.weight'):   # GroupNorm / LayerNorm gain
            return 1.0 + 0.1 * r
        return 0.1 * r                # biases
    fan_out = shape[0]
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    rf = 1
    for s in shape[2:]:
        rf *= s
    std = math.sqrt(2.0 / (fan_in + fan_out * rf))
    return r * (1.5 * std)


def synth_state_dict(manifest):
    """manifest: list of (key, shape).  Returns {key: tensor}."""
    return {k: synth_tensor(k, s) for k, s in manifest}


# --------------------------------------------------------------------------
# schedule and embeddings
# --------------------------------------------------------------------------
def noise_schedule(beta1, betaT, T):
    """models.py:615-618 / sampling.py:12-15 (same torch CPU ops => bitwise)."""
    betas = torch.linspace(start=beta1, end=betaT, steps=T)
    alphas = 1 - betas
    alpha_bars = torch.cumprod(1 - torch.linspace(start=beta1, end=betaT, steps=T), dim=0)
    alpha_prev_bars = torch.cat([torch.Tensor([1]), alpha_bars[:-1]])
    return betas, alphas, alpha_bars, alpha_prev_bars


def sinusoid_table(T, d_model):
    """modules.py:13-20 -- interleaved [sin f0, cos f0, sin f1, cos f1, ...]."""
    freq = torch.arange(0, d_model, step=2) / torch.Tensor([d_model]) * math.log(10000)
    freq = torch.exp(-freq)
    ang = torch.arange(T).float()[:, None] * freq[None, :]
    return torch.stack([torch.sin(ang), torch.cos(ang)], dim=-1).view(T, d_model)


def timestep_embedding(t, dim, max_period=10000):
    """modules.py:41-60 -- concatenated [cos..., sin...] (LatentUNet only)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) *
                      torch.arange(start=0, end=half, dtype=torch.float32) / half)
    ang = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


# --------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------
def _gn(sd, p, x):
    return F.group_norm(x, GN_GROUPS, sd[p + '.weight'], sd[p + '.bias'], GN_EPS)


def _conv(sd, p, x, stride=1, pad=1):
    return F.conv2d(x, sd[p + '.weight'], sd[p + '.bias'], stride=stride, padding=pad)


def _lin(sd, p, x):
    return F.linear(x, sd[p + '.weight'], sd[p + '.bias'])


class Drop:
    """Dropout provider.  mode: None (eval), 'torch' (F.dropout, for timing), or
    a dict {site_index: keep-mask already scaled by 1/(1-p)}."""

    def __init__(self, mode=None, p=0.1):
        self.mode, self.p, self.site = mode, p, 0

    def __call__(self, h):
        i = self.site
        self.site += 1
        if self.mode is None:
            return h
        if self.mode == 'torch':
            return F.dropout(h, self.p, True)
        return h * self.mode[i]


def time_embedding(sd, p, t):
    """modules.py:22-27, 36-38."""
    e = sd[p + '.timembedding.0.weight'][t]
    e = _lin(sd, p + '.timembedding.1', e)
    return _lin(sd, p + '.timembedding.3', F.silu(e))


def attn_block(sd, p, x):
    """modules.py:145-164 -- single-head spatial self-attention, d = C."""
    B, C, H, W = x.shape
    h = _gn(sd, p + '.group_norm', x)
    q = _conv(sd, p + '.proj_q', h, pad=0).permute(0, 2, 3, 1).reshape(B, H * W, C)
    k = _conv(sd, p + '.proj_k', h, pad=0).reshape(B, C, H * W)
    v = _conv(sd, p + '.proj_v', h, pad=0).permute(0, 2, 3, 1).reshape(B, H * W, C)
    w = F.softmax(torch.bmm(q, k) * (int(C) ** (-0.5)), dim=-1)
    o = torch.bmm(w, v).view(B, H, W, C).permute(0, 3, 1, 2)
    return x + _conv(sd, p + '.proj', o, pad=0)


def aux_res_block(sd, p, x, temb, aemb, has_attn, drop):
    """modules.py:309-328 -- AdaGN/FiLM (t then a) conditioned block, 3 convs."""
    h = _conv(sd, p + '.block1.2', F.silu(_gn(sd, p + '.block1.0', x)))
    st, bt = torch.chunk(_lin(sd, p + '.temb_proj.1', F.silu(temb))[:, :, None, None], 2, dim=1)
    h = _gn(sd, p + '.block2.0', h) * (1 + st) + bt
    sa, ba = torch.chunk(_lin(sd, p + '.aemb_proj.1', F.silu(aemb))[:, :, None, None], 2, dim=1)
    h = h * (1 + sa) + ba
    h = _conv(sd, p + '.block2.3', drop(F.silu(h)))
    h = _conv(sd, p + '.block3.3', drop(F.silu(_gn(sd, p + '.block3.0', h))))
    sc = _conv(sd, p + '.shortcut', x, pad=0) if (p + '.shortcut.weight') in sd else x
    h = h + sc
    return attn_block(sd, p + '.attn', h) if has_attn else h


def res_block(sd, p, x, temb, has_attn, drop):
    """modules.py:247-258 -- vanilla ResBlock (FiLM on t only)."""
    h = _conv(sd, p + '.block1.2', F.silu(_gn(sd, p + '.block1.0', x)))
    st, bt = torch.chunk(_lin(sd, p + '.temb_proj.1', F.silu(temb))[:, :, None, None], 2, dim=1)
    h = _gn(sd, p + '.block2.0', h) * (1 + st) + bt
    h = _conv(sd, p + '.block2.3', drop(F.silu(h)))
    h = _conv(sd, p + '.block3.3', drop(F.silu(_gn(sd, p + '.block3.0', h))))
    sc = _conv(sd, p + '.shortcut', x, pad=0) if (p + '.shortcut.weight') in sd else x
    h = h + sc
    return attn_block(sd, p + '.attn', h) if has_attn else h


def res_block_encoder(sd, p, x, has_attn, drop):
    """modules.py:361-366."""
    h = _conv(sd, p + '.block1.2', F.silu(_gn(sd, p + '.block1.0', x)))
    h = _conv(sd, p + '.block2.3', drop(F.silu(_gn(sd, p + '.block2.0', h))))
    sc = _conv(sd, p + '.shortcut', x, pad=0) if (p + '.shortcut.weight') in sd else x
    h = h + sc
    return attn_block(sd, p + '.attn', h) if has_attn else h


def down_sample(sd, p, x):
    """modules.py:73-75."""
    return _conv(sd, p + '.main', x, stride=2)


def up_sample(sd, p, x):
    """modules.py:88-93."""
    return _conv(sd, p + '.main', F.interpolate(x, scale_factor=2.0, mode='nearest'))


def _unet_body(sd, p, x, layout, block_fn):
    down, mid, up, _ = layout
    h = _conv(sd, p + '.head', x)
    hs = [h]
    for i, e in enumerate(down):
        q = '%s.downblocks.%d' % (p, i)
        h = block_fn(q, h, e[3]) if e[0] == 'res' else down_sample(sd, q, h)
        hs.append(h)
    for i, e in enumerate(mid):
        h = block_fn('%s.middleblocks.%d' % (p, i), h, e[3])
    for i, e in enumerate(up):
        q = '%s.upblocks.%d' % (p, i)
        if e[0] == 'res':
            h = block_fn(q, torch.cat([h, hs.pop()], dim=1), e[3])
        else:
            h = up_sample(sd, q, h)
    assert not hs
    return _conv(sd, p + '.tail.2', F.silu(_gn(sd, p + '.tail.0', h)))


def aux_unet(sd, p, x, t, a, ch, ch_mult, drop=None):
    """AuxiliaryUNet.forward, models.py:296-326."""
    drop = drop or Drop(None)
    aemb = _lin(sd, p + '.fc_a', a)
    temb = time_embedding(sd, p + '.time_embedding', t)
    return _unet_body(sd, p, x, unet_layout(ch, ch_mult),
                      lambda q, h, at: aux_res_block(sd, q, h, temb, aemb, at, drop))


def bottleneck_unet(sd, p, x, t, a, ch, ch_mult, drop=None):
    """BottleneckAuxUNet.forward, models.py:383-421: t-only ResBlocks on the down / up paths, the two
    AuxResBlocks (conditioned on a) in the middle only; fc_a = Sequential(SiLU, Linear) (models.py:336-339)."""
    drop = drop or Drop(None)
    aemb = _lin(sd, p + '.fc_a.1', F.silu(a))
    temb = time_embedding(sd, p + '.time_embedding', t)

    def block(q, h, at):
        if '.middleblocks.' in q:
            return aux_res_block(sd, q, h, temb, aemb, at, drop)
        return res_block(sd, q, h, temb, at, drop)
    return _unet_body(sd, p, x, unet_layout(ch, ch_mult), block)


def backbone(sd, cfg, x, idx, a, drop=None):
    """InfoDiff.backbone as constructed by models.py:623-626."""
    fn = bottleneck_unet if getattr(cfg, 'is_bottleneck', False) else aux_unet
    return fn(sd, 'backbone', x, idx, a, cfg.unets_channels, ch_mult_for(cfg), drop)


def vanilla_unet(sd, p, x, t, ch, ch_mult, drop=None):
    """UNet.forward, models.py:62-88 (as it would run without the stray kwarg)."""
    drop = drop or Drop(None)
    temb = time_embedding(sd, p + '.time_embedding', t)
    return _unet_body(sd, p, x, unet_layout(ch, ch_mult),
                      lambda q, h, at: res_block(sd, q, h, temb, at, drop))


def encoder(sd, p, x, ch, ch_mult, drop=None, reparam_noise=None):
    """Encoder.forward, models.py:488-518 -> (a, a_q, mu, log_var)."""
    drop = drop or Drop(None)
    h = _unet_body(sd, p, x, unet_layout(ch, ch_mult),
                   lambda q, hh, at: res_block_encoder(sd, q, hh, at, drop))
    a = _lin(sd, p + '.fc_a', torch.flatten(h, start_dim=1))
    mu = _lin(sd, p + '.fc_mu', a)
    log_var = _lin(sd, p + '.fc_var', a)
    if reparam_noise is None:
        reparam_noise = torch.randn_like(mu)
    a_q = mu + reparam_noise * torch.exp(0.5 * log_var)
    return a, a_q, mu, log_var


def decoder(sd, p, a, ch, ch_mult, shape, drop=None):
    """Decoder.forward, models.py:570-603: fc_a(a) reshaped to an image (NCHW order) -> the UNet
    skeleton of ResBlock_encoder blocks -> reconstruction with shape[0] channels."""
    drop = drop or Drop(None)
    h = _lin(sd, p + '.fc_a', a).reshape(a.shape[0], shape[0], shape[1], shape[2])
    return _unet_body(sd, p, h, unet_layout(ch, ch_mult),
                      lambda q, hh, at: res_block_encoder(sd, q, hh, at, drop))


def vae_forward(sd, cfg, x, drop=None, reparam_noise=None):
    """VAE.forward(get_target=True), models.py:821-833.  Encoder and decoder both use the
    [1,2,4,8] widths (models.py:785-790); the decoder reads `a` only when both weights are 0."""
    drop = drop or Drop(None)
    mult = ch_mult_for(cfg, vanilla=True)
    a, a_q, mu, log_var = encoder(sd, 'encoder', x, cfg.encoder_channels, mult, drop, reparam_noise)
    z = a if (cfg.mmd_weight == 0 and cfg.kld_weight == 0) else a_q
    rec = decoder(sd, 'decoder', z, cfg.encoder_channels, mult, cfg.shape, drop)
    return rec, a_q, mu, log_var


def vae_loss(sd, cfg, x, prior=None, drop=None, reparam_noise=None, curr_epoch=0):
    """VAE.loss_fn, models.py:793-819.  MMD is taken on a_q; the KL term is a batch MEAN here
    (models.py:807) where InfoDiff's is a sum; with both weights non-zero only MMD is added."""
    rec, a_q, mu, log_var = vae_forward(sd, cfg, x, drop, reparam_noise)
    terms = {}
    loss = (rec - x).square().mean()
    terms['recon'] = loss
    if cfg.mmd_weight != 0:
        terms['mmd'] = cfg.mmd_weight * compute_mmd(prior, a_q)
        loss = loss + terms['mmd']
    elif cfg.kld_weight != 0:
        k = torch.mean(-0.5 * torch.sum(1 + log_var - mu ** 2 - log_var.exp(), dim=1), dim=0)
        if cfg.use_C:
            cmax = torch.tensor([cfg.C_max])
            C = torch.clamp(cmax / cfg.epochs * curr_epoch, torch.tensor([0.0]), cmax)
            terms['kld'] = cfg.kld_weight * (k - C.squeeze(0)).abs()
        else:
            terms['kld'] = cfg.kld_weight * k
        loss = loss + terms['kld']
    terms.update(rec=rec, a_q=a_q, mu=mu, log_var=log_var)
    return loss, terms


def latent_unet(sd, p, x, t, a_dim, num_layers=10, drop=None, time_ch=64):
    """LatentUNet.forward, models.py:223-234 with MLPLNAct.forward 147-163."""
    drop = drop or Drop(None)
    temb = timestep_embedding(t, time_ch)
    temb = _lin(sd, p + '.time_embed.2', F.silu(_lin(sd, p + '.time_embed.0', temb)))
    h = x
    for i in range(num_layers):
        q = '%s.layers.%d' % (p, i)
        if i >= 1:
            h = torch.cat([h, x], dim=1)
        h = _lin(sd, q + '.linear', h)
        last = i == num_layers - 1
        if not last:
            cond = _lin(sd, q + '.linear_emb', F.silu(temb))
            h = h * (1 + cond)
            h = F.layer_norm(h, h.shape[-1:], sd[q + '.norm.weight'], sd[q + '.norm.bias'], 1e-5)
            h = drop(F.silu(h))
    return h


# --------------------------------------------------------------------------
# diffusion model level
# --------------------------------------------------------------------------
def q_sample(alpha_bars, x, idx, eps):
    """models.py:702-704."""
    ab = alpha_bars[idx][:, None, None, None]
    return torch.sqrt(ab) * x + torch.sqrt(1 - ab) * eps


def compute_kernel(x, y):
    """utils.py:74-83 in closed form: exp(-||x_i-y_j||^2 / dim^2)."""
    dim = x.shape[1]
    d2 = ((x[:, None, :] - y[None, :, :]) ** 2).mean(dim=2)
    return torch.exp(-d2 / dim * 1.0)


def compute_mmd(x, y):
    """utils.py:85-90."""
    return compute_kernel(x, x).mean() + compute_kernel(y, y).mean() - 2 * compute_kernel(x, y).mean()


def kld(mu, log_var):
    """models.py:663 / 687."""
    return torch.sum(-0.5 * torch.sum(1 + log_var - mu ** 2 - log_var.exp(), dim=1), dim=0)


def infodiff_eps(sd, cfg, x, t_int, a):
    """InfoDiff.forward sampling path (idx given, a given), models.py:705-723."""
    idx = torch.full((x.size(0),), int(t_int), dtype=torch.long)
    return backbone(sd, cfg, x, idx, a)


def infodiff_train_forward(sd, cfg, x, idx, eps, sched, drop=None, reparam_noise=None):
    """InfoDiff.forward training path (models.py:700-723) with the random draws
    (idx, eps, reparam noise) supplied by the caller."""
    drop = drop or Drop(None)
    x_tilde = q_sample(sched[2], x, idx, eps)
    a, a_q, mu, log_var = encoder(sd, 'encoder', x, cfg.encoder_channels, ch_mult_for(cfg),
                                  drop, reparam_noise)
    use_q = cfg.kld_weight != 0   # models.py:714-721
    out = backbone(sd, cfg, x_tilde, idx, a_q if use_q else a, drop)
    return out, x_tilde, a, a_q, mu, log_var


def infodiff_loss(sd, cfg, x, idx, eps, sched, prior=None, drop=None, reparam_noise=None,
                  curr_epoch=0):
    """InfoDiff.loss_fn, models.py:632-696.  Returns (total, dict of terms)."""
    betas, alphas, alpha_bars, _ = sched
    out, x_tilde, a, a_q, mu, log_var = infodiff_train_forward(
        sd, cfg, x, idx, eps, sched, drop, reparam_noise)
    terms = {}
    loss = (out - eps).square().mean()
    terms['denoise'] = loss
    x0 = torch.sqrt(1 / alphas[0]) * (x - betas[0] / torch.sqrt(1 - alpha_bars[0]) * out)
    rec = (x0 - x).square().mean() / cfg.diffusion_steps
    terms['recon'] = rec
    loss = loss + rec

    def kl_term():
        k = kld(mu, log_var)
        if cfg.use_C:
            cmax = torch.tensor([cfg.C_max])
            C = torch.clamp(cmax / cfg.epochs * curr_epoch, torch.tensor([0.0]), cmax)
            return cfg.kld_weight * (k - C.squeeze(0)).abs()
        return cfg.kld_weight * k

    if cfg.mmd_weight != 0 and cfg.kld_weight != 0:
        terms['mmd'] = cfg.mmd_weight * compute_mmd(prior, mu)
        terms['kld'] = kl_term()
        loss = loss + terms['mmd'] + terms['kld']
    elif cfg.mmd_weight != 0:
        terms['mmd'] = cfg.mmd_weight * compute_mmd(prior, a)
        loss = loss + terms['mmd']
    elif cfg.kld_weight != 0:
        terms['kld'] = kl_term()
        loss = loss + terms['kld']
    terms.update(out=out, x_tilde=x_tilde, a=a, mu=mu, log_var=log_var)
    return loss, terms


# --------------------------------------------------------------------------
# samplers (sampling.py:23-101, identical arithmetic order)
# --------------------------------------------------------------------------
def ddpm_step(sched, x, eps_hat, idx, noise):
    """sampling.py:29-37."""
    betas, alphas, ab, apb = sched
    sqrt_tilde_beta = torch.sqrt((1 - apb[idx]) / (1 - ab[idx]) * betas[idx])
    mu = torch.sqrt(1 / alphas[idx]) * (x - betas[idx] / torch.sqrt(1 - ab[idx]) * eps_hat)
    return mu + sqrt_tilde_beta * noise


def ddim_step(sched, x, eps_hat, idx, noise):
    """sampling.py:52-59 (as written: alpha_prev_bars[idx], eta = 0.01)."""
    betas, alphas, ab, apb = sched
    x0 = (x - torch.sqrt(1 - apb[idx]) * eps_hat) / torch.sqrt(apb[idx])
    if idx == 0:
        return x0
    sigma = DDIM_ETA * torch.sqrt((1 - apb[idx - 1]) / (1 - ab[idx - 1])) * torch.sqrt(betas[idx - 1])
    x = torch.sqrt(apb[idx - 1]) * x0 + torch.sqrt(1 - apb[idx - 1] - sigma ** 2) * eps_hat
    return x + sigma * noise


def ddim_reverse_step(sched, x, eps_hat, idx):
    """sampling.py:71-72."""
    _, _, _, apb = sched
    x0 = (x - torch.sqrt(1 - apb[idx]) * eps_hat) / torch.sqrt(apb[idx])
    return torch.sqrt(apb[idx + 1]) * x0 + torch.sqrt(1 - apb[idx + 1]) * eps_hat


def sample_loop(sched, eps_fn, xT, deterministic, noises):
    """DiffusionProcess.sampling (sampling.py:89-101).  `noises[idx]` is the
    N(0,1) draw used at step idx (ignored where the reference draws none).
    Returns the list of every intermediate x."""
    T = len(sched[2])
    x, trace = xT, []
    for idx in reversed(range(T)):
        if deterministic:
            e = eps_fn(x, idx)
            x = ddim_step(sched, x, e, idx, None if idx == 0 else noises[idx])
        else:
            nz = torch.zeros_like(x) if idx == 0 else noises[idx]
            e = eps_fn(x, idx)
            x = ddpm_step(sched, x, e, idx, nz)
        trace.append(x)
    return trace


def reverse_sample_loop(sched, eps_fn, x0):
    """DiffusionProcess.reverse_sampling (sampling.py:62-73, 81-87)."""
    T = len(sched[2])
    x, trace = x0, []
    for idx in range(T - 1):
        if idx > 0:
            x = ddim_reverse_step(sched, x, eps_fn(x, idx), idx)
        trace.append(x)
    return trace
