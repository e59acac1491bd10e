"""Latent denoiser of the latent-DDIM sampler (reference models.py:91-234: MLPLNAct,
LatentUNet) on the HIP kernels: every Linear is `idf_bgemm`, and
`x*(1+cond) -> LayerNorm -> SiLU -> Dropout` is one fused row kernel."""
import torch
import torch.nn as nn
from torch.nn import init

from . import ops
from ._lib import call
from .modules import timestep_embedding


class _LnSilu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lin, cond, g, b, seed, salt, p_drop):
        lin, cond = lin.contiguous(), cond.contiguous()
        R, Wd = lin.shape
        y = torch.empty_like(lin)
        stats = torch.empty((R, 2), dtype=torch.float32, device=lin.device)
        p = p_drop if seed is not None else 0.0
        call('idf_ln_silu_fwd', ops._p(lin), ops._p(cond), ops._p(g), ops._p(b), ops._p(y), ops._p(stats), R, Wd, 1e-5,
             ops._p(seed), salt, float(p), ops._st())
        ctx.k = (salt, p)
        ctx.save_for_backward(lin, cond, g, b, stats, seed)
        return y

    @staticmethod
    def backward(ctx, dy):
        lin, cond, g, b, stats, seed = ctx.saved_tensors
        salt, p = ctx.k
        R, Wd = lin.shape
        dlin, dcond = torch.empty_like(lin), torch.empty_like(lin)
        dgb = torch.empty((R, 2 * Wd), dtype=torch.float32, device=lin.device)
        call('idf_ln_silu_bwd', ops._p(lin), ops._p(cond), ops._p(g), ops._p(b), ops._p(stats), ops._p(dy.contiguous()),
             ops._p(dlin), ops._p(dcond), ops._p(dgb), R, Wd, ops._p(seed), salt, float(p), ops._st())
        s = ops.colsum_raw(dgb)
        return dlin, dcond, s[:Wd], s[Wd:], None, None, None


class MLPLNAct(nn.Module):
    """models.py:91-163 (same parameters, incl. the doubly-registered `linear_emb`)."""

    def __init__(self, in_channels, out_channels, norm, use_cond, activation=None, cond_channels=None,
                 condition_bias=0, dropout=0):
        super().__init__()
        self.activation = activation
        self.act = nn.SiLU() if activation is not None else nn.Identity()
        self.condition_bias = condition_bias
        self.use_cond = use_cond
        self.linear = nn.Linear(in_channels, out_channels)
        if self.use_cond:
            self.linear_emb = nn.Linear(cond_channels, out_channels)
            self.cond_layers = nn.Sequential(self.act, self.linear_emb)
        self.norm = nn.LayerNorm(out_channels) if norm else nn.Identity()
        self.dropout = nn.Dropout(dropout) if dropout > 0 else nn.Identity()
        self.p_drop = dropout
        for m in self.modules():
            if isinstance(m, nn.Linear) and activation in ('relu', 'silu'):
                init.kaiming_normal_(m.weight, a=0, nonlinearity='relu')
            elif isinstance(m, nn.Linear) and activation == 'leaky_relu':
                init.kaiming_normal_(m.weight, a=0.2, nonlinearity='leaky_relu')

    def forward(self, x, cond=None, seed=None, salt=0):
        h = ops.linear(x, self.linear.weight, self.linear.bias)
        if not self.use_cond:          # last layer: plain Linear (norm / act / dropout are Identity)
            return h
        c = ops.linear(cond, self.linear_emb.weight, self.linear_emb.bias, silu_in=self.activation is not None)
        if self.condition_bias != 1:
            c = c + (self.condition_bias - 1.0)
        return _LnSilu.apply(h, c, self.norm.weight, self.norm.bias, seed if self.training else None, salt,
                             self.p_drop)


class LatentUNet(nn.Module):
    """models.py:166-234: 10-layer skip-MLP on [B, a_dim] latents."""

    def __init__(self, T, num_layers=10, dropout=0.1, shape=None, activation='silu', num_time_emb_channels=64,
                 num_time_layers=2):
        super().__init__()
        self.num_time_emb_channels = num_time_emb_channels
        self.shape = shape
        d = shape[-1]
        layers = []
        for i in range(num_time_layers):
            layers.append(nn.Linear(num_time_emb_channels if i == 0 else d, d))
            if i < num_time_layers - 1:
                layers.append(nn.SiLU())
        self.time_embed = nn.Sequential(*layers)
        self.skip_layers = list(range(1, num_layers))
        self.layers = nn.ModuleList([])
        for i in range(num_layers):
            if i == 0:
                cfg = dict(activation=activation, norm=True, use_cond=True, a=d, b=d * 4, dropout=dropout)
            elif i == num_layers - 1:
                cfg = dict(activation=None, norm=False, use_cond=False, a=d * 4, b=d, dropout=0)
            else:
                cfg = dict(activation='silu', norm=True, use_cond=True, a=d * 4, b=d * 4, dropout=dropout)
            a = cfg['a'] + (d if i in self.skip_layers else 0)
            self.layers.append(MLPLNAct(a, cfg['b'], norm=cfg['norm'], activation=cfg['activation'],
                                        cond_channels=d, use_cond=cfg['use_cond'], condition_bias=1,
                                        dropout=cfg['dropout']))

    def forward(self, x, t):
        if not x.is_cuda:
            raise RuntimeError('infodiffusion_amd runs on the GPU only: the HIP kernels have no CPU fallback')
        x = x.float().contiguous()
        temb = timestep_embedding(t, self.num_time_emb_channels)
        te = ops.linear(temb, self.time_embed[0].weight, self.time_embed[0].bias)
        temb = ops.linear(te, self.time_embed[2].weight, self.time_embed[2].bias, silu_in=True)
        seed = torch.randint(0, 2 ** 62, (1,), device=x.device, dtype=torch.int64) if self.training else None
        h = x
        for i, layer in enumerate(self.layers):
            if i in self.skip_layers:
                h = torch.cat([h, x], dim=1)
            h = layer(h, cond=temb, seed=seed, salt=1000 + i)
        return h
