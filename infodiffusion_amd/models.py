"""AVDM networks and the diffusion training objective on the HIP kernels.

Drop-in mirror of the reference's models.py surface: same class
names, constructor / forward / loss_fn signatures, attribute names and state_dict
keys.  Extra, optional knob: `args.act_dtype` ('fp32' default, or 'bf16') selects
the activation / weight-shadow storage type of the kernels.
"""
import os

import torch
import torch.nn as nn
from torch.nn import init

from . import ops
from .modules import (AuxResBlock, DownSample, ResBlock, ResBlock_encoder, RunCtx, TimeEmbedding, UpSample,
                      ShadowSet, _Shadows, _cfg, _ACT_NONE, _ACT_SILU, batched_film, bind_context, film_groups, fused_film)
from .utils import compute_mmd, gaussian_mixture, swiss_roll

_DTYPES = {'fp32': torch.float32, 'float32': torch.float32, 'bf16': torch.bfloat16, 'bfloat16': torch.bfloat16,
           None: torch.float32}


def _act_dtype(args):
    v = getattr(args, 'act_dtype', None)
    return v if isinstance(v, torch.dtype) else _DTYPES[v]


class _UNetSkeleton(nn.Module):
    """Down / middle / up scaffolding common to AuxiliaryUNet, Encoder and UNet
    (models.py:248-284, 432-468, 16-52).  `make(in_ch, out_ch, attn)` builds a block."""

    def _build(self, make, ch, ch_mult, attn, num_res_blocks, in_ch, out_ch, make_mid=None):
        assert all([i < len(ch_mult) for i in attn]), 'attn index out of bound'
        self.head = nn.Conv2d(in_ch, ch, kernel_size=3, stride=1, padding=1)
        self.downblocks = nn.ModuleList()
        widths, now = [ch], ch
        for level, mult in enumerate(ch_mult):
            for _ in range(num_res_blocks):
                self.downblocks.append(make(now, ch * mult, level in attn))
                now = ch * mult
                widths.append(now)
            if level != len(ch_mult) - 1:
                self.downblocks.append(DownSample(now))
                widths.append(now)
        make_mid = make_mid or make
        self.middleblocks = nn.ModuleList([make_mid(now, now, True), make_mid(now, now, False)])
        self.upblocks = nn.ModuleList()
        for level, mult in reversed(list(enumerate(ch_mult))):
            for _ in range(num_res_blocks + 1):
                self.upblocks.append(make(widths.pop() + now, ch * mult, level in attn))
                now = ch * mult
            if level != 0:
                self.upblocks.append(UpSample(now))
        assert len(widths) == 0
        self.tail = nn.Sequential(nn.GroupNorm(32, now), nn.SiLU(), nn.Conv2d(now, out_ch, 3, stride=1, padding=1))

    def _init_ends(self):
        init.xavier_uniform_(self.head.weight)
        init.zeros_(self.head.bias)

    def _init_tail(self):
        init.xavier_uniform_(self.tail[-1].weight, gain=1e-5)
        init.zeros_(self.tail[-1].bias)

    def _post(self):
        # 3x3 master weights live in [O][kh][kw][I] memory (logical OIHW shape kept): the weight-gradient
        # kernel writes that layout, so gradients and optimizer state are element-aligned with no copy
        for m in self.modules():
            if isinstance(m, nn.Conv2d) and m.kernel_size != (1, 1):
                m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
        self.ctx = RunCtx()
        bind_context(self, self.ctx)
        self._cfg_head = _cfg(_Shadows(self.head), ops.S1, 9, _ACT_NONE)
        self._cfg_tail = _cfg(_Shadows(self.tail[-1]), ops.S1, 9, _ACT_SILU)
        self._shadow_set = ShadowSet(self)
        for which in ('t', 'a'):        # parameter groups must exist before an optimizer lays out its arena
            blocks = [m for m in self._res_blocks() if hasattr(m, 'temb_proj' if which == 't' else 'aemb_proj')]
            if blocks:
                film_groups(blocks, which)

    def _res_blocks(self):
        return [m for m in list(self.downblocks) + list(self.middleblocks) + list(self.upblocks)
                if not isinstance(m, (DownSample, UpSample))]

    def _prep(self, x):
        if not x.is_cuda:
            raise RuntimeError('infodiffusion_amd runs on the GPU only: the HIP kernels have no CPU fallback')
        self._shadow_set.refresh(self.ctx.act_dtype, torch.is_grad_enabled())
        self.ctx.seed = None
        if self.training:
            self.ctx.seed = torch.randint(0, 2 ** 62, (1,), device=x.device, dtype=torch.int64)
        return x.to(self.ctx.act_dtype).contiguous(memory_format=torch.channels_last)

    def _run(self, x, block_call):
        """block_call(layer, h, **kw).  While gradients are recorded the skip list holds ALIASES of each
        tensor handed out by its down-path consumer (`want_alias`): the skip connection's gradient then
        arrives at that consumer's first op and is added in-kernel instead of by an autograd add pass."""
        h = ops.fused_conv(x, self.head.weight, self.head.bias, self._cfg_head, want_stats=True)
        alias_mode = torch.is_grad_enabled() and h.requires_grad
        skips = []
        for layer in self.downblocks:
            if alias_mode:
                h, a = (layer(h, want_alias=True) if isinstance(layer, DownSample)
                        else block_call(layer, h, want_alias=True))
                skips.append(a)
            else:
                skips.append(h)
                h = layer(h) if isinstance(layer, DownSample) else block_call(layer, h)
        first = True
        for layer in self.middleblocks:
            if first and alias_mode:
                h, a = block_call(layer, h, want_alias=True)
                skips.append(a)
            else:
                if first:
                    skips.append(h)
                h = block_call(layer, h)
            first = False
        for layer in self.upblocks:
            if isinstance(layer, UpSample):
                h = layer(h)
            else:
                h = block_call(layer, (h, skips.pop()))      # the block reads the pair in place (no torch.cat)
        assert len(skips) == 0
        gn, conv = self.tail[0], self.tail[-1]
        y = ops.fused_conv(h, conv.weight, conv.bias, self._cfg_tail, gn.weight, gn.bias, x_single_use=True)
        # blocks that met their kernels for the first time asked for other weight layouts (fragment-major shadows, the attention
        # fold): pack them now, so that the NEXT pass -- possibly a captured one -- runs the steady-state kernels on settled tables
        self._shadow_set.settle(self.ctx.act_dtype, torch.is_grad_enabled())
        return y



class UNet(_UNetSkeleton):
    """models.py:7-88 (vanilla epsilon-predictor; FiLM on t only)."""

    def __init__(self, T, ch=64, ch_mult=[1, 2, 4, 8], attn=[2], num_res_blocks=2, dropout=0.1, shape=None):
        super().__init__()
        tdim = ch * 4
        self.time_embedding = TimeEmbedding(T, ch, tdim)
        self._build(lambda i, o, at: ResBlock(in_ch=i, out_ch=o, tdim=tdim, dropout=dropout, attn=at),
                    ch, ch_mult, attn, num_res_blocks, shape[0], shape[0])
        self.initialize()
        self._post()

    def initialize(self):
        self._init_ends()
        self._init_tail()

    def forward(self, x, t):
        x = self._prep(x)
        blocks = self._res_blocks()
        if fused_film(self.time_embedding, t, blocks):
            temb = None
        else:
            temb = self.time_embedding(t)
            batched_film(blocks, temb, 't')
        return self._run(x, lambda blk, h, **kw: blk(h, temb, **kw))


class AuxiliaryUNet(_UNetSkeleton):
    """models.py:237-326."""

    def __init__(self, T, ch=64, ch_mult=[1, 2, 4, 8], attn=[2], num_res_blocks=2, dropout=0.1, a_dim=32,
                 shape=None):
        super().__init__()
        tdim = ch * 4
        self.a_dim = a_dim
        self.time_embedding = TimeEmbedding(T, ch, tdim)
        self.fc_a = nn.Linear(self.a_dim, tdim)
        self._build(lambda i, o, at: AuxResBlock(in_ch=i, out_ch=o, tdim=tdim, dropout=dropout, attn=at),
                    ch, ch_mult, attn, num_res_blocks, shape[0], shape[0])
        self.initialize()
        self._post()

    def initialize(self):
        self._init_ends()
        init.xavier_uniform_(self.fc_a.weight)
        init.zeros_(self.fc_a.bias)
        self._init_tail()

    # ---- sampling: the conditioning path once per trajectory instead of once per step
    def begin_trajectory(self, a, keep=None):
        """The latent is constant over the T steps of a sampling trajectory (/root/reference/sampling.py:92-95) and the timestep takes
        T values: TimeEmbedding + every block's FiLM_t projection for ALL timesteps as one table [T, sum 2C] (rebuilt when a weight
        it is made from changes), fc_a + every block's FiLM_a projection of THIS latent once [B, sum 2C] -- a step then gathers its
        row (one launch) instead of running the three conditioning launches.  Inference only; `keep`: a cache object of an earlier
        call whose buffers a kept step graph reads (they are refilled in place).  -> the cache object, or None (not applicable)."""
        if torch.is_grad_enabled() or a is None or not a.is_cuda:
            return None
        blocks = self._res_blocks()
        tab = self.time_embedding.timembedding[0].weight
        T = tab.shape[0]
        key = tuple((p.data_ptr(), p._version) for p in self.parameters())
        with torch.no_grad():
            tt = torch.arange(T, dtype=torch.long, device=a.device)
            if keep is not None and keep.get('key') == key and keep['out_a'].shape[0] == a.shape[0]:
                c = keep
            else:
                c = {'key': key, 'table_t': None, 'out_a': None}
            if c['table_t'] is None:
                if not fused_film(self.time_embedding, tt, blocks):
                    return None
                c['table_t'] = torch.cat([blk._film['t'] for blk in blocks], dim=1).contiguous()
                c['split_t'] = [blk._film['t'].shape[1] for blk in blocks]
            # FiLM_a of this latent (the entry computes the t half too: rows of timestep 0, discarded -- once per trajectory)
            if not fused_film(self.time_embedding, tt[:1].expand(a.shape[0]).contiguous(), blocks, a, self.fc_a, False, blocks):
                for blk in blocks:
                    blk._film.clear()
                return None
            out_a = torch.cat([blk._film['a'] for blk in blocks], dim=1)
            if c['out_a'] is None:
                c['out_a'] = out_a.contiguous()
                c['split_a'] = [blk._film['a'].shape[1] for blk in blocks]
            else:
                c['out_a'].copy_(out_a)
            for blk in blocks:
                blk._film.clear()
        self._traj = c
        return c

    def end_trajectory(self):
        self._traj = None

    def forward(self, x, t, a):
        x = self._prep(x)
        blocks = self._res_blocks()
        c = getattr(self, '_traj', None)
        if c is not None and not torch.is_grad_enabled() and torch.is_tensor(t) and t.dtype == torch.long and c['out_a'].shape[0] == x.shape[0]:
            out_t = ops.gather_rows(c['table_t'], t)
            for blk, ct, ca in zip(blocks, out_t.split(c['split_t'], dim=1), c['out_a'].split(c['split_a'], dim=1)):
                blk._film['t'], blk._film['a'] = ct, ca
            temb = aemb = None
        elif fused_film(self.time_embedding, t, blocks, a, self.fc_a, False, blocks):
            temb = aemb = None          # every block finds its FiLM pairs in `_film`
        else:
            aemb = ops.linear(a, self.fc_a.weight, self.fc_a.bias)
            temb = self.time_embedding(t)
            batched_film(blocks, temb, 't')
            batched_film(blocks, aemb, 'a')
        return self._run(x, lambda blk, h, **kw: blk(h, temb, aemb, **kw))


class BottleneckAuxUNet(_UNetSkeleton):
    """models.py:329-421 (--is_bottleneck): t-conditioned ResBlocks on the down / up paths, the latent `a`
    enters only through the two middle AuxResBlocks; fc_a = Sequential(SiLU, Linear), He-initialised."""

    def __init__(self, T, ch=64, ch_mult=[1, 2, 4, 8], attn=[2], num_res_blocks=2, dropout=0.1, a_dim=32,
                 shape=None):
        super().__init__()
        tdim = ch * 4
        self.a_dim = a_dim
        self.time_embedding = TimeEmbedding(T, ch, tdim)
        self.fc_a = nn.Sequential(nn.SiLU(), nn.Linear(self.a_dim, tdim))
        self._build(lambda i, o, at: ResBlock(in_ch=i, out_ch=o, tdim=tdim, dropout=dropout, attn=at),
                    ch, ch_mult, attn, num_res_blocks, shape[0], shape[0],
                    make_mid=lambda i, o, at: AuxResBlock(i, o, tdim, dropout, attn=at, crossattn=False))
        self.initialize()
        self._post()

    def initialize(self):
        self._init_ends()
        init.kaiming_normal_(self.fc_a[1].weight, a=0, nonlinearity='relu')
        self._init_tail()

    def forward(self, x, t, a):
        x = self._prep(x)
        blocks = self._res_blocks()
        mids = [m for m in blocks if isinstance(m, AuxResBlock)]
        if fused_film(self.time_embedding, t, blocks, a, self.fc_a[1], True, mids):
            temb = aemb = None
        else:
            aemb = ops.linear(a, self.fc_a[1].weight, self.fc_a[1].bias, silu_in=True)
            temb = self.time_embedding(t)
            batched_film(blocks, temb, 't')
            batched_film(mids, aemb, 'a')
        return self._run(x, lambda blk, h, **kw: (blk(h, temb, aemb, **kw) if isinstance(blk, AuxResBlock)
                                                  else blk(h, temb, **kw)))


class Encoder(_UNetSkeleton):
    """models.py:424-518: UNet-shaped encoder -> 1 channel -> fc -> (a, a_q, mu, log_var)."""

    def __init__(self, ch=64, ch_mult=[1, 2, 4, 8, 8], attn=[2], num_res_blocks=2, dropout=0.1, a_dim=32,
                 shape=None):
        super().__init__()
        self.shape = shape
        self.a_dim = a_dim
        self._build(lambda i, o, at: ResBlock_encoder(in_ch=i, out_ch=o, dropout=dropout, attn=at),
                    ch, ch_mult, attn, num_res_blocks, shape[0], 1)
        self.fc_a = nn.Linear(self.shape[1] * self.shape[2], self.a_dim)
        self.fc_mu = nn.Linear(self.a_dim, self.a_dim)
        self.fc_var = nn.Linear(self.a_dim, self.a_dim)
        self.initialize()
        self._post()

    def initialize(self):
        self._init_ends()
        for fc in (self.fc_a, self.fc_mu, self.fc_var):
            init.xavier_uniform_(fc.weight)
            init.zeros_(fc.bias)
        self._init_tail()

    def forward(self, x, want_q=True):
        """-> (a, a_q, mu, log_var) as models.py:510-516.  want_q=False (the caller does not read a_q: InfoDiff with
        kld_weight == 0): the reparameterisation noise is still DRAWN -- the reference draws it on every call, so the
        RNG stream keeps its order -- but a_q is None and its three elementwise launches are skipped."""
        x = self._prep(x)
        h = self._run(x, lambda blk, hh, **kw: blk(hh, **kw))
        h = torch.flatten(h, start_dim=1).float()
        a = ops.linear(h, self.fc_a.weight, self.fc_a.bias)
        mu = ops.linear(a, self.fc_mu.weight, self.fc_mu.bias)
        log_var = ops.linear(a, self.fc_var.weight, self.fc_var.bias)
        noise = torch.randn_like(mu)
        a_q = torch.addcmul(mu, noise, torch.exp(0.5 * log_var)) if want_q else None
        return a, a_q, mu, log_var


class Decoder(_UNetSkeleton):
    """models.py:521-603: fc_a(a) viewed as an image -> the UNet skeleton of ResBlock_encoder blocks ->
    reconstruction.  fc_a keeps nn.Linear's default initialisation (models.py:564-568 touches head/tail only)."""

    def __init__(self, ch=64, ch_mult=[1, 2, 4, 8], attn=[2], num_res_blocks=2, dropout=0.1, a_dim=10,
                 shape=None):
        super().__init__()
        self.a_dim = a_dim
        self.shape = shape
        self._build(lambda i, o, at: ResBlock_encoder(in_ch=i, out_ch=o, dropout=dropout, attn=at),
                    ch, ch_mult, attn, num_res_blocks, shape[0], shape[0])
        self.fc_a = nn.Linear(self.a_dim, self.shape[0] * self.shape[1] * self.shape[2])
        self.initialize()
        self._post()

    def initialize(self):
        self._init_ends()
        self._init_tail()

    def forward(self, a):
        if not a.is_cuda:
            raise RuntimeError('infodiffusion_amd runs on the GPU only: the HIP kernels have no CPU fallback')
        aemb = ops.linear(a.float(), self.fc_a.weight, self.fc_a.bias)
        h = aemb.reshape(a.shape[0], self.shape[0], self.shape[1], self.shape[2])
        return self._run(self._prep(h), lambda blk, hh, **kw: blk(hh, **kw))


def _schedule(args, device):
    """models.py:615-618: identical torch CPU ops => bitwise-identical tables."""
    T = args.diffusion_steps
    alpha_bars = torch.cumprod(1 - torch.linspace(start=args.beta1, end=args.betaT, steps=T), dim=0).to(device=device)
    betas = torch.linspace(start=args.beta1, end=args.betaT, steps=T).to(device=device)
    alphas = 1 - betas
    alpha_prev_bars = torch.cat([torch.Tensor([1]).to(device=device), alpha_bars[:-1]])
    return alpha_bars, betas, alphas, alpha_prev_bars


def _draw_idx(model, n):
    """models.py:701 / 754 draw the timesteps on the CPU and copy them over; under stream capture the draw must stay
    on the device (a blocking pageable H2D copy is not capturable)."""
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        return torch.randint(0, len(model.alpha_bars), (n,), device=model.device)
    return torch.randint(0, len(model.alpha_bars), (n,)).to(device=model.device)


def _kl_capacity(model, args, curr_epoch):
    """clamp(C_max / epochs * curr_epoch, 0, C_max) as a device scalar (models.py:662-671): the persistent scalar
    `set_epoch` maintains; set here too when the caller did not (eager use, not under capture)."""
    if getattr(model, '_C_dev', None) is None or (getattr(model, '_C_epoch', None) != curr_epoch
                                                  and not torch.cuda.is_current_stream_capturing()):
        model.set_epoch(args, curr_epoch)
    return model._C_dev


class InfoDiff(nn.Module):
    """models.py:605-723."""

    def __init__(self, args, device, shape):
        super().__init__()
        self.device = device
        self.alpha_bars, self.betas, self.alphas, self.alpha_prev_bars = _schedule(args, device)
        ch_mult = [1, 2, 4] if args.input_size == 28 else [1, 2, 2, 2]
        net = BottleneckAuxUNet if getattr(args, 'is_bottleneck', False) else AuxiliaryUNet     # models.py:623-626
        self.backbone = net(ch_mult=ch_mult, T=args.diffusion_steps, ch=args.unets_channels,
                            a_dim=args.a_dim, shape=shape)
        self.encoder = Encoder(ch_mult=ch_mult, ch=args.encoder_channels, a_dim=args.a_dim, shape=shape)
        self.mmd_weight: float = args.mmd_weight
        self.kld_weight: float = args.kld_weight
        self.verbose = getattr(args, 'verbose_loss', False)
        self.set_act_dtype(_act_dtype(args))
        self._qs_tables = ops.qsample_tables(self.alpha_bars)
        # t = 0 constants of the reconstruction term (models.py:644), fp32 like the reference
        ab0 = self.alpha_bars[0].cpu()
        self._rec_c0 = float(torch.sqrt(1 / self.alphas[0].cpu()))
        self._rec_c1 = float(self.betas[0].cpu() / torch.sqrt(1 - ab0))
        self.to(device)

    def set_act_dtype(self, dtype):
        self.act_dtype = dtype
        self.backbone.ctx.act_dtype = dtype
        self.encoder.ctx.act_dtype = dtype

    def attach_grad_sync(self, sync):
        """Data parallel: cut the gradient arena at the backbone | encoder boundary (dist.GradSync.attach) and, when the
        arena layout allows it and the loss reaches the encoder through the latent only (kld_weight == 0), cut the
        backward pass at the latent as well (`cut_latent`): the trainer then runs backbone backward, puts the backbone's
        slice on the wire, and runs the encoder's backward pass beside it.  Returns whether the cut is active."""
        self._dp_sync = sync if (sync is not None and sync.attach(self.backbone)) else None
        self.cut_latent = self._dp_sync is not None and self.kld_weight == 0
        self._latent_cut = None
        self._cut_armed = False
        return self.cut_latent

    def arm_latent_cut(self):
        """The cut is PER CALL: only the forward pass that follows this request detaches the latent (and its caller must
        `pop_latent_cut()` and continue the backward pass into the encoder: trainer.GraphedTrainStep).  Every other caller --
        a plain `loss_fn(...).backward()`, a validation pass with gradients, the tools -- gets the single-backward graph."""
        if getattr(self, '_latent_cut', None) is not None:
            raise RuntimeError('InfoDiff: the previous latent cut was never popped (its encoder backward pass did not run)')
        self._cut_armed = bool(getattr(self, 'cut_latent', False))
        return self._cut_armed

    def pop_latent_cut(self):
        """(latent as the encoder produced it, the leaf that replaced it downstream) of the last armed forward pass, or None."""
        cut, self._latent_cut = getattr(self, '_latent_cut', None), None
        self._cut_armed = False
        return cut

    def _draw_idx(self, n):
        return _draw_idx(self, n)

    def set_epoch(self, args, curr_epoch):
        """KL capacity C of this epoch (models.py:662-671, --use_C) into the device scalar the loss reads: a replayed
        hipGraph picks up the new value without a re-capture.  Call outside stream capture (the trainer does)."""
        if getattr(args, 'use_C', False):
            c = min(max(float(args.C_max) / args.epochs * curr_epoch, 0.0), float(args.C_max))
            if getattr(self, '_C_dev', None) is None:
                self._C_dev = torch.zeros((), dtype=torch.float32, device=self.device)
            self._C_dev.fill_(c)
            self._C_epoch = curr_epoch

    def loss_fn(self, args, x, idx=None, curr_epoch=0):
        if x.is_cuda and x.dim() == 4:
            x = x.contiguous(memory_format=torch.channels_last)     # once: q_sample, the encoder and the loss all read NHWC
        output, epsilon, a, mu, log_var = self.forward(x, idx=idx, get_target=True)
        if (self.mmd_weight != 0 and self.kld_weight == 0 and args.prior == 'regular' and output.is_cuda
                and args.mmd_weight == self.mmd_weight):
            # the benchmarked objective: both loss terms, the MMD against prior draws and the weighted sum as one op
            # (ops._Objective: 3 launches forward, one backward node) -- same arithmetic as the general path below
            loss, terms = ops.objective_mmd(output, epsilon, x, torch.randn_like(a, device=self.device), a, self._rec_c0,
                                            self._rec_c1, 1.0 / args.diffusion_steps, args.mmd_weight)
            if self.verbose:
                print('denoising loss:', terms[0])
                print('recon loss:', terms[1])
            return loss
        terms = ops.diff_loss(output, epsilon, x, self._rec_c0, self._rec_c1, 1.0 / args.diffusion_steps)
        loss = terms[0] + terms[1]
        if self.verbose:
            print('denoising loss:', terms[0])
            print('recon loss:', terms[1])

        def prior_samples(like):
            if args.prior == 'regular':
                return torch.randn_like(like, device=self.device)
            if args.prior == '10mix':
                return torch.FloatTensor(gaussian_mixture(args.batch_size, args.a_dim)).to(device=self.device)
            return torch.FloatTensor(swiss_roll(args.batch_size)).to(device=self.device)

        def kl_term():
            kld_loss = torch.sum(-0.5 * torch.sum(1 + log_var - mu ** 2 - log_var.exp(), dim=1), dim=0)
            if args.use_C:
                return args.kld_weight * (kld_loss - _kl_capacity(self, args, curr_epoch)).abs()
            return args.kld_weight * kld_loss

        if self.mmd_weight != 0 and self.kld_weight != 0:
            loss = torch.add(loss, compute_mmd(prior_samples(a), mu), alpha=args.mmd_weight) + kl_term()
        elif args.mmd_weight != 0:
            loss = torch.add(loss, compute_mmd(prior_samples(a), a), alpha=args.mmd_weight)
        elif args.kld_weight != 0:
            loss = loss + kl_term()
        return loss

    def forward(self, x, idx=None, a=None, get_target=False):
        if idx is None:
            idx = self._draw_idx(x.size(0))
            epsilon = torch.randn_like(x, memory_format=torch.channels_last if x.dim() == 4 else torch.preserve_format)
            x_tilde = ops.q_sample(x, epsilon, idx, self._qs_tables, self.act_dtype)
        else:
            if not torch.is_tensor(idx):
                idx = torch.full((x.size(0),), int(idx), dtype=torch.long, device=self.device)
            x_tilde = x
        use_q = self.kld_weight != 0     # models.py:714-721
        if a is None:
            # both networks run this call: ONE re-pack of every weight shadow (encoder + backbone: three launches instead of the
            # six of two per-network re-packs; each network's own _prep then finds nothing stale)
            if getattr(self, '_shadow_all', None) is None:
                self._shadow_all = ShadowSet(self)
            if x.is_cuda:
                self._shadow_all.refresh(self.backbone.ctx.act_dtype, torch.is_grad_enabled())
            a, a_q, mu, log_var = self.encoder(x, want_q=use_q)
        else:
            a_q = a
        lat = a_q if use_q else a
        if getattr(self, '_cut_armed', False) and torch.is_grad_enabled() and lat.requires_grad:
            self._cut_armed = False
            # data parallel (kld_weight == 0, so lat is a): the latent enters the backbone AND the loss (the MMD term
            # reads the returned `a`) as a leaf; whoever runs the backward pass (trainer.GraphedTrainStep) continues
            # from `leaf.grad` into the encoder once the backbone's share of the gradient exchange is on the wire
            leaf = lat.detach().requires_grad_(True)
            self._latent_cut = (lat, leaf)
            lat = a = leaf
        output = self.backbone(x_tilde, idx, lat)
        if getattr(self, '_shadow_all', None) is not None:
            self._shadow_all.settle(self.backbone.ctx.act_dtype, torch.is_grad_enabled())
        return (output, epsilon, a, mu, log_var) if get_target else output


class Diff(nn.Module):
    """models.py:726-779 with the image-space UNet (the latent MLP lives in latent.py)."""

    def __init__(self, args, device, shape):
        super().__init__()
        self.device = device
        self.alpha_bars, self.betas, self.alphas, self.alpha_prev_bars = _schedule(args, device)
        self.is_latent = args.is_latent or args.mode == 'train_latent_ddim'
        ch_mult = [1, 2, 4] if args.input_size == 28 else [1, 2, 4, 8]
        if self.is_latent:
            from .latent import LatentUNet
            self.backbone = LatentUNet(T=args.diffusion_steps, num_layers=10, dropout=0.1, shape=shape,
                                       activation='silu')
        else:
            self.backbone = UNet(ch_mult=ch_mult, T=args.diffusion_steps, ch=args.unets_channels, shape=shape)
            self.backbone.ctx.act_dtype = _act_dtype(args)
        self.act_dtype = torch.float32 if self.is_latent else _act_dtype(args)
        self._qs_tables = ops.qsample_tables(self.alpha_bars)
        self.to(device)

    def loss_fn(self, args, x, idx=None, curr_epoch=0):
        output, epsilon = self.forward(x, idx=idx, get_target=True)
        return (output.float() - epsilon).square().mean()

    def forward(self, x, idx=None, get_target=False):
        if idx is None:
            idx = _draw_idx(self, x.size(0))
            epsilon = torch.randn_like(x)
            if self.is_latent:
                x_tilde = ops.q_sample(x[:, :, None, None], epsilon[:, :, None, None], idx, self._qs_tables,
                                       torch.float32)[:, :, 0, 0]
            else:
                x_tilde = ops.q_sample(x, epsilon, idx, self._qs_tables, self.act_dtype)
        else:
            if not torch.is_tensor(idx):
                idx = torch.full((x.size(0),), int(idx), dtype=torch.long, device=self.device)
            x_tilde = x
        output = self.backbone(x_tilde, idx)
        return (output, epsilon) if get_target else output


class VAE(nn.Module):
    """models.py:781-833 (--model vae baseline): Encoder -> latent -> Decoder, both with the [1,2,4,8] widths."""

    def __init__(self, args, device, shape):
        super().__init__()
        self.device = device
        ch_mult = [1, 2, 4] if args.input_size == 28 else [1, 2, 4, 8]
        self.encoder = Encoder(ch_mult=ch_mult, ch=args.encoder_channels, a_dim=args.a_dim, shape=shape)
        self.decoder = Decoder(ch_mult=ch_mult, ch=args.encoder_channels, a_dim=args.a_dim, shape=shape)
        self.mmd_weight: float = args.mmd_weight
        self.kld_weight: float = args.kld_weight
        self.verbose = getattr(args, 'verbose_loss', False)
        self.set_act_dtype(_act_dtype(args))
        self.to(device)

    def set_act_dtype(self, dtype):
        self.act_dtype = dtype
        self.encoder.ctx.act_dtype = dtype
        self.decoder.ctx.act_dtype = dtype

    set_epoch = InfoDiff.set_epoch

    def loss_fn(self, args, x, curr_epoch=0):
        reconstruction, a_q, mu, log_var = self.forward(x, get_target=True)
        loss = ops.diff_loss(reconstruction, x, x, 1.0, 0.0, 0.0)[0]        # mean((rec - x)^2), models.py:796
        if self.verbose:
            print('reconstruction loss:', loss)
        if args.mmd_weight != 0:
            true_samples = torch.randn_like(a_q, device=self.device)
            loss = loss + args.mmd_weight * compute_mmd(true_samples, a_q)
        elif args.kld_weight != 0:
            # a batch MEAN here (models.py:807), unlike InfoDiff's sum
            kld_loss = torch.mean(-0.5 * torch.sum(1 + log_var - mu ** 2 - log_var.exp(), dim=1), dim=0)
            if args.use_C:
                loss = loss + args.kld_weight * (kld_loss - _kl_capacity(self, args, curr_epoch)).abs()
            else:
                loss = loss + args.kld_weight * kld_loss
        return loss

    def forward(self, x, get_target=False):
        a, a_q, mu, log_var = self.encoder(x)
        z = a if (self.mmd_weight == 0 and self.kld_weight == 0) else a_q      # models.py:824-831
        reconstruction = self.decoder(z)
        return (reconstruction, a_q, mu, log_var) if get_target else reconstruction
