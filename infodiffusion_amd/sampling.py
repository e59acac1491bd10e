"""DDPM / DDIM denoising loops on the HIP kernels (mirror of the reference's
sampling.py surface: same classes, constructor and method signatures).

Each step = one network evaluation + ONE fused update kernel (`idf_sampler_step`)
that gathers the step's scalars from a table built once with the reference's own fp32
expressions and applies them in the reference's operation order, so no host scalar math, no tiny-op swarm and no
per-step H2D copies remain.  The state `x` is kept in fp32.

Small batches (the reference's own eval / interpolate / disentangle flows run 10-16 images) are host-bound when
every step's ~250 launches are issued from Python, so there ONE step (timestep read from a device counter that
the step itself advances) is captured into a hipGraph and replayed for the inner steps; first and last step run
eagerly.  Large batches are GPU-bound and stay eager (a capture would only cost its private memory pool).
"""
import os
import sys

import torch

from . import knobs, ops

_DDPM, _DDIM, _REV = 0, 1, 2
ETA = 0.01     # sampling.py:45
GRAPH = knobs.flag('IDF_SAMPLER_GRAPH')
TRAJ_CACHE = knobs.flag('IDF_TRAJ_CACHE')     # the backbone's conditioning path once per trajectory (models.AuxiliaryUNet.begin_trajectory)
STRICT_GRAPH = knobs.flag('IDF_SAMPLER_GRAPH_STRICT')     # a failed capture raises instead of stepping eagerly
GRAPH_MIN_STEPS = 8
GRAPH_MAX_PIXELS = knobs.num('IDF_SAMPLER_GRAPH_MAXPIX')   # batch x H x W up to which a step is replayed (256 CelebA images:
                                                                                     # 5.5 % of an eagerly issued B = 256 step is host gap)


def _tables(args, device):
    """sampling.py:12-15 (same torch CPU ops; betas/alphas also moved to the device)."""
    T = args.diffusion_steps
    betas = torch.linspace(start=args.beta1, end=args.betaT, steps=T)
    alphas = 1 - betas
    alpha_bars = torch.cumprod(1 - torch.linspace(start=args.beta1, end=args.betaT, steps=T), dim=0)
    alpha_prev_bars = torch.cat([torch.Tensor([1]), alpha_bars[:-1]])
    return tuple(t.to(device=device).contiguous() for t in (betas, alphas, alpha_bars, alpha_prev_bars))


class _ProcessBase:
    def _init_common(self, args, device):
        self.betas, self.alphas, self.alpha_bars, self.alpha_prev_bars = _tables(args, device)
        self.deterministic = args.deterministic
        self.a_dim = args.a_dim
        self.model = args.model
        self.device = device
        self._steps = torch.arange(len(self.alpha_bars), dtype=torch.long, device=device)
        # what the loops did with their inner steps: captures that worked, captures that fell back to eager stepping, replays
        self.graph_stats = {'captured': 0, 'fallback': 0, 'replays': 0}
        self._graphs = {}
        self._coef = ops.sampler_coef_table(self.betas, self.alphas, self.alpha_bars, self.alpha_prev_bars, ETA)

    # hooks (tests override to inject the reference's noise draws)
    def _randn_like(self, x):
        return torch.randn_like(x)

    def _sched(self):
        return (self.betas, self.alphas, self.alpha_bars, self.alpha_prev_bars)

    def _update(self, x, eps_hat, idx, mode, noise):
        x = x.float()
        if x.dim() == 4:
            x = x.contiguous(memory_format=torch.channels_last)
            eps_hat = eps_hat.contiguous(memory_format=torch.channels_last)
            if noise is not None:
                noise = noise.float().contiguous(memory_format=torch.channels_last)
        else:
            x, eps_hat = x.contiguous(), eps_hat.contiguous()
        xo, _ = ops.sampler_step(x, eps_hat, noise, self._steps[idx:idx + 1], self._coef[mode], mode)
        return xo

    def _update_t(self, x, eps_hat, idx_t, mode, noise):
        """As _update with the timestep in a 1-element device tensor."""
        x = x.float()
        if x.dim() == 4:
            x = x.contiguous(memory_format=torch.channels_last)
            eps_hat = eps_hat.contiguous(memory_format=torch.channels_last)
            if noise is not None:
                noise = noise.float().contiguous(memory_format=torch.channels_last)
        else:
            x, eps_hat = x.contiguous(), eps_hat.contiguous()
        return ops.sampler_step(x, eps_hat, noise, idx_t, self._coef[mode], mode)[0]

    def _graph_ok(self, x, steps):
        if not (GRAPH and x.is_cuda) or torch.is_grad_enabled() or '_randn_like' in self.__dict__:
            return False
        per = x[0].numel() // (x.shape[1] if x.dim() == 4 else 1)
        return steps >= GRAPH_MIN_STEPS and x.shape[0] * per <= GRAPH_MAX_PIXELS \
            and not torch.cuda.is_current_stream_capturing()

    def _graph_ident(self):
        """What a kept step graph bakes in besides the shapes: the network object(s), their activation dtype, the addresses of
        their weight shadows (ShadowSet.tkey changes when shadows are reallocated: dtype / device / layout changes) and the
        kernel-selection switches -- a change of any of them must not replay the stale graph."""
        sig = [ops.switch_state()]
        keep = getattr(self, '_traj_keep', None)        # the conditioning cache a kept graph reads (refilled in place per trajectory)
        if keep is not None:
            sig.append((keep['table_t'].data_ptr(), keep['out_a'].data_ptr()))
        for v in vars(self).values():
            if isinstance(v, torch.nn.Module):
                sig.append(id(v))
                for m in v.modules():
                    sig.append(getattr(m, 'act_dtype', None) if hasattr(m, 'act_dtype') else None)
                    for name in ('_shadow_set', '_shadow_all'):
                        ss = m.__dict__.get(name)
                        if ss is not None:
                            sig.append(hash(ss.tkey))
        return tuple(sig)

    def release_graphs(self):
        """Drop the kept step graph(s) and with them the private pool that holds a network evaluation's activations (a B = 256
        CelebA evaluation: GBs) -- call when a loop that samples now and then goes back to training."""
        self._graphs.clear()

    def _graphed(self, x, eps_fn, mode, first, delta, count, eps_of=None, a=None):
        """Generator over `count` steps idx = first, first+delta, ...: the step is captured once and replayed.
        x must come from an eagerly executed step (weight shadows / allocator already warm).
        eps_of(a) -> eps_fn: given, the captured step reads the latent from a buffer of its own and the graph is kept for the
        next call with the same shapes (the reference's eval flows sample batch after batch: run.py:255-259, 284-287)."""
        key = (mode, first, delta, count, tuple(x.shape), None if a is None else tuple(a.shape))
        if eps_of is not None:
            key = key + (self._graph_ident(),)
        ent = self._graphs.get(key) if eps_of is not None else None
        if ent is not None:
            xs, idx_t, a_s, g = ent
            xs.copy_(x)
            idx_t.fill_(first)
            if a_s is not None:
                a_s.copy_(a)
            for _ in range(count):
                g.replay()
                self.graph_stats['replays'] += 1
                yield xs.clone()
            return
        xs = x.float().clone()
        if xs.dim() == 4:
            xs = xs.contiguous(memory_format=torch.channels_last)
        idx_t = torch.full((1,), first, dtype=torch.long, device=xs.device)
        n = xs.shape[0]
        a_s = None
        if eps_of is not None:
            a_s = a.clone() if a is not None else None
            eps_fn = eps_of(a_s)

        def body():
            t = idx_t.expand(n).contiguous()
            if mode == _DDPM:
                noise = torch.randn_like(xs)          # reference order: noise before the network call
                eps_hat = eps_fn(xs, t)
            else:
                eps_hat = eps_fn(xs, t)
                noise = torch.randn_like(xs) if mode == _DDIM else None
            xs.copy_(self._update_t(xs, eps_hat, idx_t, mode, noise))
            idx_t.add_(delta)

        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, capture_error_mode='thread_local'):
                body()
        except Exception as e:  # noqa: BLE001
            print('sampler step capture failed (%s: %s); stepping eagerly' % (type(e).__name__, str(e)[:200]),
                  file=sys.stderr)
            torch.cuda.synchronize()
            g = None
            idx_t.fill_(first)
            self.graph_stats['fallback'] += 1
            if STRICT_GRAPH:
                raise
        else:
            self.graph_stats['captured'] += 1
            if eps_of is not None:
                self._graphs.clear()            # one resident graph (its private pool holds a network evaluation's activations)
                self._graphs[key] = (xs, idx_t, a_s, g)
        for _ in range(count):
            if g is not None:
                g.replay()
                self.graph_stats['replays'] += 1
            else:
                body()
            yield xs.clone()

    def _loop(self, x, eps_fn, deterministic, eps_of=None, a=None):
        T = len(self.alpha_bars)
        mode = _DDIM if deterministic else _DDPM
        graphed = self._graph_ok(x, T - 2)
        for idx in reversed(range(T)):
            if graphed and idx == T - 2:
                for x in self._graphed(x, eps_fn, mode, idx, -1, T - 2, eps_of, a):      # steps T-2 ... 1
                    yield x
            if graphed and 0 < idx < T - 1:
                continue
            if deterministic:
                eps_hat = eps_fn(x, idx)
                noise = None if idx == 0 else self._randn_like(x)
                x = self._update(x, eps_hat, idx, _DDIM, noise)
            else:
                noise = None if idx == 0 else self._randn_like(x)   # reference draws before the net call
                eps_hat = eps_fn(x, idx)
                x = self._update(x, eps_hat, idx, _DDPM, noise)
            yield x

    def _reverse_loop(self, x, eps_fn):
        T = len(self.alpha_bars)
        graphed = self._graph_ok(x, T - 3)
        for idx in range(T - 1):
            if graphed and idx == 2:
                for x in self._graphed(x, eps_fn, _REV, idx, 1, T - 3):       # steps 2 ... T-2
                    yield x
            if graphed and idx >= 2:
                continue
            if idx > 0:
                x = self._update(x, eps_fn(x, idx), idx, _REV, None)
            yield x


class DiffusionProcess(_ProcessBase):
    """sampling.py:3-101."""

    def __init__(self, args, diffusion_fn, device, shape):
        self._init_common(args, device)
        self.shape = shape
        self.diffusion_fn = diffusion_fn.to(device=device)

    def _eps(self, a):
        if self.model == 'vanilla':
            return lambda x, idx: self.diffusion_fn(x, idx)
        return lambda x, idx: self.diffusion_fn(x, idx, a)

    def _ddpm_one_diffusion_step(self, x, a=None):
        return self._loop(x, self._eps(a), False)

    def _ddim_one_diffusion_step(self, x, a=None):
        return self._loop(x, self._eps(a), True)

    def _ddim_one_reverse_diffusion_step(self, x, a=None):
        return self._reverse_loop(x, self._eps(a))

    def _one_diffusion_step(self, sample, a=None, deterministic=False):
        """The trajectory x_T -> x_0 under ONE latent: the backbone's conditioning path (TimeEmbedding, fc_a, every block's FiLM
        projections) runs once per trajectory, not once per step (models.AuxiliaryUNet.begin_trajectory; the latent is constant over
        the T steps, sampling.py:92-95)."""
        bb = getattr(self.diffusion_fn, 'backbone', None)
        if not TRAJ_CACHE or a is None or bb is None or not hasattr(bb, 'begin_trajectory') or torch.is_grad_enabled():
            return self._loop(sample, self._eps(a), bool(deterministic), self._eps, a)

        def run():
            cache = bb.begin_trajectory(a, getattr(self, '_traj_keep', None))
            self._traj_keep = cache
            try:
                yield from self._loop(sample, self._eps(a), bool(deterministic), self._eps, a)
            finally:
                bb.end_trajectory()
        return run()

    @torch.no_grad()
    def reverse_sampling(self, x0, a=None):
        # sampling.py:84 passes only `sample`: `a` is dropped and the model re-encodes x_t each step
        final = x0
        for sample in self._ddim_one_reverse_diffusion_step(x0):
            final = sample
        return final

    @torch.no_grad()
    def sampling(self, sampling_number=16, xT=None, a=None):
        if xT is None:
            xT = torch.randn([sampling_number, *self.shape]).to(device=self.device)
        if self.model != 'vanilla' and a is None:
            a = torch.randn([sampling_number, self.a_dim]).to(device=self.device)
        final = xT
        for sample in self._one_diffusion_step(sample=xT, a=a, deterministic=self.deterministic):
            final = sample
        return final


class TwoPhaseDiffusionProcess(_ProcessBase):
    """sampling.py:104-204.  As executed by the reference, `t` is captured by value
    (sampling.py:199-202) and stays 0 <= split_step, so EVERY step calls
    diffusion_fn_2(x, idx) -- reproduced here."""

    def __init__(self, args, diffusion_fn_1, diffusion_fn_2, device, shape):
        self._init_common(args, device)
        self.shape = shape
        self.split_step = args.split_step
        self.mode = args.mode
        self.diffusion_fn_1 = diffusion_fn_1.to(device=device)
        self.diffusion_fn_2 = diffusion_fn_2.to(device=device)

    def _eps(self, a, t):
        if t <= self.split_step:
            return lambda x, idx: self.diffusion_fn_2(x, idx)
        return lambda x, idx: self.diffusion_fn_1(x, idx, a)

    def _one_diffusion_step(self, sample, a=None, deterministic=False, t=None):
        return self._loop(sample, self._eps(a, t), bool(deterministic))

    @torch.no_grad()
    def reverse_sampling(self, x0, a=None):
        final = x0
        for sample in self._reverse_loop(x0, lambda x, idx: self.diffusion_fn_1(x, idx, None)):
            final = sample
        return final

    @torch.no_grad()
    def sampling(self, sampling_number=16, xT=None, a=None):
        if xT is None:
            xT = torch.randn([sampling_number, *self.shape]).to(device=self.device)
        if a is None:
            a = torch.randn([sampling_number, self.a_dim]).to(device=self.device)
        final = xT
        for sample in self._one_diffusion_step(sample=xT, a=a, deterministic=self.deterministic, t=0):
            final = sample
        return final


class LatentDiffusionProcess(_ProcessBase):
    """sampling.py:207-292: the same loops on [B, a_dim] latents."""

    def __init__(self, args, diffusion_fn, device):
        self._init_common(args, device)
        self.split_step = args.split_step
        self.mode = args.mode
        self.diffusion_fn = diffusion_fn.to(device=device)

    def _one_diffusion_step(self, sample, deterministic=False):
        return self._loop(sample, lambda x, idx: self.diffusion_fn(x, idx), bool(deterministic))

    @torch.no_grad()
    def reverse_sampling(self, x0):
        final = x0
        for sample in self._reverse_loop(x0, lambda x, idx: self.diffusion_fn(x, idx)):
            final = sample
        return final

    @torch.no_grad()
    def sampling(self, sampling_number=16, xT=None):
        if xT is None:
            xT = torch.randn([sampling_number, self.a_dim]).to(device=self.device)
        final = xT
        for sample in self._one_diffusion_step(sample=xT, deterministic=self.deterministic):
            final = sample
        return final
