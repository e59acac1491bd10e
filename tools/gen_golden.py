#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (imported from
/root/reference, CPU) on seeded inputs with the synthetic-weights protocol, and
cross-check the oracle restatement (oracle/infodiff_oracle.py) against it.

Runs only in the build container (the reference does not travel).  Fixtures hold
data only: inputs, expected outputs, and (key, shape) manifests.

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py
"""
import contextlib
import io
import json
import os
import sys
import types

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference')

import numpy as np
import torch

import models as R_models          # reference
import modules as R_modules        # reference
import sampling as R_sampling      # reference
import utils as R_utils            # reference

from oracle import infodiff_oracle as O

GOLD = os.path.join(ROOT, 'tests', 'golden')
os.makedirs(GOLD, exist_ok=True)
torch.set_num_threads(8)


def args_for(cfg):
    return types.SimpleNamespace(**cfg.__dict__)


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def load_synth(module):
    sd = module.state_dict()
    man = [(k, tuple(v.shape)) for k, v in sd.items()]
    syn = O.synth_state_dict(man)
    module.load_state_dict(syn, strict=True)
    return man, syn


def maxrel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def check(name, got, want, tol=2e-5):
    e = maxrel(got, want)
    print('  %-40s rel err %.2e' % (name, e))
    assert e <= tol, (name, e)


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        out[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    np.savez_compressed(os.path.join(GOLD, name + '.npz'), **out)
    print('wrote', name, '%.1f KB' % (os.path.getsize(os.path.join(GOLD, name + '.npz')) / 1024))


def rnd(seed, *shape):
    g = torch.Generator(device='cpu')
    g.manual_seed(seed)
    return torch.randn(*shape, generator=g)


# ------------------------------------------------------------------ schedule
def gen_schedule():
    out = {}
    for T in (100, 1000):
        a = types.SimpleNamespace(beta1=1e-5, betaT=1e-2, diffusion_steps=T, deterministic=True,
                                  a_dim=4, model='diff')
        p = R_sampling.DiffusionProcess(a, torch.nn.Identity(), 'cpu', (1, 2, 2))
        b, al, ab, apb = O.noise_schedule(1e-5, 1e-2, T)
        assert torch.equal(b, p.betas) and torch.equal(ab, p.alpha_bars) and torch.equal(apb, p.alpha_prev_bars)
        out['betas_%d' % T] = p.betas
        out['alpha_bars_%d' % T] = p.alpha_bars
        out['alpha_prev_bars_%d' % T] = p.alpha_prev_bars
        te = R_modules.TimeEmbedding(T, 64, 256)
        tab = te.timembedding[0].weight.detach()
        assert torch.equal(tab, O.sinusoid_table(T, 64))
        rows = [0, 1, 2, T // 2, T - 1]
        out['table_rows_%d' % T] = torch.tensor(rows)
        out['table_%d' % T] = tab[rows]
    tt = torch.tensor([0, 1, 17, 999])
    assert torch.equal(R_modules.timestep_embedding(tt, 64), O.timestep_embedding(tt, 64))
    out['tse_t'] = tt
    out['tse'] = R_modules.timestep_embedding(tt, 64)
    save('schedule', **out)


# -------------------------------------------------------------------- blocks
def gen_blocks():
    out = {}
    man_all = {}
    tdim = 256

    def run_block(tag, mod, fn_oracle, x, extra=(), seed=0):
        man, syn = load_synth(mod)
        man_all[tag] = [(k, list(s)) for k, s in man]
        mod.eval()
        x = x.clone().requires_grad_(True)
        y = mod(x, *extra)
        gy = rnd(seed + 7, *y.shape)
        (y * gy).sum().backward()
        yo = fn_oracle(syn, x.detach())
        check(tag, yo, y.detach())
        out[tag + '.x'] = x.detach()
        out[tag + '.y'] = y.detach()
        out[tag + '.gy'] = gy
        out[tag + '.gx'] = x.grad
        # two representative parameter grads
        named = dict(mod.named_parameters())
        small = ('block2.0.weight', 'block2.0.bias', 'block3.3.bias', 'shortcut.weight', 'main.bias',
                 'temb_proj.1.bias', 'aemb_proj.1.bias', 'proj_k.bias', 'group_norm.weight')
        big = {'aux64': ('block1.2.weight',), 'attn128': ('proj_q.weight',), 'down64': ('main.weight',),
               'up64': ('main.weight',), 'enc64_128': ('block2.3.weight',)}.get(tag, ())
        for k, prm in named.items():
            if prm.grad is not None and 'crossattn' not in k and any(k.endswith(e) for e in small + big):
                out[tag + '.g.' + k] = prm.grad.detach()

    temb = rnd(11, 2, tdim)
    aemb = rnd(12, 2, tdim)
    out['temb'] = temb
    out['aemb'] = aemb
    nod = O.Drop(None)

    m = R_modules.AuxResBlock(64, 64, tdim, 0.1, attn=False)
    run_block('aux64', m, lambda sd, x: O.aux_res_block(sd, '', x, temb, aemb, False, nod)
              if False else O.aux_res_block({('.' + k): v for k, v in sd.items()}, '', x, temb, aemb, False, nod),
              rnd(1, 2, 64, 8, 8), (temb, aemb))
    m = R_modules.AuxResBlock(192, 64, tdim, 0.1, attn=False)
    run_block('aux192_64', m, lambda sd, x: O.aux_res_block({('.' + k): v for k, v in sd.items()}, '', x, temb, aemb, False, nod),
              rnd(2, 2, 192, 8, 8), (temb, aemb))
    m = R_modules.AuxResBlock(128, 128, tdim, 0.1, attn=True)
    run_block('aux128_attn', m, lambda sd, x: O.aux_res_block({('.' + k): v for k, v in sd.items()}, '', x, temb, aemb, True, nod),
              rnd(3, 2, 128, 8, 8), (temb, aemb))
    m = R_modules.ResBlock_encoder(64, 128, 0.1, attn=False)
    run_block('enc64_128', m, lambda sd, x: O.res_block_encoder({('.' + k): v for k, v in sd.items()}, '', x, False, nod),
              rnd(4, 2, 64, 8, 8))
    m = R_modules.ResBlock(64, 64, tdim, 0.1, attn=False)
    run_block('res64', m, lambda sd, x: O.res_block({('.' + k): v for k, v in sd.items()}, '', x, temb, False, nod),
              rnd(5, 2, 64, 8, 8), (temb,))
    m = R_modules.AttnBlock(128)
    run_block('attn128', m, lambda sd, x: O.attn_block({('.' + k): v for k, v in sd.items()}, '', x),
              rnd(6, 2, 128, 8, 8))
    m = R_modules.DownSample(64)
    run_block('down64', m, lambda sd, x: O.down_sample({('.' + k): v for k, v in sd.items()}, '', x),
              rnd(7, 2, 64, 8, 8))
    m = R_modules.UpSample(64)
    run_block('up64', m, lambda sd, x: O.up_sample({('.' + k): v for k, v in sd.items()}, '', x),
              rnd(8, 2, 64, 4, 4))
    save('blocks', **out)
    with open(os.path.join(GOLD, 'blocks_manifest.json'), 'w') as f:
        json.dump(man_all, f)


# ----------------------------------------------------------------------- MMD
def gen_mmd():
    out = {}
    for tag, (n, m, d) in {'b32d32': (32, 32, 32), 'b32d256': (32, 32, 256), 'b7d5': (7, 7, 5)}.items():
        x = rnd(100 + d, n, d)
        y = (rnd(200 + d, m, d) * 0.7 + 0.1).requires_grad_(True)
        v = R_utils.compute_mmd(x, y)
        v.backward()
        check('mmd ' + tag, O.compute_mmd(x, y.detach()), v.detach(), 1e-5)
        out[tag + '.x'], out[tag + '.y'], out[tag + '.v'], out[tag + '.gy'] = x, y.detach(), v.detach(), y.grad
    save('mmd', **out)


# ---------------------------------------------------------------- full model
def gen_model(tag, cfg, B, seed, grads=True):
    a = args_for(cfg)
    shape = cfg.shape
    torch.manual_seed(0)
    model = R_models.InfoDiff(a, 'cpu', shape)
    man, syn = load_synth(model)
    with open(os.path.join(GOLD, 'manifest_%s.json' % tag), 'w') as f:
        json.dump([(k, list(s)) for k, s in man], f)
    model.eval()   # dropout off: bitwise-comparable training arithmetic
    g = torch.Generator(device='cpu')
    g.manual_seed(seed)
    x = torch.rand(B, *shape, generator=g) * 2 - 1

    # reproduce the reference's RNG draw order
    torch.manual_seed(seed)
    idx = torch.randint(0, cfg.diffusion_steps, (B,))
    eps = torch.randn_like(x)
    reparam = torch.randn(B, cfg.a_dim)
    prior = torch.randn(B, cfg.a_dim)

    torch.manual_seed(seed)
    model.zero_grad()
    loss = quiet(model.loss_fn, args=a, x=x)
    if grads:
        loss.backward()
    sched = O.noise_schedule(cfg.beta1, cfg.betaT, cfg.diffusion_steps)
    lo, terms = O.infodiff_loss(syn, cfg, x, idx, eps, sched, prior=prior, reparam_noise=reparam)
    check(tag + ' loss', lo, loss.detach(), 1e-5)

    torch.manual_seed(seed)
    out_ref, eps_ref, a_ref, mu_ref, lv_ref = model.forward(x, get_target=True)
    assert torch.equal(eps_ref, eps)
    check(tag + ' out', terms['out'], out_ref.detach())
    check(tag + ' a', terms['a'], a_ref.detach())

    res = dict(x=x, idx=idx, eps=eps, reparam=reparam, prior=prior, loss=loss.detach(),
               out=out_ref.detach(), a=a_ref.detach(), mu=mu_ref.detach(), log_var=lv_ref.detach(),
               x_tilde=terms['x_tilde'], denoise=terms['denoise'], recon=terms['recon'],
               mmd=terms.get('mmd', torch.zeros(())))
    if grads:
        named = dict(model.named_parameters())
        pick = ['backbone.head.weight', 'backbone.head.bias', 'backbone.fc_a.weight', 'backbone.fc_a.1.weight',
                'backbone.middleblocks.0.aemb_proj.1.weight', 'backbone.middleblocks.1.block2.3.weight',
                'backbone.downblocks.0.block1.2.weight', 'backbone.downblocks.0.block2.0.weight',
                'backbone.downblocks.0.temb_proj.1.weight', 'backbone.downblocks.0.aemb_proj.1.bias',
                'backbone.downblocks.3.shortcut.weight', 'backbone.downblocks.2.main.weight',
                'backbone.middleblocks.0.attn.proj_q.weight', 'backbone.middleblocks.0.attn.proj.weight',
                'backbone.upblocks.3.main.weight', 'backbone.upblocks.14.block3.3.weight',
                'backbone.tail.0.weight', 'backbone.tail.2.weight', 'backbone.tail.2.bias',
                'backbone.time_embedding.timembedding.1.weight',
                'encoder.head.weight', 'encoder.fc_a.weight', 'encoder.tail.2.weight',
                'encoder.downblocks.0.block1.2.weight', 'encoder.middleblocks.0.attn.proj_v.bias']
        gn = 0.0
        for k, prm in named.items():
            if prm.grad is not None:
                gn += float(prm.grad.double().pow(2).sum())
        res['grad_norm'] = torch.tensor(gn ** 0.5)
        for k in pick:
            if k in named and named[k].grad is not None:
                res['g.' + k] = named[k].grad.detach()
        nograd = [k for k, prm in named.items() if prm.requires_grad and prm.grad is None]
        with open(os.path.join(GOLD, 'nograd_%s.json' % tag), 'w') as f:
            json.dump(nograd, f)

    # sampling-path epsilon prediction: model(x, idx:int, a)
    a_in = rnd(seed + 1, B, cfg.a_dim)
    xs = rnd(seed + 2, B, *shape)
    with torch.no_grad():
        e17 = model(xs, 17, a_in)
    check(tag + ' eps(t=17)', O.infodiff_eps(syn, cfg, xs, 17, a_in), e17)
    res.update(samp_x=xs, samp_a=a_in, samp_eps17=e17)

    # real-model sampler traces on a short schedule
    for det in (True, False):
        cfg_s = O.Cfg(**{**cfg.__dict__, 'diffusion_steps': 4, 'deterministic': det})
        a_s = args_for(cfg_s)
        torch.manual_seed(0)
        m_s = R_models.InfoDiff(a_s, 'cpu', shape)
        man_s, syn_s = load_synth(m_s)
        m_s.eval()
        proc = R_sampling.DiffusionProcess(a_s, m_s, 'cpu', shape)
        xT = rnd(seed + 3, 2, *shape)
        a2 = rnd(seed + 4, 2, cfg.a_dim)
        torch.manual_seed(seed + 5)
        with torch.no_grad():
            trace_ref = list(proc._one_diffusion_step(xT, a2, det))
        # replay the noise draws
        torch.manual_seed(seed + 5)
        noises = {}
        for i in reversed(range(4)):
            if i != 0:
                noises[i] = torch.randn_like(xT)
        sched_s = O.noise_schedule(cfg.beta1, cfg.betaT, 4)
        with torch.no_grad():
            trace_o = O.sample_loop(sched_s, lambda xx, i: O.infodiff_eps(syn_s, cfg_s, xx, i, a2), xT, det, noises)
        for k in range(4):
            check('%s sampler det=%s step%d' % (tag, det, k), trace_o[k], trace_ref[k], 5e-5)
        key = 'ddim' if det else 'ddpm'
        res[key + '.xT'] = xT
        res[key + '.a'] = a2
        res[key + '.noise'] = torch.stack([noises[i] for i in (3, 2, 1)])
        res[key + '.trace'] = torch.stack(trace_ref)
        if det and cfg.kld_weight == 0:   # with kld != 0 the re-encode draws fresh reparam noise
            with torch.no_grad():
                rt = list(proc._ddim_one_reverse_diffusion_step(xT))   # a=None: encoder re-run (quirk 3)
            res['ddim.rev_trace'] = torch.stack(rt)
            with torch.no_grad():
                ro = O.reverse_sample_loop(
                    sched_s, lambda xx, i: O.backbone(syn_s, cfg, xx, torch.full((2,), i, dtype=torch.long),
                                                      O.encoder(syn_s, 'encoder', xx, cfg.encoder_channels,
                                                                O.ch_mult_for(cfg))[0]), xT)
            for k in range(len(rt)):
                check('%s reverse step%d' % (tag, k), ro[k], rt[k], 5e-5)
    save('model_' + tag, **res)


# ------------------------------------------------------------------ VAE baseline
def gen_vae(tag, cfg, B, seed):
    """models.py:781-833 + Decoder 521-603: loss terms, reconstruction, picked gradients, decoder(a)."""
    a = args_for(cfg)
    shape = cfg.shape
    torch.manual_seed(0)
    model = R_models.VAE(a, 'cpu', shape)
    man, syn = load_synth(model)
    with open(os.path.join(GOLD, 'manifest_%s.json' % tag), 'w') as f:
        json.dump([(k, list(s)) for k, s in man], f)
    model.eval()
    g = torch.Generator(device='cpu')
    g.manual_seed(seed)
    x = torch.rand(B, *shape, generator=g) * 2 - 1
    torch.manual_seed(seed)     # draw order in loss_fn: reparam noise (encoder), then the prior samples
    reparam = torch.randn(B, cfg.a_dim)
    prior = torch.randn(B, cfg.a_dim)
    torch.manual_seed(seed)
    model.zero_grad()
    loss = quiet(model.loss_fn, args=a, x=x)
    loss.backward()
    lo, terms = O.vae_loss(syn, cfg, x, prior=prior, reparam_noise=reparam)
    check(tag + ' loss', lo, loss.detach(), 1e-5)
    torch.manual_seed(seed)
    rec, a_q, mu, lv = model.forward(x, get_target=True)
    check(tag + ' rec', terms['rec'], rec.detach())
    check(tag + ' a_q', terms['a_q'], a_q.detach())
    res = dict(x=x, reparam=reparam, prior=prior, loss=loss.detach(), rec=rec.detach(), a_q=a_q.detach(),
               mu=mu.detach(), log_var=lv.detach(), recon=terms['recon'],
               mmd=terms.get('mmd', torch.zeros(())), kld=terms.get('kld', torch.zeros(())))
    named = dict(model.named_parameters())
    gn = 0.0
    for k, prm in named.items():
        if prm.grad is not None:
            gn += float(prm.grad.double().pow(2).sum())
    res['grad_norm'] = torch.tensor(gn ** 0.5)
    for k in ['decoder.fc_a.weight', 'decoder.head.weight', 'decoder.downblocks.0.block1.2.weight',
              'decoder.middleblocks.0.attn.proj.weight', 'decoder.upblocks.0.shortcut.weight',
              'decoder.upblocks.11.main.weight', 'decoder.tail.2.weight', 'decoder.tail.0.bias',
              'encoder.head.weight', 'encoder.fc_a.weight', 'encoder.fc_mu.weight', 'encoder.fc_var.bias',
              'encoder.downblocks.3.block2.3.weight', 'encoder.tail.2.weight']:
        if k in named and named[k].grad is not None:
            res['g.' + k] = named[k].grad.detach()
    nograd = [k for k, prm in named.items() if prm.requires_grad and prm.grad is None]
    with open(os.path.join(GOLD, 'nograd_%s.json' % tag), 'w') as f:
        json.dump(nograd, f)
    # generation path used by run.py:261-263: decoder(randn)
    a_in = rnd(seed + 1, B, cfg.a_dim)
    with torch.no_grad():
        dec = model.decoder(a_in)
    check(tag + ' decoder(a)', O.decoder(syn, 'decoder', a_in, cfg.encoder_channels,
                                         O.ch_mult_for(cfg, vanilla=True), shape), dec)
    res.update(dec_a=a_in, dec_out=dec)
    save('model_' + tag, **res)


# ------------------------------------------------------------------ priors / KL capacity
def gen_priors_and_capacity():
    """utils.py:11-40 (the '10mix' / 'roll' priors, host numpy RNG) and the --use_C branch of
    InfoDiff.loss_fn (models.py:662-671) at a non-zero epoch."""
    np.random.seed(5)
    mix = R_utils.gaussian_mixture(6, 8)
    np.random.seed(6)
    roll = R_utils.swiss_roll(7)
    res = dict(mix=mix, roll=roll)
    cfg = O.dataset_cfg('fmnist', a_dim=16, mmd_weight=0.1, kld_weight=0.01, use_C=True, C_max=25.0, epochs=20)
    a = args_for(cfg)
    torch.manual_seed(0)
    model = R_models.InfoDiff(a, 'cpu', cfg.shape)
    man, syn = load_synth(model)
    model.eval()
    B, seed, epoch = 2, 71, 3
    g = torch.Generator(device='cpu')
    g.manual_seed(seed)
    x = torch.rand(B, *cfg.shape, generator=g) * 2 - 1
    torch.manual_seed(seed)
    idx = torch.randint(0, cfg.diffusion_steps, (B,))
    eps = torch.randn_like(x)
    reparam = torch.randn(B, cfg.a_dim)
    prior = torch.randn(B, cfg.a_dim)
    torch.manual_seed(seed)
    loss = quiet(model.loss_fn, args=a, x=x, curr_epoch=epoch)
    loss.backward()
    sched = O.noise_schedule(cfg.beta1, cfg.betaT, cfg.diffusion_steps)
    lo, terms = O.infodiff_loss(syn, cfg, x, idx, eps, sched, prior=prior, reparam_noise=reparam, curr_epoch=epoch)
    check('use_C loss', lo, loss.detach(), 1e-5)
    named = dict(model.named_parameters())
    res.update(x=x, idx=idx, eps=eps, reparam=reparam, prior=prior, loss=loss.detach(), kld=terms['kld'],
               epoch=torch.tensor(epoch))
    res['g.encoder.fc_mu.weight'] = named['encoder.fc_mu.weight'].grad
    res['g.encoder.fc_var.bias'] = named['encoder.fc_var.bias'].grad
    save('priors_capacity', **res)


# ------------------------------------------------------------------ remaining loss / forward branches
BRANCHES = [('plain', dict(mmd_weight=0.0, kld_weight=0.0)), ('kld_only', dict(mmd_weight=0.0, kld_weight=0.01))]


def gen_branches():
    """models.py:648-696 / 714-721: no auxiliary term (backbone on a), KL only (backbone on a_q)."""
    res = {}
    for tag, kw in BRANCHES:
        cfg = O.dataset_cfg('fmnist', a_dim=16, **kw)
        a = args_for(cfg)
        torch.manual_seed(0)
        model = R_models.InfoDiff(a, 'cpu', cfg.shape)
        man, syn = load_synth(model)
        model.eval()
        B, seed = 2, 81
        g = torch.Generator(device='cpu')
        g.manual_seed(seed)
        x = torch.rand(B, *cfg.shape, generator=g) * 2 - 1
        torch.manual_seed(seed)
        idx = torch.randint(0, cfg.diffusion_steps, (B,))
        eps = torch.randn_like(x)
        reparam = torch.randn(B, cfg.a_dim)
        torch.manual_seed(seed)
        loss = quiet(model.loss_fn, args=a, x=x)
        loss.backward()
        sched = O.noise_schedule(cfg.beta1, cfg.betaT, cfg.diffusion_steps)
        lo, terms = O.infodiff_loss(syn, cfg, x, idx, eps, sched, prior=None, reparam_noise=reparam)
        check(tag + ' loss', lo, loss.detach(), 1e-5)
        named = dict(model.named_parameters())
        res.update({tag + '.x': x, tag + '.idx': idx, tag + '.eps': eps, tag + '.reparam': reparam,
                    tag + '.loss': loss.detach(), tag + '.out': terms['out'],
                    tag + '.g.encoder.fc_a.weight': named['encoder.fc_a.weight'].grad,
                    tag + '.g.backbone.fc_a.weight': named['backbone.fc_a.weight'].grad})
        res[tag + '.has_mu_grad'] = torch.tensor(named['encoder.fc_mu.weight'].grad is not None)
    save('loss_branches', **res)


# ------------------------------------------------------------ stub samplers
def gen_sampler_stub():
    out = {}
    for T in (4, 10):
        for det in (True, False):
            a = types.SimpleNamespace(beta1=1e-5, betaT=1e-2, diffusion_steps=T, deterministic=det,
                                      a_dim=4, model='diff')

            class Stub(torch.nn.Module):
                def forward(self, x, idx, a=None):
                    return 0.1 * x + 0.01 * idx
            proc = R_sampling.DiffusionProcess(a, Stub(), 'cpu', (3, 8, 8))
            xT = rnd(T, 3, 3, 8, 8)
            torch.manual_seed(T + 1)
            tr = list(proc._one_diffusion_step(xT, None, det))
            torch.manual_seed(T + 1)
            noises = {i: torch.randn_like(xT) for i in reversed(range(T)) if i != 0}
            sched = O.noise_schedule(1e-5, 1e-2, T)
            to = O.sample_loop(sched, lambda x, i: 0.1 * x + 0.01 * i, xT, det, noises)
            for k in range(T):
                check('stub T=%d det=%s step %d' % (T, det, k), to[k], tr[k], 1e-6)
            tag = 'T%d_%s' % (T, 'ddim' if det else 'ddpm')
            out[tag + '.xT'] = xT
            out[tag + '.noise'] = torch.stack([noises[i] for i in reversed(range(1, T))])
            out[tag + '.trace'] = torch.stack(tr)
            if det:
                rt = list(proc._ddim_one_reverse_diffusion_step(xT))
                ro = O.reverse_sample_loop(sched, lambda x, i: 0.1 * x + 0.01 * i, xT)
                for k in range(len(rt)):
                    check('stub reverse T=%d step %d' % (T, k), ro[k], rt[k], 1e-6)
                out[tag + '.rev_trace'] = torch.stack(rt)
    save('sampler_stub', **out)


# ---------------------------------------------------------- latent denoiser
def gen_latent():
    cfg = O.Cfg(a_dim=32, is_latent=True, diffusion_steps=1000, input_size=32)
    a = args_for(cfg)
    torch.manual_seed(0)
    m = R_models.Diff(a, 'cpu', (1, 32, 32))
    man, syn = load_synth(m)
    with open(os.path.join(GOLD, 'manifest_latent32.json'), 'w') as f:
        json.dump([(k, list(s)) for k, s in man], f)
    m.eval()
    x = rnd(31, 6, 32)
    with torch.no_grad():
        y = m(x, 123)
    yo = O.latent_unet(syn, 'backbone', x, torch.full((6,), 123, dtype=torch.long), 32)
    check('latent unet', yo, y, 1e-5)
    out = dict(x=x, y123=y)
    # training-path loss with replayed draws
    torch.manual_seed(9)
    idx = torch.randint(0, 1000, (6,))
    eps = torch.randn_like(x)
    torch.manual_seed(9)
    loss = m.loss_fn(a, x)
    sched = O.noise_schedule(1e-5, 1e-2, 1000)
    ab = sched[2][idx][:, None]
    xt = torch.sqrt(ab) * x + torch.sqrt(1 - ab) * eps
    lo = (O.latent_unet(syn, 'backbone', xt, idx, 32) - eps).square().mean()
    check('latent loss', lo, loss.detach(), 1e-5)
    out.update(idx=idx, eps=eps, loss=loss.detach())
    # sampler on a short schedule
    for det in (True, False):
        cfg_s = O.Cfg(a_dim=32, is_latent=True, diffusion_steps=5, input_size=32, deterministic=det)
        a_s = args_for(cfg_s)
        m_s = R_models.Diff(a_s, 'cpu', (1, 32, 32))
        _, syn_s = load_synth(m_s)
        m_s.eval()
        proc = R_sampling.LatentDiffusionProcess(a_s, m_s, 'cpu')
        xT = rnd(32, 3, 32)
        torch.manual_seed(77)
        with torch.no_grad():
            tr = list(proc._one_diffusion_step(xT, det))
        torch.manual_seed(77)
        noises = {i: torch.randn_like(xT) for i in reversed(range(5)) if i != 0}
        sched_s = O.noise_schedule(1e-5, 1e-2, 5)
        with torch.no_grad():
            to = O.sample_loop(sched_s, lambda xx, i: O.latent_unet(syn_s, 'backbone', xx,
                                                                   torch.full((3,), i, dtype=torch.long), 32),
                               xT, det, noises)
        for k in range(5):
            check('latent sampler det=%s step %d' % (det, k), to[k], tr[k], 2e-5)
        key = 'ddim' if det else 'ddpm'
        out[key + '.xT'] = xT
        out[key + '.noise'] = torch.stack([noises[i] for i in (4, 3, 2, 1)])
        out[key + '.trace'] = torch.stack(tr)
    save('latent', **out)


# -------------------------------------------------- vanilla UNet + two-phase
def gen_vanilla_twophase():
    """A15: the reference's UNet passes a stray `crossattn` kwarg to ResBlock
    (models.py:32-33).  Swallow it at import time in THIS tool only."""
    orig = R_modules.ResBlock.__init__

    def patched(self, in_ch, out_ch, tdim, dropout, attn=False, crossattn=False):
        orig(self, in_ch, out_ch, tdim, dropout, attn=attn)
    R_modules.ResBlock.__init__ = patched
    R_models.ResBlock.__init__ = patched
    try:
        cfg = O.dataset_cfg('fmnist', a_dim=8, diffusion_steps=3, deterministic=True, model='diff',
                            is_latent=False, mode='eval_fid', split_step=1)
        a = args_for(cfg)
        torch.manual_seed(0)
        m2 = R_models.Diff(a, 'cpu', cfg.shape)
        man2, syn2 = load_synth(m2)
        with open(os.path.join(GOLD, 'manifest_vanilla_fmnist.json'), 'w') as f:
            json.dump([(k, list(s)) for k, s in man2], f)
        m2.eval()
        m1 = R_models.InfoDiff(a, 'cpu', cfg.shape)
        load_synth(m1)
        m1.eval()
        x = rnd(41, 2, *cfg.shape)
        with torch.no_grad():
            y = m2(x, 2)
        yo = O.vanilla_unet(syn2, 'backbone', x, torch.full((2,), 2, dtype=torch.long),
                            cfg.unets_channels, O.ch_mult_for(cfg, vanilla=True))
        check('vanilla unet', yo, y)
        calls = []

        class Spy(torch.nn.Module):
            def __init__(self, inner, name):
                super().__init__()
                self.inner, self.name = inner, name

            def forward(self, *aa):
                calls.append(self.name)
                return self.inner(*aa)
        proc = R_sampling.TwoPhaseDiffusionProcess(a, Spy(m1, 'f1'), Spy(m2, 'f2'), 'cpu', cfg.shape)
        xT = rnd(42, 2, *cfg.shape)
        a2 = rnd(43, 2, cfg.a_dim)
        torch.manual_seed(5)
        with torch.no_grad():
            fin = proc.sampling(2, xT=xT, a=a2)
        assert calls == ['f2'] * 3, calls   # quirk 2: always model 2
        torch.manual_seed(5)
        noises = {i: torch.randn_like(xT) for i in reversed(range(3)) if i != 0}
        save('vanilla_twophase', x=x, y2=y, xT=xT, a=a2, final=fin,
             noise=torch.stack([noises[2], noises[1]]), calls=np.array([2, 2, 2]))
    finally:
        R_modules.ResBlock.__init__ = orig
        R_models.ResBlock.__init__ = orig


# ------------------------------ BASELINE configs[4]: CIFAR-10 shape, two-phase + latent-DDIM
def gen_config5():
    """The CIFAR-10 32x32 pipeline of eval_fid.sh:11 at its own shape: vanilla [1,2,4,8] UNet (512-channel 4x4
    maps), latent denoiser at a_dim = 256 (hidden 1024) with a DDIM trace, and the two-phase sampler's result."""
    orig = R_modules.ResBlock.__init__

    def patched(self, in_ch, out_ch, tdim, dropout, attn=False, crossattn=False):
        orig(self, in_ch, out_ch, tdim, dropout, attn=attn)
    R_modules.ResBlock.__init__ = patched
    R_models.ResBlock.__init__ = patched
    try:
        cfg = O.dataset_cfg('cifar10', a_dim=256, diffusion_steps=4, deterministic=True, model='diff',
                            is_latent=False, mode='eval_fid', split_step=1)
        a = args_for(cfg)
        torch.manual_seed(0)
        m2 = R_models.Diff(a, 'cpu', cfg.shape)
        man2, syn2 = load_synth(m2)
        with open(os.path.join(GOLD, 'manifest_vanilla_cifar.json'), 'w') as f:
            json.dump([(k, list(s)) for k, s in man2], f)
        m2.eval()
        x = rnd(51, 2, *cfg.shape)
        with torch.no_grad():
            y = m2(x, 2)
        yo = O.vanilla_unet(syn2, 'backbone', x, torch.full((2,), 2, dtype=torch.long),
                            cfg.unets_channels, O.ch_mult_for(cfg, vanilla=True))
        check('vanilla unet (cifar)', yo, y)
        m1 = R_models.InfoDiff(a, 'cpu', cfg.shape)
        load_synth(m1)
        m1.eval()
        proc = R_sampling.TwoPhaseDiffusionProcess(a, m1, m2, 'cpu', cfg.shape)
        xT = rnd(52, 2, *cfg.shape)
        a2 = rnd(53, 2, cfg.a_dim)
        torch.manual_seed(5)
        with torch.no_grad():
            fin = proc.sampling(2, xT=xT, a=a2)
        torch.manual_seed(5)
        noises = {i: torch.randn_like(xT) for i in reversed(range(4)) if i != 0}
        out = dict(x=x, y2=y, xT=xT, a=a2, final=fin, noise=torch.stack([noises[3], noises[2], noises[1]]))
    finally:
        R_modules.ResBlock.__init__ = orig
        R_models.ResBlock.__init__ = orig
    # latent denoiser at a_dim 256
    cfgl = O.Cfg(a_dim=256, is_latent=True, diffusion_steps=4, input_size=32, deterministic=True)
    al = args_for(cfgl)
    torch.manual_seed(0)
    ml = R_models.Diff(al, 'cpu', (1, 256, 256))
    manl, synl = load_synth(ml)
    with open(os.path.join(GOLD, 'manifest_latent256.json'), 'w') as f:
        json.dump([(k, list(s)) for k, s in manl], f)
    ml.eval()
    xl = rnd(54, 3, 256)
    with torch.no_grad():
        yl = ml(xl, 3)
    yo = O.latent_unet(synl, 'backbone', xl, torch.full((3,), 3, dtype=torch.long), 256)
    check('latent unet a_dim 256', yo, yl, 1e-5)
    procl = R_sampling.LatentDiffusionProcess(al, ml, 'cpu')
    xTl = rnd(55, 3, 256)
    torch.manual_seed(78)
    with torch.no_grad():
        trl = list(procl._one_diffusion_step(xTl, True))
    torch.manual_seed(78)
    nzl = {i: torch.randn_like(xTl) for i in reversed(range(4)) if i != 0}
    out.update({'lat.x': xl, 'lat.y3': yl, 'lat.xT': xTl, 'lat.noise': torch.stack([nzl[3], nzl[2], nzl[1]]),
                'lat.trace': torch.stack(trl)})
    save('config5_cifar', **out)


def gen_latent_archive():
    """A `{model}_{exp}_latent.npz` archive as the REFERENCE writes it (run.py:415-443, save_latent: the wire format between the two
    phases of config 5): the reference's own encoder (synthetic weights, fmnist configuration) on seeded batches, `all_a` / `all_attr`
    collected and saved with the reference's statements; read back through the reference's own LatentDataset (utils.py:163-172) for
    the expected rows.  The archive is committed as a data fixture (tests/golden/ref_diff_latent.npz) with its inputs beside it."""
    cfg = O.dataset_cfg('fmnist', a_dim=32, mmd_weight=0.1)
    a = args_for(cfg)
    torch.manual_seed(0)
    model = R_models.InfoDiff(a, 'cpu', cfg.shape)
    man, syn = load_synth(model)
    model.eval()
    g = torch.Generator(device='cpu')
    g.manual_seed(71)
    batches = [(torch.rand(5, *cfg.shape, generator=g) * 2 - 1, torch.randint(0, 10, (5,), generator=g)) for _ in range(3)]
    all_a, all_attr = [], []
    for data_all in batches:                      # run.py:418-439 with dataset == 'fmnist', kld_weight == 0, mmd_weight != 0
        data = data_all[0]
        latents_classes = data_all[1]
        with torch.no_grad():
            av, _, _, _ = model.encoder(data)
        all_a.append(av.cpu().numpy())
        all_attr.append(latents_classes)
    all_a = np.concatenate(all_a)
    all_attr = np.concatenate(all_attr)
    path = os.path.join(GOLD, 'ref_diff_latent')
    np.savez(path, all_a=all_a, all_attr=all_attr)                 # run.py:442 (np.savez appends .npz)
    ds = R_utils.LatentDataset(path + '.npz')
    rows = torch.stack([ds[i] for i in range(len(ds))])
    with torch.no_grad():
        ours = torch.cat([O.encoder(syn, 'encoder', b[0], cfg.encoder_channels, O.ch_mult_for(cfg))[0] for b in batches])
    check('latent archive rows', ours, rows)
    save('ref_diff_latent_inputs', x=torch.cat([b[0] for b in batches]), attr=torch.cat([b[1] for b in batches]), rows=rows,
         n=len(ds))
    print('wrote ref_diff_latent.npz %.1f KB' % (os.path.getsize(path + '.npz') / 1024))


if __name__ == '__main__':
    which = sys.argv[1:] or ['schedule', 'blocks', 'mmd', 'stub', 'latent', 'vanilla', 'fmnist', 'bneck', 'vae', 'priors', 'branches', 'size28', 'latent_archive', 'celeba', 'config5']
    if 'schedule' in which:
        gen_schedule()
    if 'blocks' in which:
        gen_blocks()
    if 'mmd' in which:
        gen_mmd()
    if 'stub' in which:
        gen_sampler_stub()
    if 'latent' in which:
        gen_latent()
    if 'vanilla' in which:
        gen_vanilla_twophase()
    if 'fmnist' in which:
        gen_model('fmnist', O.dataset_cfg('fmnist', a_dim=32, mmd_weight=0.1), B=4, seed=64)
        gen_model('fmnist_kld', O.dataset_cfg('fmnist', a_dim=16, mmd_weight=0.1, kld_weight=0.01), B=3, seed=65)
    if 'bneck' in which:
        gen_model('fmnist_bneck', O.dataset_cfg('fmnist', a_dim=32, mmd_weight=0.1, is_bottleneck=True), B=3, seed=66)
    if 'vae' in which:
        gen_vae('fmnist_vae', O.dataset_cfg('fmnist', a_dim=32, mmd_weight=0.1), B=3, seed=67)
        gen_vae('fmnist_vae_kld', O.dataset_cfg('fmnist', a_dim=32, mmd_weight=0.0, kld_weight=0.01), B=2, seed=68)
    if 'priors' in which:
        gen_priors_and_capacity()
    if 'branches' in which:
        gen_branches()
    if 'size28' in which:
        # the `input_size == 28` branch (models.py:619-622: ch_mult [1, 2, 4], maps 28 / 14 / 7 -- no named dataset reaches it, data.py:63-102
        # sets 32 or 64 everywhere, but `InfoDiff(args, ...)` with input_size 28 is constructible and runs)
        gen_model('size28', O.Cfg(input_channels=1, unets_channels=32, encoder_channels=32, input_size=28, a_dim=32, mmd_weight=0.1), B=3, seed=69)
    if 'latent_archive' in which:
        gen_latent_archive()
    if 'celeba' in which:
        gen_model('celeba', O.dataset_cfg('celeba', a_dim=32, mmd_weight=0.1), B=2, seed=64)
    if 'config5' in which:
        gen_config5()
    print('ALL ORACLE-vs-REFERENCE CHECKS PASSED')
