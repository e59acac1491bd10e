"""Shared test helpers: build the product model with the synthetic-weights
protocol, and the smoke check used by __graft_entry__.smoke()."""
import json
import os
import types

import numpy as np
import torch

from oracle import infodiff_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def rel_l2(a, b):
    """||a - b||_2 / ||b||_2 (the stricter reading of "relative error" next to rel()'s max-norm)."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def gold(name):
    z = np.load(os.path.join(GOLD, name + '.npz'))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def manifest(name):
    with open(os.path.join(GOLD, name + '.json')) as f:
        return json.load(f)


def args_of(cfg, **kw):
    d = dict(cfg.__dict__)
    d.update(kw)
    return types.SimpleNamespace(**d)


def make_infodiff(cfg, device, act_dtype='fp32', manifest_name=None, **kw):
    """Product InfoDiff with synthetic weights (same protocol as the fixtures)."""
    from infodiffusion_amd.models import InfoDiff
    args = args_of(cfg, act_dtype=act_dtype, **kw)
    model = InfoDiff(args, device, cfg.shape)
    man = manifest(manifest_name) if manifest_name else [(k, list(v.shape)) for k, v in model.state_dict().items()]
    sd = O.synth_state_dict(man)
    model.load_state_dict(sd, strict=True)
    return model, args, sd


def smoke_check(device):
    """One tiny training step (fwd + bwd) and one DDIM step of the product on
    `device`, checked against the CPU oracle."""
    cfg = O.dataset_cfg('fmnist', a_dim=32, mmd_weight=0.1, diffusion_steps=1000)
    model, args, sd = make_infodiff(cfg, device, 'fp32')
    model.eval()
    B = 2
    g = torch.Generator(device='cpu')
    g.manual_seed(7)
    x = torch.rand(B, *cfg.shape, generator=g) * 2 - 1
    idx = torch.randint(0, 1000, (B,), generator=g)
    eps = torch.randn(B, *cfg.shape, generator=g)
    prior = torch.randn(B, cfg.a_dim, generator=g)
    sched = O.noise_schedule(cfg.beta1, cfg.betaT, cfg.diffusion_steps)
    with torch.no_grad():
        lo, terms = O.infodiff_loss(sd, cfg, x, idx, eps, sched, prior=prior, reparam_noise=torch.zeros(B, cfg.a_dim))
    from infodiffusion_amd import ops
    from infodiffusion_amd.utils import compute_mmd
    xd = x.to(device)
    xt = ops.q_sample(xd, eps.to(device), idx.to(device), model._qs_tables, torch.float32)
    a, _, _, _ = model.encoder(xd)
    out = model.backbone(xt, idx.to(device), a)
    t = ops.diff_loss(out, eps.to(device), xd, model._rec_c0, model._rec_c1, 1.0 / cfg.diffusion_steps)
    loss = t[0] + t[1] + cfg.mmd_weight * compute_mmd(prior.to(device), a)
    loss.backward()
    assert rel(out, terms['out']) < 1e-4, rel(out, terms['out'])
    assert rel(loss, lo) < 1e-4, (float(loss), float(lo))
    assert model.backbone.head.weight.grad is not None
    # one DDIM update against the oracle
    from infodiffusion_amd.sampling import DiffusionProcess
    cfg_s = O.Cfg(**{**cfg.__dict__, 'diffusion_steps': 4, 'deterministic': True})
    proc = DiffusionProcess(args_of(cfg_s), torch.nn.Identity(), device, cfg.shape)
    nz = torch.randn(B, *cfg.shape, generator=g)
    xo = proc._update(xd, out.detach(), 2, 1, nz.to(device))
    ref = O.ddim_step(O.noise_schedule(cfg.beta1, cfg.betaT, 4), x, terms['out'], 2, nz)
    assert rel(xo, ref) < 1e-4
    torch.cuda.synchronize()
    assert ops.rs_sync_timeouts(False) == 0
    print('smoke ok: loss %.6f (oracle %.6f)' % (float(loss), float(lo)))
