#!/usr/bin/env python3
"""Checkpoint interchange with the REAL reference, both halves of the round trip:

  make  (GPU box):        train the product for a few steps, save `model.state_dict()` the way run.py does, plus a
                          probe (x_t, t, a) and the product's eps-hat / encoder outputs for it (fp32 activations).
  check (build container): load that file into the reference's own InfoDiff (strict=True, CPU) and compare the
                          reference's outputs on the probe with the product's.

    gpurun -- python tools/ckpt_interchange.py make gpurun_out/ckpt_probe.pt
    PYTHONDONTWRITEBYTECODE=1 python tools/ckpt_interchange.py check gpurun_out/ckpt_probe.pt
"""
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CFG = dict(beta1=1e-5, betaT=1e-2, diffusion_steps=1000, input_size=32, input_channels=1, is_bottleneck=False,
           unets_channels=32, encoder_channels=32, a_dim=32, mmd_weight=0.1, kld_weight=0.0, prior='regular',
           batch_size=16, use_C=False, C_max=25.0, epochs=2, deterministic=True, model='diff', split_step=500,
           mode='train', is_latent=False, dataset='fmnist')
SHAPE = (1, 32, 32)


def make(path):
    from infodiffusion_amd.models import InfoDiff
    from infodiffusion_amd.optim import FusedClipAdamW
    from infodiffusion_amd.trainer import GraphedTrainStep
    dev = torch.device('cuda')
    args = types.SimpleNamespace(act_dtype='fp32', **CFG)
    torch.manual_seed(12)
    model = InfoDiff(args, dev, SHAPE).train()
    opt = FusedClipAdamW(model.parameters(), lr=2e-4, weight_decay=1e-5, max_norm=1.0)
    step = GraphedTrainStep(model, args, opt)
    g = torch.Generator(device=dev)
    g.manual_seed(13)
    losses = [float(step(torch.rand(16, *SHAPE, generator=g, device=dev) * 2 - 1, 0)) for _ in range(40)]
    model.eval()
    cg = torch.Generator(device='cpu')
    cg.manual_seed(14)
    x, a = torch.randn(4, *SHAPE, generator=cg), torch.randn(4, 32, generator=cg)
    x0 = torch.rand(4, *SHAPE, generator=cg) * 2 - 1
    with torch.no_grad():
        eps = model(x.to(dev), 321, a.to(dev)).float().cpu()
        enc_a = model.encoder(x0.to(dev))[0].float().cpu()
    # a 60-step DDIM and DDPM trajectory of the product with the noise draws recorded, for replay in the reference
    from infodiffusion_amd.sampling import DiffusionProcess
    traj = {}
    for det in (True, False):
        sargs = types.SimpleNamespace(act_dtype='fp32', **{**CFG, 'diffusion_steps': 60, 'deterministic': det})
        proc = DiffusionProcess(sargs, model, dev, SHAPE)
        noises = []

        def rec(t, noises=noises):
            z = torch.randn_like(t)
            noises.append(z.float().cpu())
            return z
        proc._randn_like = rec
        xT = torch.randn(2, *SHAPE, generator=cg)
        out = proc.sampling(xT=xT.to(dev), a=a[:2].to(dev)).float().cpu()
        traj[det] = dict(xT=xT, noises=noises, out=out)
    # five optimisation steps (dropout off, eager) with every random draw recorded, for replay in the reference
    start = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    opt2 = FusedClipAdamW(model.parameters(), lr=1e-3, weight_decay=1e-5, max_norm=1.0)
    draws, opt_losses, opt_norms = [], [], []
    orig_rl, orig_ri = torch.randn_like, torch.randint
    for k in range(5):
        xb = torch.rand(8, *SHAPE, generator=cg) * 2 - 1
        step_draws = dict(x=xb, idx=torch.randint(0, 1000, (8,), generator=cg), eps=torch.randn(8, *SHAPE, generator=cg),
                          reparam=torch.randn(8, 32, generator=cg), prior=torch.randn(8, 32, generator=cg))
        feed = iter([step_draws['eps'], step_draws['reparam'], step_draws['prior']])
        torch.randn_like = lambda t, **kw: next(feed).to(t.device)
        torch.randint = lambda *a_, **kw: step_draws['idx'].clone()
        try:
            loss = model.loss_fn(args, xb.to(dev))
        finally:
            torch.randn_like, torch.randint = orig_rl, orig_ri
        opt2.zero_grad()
        loss.backward()
        opt2.step()
        draws.append(step_draws)
        opt_losses.append(float(loss))
        opt_norms.append(float(opt2.total_norm()))
    end = {k: model.state_dict()[k].detach().cpu().clone() for k in
           ('backbone.head.weight', 'backbone.downblocks.0.block1.2.weight', 'backbone.middleblocks.0.attn.proj_q.weight',
            'backbone.fc_a.weight', 'backbone.tail.2.weight', 'encoder.fc_a.weight', 'encoder.upblocks.3.main.weight')}
    torch.save({'state_dict': {k: v.cpu() for k, v in start.items()}, 'x': x, 'a': a, 'x0': x0, 't': 321,
                'eps': eps, 'enc_a': enc_a, 'losses': losses, 'traj': traj,
                'opt': dict(draws=draws, losses=opt_losses, norms=opt_norms, end=end)}, path)
    print('trained 40 steps (loss %.4f -> %.4f); wrote %s (%.1f MB)' % (losses[0], losses[-1], path,
                                                                      os.path.getsize(path) / 1e6))


def check(path):
    sys.dont_write_bytecode = True
    sys.path.insert(0, '/root/reference')
    import models as R_models      # the reference itself
    blob = torch.load(path, map_location='cpu')
    ref = R_models.InfoDiff(types.SimpleNamespace(**CFG), 'cpu', SHAPE)
    ref.load_state_dict(blob['state_dict'], strict=True)          # every key, every shape
    ref.eval()
    with torch.no_grad():
        eps = ref(blob['x'], blob['t'], blob['a'])
        enc_a = ref.encoder(blob['x0'])[0]
    rel = lambda u, v: float((u - v).abs().max() / (v.abs().max() + 1e-30))
    e1, e2 = rel(blob['eps'], eps), rel(blob['enc_a'], enc_a)
    print('reference loaded the product checkpoint strictly (%d entries)' % len(blob['state_dict']))
    print('eps-hat(t=321): max rel err product-GPU vs reference-CPU %.2e   encoder a: %.2e' % (e1, e2))
    assert e1 < 1e-4 and e2 < 1e-4
    import sampling as R_sampling
    for det, name in ((True, 'DDIM'), (False, 'DDPM')):
        tr = blob['traj'][det]
        sargs = types.SimpleNamespace(**{**CFG, 'diffusion_steps': 60, 'deterministic': det})
        proc = R_sampling.DiffusionProcess(sargs, ref, 'cpu', SHAPE)
        feed = iter(tr['noises'])
        orig = torch.randn_like
        torch.randn_like = lambda t, **k: next(feed)
        try:
            with torch.no_grad():
                out = proc.sampling(xT=tr['xT'], a=blob['a'][:2])
        finally:
            torch.randn_like = orig
        e = rel(tr['out'], out)
        print('%s, 60 steps, same noise draws: final sample max rel err product-GPU vs reference-CPU %.2e' % (name, e))
        assert e < 1e-4
    # five optimisation steps in the reference (run.py:195-200: loss_fn, zero_grad, backward, clip 1.0, AdamW) on the
    # recorded draws; dropout off on both sides
    import contextlib
    import io
    o = blob['opt']
    args = types.SimpleNamespace(**CFG)
    ropt = torch.optim.AdamW(ref.parameters(), lr=1e-3, weight_decay=1e-5)
    orig_rl, orig_ri = torch.randn_like, torch.randint
    for k, d in enumerate(o['draws']):
        feed = iter([d['eps'], d['reparam'], d['prior']])
        torch.randn_like = lambda t, **kw: next(feed)
        torch.randint = lambda *a_, **kw: d['idx'].clone()
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                loss = ref.loss_fn(args=args, x=d['x'])
        finally:
            torch.randn_like, torch.randint = orig_rl, orig_ri
        ropt.zero_grad()
        loss.backward()
        norm = torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0)
        ropt.step()
        print('step %d  loss reference %.6f product %.6f   grad norm reference %.5f product %.5f' %
              (k, float(loss), o['losses'][k], float(norm), o['norms'][k]))
        assert abs(float(loss) - o['losses'][k]) < 1e-4 * abs(float(loss)) + 1e-6
        assert abs(float(norm) - o['norms'][k]) < 2e-3 * float(norm)
    sd = ref.state_dict()
    worst = max(rel(o['end'][k], sd[k]) for k in o['end'])
    print('weights after 5 steps (7 tensors): max rel err product-GPU vs reference-CPU %.2e' % worst)
    assert worst < 2e-3
    return e1, e2


if __name__ == '__main__':
    {'make': make, 'check': check}[sys.argv[1]](sys.argv[2])
