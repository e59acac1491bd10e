#!/usr/bin/env python3
"""Per-parameter gradient error of the bf16 CelebA training step against the reference fixture (B = 2), by kernel path:
is a large per-parameter deviation bf16 noise or a defect of a fused kernel?  Each configuration runs in a child process
(the switches are read at import): default, the GroupNorm-backward epilogue off, the GroupNorm prologue off, both off, and
the fp32 path.  Prints the worst parameters (max-abs error / the gradient's own max-abs) and the same error measured
against the FP32 product path (which matches the reference to 2e-3), plus the error distribution over all parameters.
Usage: python tools/bf16_grad_profile.py            (driver)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(dtype, out, B=2):
    import torch
    from oracle import infodiff_oracle as O
    from tests.helpers import args_of, gold, make_infodiff
    from tests.test_gpu_model import _ReplayedDraws
    cfg = O.dataset_cfg('celeba', a_dim=32, mmd_weight=0.1)
    g = gold('model_celeba')
    if B != 2:
        gen = torch.Generator(device='cpu')
        gen.manual_seed(5)
        g = {'x': torch.rand(B, *cfg.shape, generator=gen) * 2 - 1, 'idx': torch.randint(0, 1000, (B,), generator=gen),
             'eps': torch.randn(B, *cfg.shape, generator=gen), 'reparam': torch.zeros(B, 32),
             'prior': torch.randn(B, 32, generator=gen)}
    model, args, sd = make_infodiff(cfg, 'cuda', dtype, 'manifest_celeba')
    model.eval()
    from infodiffusion_amd.optim import FusedClipAdamW
    opt = FusedClipAdamW(model.parameters(), lr=0.0, weight_decay=0.0, max_norm=1e9)
    with _ReplayedDraws(g):
        loss = model.loss_fn(args_of(cfg), g['x'].to('cuda'))
        opt.zero_grad()
        loss.backward()
    torch.save({'loss': float(loss), 'grads': {k: p.grad.detach().float().cpu() for k, p in model.named_parameters()
                                               if p.grad is not None}}, out)


def main():
    import torch
    from tests.helpers import gold
    runs = [('fp32', 'fp32', {}), ('bf16 default', 'bf16', {}), ('bf16 IDF_BWD_LAZY=0', 'bf16', {'IDF_BWD_LAZY': '0'}),
            ('bf16 IDF_BWD_CHAIN=0 (round 2 backward)', 'bf16', {'IDF_BWD_CHAIN': '0'}),
            ('bf16 IDF_BWD_CHAIN=0 IDF_DGRAD_GN=0 IDF_GN_FUSE=0 (no fused GroupNorm)', 'bf16',
             {'IDF_BWD_CHAIN': '0', 'IDF_DGRAD_GN': '0', 'IDF_GN_FUSE': '0'})]
    g = gold('model_celeba')
    ref = {k[2:]: v for k, v in g.items() if k.startswith('g.')}
    for B in (2, 32):
        res = {}
        for name, dt, env in runs:
            out = '/tmp/_gradprof.pt'
            subprocess.check_call([sys.executable, os.path.abspath(__file__), '--worker', dt, out, str(B)],
                                  env=dict(os.environ, **env))
            res[name] = torch.load(out)
        f32 = res['fp32']['grads']
        print('\n================ B = %d: per-parameter gradient error against the fp32 product path (which matches the '
              'reference to 3e-5) ================' % B)
        print('loss:', ' '.join('%s %.6f' % (n, r['loss']) for n, r in res.items()))
        for name, r in res.items():
            if name == 'fp32':
                continue
            errs = []
            for k, gr in f32.items():
                sc = float(gr.abs().max())
                if sc < 1e-7:
                    continue
                d = r['grads'][k] - gr
                errs.append((float(d.abs().max()) / sc, float(d.norm() / gr.norm()), k, sc))
            errs.sort(reverse=True)
            e, l = torch.tensor([x[0] for x in errs]), torch.tensor([x[1] for x in errs])
            q = lambda t, v: float(t.quantile(v))
            print('\n%s  (%d parameters)' % (name, len(errs)))
            print('  max-abs / max-abs : median %.2e  p90 %.2e  p99 %.2e  max %.2e' % (q(e, .5), q(e, .9), q(e, .99), float(e.max())))
            print('  rel-L2            : median %.2e  p90 %.2e  p99 %.2e  max %.2e' % (q(l, .5), q(l, .9), q(l, .99), float(l.max())))
            for x in errs[:6]:
                print('    %.3e  L2 %.3e  %-58s (|g|max %.2e)' % x)
            if B == 2:
                fx = sorted(((float((r['grads'][k] - v).abs().max()) / float(v.abs().max()), k) for k, v in ref.items()), reverse=True)
                print('  vs the reference fixture (22 gradients): ' + ', '.join('%.2e %s' % (a, b.split('.', 1)[1][-28:]) for a, b in fx[:4]))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--worker':
        worker(sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 2)
    else:
        main()
