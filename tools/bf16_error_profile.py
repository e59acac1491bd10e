#!/usr/bin/env python3
"""Where the bf16 path loses precision: the backbone evaluated in fp32 and in bf16 on the CelebA fixture's sampling
input, relative error (max-abs / max-abs, and L2) of every block's output, of epsilon-hat against the fp32 product
and against the reference fixture."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle import infodiff_oracle as O
from tests.helpers import gold, make_infodiff, rel

DEV = 'cuda'


def l2(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def run(dtype, tag='celeba'):
    cfg = O.dataset_cfg(tag, a_dim=32, mmd_weight=0.1)
    model, args, sd = make_infodiff(cfg, DEV, dtype, 'manifest_' + tag)
    model.eval()
    outs = []
    hooks = []
    bb = model.backbone
    for name, m in list(bb.downblocks.named_children()) + list(bb.middleblocks.named_children()) + list(bb.upblocks.named_children()):
        hooks.append(m.register_forward_hook(lambda mod, i, o, n=name: outs.append((type(mod).__name__, (o[0] if isinstance(o, tuple) else o).detach().float().cpu()))))
    g = gold('model_' + tag)
    with torch.no_grad():
        e = model(g['samp_x'].to(DEV), 17, g['samp_a'].to(DEV))
    for h in hooks:
        h.remove()
    return e.float().cpu(), outs, g


e32, o32, g = run('fp32')
e16, o16, _ = run('bf16')
print('block outputs, bf16 vs fp32 product path (max-abs/max-abs, L2):')
for i, ((n, a), (_, b)) in enumerate(zip(o16, o32)):
    print('  %2d %-18s %-22s %.2e  %.2e' % (i, n, tuple(a.shape), rel(a, b), l2(a, b)))
print('eps-hat bf16 vs fp32 product: max %.2e  L2 %.2e' % (rel(e16, e32), l2(e16, e32)))
print('eps-hat fp32 vs reference:    max %.2e  L2 %.2e' % (rel(e32, g['samp_eps17']), l2(e32, g['samp_eps17'])))
print('eps-hat bf16 vs reference:    max %.2e  L2 %.2e' % (rel(e16, g['samp_eps17']), l2(e16, g['samp_eps17'])))
