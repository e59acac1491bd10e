#!/usr/bin/env python3
"""Which bf16 roundings cost the epsilon-hat accuracy (north_star: 1e-2; measured 1.5e-2 max-norm / 1.8e-2 rel-L2)?

The backbone is evaluated in the fp32 kernels on the CelebA fixture's sampling input with bf16 roundings INJECTED at
chosen classes of tensors (x.bfloat16().float() on the op's output), which prices a storage policy before any kernel
is written for it:

  w      conv / linear master weights rounded to bf16 (the weight shadows of the bf16 path)
  a      the activated tensor a = dropout(SiLU(GroupNorm(x))) a conv contracts over (the MFMA's bf16 operand)
  inner  conv outputs that only a GroupNorm reads (block1 / block2 outputs), q|k|v, the attention output, 1x1 shortcuts
  stream the residual stream: head / DownSample / UpSample outputs, block outputs (conv3 + residual), AttnBlock outputs
         -- these are also the 12 skip tensors

`all` = the bf16 path's storage policy.  The error is epsilon-hat against the reference fixture (max-abs / max-abs and
rel-L2).  Usage: python tools/bf16_rounding_study.py [celeba|fmnist]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle import infodiff_oracle as O
from tests.helpers import gold, make_infodiff, rel, rel_l2
from infodiffusion_amd import modules, ops

DEV = 'cuda'
ACTIVE = set()


def r(t):
    return t.bfloat16().float().contiguous(memory_format=torch.channels_last) if t.dim() == 4 else t.bfloat16().float()


_fused_conv, _attention, _gn_fwd, _gn_apply = ops.fused_conv, ops.attention, ops.gn_fused_fwd_raw, ops.gn_apply_raw


def fused_conv(x, weight, bias, cfg, gn_w=None, gn_b=None, film_t=None, film_a=None, residual=None, seed=None,
               passthrough=False, want_stats=False):
    out = _fused_conv(x, weight, bias, cfg, gn_w, gn_b, film_t, film_a, residual, seed, passthrough, want_stats)
    y = out[0] if isinstance(out, tuple) else out
    stream = residual is not None or cfg['mode'] != ops.S1 or (cfg['act'] == 0 and cfg['taps'] == 9)
    if ('stream' if stream else 'inner') in ACTIVE and weight.shape[0] > 3:      # the epsilon-hat head itself stays fp32
        y = r(y)
    return (y,) + tuple(out[1:]) if isinstance(out, tuple) else y


def attention(qkv):
    o = _attention(qkv)
    return r(o) if 'inner' in ACTIVE else o


def gn_fwd(*a, **k):
    out = _gn_fwd(*a, **k)
    return (r(out[0]),) + tuple(out[1:]) if 'a' in ACTIVE else out


def gn_apply(*a, **k):
    out = _gn_apply(*a, **k)
    return r(out) if 'a' in ACTIVE else out


ops.fused_conv, ops.attention, ops.gn_fused_fwd_raw, ops.gn_apply_raw = fused_conv, attention, gn_fwd, gn_apply


def run(tag, classes):
    ACTIVE.clear()
    ACTIVE.update(classes)
    cfg = O.dataset_cfg(tag, a_dim=32, mmd_weight=0.1)
    model, args, sd = make_infodiff(cfg, DEV, 'fp32', 'manifest_' + tag)
    model.eval()
    if 'w' in ACTIVE:
        with torch.no_grad():
            for n, p in model.backbone.named_parameters():
                if p.dim() >= 2 and 'timembedding.0' not in n:
                    p.copy_(p.bfloat16().float())
    g = gold('model_' + tag)
    with torch.no_grad():
        e = model(g['samp_x'].to(DEV), 17, g['samp_a'].to(DEV))
    return rel(e, g['samp_eps17']), rel_l2(e, g['samp_eps17'])


if __name__ == '__main__':
    tag = sys.argv[1] if len(sys.argv) > 1 else 'celeba'
    print('epsilon-hat vs the reference fixture (%s), fp32 kernels with bf16 roundings injected:' % tag)
    print('  %-44s %-10s %s' % ('rounded tensor classes', 'max-norm', 'rel-L2'))
    for name, cl in (('none (the fp32 path)', ()), ('all = the bf16 storage policy', ('w', 'a', 'inner', 'stream')),
                     ('w + a only (fp32 storage, bf16 MFMA operands)', ('w', 'a')),
                     ('w + a + inner (fp32 residual stream / skips)', ('w', 'a', 'inner')),
                     ('w + a + stream (fp32 GroupNorm inputs)', ('w', 'a', 'stream')),
                     ('stream only', ('stream',)), ('inner only', ('inner',)), ('a only', ('a',)), ('w only', ('w',))):
        m, l = run(tag, cl)
        print('  %-44s %.2e   %.2e' % (name, m, l))
