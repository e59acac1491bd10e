#!/usr/bin/env python3
"""Where do the torch (non-library) kernels of a training step come from?  One eager step under torch.profiler with python
stacks; prints the innermost infodiffusion_amd / bench frames of every aten op that launches a kernel.
Usage: python tools/glue_sources.py"""
import os
import sys
from collections import Counter

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

import bench
from infodiffusion_amd.models import InfoDiff
from infodiffusion_amd.optim import FusedClipAdamW

sys.argv = sys.argv[:1]
a = bench.parse()
margs = bench.make_args(a)
dev = torch.device('cuda', 0)
torch.manual_seed(1)
model = InfoDiff(margs, dev, (3, 64, 64)).train()
opt = FusedClipAdamW(model.parameters(), lr=1e-4, weight_decay=1e-5, max_norm=1.0)
x = (torch.rand(32, 3, 64, 64) * 2 - 1).to(dev).contiguous(memory_format=torch.channels_last)


def step():
    loss = model.loss_fn(margs, x)
    opt.zero_grad()
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()

# python-level trace of the torch calls that launch glue kernels: innermost frame inside this repository
import traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cnt = Counter()


def where():
    fr = [f for f in traceback.extract_stack()[:-2] if f.filename.startswith(ROOT) and 'glue_sources' not in f.filename]
    return '%s:%d %s' % (os.path.relpath(fr[-1].filename, ROOT), fr[-1].lineno, fr[-1].line[:70]) if fr else '(torch internals / autograd engine)'


def wrap(owner, name):
    orig = getattr(owner, name)

    def f(*a, **k):
        cnt[(name, where())] += 1
        return orig(*a, **k)
    setattr(owner, name, f)


for owner, names in ((torch, ('cat', 'zeros', 'zeros_like', 'randn', 'randn_like', 'randint', 'full', 'ones', 'empty_like')),
                     (torch.Tensor, ('copy_', 'fill_', 'zero_', 'add_', 'mul_', 'to', 'clone', 'contiguous', 'float', 'bfloat16',
                                     '__add__', '__mul__', '__sub__', '__truediv__', 'sum', 'mean', 'normal_', 'random_'))):
    for n in names:
        wrap(owner, n)
step()
torch.cuda.synchronize()
for (name, w), n in sorted(cnt.items(), key=lambda kv: kv[0][1]):
    if name in ('empty_like',):
        continue
    print('%3d  %-12s %s' % (n, name, w))
