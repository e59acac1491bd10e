#!/usr/bin/env python3
"""Every non-library (ATen / runtime) kernel launch of one eager CelebA B = 32 training step, with the torch op that issued it and the
innermost frame of this repository on its stack (torch.profiler, with_stack).  Usage: python tools/aten_launches.py"""
import os
import sys
from collections import Counter

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

import bench
from infodiffusion_amd.models import InfoDiff
from infodiffusion_amd.optim import FusedClipAdamW
from infodiffusion_amd.trainer import GraphedTrainStep

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = sys.argv[:1]
a = bench.parse()
margs = bench.make_args(a)
dev = torch.device('cuda', 0)
torch.manual_seed(1)
model = InfoDiff(margs, dev, (3, 64, 64)).train()
opt = FusedClipAdamW(model.parameters(), lr=1e-4, weight_decay=1e-5, max_norm=1.0)
step = GraphedTrainStep(model, margs, opt, use_graph=False, health_every=0)
x = (torch.rand(32, 3, 64, 64, device=dev) * 2 - 1).contiguous(memory_format=torch.channels_last)
for _ in range(3):
    step(x, 0)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(x, 0)
    torch.cuda.synchronize()
rows = Counter()
for ev in prof.events():
    if ev.device_type is not None and str(ev.device_type).endswith('CPU') and ev.kernels:
        if not ev.name.startswith('aten::'):
            continue
        # leaf aten ops only (an op whose children also launched is a wrapper)
        if any(c.kernels for c in (ev.cpu_children or [])):
            continue
        frames = [f for f in (ev.stack or []) if ROOT in f and 'tools/aten_launches' not in f]
        where = frames[0].replace(ROOT + '/', '')[:110] if frames else '(autograd engine / torch internals)'
        for k in ev.kernels:
            rows[(ev.name, k.name[:60], where)] += 1
tot = 0
for (op, kern, where), n in sorted(rows.items(), key=lambda t: (t[0][2], t[0][0])):
    print('%2d  %-22s %-62s %s' % (n, op, kern, where))
    tot += n
print('total non-library launches per step: %d' % tot)
