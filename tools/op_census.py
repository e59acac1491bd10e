#!/usr/bin/env python3
"""Census of ATen ops issued by one eager training step (to find stray tiny launches)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import types

import torch
from torch.profiler import ProfilerActivity, profile

import bench
from infodiffusion_amd.models import InfoDiff
from infodiffusion_amd.optim import FusedClipAdamW

a = types.SimpleNamespace(a_dim=32, batch=32, dtype='bf16')
margs = bench.make_args(a)
dev = torch.device('cuda', 0)
model = InfoDiff(margs, dev, (3, 64, 64)).train()
opt = FusedClipAdamW(model.parameters(), lr=1e-4, weight_decay=1e-5)
x = torch.rand(32, 3, 64, 64, device=dev) * 2 - 1


def step():
    loss = model.loss_fn(margs, x)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='count', row_limit=25, max_name_column_width=50))
for name in ('aten::fill_', 'aten::zero_', 'aten::copy_', 'aten::zeros', 'aten::add', 'aten::cat', 'aten::clone'):
    print('==', name)
    rows = [e for e in prof.key_averages(group_by_stack_n=6) if e.key == name]
    rows.sort(key=lambda e: -e.count)
    for e in rows[:6]:
        print('  count', e.count, '|', ' <- '.join(s.split('/')[-1] for s in e.stack[:5]))
