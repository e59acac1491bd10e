#!/usr/bin/env python3
"""Fused clip + AdamW (idf_clip_adamw: squared-norm, scalars, update) on the CelebA model's parameters: time per step from a
captured hipGraph and the bytes it moves (g read twice; p, m, v read and written).  Usage: python tools/bench_optim.py"""
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from infodiffusion_amd.models import InfoDiff
from infodiffusion_amd.optim import FusedClipAdamW

a = types.SimpleNamespace(a_dim=32, batch=32, dtype='bf16')
margs = bench.make_args(a)
dev = torch.device('cuda:0')
model = InfoDiff(margs, dev, (3, 64, 64))
opt = FusedClipAdamW(model.parameters(), lr=1e-4, weight_decay=1e-5, max_norm=1.0)
x = torch.rand(32, 3, 64, 64, device=dev) * 2 - 1
for _ in range(2):
    loss = model.loss_fn(args=margs, x=x)
    opt.zero_grad()
    loss.backward()
    opt.step()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
with torch.cuda.stream(side), torch.cuda.graph(g, stream=side):
    for _ in range(10):
        opt.step()
g.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    g.replay()
e1.record()
torch.cuda.synchronize()
n = sum(p.numel() for p in model.parameters() if p.grad is not None)
us = e0.elapsed_time(e1) / 50 * 1e3
print('parameters with gradients %d  step %.1f us  %.0f GB/s (32 bytes per parameter)' % (n, us, n * 32 / us / 1e3))
