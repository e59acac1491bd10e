#!/usr/bin/env python3
"""hipGraph-replayed time of the training step's phases (forward only / forward+backward / full)."""
import os
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from infodiffusion_amd.models import InfoDiff
from infodiffusion_amd.optim import FusedClipAdamW

a = types.SimpleNamespace(a_dim=32, batch=32, dtype='bf16')
margs = bench.make_args(a)
dev = torch.device('cuda', 0)
model = InfoDiff(margs, dev, (3, 64, 64)).train()
opt = FusedClipAdamW(model.parameters(), lr=1e-4, weight_decay=1e-5)
x = torch.rand(32, 3, 64, 64, device=dev) * 2 - 1


def fwd():
    with torch.no_grad():
        return model.loss_fn(margs, x)


def fwd_grad():
    return model.loss_fn(margs, x)


def fwd_bwd():
    loss = model.loss_fn(margs, x)
    opt.zero_grad(set_to_none=True)
    loss.backward()


def full():
    fwd_bwd()
    opt.step()


def enc_only():
    with torch.no_grad():
        return model.encoder(x)


def timed(fn, name, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    print('%-28s %.3f ms' % (name, (time.time() - t0) / n * 1e3), flush=True)


full()
timed(enc_only, 'encoder forward (no grad)')
timed(fwd, 'loss forward (no grad)')
timed(fwd_grad, 'loss forward (grad mode)')
timed(fwd_bwd, 'forward + backward')
timed(full, 'full step')
