#!/usr/bin/env python3
"""In-situ timing of the batched weight-gradient launches of a CelebA B = 32 training step (idf_conv_wgrad_bf16_batched per
(taps, mode) class + idf_wgrad_reduce_batched): the hook re-issues each launch REPS times right behind the original call
(operands still alive, warm) between two HIP events.  Usage: python tools/bench_wgrad.py [B] [reps]   (IDF_LIB selects a variant)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from infodiffusion_amd import ops
from infodiffusion_amd.models import InfoDiff
from infodiffusion_amd.optim import FusedClipAdamW
from infodiffusion_amd.trainer import GraphedTrainStep

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device('cuda', 0)
a = type('A', (), dict(a_dim=32, batch=B, dtype='bf16'))()
margs = bench.make_args(a)
torch.manual_seed(64)
model = InfoDiff(margs, dev, (3, 64, 64))
model.train()
opt = FusedClipAdamW(model.parameters(), lr=1e-4, weight_decay=1e-5, max_norm=1.0)
step = GraphedTrainStep(model, margs, opt, use_graph=False)
x = (torch.rand(B, 3, 64, 64, device=dev) * 2 - 1).contiguous(memory_format=torch.channels_last)
for _ in range(2):
    step(x, 0)
torch.cuda.synchronize()
orig = ops.call
rows = []


def hooked(name, *args):
    orig(name, *args)
    if name in ('idf_conv_wgrad_bf16_batched', 'idf_wgrad_reduce_batched'):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REPS):
            orig(name, *args)
        e1.record()
        rows.append((name, args[1:6], e0, e1))


_flush = ops.WgradBatch._flush_pending.__func__


def flush_logged(cls):
    if os.environ.get('WG_LIST'):
        for it in cls.pending:
            print('item B%d H%d W%d Cin%d Cout%d taps%d mode%d a2=%s' % (it[4], it[5], it[6], it[7], it[8], it[9], it[10], it[11] is not None))
    _flush(cls)
    if os.environ.get('WG_LIST'):
        for buf in cls._bufs.values():
            print('plan', buf[3])


ops.WgradBatch._flush_pending = classmethod(flush_logged)
ops.call = hooked
step(x, 0)
torch.cuda.synchronize()
ops.call = orig
tot = 0.0
for name, meta, e0, e1 in rows:
    us = e0.elapsed_time(e1) * 1e3 / REPS
    tot += us
    print('%-30s %-28s %8.1f us' % (name, meta, us))
print('total %.1f us' % tot)
