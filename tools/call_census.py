#!/usr/bin/env python3
"""Census of C-ABI calls (entry point + integer arguments) issued by one eager training step,
each with its average GPU time (HIP events around every call).  Usage: python tools/call_census.py [batch]"""
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import types

import torch

import bench
from infodiffusion_amd import ops, _lib
from infodiffusion_amd.models import InfoDiff
from infodiffusion_amd.optim import FusedClipAdamW

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
a = types.SimpleNamespace(a_dim=32, batch=B, dtype='bf16')
margs = bench.make_args(a)
dev = torch.device('cuda', 0)
model = InfoDiff(margs, dev, (3, 64, 64)).train()
opt = FusedClipAdamW(model.parameters(), lr=1e-4, weight_decay=1e-5)
x = torch.rand(B, 3, 64, 64, device=dev) * 2 - 1


def step():
    loss = model.loss_fn(margs, x)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()

records = []
orig = _lib.call


def traced(name, *args):
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    orig(name, *args)
    e1.record()
    key = (name,) + tuple(v for v in args if isinstance(v, int) and not isinstance(v, bool) and abs(v) < (1 << 20))
    records.append((key, e0, e1))


ops.call = traced
_lib.call = traced
REP = 5
for _ in range(REP):
    step()
torch.cuda.synchronize()
agg = defaultdict(lambda: [0, 0.0])
for key, e0, e1 in records:
    agg[key][0] += 1
    agg[key][1] += e0.elapsed_time(e1) * 1e3
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for v in agg.values()) / REP
print('total us/step (event-bracketed, includes launch gaps): %.0f' % tot)
for key, (n, us) in rows[:70]:
    print('%7.1f us/step  n=%3d  avg %6.1f us  %s %s' % (us / REP, n // REP, us / n, key[0], key[1:]))
