#!/usr/bin/env python3
"""Is the B = 32 training step a latency chain that a second, independent chain could hide?  (step time against batch:
6.7 / 7.8 / 10.0 / 15.3 ms at B = 8 / 16 / 32 / 64 -- an intercept of ~5.6 ms.)  Two INDEPENDENT CelebA models, B = 16 each,
stepped (a) one after the other on one stream and (b) on two streams forked and joined inside ONE hipGraph, against one
model at B = 32.  Nothing here is a product path: it prices the idea before any kernel or autograd plumbing is written.
Usage: python tools/two_chain_probe.py [half_batch]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from infodiffusion_amd.models import InfoDiff
from infodiffusion_amd.optim import FusedClipAdamW

HB = int(sys.argv[1]) if len(sys.argv) > 1 else 16
sys.argv = sys.argv[:1]
a = bench.parse()
margs = bench.make_args(a)
dev = torch.device('cuda', 0)


def make(batch, seed):
    torch.manual_seed(seed)
    m = InfoDiff(margs, dev, (3, 64, 64)).train()
    o = FusedClipAdamW(m.parameters(), lr=1e-4, weight_decay=1e-5, max_norm=1.0)
    g = torch.Generator(device='cpu')
    g.manual_seed(seed)
    x = (torch.rand(batch, 3, 64, 64, generator=g) * 2 - 1).to(dev).contiguous(memory_format=torch.channels_last)
    return m, o, x


def step(m, o, x):
    loss = m.loss_fn(margs, x)
    o.zero_grad()
    loss.backward()
    o.step()


def timed(graph, n=40):
    for _ in range(5):
        graph.replay()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        graph.replay()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3


def capture(fn):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn(False)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn(True)
    return g


A, B, F = make(HB, 1), make(HB, 2), make(2 * HB, 3)
s2 = torch.cuda.Stream()


def full(_):
    step(*F)


def serial(_):
    step(*A)
    step(*B)


def forked(_):
    cur = torch.cuda.current_stream()
    s2.wait_stream(cur)
    with torch.cuda.stream(s2):
        step(*B)
    step(*A)
    cur.wait_stream(s2)


for name, fn in (('one model, B = %d' % (2 * HB), full), ('two models, B = %d each, one stream' % HB, serial),
                 ('two models, B = %d each, two streams in one graph' % HB, forked)):
    print('%-52s %.3f ms per replay' % (name, timed(capture(fn))), flush=True)
