#!/usr/bin/env python3
"""Soak of the bit-reproducible training step (and with it of every in-launch hand-off: a stale read of another workgroup's partial
sums in idf_conv_rs_dgrad_gn_bf16 would change bits): the benchmarked CelebA configuration, B = 32, bf16, dropout on, N graph-replayed
steps, run TWICE from the same seeds -- every loss, every gradient norm and every parameter must agree bit for bit, and no workgroup
may have given up waiting.  usage: tools/soak_determinism.py [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle import infodiff_oracle as O
from tests.helpers import args_of, make_infodiff
from infodiffusion_amd import ops
from infodiffusion_amd.optim import FusedClipAdamW
from infodiffusion_amd.trainer import GraphedTrainStep

N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
DEV = torch.device('cuda')
cfg = O.dataset_cfg('celeba', a_dim=32, mmd_weight=0.1)
gx = torch.Generator(device='cpu')
gx.manual_seed(5)
xs = [(torch.rand(32, *cfg.shape, generator=gx) * 2 - 1).to(DEV) for _ in range(4)]


def run():
    torch.manual_seed(321)
    torch.cuda.manual_seed_all(321)
    model, args, sd = make_infodiff(cfg, DEV, 'bf16', 'manifest_celeba')
    model.train()
    opt = FusedClipAdamW(model.parameters(), lr=1e-4, weight_decay=1e-5, max_norm=1.0)
    step = GraphedTrainStep(model, args_of(cfg), opt)
    out = []
    for k in range(N):
        lv = step(xs[k % len(xs)], 0)
        if k % 20 == 0 or k == N - 1:
            out.append((float(lv), float(opt.total_norm())))
    assert step.graph is not None
    torch.cuda.synchronize()
    return out, [p.detach().clone() for p in model.parameters()]


assert ops._WGRAD_DET
a, pa = run()
b, pb = run()
bad = sum(not torch.equal(u, v) for u, v in zip(pa, pb))
print('steps %d  loss %.5f -> %.5f  records equal: %s  parameters differing: %d of %d  sync time-outs: %d' % (
    N, a[0][0], a[-1][0], a == b, bad, len(pa), ops.rs_sync_timeouts(False)))
assert a == b and bad == 0 and ops.rs_sync_timeouts(False) == 0
assert all(x[0] == x[0] and x[0] < 1e3 for x in a)
