#!/usr/bin/env python3
"""Where the small-map GroupNorm-prologue conv's time goes (B = 32, cold buffer sets, hipGraph-timed): plain conv, + output
statistics, + affine prologue (act 1), + SiLU (act 2, eval), + training outputs (dropout, activated tensor, coefficients).
Usage: python tools/bench_pro_breakdown.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench_gnconv import timeit
from infodiffusion_amd import ops

DEV, CL = 'cuda', torch.channels_last
B = 32
for Cin, Cout, H in [(128, 128, 16), (256, 128, 16), (128, 128, 8), (256, 128, 8), (128, 128, 32)]:
    K = 24
    sets = []
    for k in range(K):
        x = torch.randn(B, Cin, H, H, device=DEV).bfloat16().contiguous(memory_format=CL)
        sets.append((x, ops.gn_partials_raw(x)))
    w = torch.randn(Cout, Cin, 3, 3, device=DEV) * 0.05
    wf, _ = ops.pack_weight(w, torch.bfloat16, True, False)
    bias = torch.zeros(Cout, device=DEV)
    g, b_ = torch.ones(Cin, device=DEV), torch.zeros(Cin, device=DEV)
    ft, fa = torch.randn(B, 2 * Cin, device=DEV) * 0.1, torch.randn(B, 2 * Cin, device=DEV) * 0.1
    seed = torch.tensor([1234], dtype=torch.int64, device=DEV)

    def pro(x, st, act, train, film=True):
        sd, p = (seed, 0.1) if train else (None, 0.0)
        return ops.conv_gn_raw(x, None, st, None, g, b_, ft if film else None, fa if film else None, sd, 3, p, act, wf, bias, None,
                               Cout, 9, keep_a=train, keep_coef=train, want_stats=True)
    t = {}
    t['plain'] = timeit([lambda x=x: ops.conv_raw(x, wf, bias, None, None, None, None, 0, 0.0, ops.S1, 9, 0, Cout) for x, _ in sets])
    t['+stats'] = timeit([lambda x=x: ops.conv_raw(x, wf, bias, None, None, None, None, 0, 0.0, ops.S1, 9, 0, Cout, want_stats=True)
                          for x, _ in sets])
    t['affine'] = timeit([lambda x=x, st=st: pro(x, st, 1, False, False) for x, st in sets])
    t['affine+film'] = timeit([lambda x=x, st=st: pro(x, st, 1, False) for x, st in sets])
    t['silu'] = timeit([lambda x=x, st=st: pro(x, st, 2, False) for x, st in sets])
    t['train'] = timeit([lambda x=x, st=st: pro(x, st, 2, True) for x, st in sets])
    print('B %d %3d->%3d @%2dx%2d  ' % (B, Cin, Cout, H, H) + '  '.join('%s %.1f' % kv for kv in t.items()), flush=True)
