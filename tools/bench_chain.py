#!/usr/bin/env python3
"""The big-map backward chain at the CelebA training shapes, COLD (every call on the next of N buffer sets, > 512 MB in
rotation, captured in a hipGraph): idf_gn_bwd_apply (algorithmic GB/s: du, x (, dres) read, dx written) and the data-gradient
conv with / without the du epilogue.  Usage: python tools/bench_chain.py [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from infodiffusion_amd import ops
from tools.bench_gnbwd import Slot, timeit

DEV, CL = 'cuda', torch.channels_last
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
SHAPES = [(64, 0, 64), (128, 0, 32), (128, 64, 64), (64, 64, 64), (128, 128, 32), (128, 0, 16), (128, 128, 16), (128, 128, 8)]   # (C1, C2, H)


def main():
    for C1, C2, H in SHAPES:
        C = C1 + C2
        for with_res in (False, True):
            per = B * C * H * H * 2 * (4 if with_res else 3)
            n = max(4, min(64, (768 << 20) // per))
            g, b_ = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
            sets = []
            T = max(1, H * H // 256)
            for _ in range(n):
                x1 = torch.randn(B, C1, H, H, device=DEV).bfloat16().contiguous(memory_format=CL)
                x2 = torch.randn(B, C2, H, H, device=DEV).bfloat16().contiguous(memory_format=CL) if C2 else None
                du = torch.randn(B, C, H, H, device=DEV).bfloat16().contiguous(memory_format=CL)
                dres = torch.randn_like(du) if with_res else None
                part = torch.randn(B, T, C, 2, device=DEV)
                m, r, sc = torch.randn(B, 32, device=DEV), torch.rand(B, 32, device=DEV) + 0.5, torch.randn(B, C, device=DEV)
                sets.append((du, part, x1, x2, dres, m, r, sc))
            acc = (Slot(C), Slot(C))
            calls = [(lambda s=s: ops.gn_bwd_apply_raw(s[0], s[1], s[2], g, b_, None, None, s[5], s[6], s[7], acc=acc, dres=s[4],
                                                       x2=s[3])) for s in sets]
            t = timeit(calls)
            print('apply  B %3d C %3d+%3d %2dx%2d res %d  sets %2d  %7.1f us  %6.0f GB/s' % (B, C1, C2, H, H, with_res, n, t, per / t / 1e3),
                  flush=True)
    for Cin, C, H in [(64, 64, 64), (128, 128, 32), (64, 128, 64)]:
        per = B * (Cin + 2 * C) * H * H * 2
        n = max(4, min(32, (768 << 20) // per))
        w = (torch.randn(Cin, C, 3, 3, device=DEV) / (9 * C) ** 0.5)
        _, wd = ops.pack_weight(w, torch.bfloat16, True, True)
        seed = torch.tensor([1234567], dtype=torch.int64, device=DEV)
        sets = []
        for _ in range(n):
            x = torch.randn(B, C, H, H, device=DEV).bfloat16().contiguous(memory_format=CL)
            dy = torch.randn(B, Cin, H, H, device=DEV).bfloat16().contiguous(memory_format=CL)
            sets.append((x, dy, torch.randn(B, C, device=DEV), torch.randn(B, C, device=DEV)))
        t0 = timeit([(lambda s=s: ops.conv_dgrad_raw(s[1], wd, ops.S1, 9, s[0].shape)) for s in sets])
        t1 = timeit([(lambda s=s: ops.conv_dgrad_chain_raw(s[1], wd, 9, C, x=s[0], sc=s[2], sh=s[3], seed=seed, salt=3, p_drop=0.1, act=2)) for s in sets])
        t2 = timeit([(lambda s=s: ops.conv_dgrad_chain_raw(s[1], wd, 9, C, x=s[0], sc=s[2], sh=s[3], act=2)) for s in sets])
        print('dgrad  B %3d %3d->%3d %2dx%2d  plain %6.1f us   du epilogue (dropout) %6.1f us   du epilogue (no dropout) %6.1f us' % (
            B, Cin, C, H, H, t0, t1, t2), flush=True)


if __name__ == '__main__':
    main()
