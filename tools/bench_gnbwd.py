#!/usr/bin/env python3
"""GroupNorm/FiLM/SiLU backward (idf_gn_fused_bwd) at the CelebA training shapes, COLD: every call works on the next of
N buffer sets (> 512 MB in rotation, so nothing is L2 / MALL resident), captured in a hipGraph.  Prints per-shape time
and algorithmic GB/s (reads x, dA, dres; writes dx).  Usage: python tools/bench_gnbwd.py [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from infodiffusion_amd import ops

DEV, CL = 'cuda', torch.channels_last
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
SHAPES = [(64, 64), (128, 64), (128, 32), (256, 32), (128, 16), (256, 16), (128, 8), (256, 8)]     # (C, H)


class Slot:
    """Stand-in for a gradient-arena slot (ops._gn_acc)."""
    def __init__(self, n):
        self.t = torch.zeros(n, device=DEV)

    def available(self):
        return True

    def take(self):
        return self.t


def timeit(calls, reps=3):
    for f in calls[:2]:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side), torch.cuda.graph(g, stream=side):
        for f in calls:
            f()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * len(calls)) * 1e3


def main():
    for C, H in SHAPES:
        for with_res in (False, True):
            per = B * C * H * H * 2 * (4 if with_res else 3)
            n = max(4, min(64, (768 << 20) // per))
            g, b_ = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
            ft = 0.1 * torch.randn(B, 2 * C, device=DEV)
            seed = torch.tensor([1234567], dtype=torch.int64, device=DEV)
            sets = []
            for _ in range(n):
                x = torch.randn(B, C, H, H, device=DEV).bfloat16().contiguous(memory_format=CL)
                dA = torch.randn_like(x)
                dres = torch.randn_like(x) if with_res else None
                _, m, r, sc, sh = ops.gn_fused_fwd_raw(x, g, b_, ft, None, seed, 3, 0.1, 2)
                sets.append((x, dA, dres, m, r, sc, sh))
            acc = (Slot(C), Slot(C))
            calls = [(lambda s=s: ops.gn_fused_bwd_raw(s[1], s[0], g, b_, ft, None, s[3], s[4], s[5], s[6], seed, 3, 0.1, 2,
                                                       acc=acc, dres=s[2])) for s in sets]
            t = timeit(calls)
            print('B %3d C %3d %2dx%2d res %d  sets %2d  %7.1f us  %6.0f GB/s' % (B, C, H, H, with_res, n, t, per / t / 1e3),
                  flush=True)


if __name__ == '__main__':
    main()
