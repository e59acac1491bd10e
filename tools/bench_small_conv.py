#!/usr/bin/env python3
"""Per-chunk cost of the small-map convs: plain 3x3 conv Cin -> 128 at 8x8 / 16x16, B = 32, for Cin = 32 .. 256 (1 .. 8
32-channel chunks), cold buffers, graph-timed.  The slope is what one more chunk's load -> LDS -> MFMA round costs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from infodiffusion_amd import ops
from tools.bench_gnbwd import timeit

DEV, CL = 'cuda', torch.channels_last
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for H in (8, 16, 32):
    for Cin in (32, 64, 128, 256):
        Cout = 128
        w = torch.randn(Cout, Cin, 3, 3, device=DEV) / (9 * Cin) ** 0.5
        wf, _ = ops.pack_weight(w, torch.bfloat16, True, False)
        sets = [torch.randn(B, Cin, H, H, device=DEV).bfloat16().contiguous(memory_format=CL) for _ in range(32)]
        t = timeit([(lambda x=x: ops.conv_raw(x, wf, None, None, None, None, None, 0, 0.0, ops.S1, 9, 0, Cout)) for x in sets])
        print('B %d  %3d -> %d @ %2dx%2d  %6.1f us' % (B, Cin, Cout, H, H, t), flush=True)
