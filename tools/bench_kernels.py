#!/usr/bin/env python3
"""Micro-benchmarks of the hot kernels at the CelebA training shapes (B = 32, bf16):
per-shape time, algorithmic TFLOP/s and GB/s.  Usage: python tools/bench_kernels.py [conv|wgrad|gn|all]

HOT numbers: every call of a shape works on the same buffers, which stay L2 / Infinity-Cache resident between calls
(the largest tensor here is 34 MB against 256 MB of Infinity Cache), so the GB/s columns are cache rates, not HBM
rates.  The cold counterparts, which bench.py's rooflines are built from: tools/bench_gnbwd.py (GroupNorm backward),
tools/bench_gnconv.py (GroupNorm + conv, both forms), bench.py's LaunchRecorder (the step's own launches).
GB/s counts the algorithmic passes: GroupNorm forward = x read + a written (2 passes), backward = x, dA read + dx
written (3 passes)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from infodiffusion_amd import ops

DEV, CL = 'cuda', torch.channels_last
B = 32
SHAPES = [(64, 64, 64), (128, 64, 64), (192, 64, 64), (128, 128, 64), (128, 128, 32), (256, 128, 32), (128, 128, 16),
          (256, 128, 16), (128, 128, 8)]   # (Cin, Cout, H)


def timeit(fn, n=20):
    """Average GPU time of one call: n calls captured in a hipGraph (no host launch gaps), replayed."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side), torch.cuda.graph(g, stream=side):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * n) * 1e3   # us


def main(which):
    for Cin, Cout, H in SHAPES:
        x = torch.randn(B, Cin, H, H, device=DEV).bfloat16().contiguous(memory_format=CL)
        dy = torch.randn(B, Cout, H, H, device=DEV).bfloat16().contiguous(memory_format=CL)
        w = torch.randn(Cout, Cin, 3, 3, device=DEV) * 0.05
        wf, wd = ops.pack_weight(w, torch.bfloat16, True, True)
        bias = torch.zeros(Cout, device=DEV)
        fl = 2.0 * B * H * H * Cin * Cout * 9
        by = (x.numel() + dy.numel()) * 2
        line = 'Cin %3d Cout %3d %2dx%2d  %6.2f GF %6.1f MB |' % (Cin, Cout, H, H, fl / 1e9, by / 1e6)
        if which in ('conv', 'all'):
            t = timeit(lambda: ops.conv_raw(x, wf, bias, None, None, None, None, 0, 0.0, ops.S1, 9, 0, Cout))
            line += ' fwd %7.1f us %6.1f TF/s %6.0f GB/s |' % (t, fl / t / 1e6, by / t / 1e3)
            t = timeit(lambda: ops.conv_dgrad_raw(dy, wd, ops.S1, 9, x.shape))
            line += ' dgrad %7.1f us %6.1f TF/s |' % (t, fl / t / 1e6)
        if which in ('wgrad', 'all'):
            t = timeit(lambda: ops.conv_wgrad_bias_raw(x, dy, ops.S1, 9, True))
            line += ' wgrad %7.1f us %6.1f TF/s %6.0f GB/s |' % (t, fl / t / 1e6, by / t / 1e3)
        if which in ('gn', 'all'):
            g, b_ = torch.ones(Cin, device=DEV), torch.zeros(Cin, device=DEV)

            def gn():
                if ops.gn_small_ok(x):
                    return ops.gn_fused_fwd_raw(x, g, b_, None, None, None, 0, 0.0, 2)
                m, r, sc, sh = ops.gn_coef_fwd_raw(x, g, b_, None, None)
                return ops.gn_apply_raw(x, sc, sh, None, 0, 0.0, 2)
            t = timeit(gn)
            line += ' gn-fwd %6.1f us %5.0f GB/s' % (t, 2 * x.numel() * 2 / t / 1e3)
            dA = torch.randn_like(x)
            if ops.gn_small_ok(x):
                _, m, r, sc, sh = ops.gn_fused_fwd_raw(x, g, b_, None, None, None, 0, 0.0, 2)
                t = timeit(lambda: ops.gn_fused_bwd_raw(dA, x, g, b_, None, None, m, r, sc, sh, None, 0, 0.0, 2))
            else:
                m, r, sc, sh = ops.gn_coef_fwd_raw(x, g, b_, None, None)
                t = timeit(lambda: ops.gn_coef_bwd_raw(dA, x, None, g, b_, None, None, m, r, sc, sh, None, 0, 0.0, 2))
            line += ' gn-bwd %6.1f us %5.0f GB/s' % (t, 3 * x.numel() * 2 / t / 1e3)
        print(line, flush=True)


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'all')
