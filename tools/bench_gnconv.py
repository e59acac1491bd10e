#!/usr/bin/env python3
"""GroupNorm+SiLU(+dropout) -> conv3x3 at the CelebA shapes: the two-launch path (one-launch GroupNorm kernel,
then the conv on the materialised activated tensor) against the one-launch GroupNorm-prologue conv
(idf_conv_gn_bf16), inference form and training form (dropout, activated tensor + coefficients written,
statistics of the output).  Every timed call works on its own buffer set (the sets together exceed the 256 MB
Infinity Cache where memory allows), so producers' leftovers in L2 / MALL do not flatter the numbers.
Usage: bench_gnconv.py [B ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from infodiffusion_amd import ops

DEV, CL = 'cuda', torch.channels_last
SHAPES = [(64, 64, 64), (128, 64, 64), (128, 128, 32), (256, 128, 32), (128, 128, 16), (256, 128, 16), (128, 128, 8)]


def timeit(fns, n_rep=2):
    """Average GPU time of one call; fns = one closure per buffer set, captured back to back in a hipGraph."""
    for f in fns[:2]:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side), torch.cuda.graph(g, stream=side):
        for _ in range(n_rep):
            for f in fns:
                f()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * n_rep * len(fns)) * 1e3   # us


def main(Bs):
    for B in Bs:
        for Cin, Cout, H in SHAPES:
            per_set = B * H * H * (Cin + Cout) * 2 * 2
            K = max(4, min(24, -(-(320 << 20) // per_set)))
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

            def old(x, train):
                sd, p = (seed, 0.1) if train else (None, 0.0)
                if ops.gn_small_ok(x):
                    a = ops.gn_fused_fwd_raw(x, g, b_, ft, fa, sd, 3, p, 2)[0]
                else:
                    m, r, sc, sh = ops.gn_coef_fwd_raw(x, g, b_, ft, fa)
                    a = ops.gn_apply_raw(x, sc, sh, sd, 3, p, 2)
                return ops.conv_raw(a, wf, bias, None, None, None, None, 0, 0.0, ops.S1, 9, 0, Cout)

            def new(x, st, train):
                sd, p = (seed, 0.1) if train else (None, 0.0)
                return ops.conv_gn_raw(x, None, st, None, g, b_, ft, fa, sd, 3, p, 2, wf, bias, None, Cout, 9,
                                       keep_a=train, keep_coef=train, want_stats=True)

            t_conv = timeit([lambda x=x: ops.conv_raw(x, wf, bias, None, None, None, None, 0, 0.0, ops.S1, 9, 0, Cout)
                             for x, _ in sets])
            t_oe = timeit([lambda x=x: old(x, False) for x, _ in sets])
            t_ot = timeit([lambda x=x: old(x, True) for x, _ in sets])
            t_ne = timeit([lambda x=x, st=st: new(x, st, False) for x, st in sets])
            t_nt = timeit([lambda x=x, st=st: new(x, st, True) for x, st in sets])
            fl = 2.0 * B * H * H * Cin * Cout * 9
            print('B %3d Cin %3d Cout %3d %2dx%2d sets %2d | conv %6.1f us (%5.0f TF/s) | GN+conv eval %6.1f train %6.1f | '
                  'one launch eval %6.1f (%5.0f TF/s) train %6.1f' % (B, Cin, Cout, H, H, K, t_conv, fl / t_conv / 1e6,
                                                                      t_oe, t_ot, t_ne, fl / t_ne / 1e6, t_nt), flush=True)
            del sets
            torch.cuda.empty_cache()


if __name__ == '__main__':
    main([int(v) for v in sys.argv[1:]] or [32, 256])
