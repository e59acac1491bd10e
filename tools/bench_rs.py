#!/usr/bin/env python3
"""The register-weights / row-reuse conv (idf_conv_rs_*) against the halo / direct-to-LDS kernels it replaces, at the CelebA
training shapes, COLD (every call on the next of N buffer sets, > 512 MB in rotation, captured in a hipGraph and replayed):
the GroupNorm-prologue forward conv (training: a_out + coefficients + statistics) and the du-epilogue data-gradient conv.
Usage: python tools/bench_rs.py [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from infodiffusion_amd import ops
from tools.bench_gnbwd import timeit

DEV, CL = 'cuda', torch.channels_last
ops._RS_FWD_ALL = True      # measure the form at every covered forward shape
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32


class Frag:
    def __init__(self, w):
        wf, wd = ops.pack_weight(w, torch.bfloat16, True, True)

        def frag(m):
            N, taps, K = m.shape
            return m.view(N // 16, 16, taps, K // 64, 2, 4, 8).permute(3, 0, 2, 4, 5, 1, 6).contiguous().view(-1)
        self.val = [wf, wd, frag(wf), frag(wd)]

    def request_frag(self):
        raise AssertionError


def main():
    for Cin, Cout, H in [(64, 64, 64), (128, 128, 32), (64, 128, 32), (128, 64, 32)]:
        if not ops.rs_tiles(B, H, H, Cin, Cout):
            print('B %d %d->%d @%d: not covered' % (B, Cin, Cout, H))
            continue
        per = B * (2 * Cin + 2 * Cout) * H * H * 2
        n = max(4, min(32, (768 << 20) // per))
        w = torch.randn(Cout, Cin, 3, 3, device=DEV) / (9 * Cin) ** 0.5
        sh = Frag(w)
        gam, bet = torch.ones(Cin, device=DEV), torch.zeros(Cin, device=DEV)
        bias = torch.zeros(Cout, device=DEV)
        ft, fa = 0.1 * torch.randn(B, 2 * Cin, device=DEV), 0.1 * torch.randn(B, 2 * Cin, device=DEV)
        seed = torch.tensor([1234567], dtype=torch.int64, device=DEV)
        sets = []
        for _ in range(n):
            x = torch.randn(B, Cin, H, H, device=DEV).bfloat16().contiguous(memory_format=CL)
            st = ops.gn_partials_raw(x)
            if st.shape[1] > 16:
                st = st.view(B, 16, st.shape[1] // 16, Cin, 2).sum(dim=2).contiguous()
            sets.append((x, st))

        def fwd(s, shadows):
            return ops.conv_gn_raw(s[0], None, s[1], None, gam, bet, ft, fa, seed, 7, 0.1, 2, sh.val[0], bias, None, Cout, 9,
                                   keep_a=True, keep_coef=True, want_stats=True, shadows=shadows)
        t0 = timeit([(lambda s=s: fwd(s, None)) for s in sets])
        t1 = timeit([(lambda s=s: fwd(s, sh)) for s in sets])
        gf = 2.0 * B * H * H * Cout * 9 * Cin / 1e9
        print('fwd GN-prologue conv  B %3d %3d->%3d @%2d  halo / dlds %6.1f us   rs %6.1f us  (%.0f -> %.0f TF/s)' % (
            B, Cin, Cout, H, t0, t1, gf / t0 * 1e3, gf / t1 * 1e3), flush=True)
    for Cin, C, H in [(64, 64, 64), (128, 128, 32), (64, 128, 32), (128, 64, 32)]:
        if not ops.rs_tiles(B, H, H, Cin, C):
            continue
        per = B * (Cin + 2 * C) * H * H * 2
        n = max(4, min(32, (768 << 20) // per))
        w = torch.randn(Cin, C, 3, 3, device=DEV) / (9 * C) ** 0.5
        sh = Frag(w)
        seed = torch.tensor([1234567], dtype=torch.int64, device=DEV)
        sets = []
        for _ in range(n):
            x = torch.randn(B, C, H, H, device=DEV).bfloat16().contiguous(memory_format=CL)
            dy = torch.randn(B, Cin, H, H, device=DEV).bfloat16().contiguous(memory_format=CL)
            sets.append((x, dy, torch.randn(B, C, device=DEV), torch.randn(B, C, device=DEV)))

        def bwd(s, shadows):
            return ops.conv_dgrad_chain_raw(s[1], sh.val[1], 9, C, x=s[0], sc=s[2], sh=s[3], seed=seed, salt=3, p_drop=0.1, act=2,
                                            shadows=shadows)
        t0 = timeit([(lambda s=s: bwd(s, None)) for s in sets])
        t1 = timeit([(lambda s=s: bwd(s, sh)) for s in sets])
        gf = 2.0 * B * H * H * C * 9 * Cin / 1e9
        print('dgrad, du epilogue    B %3d %3d->%3d @%2d  halo        %6.1f us   rs %6.1f us  (%.0f -> %.0f TF/s)' % (
            B, Cin, C, H, t0, t1, gf / t0 * 1e3, gf / t1 * 1e3), flush=True)
        # the whole GroupNorm-stage backward: du-epilogue conv + streaming apply (two launches) against the group-synchronised launch
        gam, bet = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
        ft, fa = 0.1 * torch.randn(B, 2 * C, device=DEV), 0.1 * torch.randn(B, 2 * C, device=DEV)
        mean, rstd = torch.zeros(B, 32, device=DEV), torch.ones(B, 32, device=DEV)
        for nres in (0, 1):
            res = [torch.randn(B, C, H, H, device=DEV).bfloat16().contiguous(memory_format=CL) for _ in range(n)] if nres else [None] * n

            def two(s, r):
                du, part, _ = bwd(s, sh)
                return ops.gn_bwd_apply_raw(du, part, s[0], gam, bet, ft, fa, mean, rstd, s[2], dres=r)

            def one(s, r):
                return ops.conv_dgrad_gn_sync_raw(s[1], C, s[0], gam, bet, ft, fa, mean, rstd, s[2], s[3], seed, 3, 0.1, 2, None, r, None,
                                                  shadows=sh)
            if one(sets[0], res[0]) is None:
                continue
            t2 = timeit([(lambda s=s, r=r: two(s, r)) for s, r in zip(sets, res)])
            t3 = timeit([(lambda s=s, r=r: one(s, r)) for s, r in zip(sets, res)])
            print('   GroupNorm-stage backward (dres %d)   conv + apply %6.1f us   one synchronised launch %6.1f us' % (nres, t2, t3), flush=True)
        assert ops.rs_sync_timeouts() == 0


if __name__ == '__main__':
    main()
