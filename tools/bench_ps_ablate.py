#!/usr/bin/env python3
"""One shape of the persistent conv under the IDF_CONV_PS_DBG ablations (timing only). Usage: B Cin Cout H pro"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infodiffusion_amd import ops
from tools.bench_gnconv import timeit
B, Cin, Cout, H, pro = [int(v) for v in sys.argv[1:6]]
DEV, CL = 'cuda', torch.channels_last
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
if pro:
    t = timeit([lambda x=x, st=st: ops.conv_gn_raw(x, None, st, None, g, b_, None, None, None, 3, 0.0, 2, wf, bias, None, Cout, 9,
                                                   want_stats=True) for x, st in sets])
else:
    t = timeit([lambda x=x: ops.conv_raw(x, wf, bias, None, None, None, None, 0, 0.0, ops.S1, 9, 0, Cout, want_stats=True)
                for x, _ in sets])
print('dbg %s B %d %d->%d %dx%d pro %d: %.1f us' % (os.environ.get('IDF_CONV_PS_DBG', '0'), B, Cin, Cout, H, H, pro, t))
