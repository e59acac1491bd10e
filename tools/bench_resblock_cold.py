#!/usr/bin/env python3
"""The image-resident 8x8 ResBlock forward with WARM operands (the same launch replayed back to back) against COLD ones (a 640-MB
fill between the launches: nothing of the block's weights or inputs survives in L2 / MALL; the fill's own time is subtracted).
usage: tools/bench_resblock_cold.py [B]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infodiffusion_amd import modules, ops
from tools.bench_resblock import make, DEV, CL

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
big = torch.empty(640 << 20, dtype=torch.uint8, device=DEV)


def timed(fns, reps=10):
    for _ in range(3):
        for f in fns:
            f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
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
    return e0.elapsed_time(e1) / (3 * reps) * 1e3


for kind, cin in (('aux', 128), ('enc', 128)):
    blk = make(kind, cin)
    x1 = torch.randn(B, 128, 8, 8, device=DEV).bfloat16().contiguous(memory_format=CL).requires_grad_(True)
    x1._gn = ops.gn_partials_raw(x1)
    ft = torch.randn(B, 256, device=DEV) * 0.1
    fa = torch.randn(B, 256, device=DEV) * 0.1

    def fwd():
        if kind == 'aux':
            blk._film = {'t': ft, 'a': fa}
            return blk(x1, None, None)
        return blk(x1)

    def flush():
        big.fill_(1)
    ops._RB_SMALL = True
    sset = modules.ShadowSet(blk)
    for _ in range(2):           # what a network's forward pass does around its blocks: layouts requested on the way settle behind it
        sset.refresh(torch.bfloat16, True)
        fwd()
        sset.settle(torch.bfloat16, True)
    warm = timed([fwd])
    fl = timed([flush])
    cold = timed([flush, fwd]) - fl
    print('%s 128->128 B %d: warm %.1f us   cold %.1f us   (fill alone %.1f us)' % (kind, B, warm, cold, fl))
