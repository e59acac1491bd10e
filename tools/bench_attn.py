import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import torch
from bench_gnconv import timeit
from infodiffusion_amd import ops
DEV, CL = 'cuda', torch.channels_last
for B in (32, 256):
    sets = [torch.randn(B, 384, 16, 16, device=DEV).bfloat16().contiguous(memory_format=CL) for _ in range(8)]
    with torch.no_grad():
        t = timeit([lambda q=q: ops.attention(q) for q in sets])
    print('B %d attention fwd N=256 d=128: %.1f us  (%.0f TF/s)' % (B, t, 4.0 * B * 256 * 256 * 128 / t / 1e6))
