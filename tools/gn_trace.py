#!/usr/bin/env python3
"""GroupNorm-apply variants at CelebA L0/L1 shapes (run under rocprofv3 --kernel-trace):
act=1 (affine), act=2 (SiLU), act=2 + dropout; then backward partial / apply."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from infodiffusion_amd import ops

DEV, CL = 'cuda', torch.channels_last
for C, H in [(64, 64), (128, 32)]:
    x = torch.randn(32, C, H, H, device=DEV).bfloat16().contiguous(memory_format=CL)
    dA = torch.randn_like(x)
    g, b = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    seed = torch.tensor([7], dtype=torch.int64, device=DEV)
    mean, rstd, sc, sh = ops.gn_coef_fwd_raw(x, g, b, None, None)
    for _ in range(5):
        ops.gn_apply_raw(x, sc, sh, None, 3, 0.0, 1)
    torch.cuda.synchronize()
    for _ in range(5):
        ops.gn_apply_raw(x, sc, sh, None, 3, 0.0, 2)
    torch.cuda.synchronize()
    for _ in range(5):
        ops.gn_apply_raw(x, sc, sh, seed, 3, 0.1, 2)
    torch.cuda.synchronize()
    for _ in range(5):
        ops.gn_coef_bwd_raw(dA, x, None, g, b, None, None, mean, rstd, sc, sh, None, 3, 0.0, 2)
    torch.cuda.synchronize()
    for _ in range(5):
        ops.gn_coef_bwd_raw(dA, x, None, g, b, None, None, mean, rstd, sc, sh, seed, 3, 0.1, 2)
    torch.cuda.synchronize()
