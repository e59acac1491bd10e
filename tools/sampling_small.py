#!/usr/bin/env python3
"""DDIM-100 sampling time at small batches, eager steps vs the replayed captured step."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import infodiff_oracle as O          # noqa: E402  (dataset table only)
from tests.helpers import args_of                 # noqa: E402
from infodiffusion_amd import sampling as S       # noqa: E402
from infodiffusion_amd.models import InfoDiff     # noqa: E402

cfg = O.dataset_cfg('celeba', a_dim=32, mmd_weight=0.1, diffusion_steps=100, deterministic=True)
args = args_of(cfg, act_dtype='bf16')
model = InfoDiff(args, torch.device('cuda'), cfg.shape).eval()
proc = S.DiffusionProcess(args, model, torch.device('cuda'), cfg.shape)
for B in ([int(v) for v in sys.argv[1:]] or [1, 16, 32, 64, 128, 256]):
    row = []
    for graph in (False, True):
        S.GRAPH = graph
        S.GRAPH_MAX_PIXELS = 1 << 40
        proc.sampling(B)
        torch.cuda.synchronize()
        t0 = time.time()
        proc.sampling(B)
        torch.cuda.synchronize()
        row.append(time.time() - t0)
    print('B=%3d  eager %.3f s (%.1f img/s)   graphed %.3f s (%.1f img/s)' % (B, row[0], B / row[0], row[1], B / row[1]))
