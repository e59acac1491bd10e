#!/usr/bin/env python3
"""DDIM sampling only (CelebA shape, bf16): `run_sampling.py [B] [T] [reps]` -- the workload to put under
rocprofv3 for tools/eval_inventory.py (every kernel launch belongs to a network evaluation or a sampler update)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import infodiff_oracle as O          # noqa: E402  (dataset table only)
from tests.helpers import args_of                 # noqa: E402
from infodiffusion_amd import sampling as S       # noqa: E402
from infodiffusion_amd.models import InfoDiff     # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
cfg = O.dataset_cfg('celeba', a_dim=32, mmd_weight=0.1, diffusion_steps=T, deterministic=True)
args = args_of(cfg, act_dtype='bf16')
model = InfoDiff(args, torch.device('cuda'), cfg.shape).eval()
proc = S.DiffusionProcess(args, model, torch.device('cuda'), cfg.shape)
for r in range(reps):
    torch.cuda.synchronize()
    t0 = time.time()
    proc.sampling(B)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print('B=%d T=%d: %.3f s  %.1f img/s  %.2f ms / evaluation' % (B, T, dt, B / dt, dt / T * 1e3))
