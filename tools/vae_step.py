#!/usr/bin/env python3
"""Sanity + timing of the --model vae training step at CelebA scale (widths 64..512) on one MI355X."""
import sys
import time
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import infodiff_oracle as O          # noqa: E402  (dataset table only)
from tests.helpers import args_of                 # noqa: E402
from infodiffusion_amd.models import VAE          # noqa: E402
from infodiffusion_amd.optim import FusedClipAdamW  # noqa: E402
from infodiffusion_amd.trainer import GraphedTrainStep  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
cfg = O.dataset_cfg('celeba', a_dim=32, mmd_weight=0.1)
args = args_of(cfg, act_dtype='bf16', batch_size=B)
torch.manual_seed(0)
model = VAE(args, torch.device('cuda'), cfg.shape).train()
opt = FusedClipAdamW(model.parameters(), lr=1e-4, weight_decay=1e-5, max_norm=1.0)
step = GraphedTrainStep(model, args, opt)
x = (torch.rand(B, *cfg.shape, device='cuda') * 2 - 1).contiguous(memory_format=torch.channels_last)
losses = [float(step(x, 0)) for _ in range(6)]
torch.cuda.synchronize()
t0 = time.time()
for _ in range(20):
    step(x, 0)
torch.cuda.synchronize()
dt = (time.time() - t0) / 20
print('losses', ['%.6f' % v for v in losses])
print('vae celeba B=%d bf16: %.2f ms/step, %.0f img/s' % (B, dt * 1e3, B / dt))
assert all(v == v and abs(v) < 1e3 for v in losses)
