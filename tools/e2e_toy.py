#!/usr/bin/env python3
"""End-to-end behaviour check on one MI355X (not a parity test): train the product on a toy distribution the
objective can learn -- 32x32 one-channel images that are CONSTANT (one random grey level per image) -- through the
graphed training step, then draw images with the product's DDIM sampler (T = 1000, eta = 0.01 as the reference) and
with DDPM, from x_T ~ N(0, I) and latents a ~ N(0, I).  Untrained, the samplers return noise-like images (per-image
pixel std ~ 1); trained, the images must come out nearly constant, like the data.  Usage: e2e_toy.py [train_steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import infodiff_oracle as O          # noqa: E402  (dataset table only)
from tests.helpers import args_of                 # noqa: E402
from infodiffusion_amd.models import InfoDiff     # noqa: E402
from infodiffusion_amd.optim import FusedClipAdamW  # noqa: E402
from infodiffusion_amd.sampling import DiffusionProcess  # noqa: E402
from infodiffusion_amd.trainer import GraphedTrainStep  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
dev = torch.device('cuda')
cfg = O.dataset_cfg('fmnist', a_dim=32, mmd_weight=0.1, diffusion_steps=1000, deterministic=True)
args = args_of(cfg, act_dtype='bf16', batch_size=64)
torch.manual_seed(3)
model = InfoDiff(args, dev, cfg.shape)


def image_std(x):
    return float(x.float().flatten(1).std(dim=1).mean())


def draw(det):
    args.deterministic = det
    model.eval()
    out = DiffusionProcess(args, model, dev, cfg.shape).sampling(sampling_number=32)
    model.train()
    return out


before = image_std(draw(True))
model.train()
opt = FusedClipAdamW(model.parameters(), lr=2e-4, weight_decay=1e-5, max_norm=1.0)
step = GraphedTrainStep(model, args, opt)
gd = torch.Generator(device=dev)
gd.manual_seed(4)
acc, hist = torch.zeros((), device=dev), []
t0 = time.time()
for i in range(steps):
    level = torch.rand(64, 1, 1, 1, generator=gd, device=dev) * 1.6 - 0.8
    x = level.expand(64, 1, 32, 32).contiguous()
    acc += step(x, 0)
    if (i + 1) % 250 == 0:
        hist.append(float(acc) / 250)
        acc.zero_()
print('train: %d steps in %.1f s, mean loss per 250 steps: %s' % (steps, time.time() - t0, ' '.join('%.4f' % v for v in hist)))
ddim, ddpm = draw(True), draw(False)
print('per-image pixel std  untrained DDIM %.3f | trained DDIM %.3f | trained DDPM %.3f   (data: 0.000)'
      % (before, image_std(ddim), image_std(ddpm)))
print('image means of 8 DDIM samples:', ' '.join('%.2f' % v for v in ddim.float().mean(dim=(1, 2, 3))[:8].tolist()))
assert hist[-1] < 0.2 * hist[0]
assert image_std(ddim) < 0.25 * before and image_std(ddpm) < 0.25 * before
