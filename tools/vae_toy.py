#!/usr/bin/env python3
"""--model vae end to end on one MI355X: the Encoder -> Decoder baseline trained through the graphed step on constant-grey
32x32 images must learn to reconstruct them (reconstruction MSE falls from ~0.2 towards 0)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import infodiff_oracle as O          # noqa: E402  (dataset table only)
from tests.helpers import args_of                 # noqa: E402
from infodiffusion_amd.models import VAE          # noqa: E402
from infodiffusion_amd.optim import FusedClipAdamW  # noqa: E402
from infodiffusion_amd.trainer import GraphedTrainStep  # noqa: E402

dev = torch.device('cuda')
cfg = O.dataset_cfg('fmnist', a_dim=32, mmd_weight=0.1)
args = args_of(cfg, act_dtype='bf16', batch_size=64)
torch.manual_seed(0)
model = VAE(args, dev, cfg.shape).train()
opt = FusedClipAdamW(model.parameters(), lr=2e-4, weight_decay=1e-5, max_norm=1.0)
step = GraphedTrainStep(model, args, opt)
gd = torch.Generator(device=dev)
gd.manual_seed(1)
acc, hist = torch.zeros((), device=dev), []
for i in range(1500):
    x = (torch.rand(64, 1, 1, 1, generator=gd, device=dev) * 1.6 - 0.8).expand(64, 1, 32, 32).contiguous()
    acc += step(x, 0)
    if (i + 1) % 100 == 0:
        hist.append(float(acc) / 100)
        acc.zero_()
print('vae mean loss per 100 steps:', ' '.join('%.4f' % v for v in hist))
model.eval()
with torch.no_grad():
    x = (torch.linspace(-0.8, 0.8, 8, device=dev).view(8, 1, 1, 1)).expand(8, 1, 32, 32).contiguous()
    rec = model(x)
print('grey level in  :', ' '.join('%.2f' % v for v in x.float().mean(dim=(1, 2, 3)).tolist()))
print('reconstruction :', ' '.join('%.2f' % v for v in rec.float().mean(dim=(1, 2, 3)).tolist()))
assert hist[-1] < 0.5 * hist[0]      # runs plateau at different levels (0.001 ... 0.05): atomics-order noise + Adam
