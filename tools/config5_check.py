#!/usr/bin/env python3
"""BASELINE configs[4] on one GPU's shard: CIFAR-10 shape, latent DDIM (LatentUNet) -> `a`, then the image sampler
(DiffusionProcess with the latent, and TwoPhaseDiffusionProcess), B = 64 per GPU, T = 100, bf16; graphed vs eager."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import infodiff_oracle as O          # noqa: E402  (dataset table only)
from tests.helpers import args_of                 # noqa: E402
from infodiffusion_amd import sampling as S       # noqa: E402
from infodiffusion_amd.models import Diff, InfoDiff  # noqa: E402

dev = torch.device('cuda')
cfg = O.dataset_cfg('cifar10', a_dim=256, mmd_weight=0.1, diffusion_steps=100, deterministic=True)
args = args_of(cfg, act_dtype='bf16', is_latent=False, mode='eval_fid', split_step=50)
model = InfoDiff(args, dev, cfg.shape).eval()
vanilla = Diff(args, dev, cfg.shape).eval()
largs = args_of(cfg, act_dtype='bf16', is_latent=True, mode='eval_fid', split_step=50)
latent = Diff(largs, dev, (1, cfg.a_dim, cfg.a_dim)).eval()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for graph in (False, True):
    S.GRAPH = graph
    pl = S.LatentDiffusionProcess(largs, latent, dev)
    pi = S.DiffusionProcess(args, model, dev, cfg.shape)
    p2 = S.TwoPhaseDiffusionProcess(args, model, vanilla, dev, cfg.shape)
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.time()
        a = pl.sampling(sampling_number=B)
        torch.cuda.synchronize()
        t1 = time.time()
        img = pi.sampling(sampling_number=B, a=a)
        torch.cuda.synchronize()
        t2 = time.time()
        img2 = p2.sampling(sampling_number=B)
        torch.cuda.synchronize()
        t3 = time.time()
    assert a.shape == (B, cfg.a_dim) and img.shape == (B, 3, 32, 32) and img2.shape == (B, 3, 32, 32)
    assert torch.isfinite(a).all() and torch.isfinite(img.float()).all() and torch.isfinite(img2.float()).all()
    print('graph=%d  latent DDIM-100 %.3f s | image sampler %.3f s (%.0f img/s) | two-phase %.3f s (%.0f img/s)' %
          (graph, t1 - t0, t2 - t1, B / (t2 - t1), t3 - t2, B / (t3 - t2)))
