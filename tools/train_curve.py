#!/usr/bin/env python3
"""Loss trajectory of the product's training loop (GraphedTrainStep, fused clip+AdamW, bf16, dropout on) on a FIXED
pool of 8 random-pixel CelebA-shaped batches: the network can memorise the pool's noise statistics only through
the denoising objective, so the loss must fall from ~1 (epsilon-MSE of an untrained net) and stay finite.
Also runs the same steps with --act_dtype fp32 for comparison.  Usage: train_curve.py [steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import infodiff_oracle as O          # noqa: E402  (dataset table only)
from tests.helpers import args_of                 # noqa: E402
from infodiffusion_amd.models import InfoDiff     # noqa: E402
from infodiffusion_amd.optim import FusedClipAdamW  # noqa: E402
from infodiffusion_amd.trainer import GraphedTrainStep  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dev = torch.device('cuda')
cfg = O.dataset_cfg('celeba', a_dim=32, mmd_weight=0.1)
g = torch.Generator(device='cpu')
g.manual_seed(64)
pool = [(torch.rand(32, 3, 64, 64, generator=g) * 2 - 1).to(dev).contiguous(memory_format=torch.channels_last)
        for _ in range(8)]
sync_mode = os.environ.get('IDF_FORCE_SYNC') == '1'      # 1-rank RCCL group: the whole data-parallel code path on one GPU
if sync_mode:
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29534')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
for dt in ('bf16', 'fp32'):
    args = args_of(cfg, act_dtype=dt, batch_size=32)
    torch.manual_seed(64)
    model = InfoDiff(args, dev, cfg.shape).train()
    opt = FusedClipAdamW(model.parameters(), lr=2e-4, weight_decay=1e-5, max_norm=1.0)
    sync = None
    if sync_mode:
        from infodiffusion_amd.dist import GradSync
        sync = GradSync(model, 1, force=True, arena=opt.arena)
        sync.broadcast_parameters()
    step = GraphedTrainStep(model, args, opt, sync)
    hist = []
    acc = torch.zeros((), device=dev)
    for i in range(steps):
        acc += step(pool[i % 8], 0)
        if (i + 1) % 50 == 0:
            hist.append(float(acc) / 50)
            acc.zero_()
    print(dt + (' +sync' if sync_mode else ''), 'mean loss per 50 steps:', ' '.join('%.4f' % v for v in hist))
    assert all(v == v for v in hist) and hist[-1] < 0.5 * hist[0], hist
