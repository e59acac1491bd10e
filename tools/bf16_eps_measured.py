#!/usr/bin/env python3
"""epsilon-hat of the bf16 path against the reference fixtures (fp32 reference outputs): the numbers tests/test_gpu_model.py bounds
with BF16_EPS_TOL (CelebA, config5) and BF16_EPS_TOL_FMNIST.  Each model is evaluated three times (first pass = the layouts a
network starts with, later passes = steady-state kernels).  Usage: python tools/bf16_eps_measured.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle import infodiff_oracle as O
from tests.helpers import args_of, gold, make_infodiff, manifest, rel, rel_l2

DEV = 'cuda'
for tag, ds, man in (('model_celeba', 'celeba', 'manifest_celeba'), ('model_fmnist', 'fmnist', 'manifest_fmnist')):
    cfg = O.dataset_cfg(ds, a_dim=32, mmd_weight=0.1)
    g = gold(tag)
    model, args, sd = make_infodiff(cfg, DEV, 'bf16', man)
    model.eval()
    for k in range(3):
        with torch.no_grad():
            e17 = model(g['samp_x'].to(DEV), 17, g['samp_a'].to(DEV))
        print('%-13s pass %d  eps-hat(t = 17): max-norm %.4e  rel-L2 %.4e' % (tag, k, rel(e17, g['samp_eps17']), rel_l2(e17, g['samp_eps17'])))
from infodiffusion_amd.models import Diff
g = gold('config5_cifar')
cfg = O.dataset_cfg('cifar10', a_dim=256, diffusion_steps=4, deterministic=True, model='diff', is_latent=False, mode='eval_fid', split_step=1)
m2 = Diff(args_of(cfg, act_dtype='bf16'), DEV, cfg.shape)
m2.load_state_dict(O.synth_state_dict(manifest('manifest_vanilla_cifar')), strict=True)
m2.eval()
for k in range(3):
    with torch.no_grad():
        y = m2(g['x'].to(DEV), 2)
    print('config5 vanilla UNet pass %d  eps-hat(t = 2): max-norm %.4e  rel-L2 %.4e' % (k, rel(y, g['y2']), rel_l2(y, g['y2'])))
