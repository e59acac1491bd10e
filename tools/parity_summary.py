#!/usr/bin/env python3
"""Measured parity of the benchmarked configuration against the REFERENCE's fp32 outputs (fixture tests/golden/model_celeba.npz,
generated from the real reference by tools/gen_golden.py): epsilon-hat and the p_losses value in bf16 and in fp32, next to
north_star's tolerances.  Writes the JSON object bench.py attaches to its line as `parity` (profiles/r06_parity.json).
Usage (GPU box): python tools/parity_summary.py [out.json]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle import infodiff_oracle as O          # (test infrastructure: the checker, as in tests/)
from tests.helpers import gold, make_infodiff, rel, rel_l2
from tests.test_gpu_model import _replay

DEV = 'cuda'
cfg = O.dataset_cfg('celeba', a_dim=32, mmd_weight=0.1)
g = gold('model_celeba')
res = {'fixture': 'tests/golden/model_celeba.npz (reference fp32 CPU outputs, B = 2, CelebA 3x64x64, a_dim 32, mmd 0.1)',
       'north_star': {'eps_hat_bf16': 1e-2, 'loss_bf16': 1e-2, 'fp32': 1e-4}}
for dt in ('bf16', 'fp32'):
    model, args, sd = make_infodiff(cfg, DEV, dt, 'manifest_celeba')
    model.eval()
    e = None
    for _ in range(3):          # steady-state kernels (the first pass runs on the layouts a network starts with)
        with torch.no_grad():
            e = model(g['samp_x'].to(DEV), 17, g['samp_a'].to(DEV))
    loss = None
    for _ in range(2):
        loss = _replay(model, cfg, g, 0)
    res['eps_hat_%s_max' % dt] = float('%.3e' % rel(e, g['samp_eps17']))
    res['eps_hat_%s_rel_l2' % dt] = float('%.3e' % rel_l2(e, g['samp_eps17']))
    res['loss_%s' % dt] = float('%.3e' % rel(loss, g['loss']))
res['meets_north_star'] = {'eps_hat_bf16': res['eps_hat_bf16_max'] <= 1e-2, 'loss_bf16': res['loss_bf16'] <= 1e-2,
                           'eps_hat_fp32': res['eps_hat_fp32_max'] <= 1e-4, 'loss_fp32': res['loss_fp32'] <= 1e-4}
res['note'] = ('bf16 epsilon-hat misses north_star\'s 1e-2: rounding only the MFMA operands to bf16 (every stored tensor fp32) already '
               'costs 9.8e-3 max-norm (profiles/r03_bf16_error_profile.txt); tests pin measured + 15 % (tests/test_gpu_model.py BF16_EPS_TOL)')
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'r06_parity.json')
with open(out, 'w') as f:
    json.dump(res, f, indent=1)
print(json.dumps(res))
