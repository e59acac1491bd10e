#!/usr/bin/env python3
"""Where do the stray ATen kernels of one training step come from?  TorchDispatchMode census of
copy_/clone/fill/zero/add/cat/... with the innermost repo frames.  Usage: python tools/op_sources.py"""
import os
import sys
import traceback
import types
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.utils._python_dispatch import TorchDispatchMode

import bench
from infodiffusion_amd.models import InfoDiff
from infodiffusion_amd.optim import FusedClipAdamW

a = types.SimpleNamespace(a_dim=32, batch=32, dtype='bf16')
margs = bench.make_args(a)
dev = torch.device('cuda', 0)
model = InfoDiff(margs, dev, (3, 64, 64)).train()
opt = FusedClipAdamW(model.parameters(), lr=1e-4, weight_decay=1e-5)
x = torch.rand(32, 3, 64, 64, device=dev) * 2 - 1
SKIP = ('empty', 'as_strided', 'slice', 'detach', 'view', 'narrow', 'permute', 'alias', 'select', 'unsqueeze',
        'squeeze', 'expand', 'transpose', 't.', 'reshape', '_unsafe_view', 'split', 'unbind', 'chunk', 'stride',
        'size', 'is_', 'sym_', 'numel', 'dim', 'record_stream', '_local_scalar')
cnt = Counter()


class Census(TorchDispatchMode):
    def __torch_dispatch__(self, func, types_, args=(), kwargs=None):
        name = str(func)
        if not any(s in name for s in SKIP):
            fr = [f for f in traceback.extract_stack() if ROOT in f.filename and 'op_sources' not in f.filename]
            where = ' < '.join('%s:%d' % (os.path.basename(f.filename), f.lineno) for f in fr[-3:][::-1])
            shp = next((tuple(t.shape) for t in args if isinstance(t, torch.Tensor)), ())
            cnt[(name, where, shp if len(shp) < 3 else ())] += 1
        return func(*args, **(kwargs or {}))


def step():
    loss = model.loss_fn(margs, x)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()


for _ in range(2):
    step()
with Census():
    step()
torch.cuda.synchronize()
for (name, where, shp), n in sorted(cnt.items(), key=lambda kv: -kv[1])[:60]:
    print('%4d  %-28s %-18s %s' % (n, name.replace('aten.', ''), shp, where))
