"""Fused global-norm clip + AdamW (the optimizer tail of reference run.py:177, 199-200) as three
kernel launches over a device-side chunk table -- same arithmetic as
`clip_grad_norm_(params, max_norm)` followed by `torch.optim.AdamW(...).step()`."""
import os

import numpy as np
import torch

from . import ops
from .grad_arena import GradArena
from ._lib import call

_CHUNK = 8192    # elements per workgroup: 65536 ran at 3.3 TB/s, 8192 at 4.9 (tools/bench_optim.py)


class FusedClipAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_norm=1.0,
                 grad_arena=True):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, max_norm=max_norm))
        # fixed gradient slots the backward kernels accumulate into; cleared by zero_grad() in one memset
        self.arena = GradArena([p for g in self.param_groups for p in g['params']]) if grad_arena else None
        self._key = None
        self._table = self._partial = None
        self._state = None
        self._lr = None
        self._pinned = None

    def zero_grad(self, set_to_none=True):
        """Always drops the `.grad` tensors (they may alias arena slots) and clears the arena."""
        if self.arena is not None:
            self.arena.zero()
        super().zero_grad(set_to_none=True)

    def _dense_like(self, p, g):
        """Gradient memory must be element-aligned with the parameter's."""
        if g.dtype != torch.float32 or not g.is_cuda or g.shape != p.shape:
            return False
        # strides of size-1 dims carry no layout (a 1x1 conv weight is dense in either memory format)
        return all(n == 1 or a == b for n, a, b in zip(p.shape, g.stride(), p.stride()))

    @torch.no_grad()
    def step(self, closure=None):
        if len(self.param_groups) != 1:
            raise NotImplementedError('FusedClipAdamW: one parameter group')
        grp = self.param_groups[0]
        items = []
        for p in grp['params']:
            if p.grad is None:
                continue
            if not self._dense_like(p, p.grad):
                p.grad = torch.empty_like(p).copy_(p.grad)
            st = self.state[p]
            if not st:
                st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            items.append((p, p.grad, st['exp_avg'], st['exp_avg_sq']))
        if not items:
            return None
        dev = items[0][0].device
        key = tuple((p.data_ptr(), g.data_ptr()) for p, g, _, _ in items)
        if key != self._key:
            rows = []
            for p, g, m, v in items:
                n = p.numel()
                for off in range(0, n, _CHUNK):
                    rows.append((p.data_ptr() + 4 * off, g.data_ptr() + 4 * off, m.data_ptr() + 4 * off,
                                 v.data_ptr() + 4 * off, min(_CHUNK, n - off)))
            host = torch.from_numpy(np.asarray(rows, dtype=np.int64))
            capturing = torch.cuda.is_current_stream_capturing()
            if self._table is None or self._table.shape != host.shape:
                if capturing:
                    raise RuntimeError('FusedClipAdamW: run one eager step before graph capture')
                self._table = torch.empty(host.shape, dtype=torch.int64, device=dev)
                self._partial = torch.empty((host.shape[0],), dtype=torch.float32, device=dev)
                self._pinned = torch.empty(host.shape, dtype=torch.int64).pin_memory()
            if capturing:
                self._pinned.copy_(host)                       # read at replay time; pointers are replay-invariant
                self._table.copy_(self._pinned, non_blocking=True)
            else:
                self._table.copy_(host.to(dev))
            self._key = key
        if self._state is None:
            self._state = torch.zeros((8,), dtype=torch.float32, device=dev)
            self._lr = torch.empty((1,), dtype=torch.float32, device=dev)
        lr = grp['lr']
        if torch.is_tensor(lr):
            self._lr.copy_(lr.reshape(1))
        elif getattr(self, '_lr_host', None) != lr:
            self._lr.fill_(float(lr))
            self._lr_host = lr
        call('idf_clip_adamw', ops._p(self._table), self._table.shape[0], ops._p(self._partial), ops._p(self._state),
             ops._p(self._lr), float(grp['max_norm']), float(grp['betas'][0]), float(grp['betas'][1]),
             float(grp['eps']), float(grp['weight_decay']), 1, ops._st())
        # the kernel wrote the parameters behind autograd's back: bump their version counters so every
        # consumer that caches derived data (conv weight shadows) sees the update
        torch.autograd.graph.increment_version([p for p, _, _, _ in items])
        return None

    def refresh_lr(self):
        """Push the param group's learning rate to the device scalar the kernels read.  step() does it when
        it runs in Python; call this after an LR-scheduler step when step() is replayed from a hipGraph."""
        if self._lr is not None:
            lr = self.param_groups[0]['lr']
            if torch.is_tensor(lr):
                self._lr.copy_(lr.reshape(1))
            else:
                self._lr.fill_(float(lr))
                self._lr_host = lr

    def total_norm(self):
        """Gradient norm before clipping of the last step (device scalar)."""
        return self._state[4]
