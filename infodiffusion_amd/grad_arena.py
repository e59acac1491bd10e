"""Gradient arena: one flat fp32 buffer with a fixed slot per parameter.

The weight-gradient kernels accumulate with fp32 atomics, so their output has to be zero before they
run; per call that is a memset launch per convolution, and the GroupNorm gamma/beta gradients need a
batch column-sum launch each (~370 tiny launches per CelebA step).  With the arena the optimizer's
`zero_grad()` clears every slot with ONE memset, the kernels accumulate straight into the slots and
the autograd functions hand the slot views back as the gradients (so `p.grad` lives at a fixed
address: the fused optimizer's chunk table and a captured graph never change).

A slot may be handed out once per zeroing: a second request before the next `zero()` (shared weights,
gradient accumulation without `zero_grad`) returns None and the caller takes its stand-alone path,
which is always correct.
"""
import weakref

import torch

_ALIGN = 64          # floats (256 B)
_SLOTS = {}          # id(param) -> _Slot


class _Slot:
    __slots__ = ('arena', 'ref', 'view', 'used')

    def __init__(self, arena, p, view):
        self.arena, self.ref, self.view, self.used = arena, weakref.ref(p), view, -1

    def available(self):
        return self.used != self.arena.epoch

    def take(self):
        """The (zeroed) gradient view, or None when it was already handed out since the last zero()."""
        if self.used == self.arena.epoch:
            return None
        self.used = self.arena.epoch
        # a fresh alias: AccumulateGrad adopts a gradient only when nobody else holds the tensor object
        return self.view.detach()


def _dense(p):
    n, expect = p.numel(), 1
    for size, stride in sorted(zip(p.shape, p.stride()), key=lambda t: t[1]):
        if size == 1:
            continue
        if stride != expect:
            return False
        expect *= size
    return expect == n


class GradArena:
    def __init__(self, params):
        params = [p for p in params if p.requires_grad and p.dtype == torch.float32 and _dense(p)]
        self.epoch = 0
        self.flat = None
        if not params:
            return
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += -(-p.numel() // _ALIGN) * _ALIGN
        self.flat = torch.zeros((total,), dtype=torch.float32, device=params[0].device)
        for p, off in zip(params, offs):
            _SLOTS[id(p)] = _Slot(self, p, torch.as_strided(self.flat, p.shape, p.stride(), off))

    def zero(self):
        """Clear every slot (one memset) and make them available again."""
        if self.flat is not None:
            self.flat.zero_()
        self.epoch += 1

    def holds(self, t):
        """`t`'s memory lies inside the arena (a gradient the kernels accumulated in place)."""
        if self.flat is None or t.device != self.flat.device:
            return False
        lo = self.flat.data_ptr()
        return lo <= t.data_ptr() < lo + 4 * self.flat.numel()

    def covers(self, p):
        s = _SLOTS.get(id(p))
        return s is not None and s.arena is self and s.ref() is p


def slot_of(p):
    """The parameter's slot (or None): looked up at forward time, taken in backward."""
    if p is None:
        return None
    s = _SLOTS.get(id(p))
    if s is None:
        return None
    if s.ref() is not p:          # stale id
        del _SLOTS[id(p)]
        return None
    return s
