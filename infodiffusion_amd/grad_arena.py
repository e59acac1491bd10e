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


class ParamGroup:
    """Parameters that are always used concatenated along dim 0 (the q/k/v 1x1 convs of an attention
    block, every block's FiLM projection): their storage is kept adjacent in one flat buffer, so the
    concatenation is a VIEW (no per-step torch.cat), and the gradient arena lays their slots out
    adjacently too, so the gradient of the concatenation is written once, in place, with no split copies."""

    def __init__(self, params):
        self.params = list(params)
        self.flat = None
        self.rows = [p.shape[0] for p in self.params]
        self.rest = tuple(self.params[0].shape[1:])
        assert all(tuple(p.shape[1:]) == self.rest for p in self.params)
        for i, p in enumerate(self.params):
            p._idf_group = (self, i)

    def ensure(self):
        """(Re-)establish adjacency: module.to() / .float() give every parameter its own allocation."""
        ps, off, ok = self.params, 0, self.flat is not None
        for p in ps:
            ok = ok and p.device == self.flat.device and p.data_ptr() == self.flat.data_ptr() + 4 * off \
                and p.is_contiguous()
            off += p.numel()
        if ok:
            return
        flat = torch.empty((off,), dtype=torch.float32, device=ps[0].device)
        off = 0
        with torch.no_grad():
            for p in ps:
                v = flat[off:off + p.numel()].view(p.shape)
                v.copy_(p.data)
                p.data = v
                off += p.numel()
        self.flat = flat

    def cat(self):
        return self.flat.view((sum(self.rows),) + self.rest)


class _GroupSlot:
    """The adjacent arena slots of a ParamGroup seen as one gradient of the concatenated shape."""

    def __init__(self, group, slots):
        self.group, self.slots = group, slots
        self.arena = slots[0].arena
        s0 = slots[0].view
        self._shape = (sum(group.rows),) + group.rest
        self._args = (self._shape, s0.stride(), s0.storage_offset())
        off, self.adjacent = s0.storage_offset(), True
        for s in slots:
            self.adjacent = self.adjacent and s.view.storage_offset() == off and s.view.stride() == s0.stride()
            off += s.view.numel()

    @property
    def view(self):
        return torch.as_strided(self.arena.flat, *self._args)

    def available(self):
        return self.adjacent and all(s.available() for s in self.slots)

    def take(self):
        if not self.available():
            return None
        for s in self.slots:
            s.used = self.arena.epoch
        return self.view

    def split(self, dcat):
        """Per-parameter gradients when `dcat` IS this region (else None)."""
        if dcat is None or not self.adjacent or dcat.data_ptr() != self.slots[0].view.data_ptr() \
                or tuple(dcat.shape) != self._shape or not self.arena.holds(dcat):
            return None
        return [s.view.detach() for s in self.slots]


def _grouped_order(params):
    """Registration order, except that the members of a ParamGroup follow each other in group order."""
    have, out, placed = {id(p) for p in params}, [], set()
    for p in params:
        if id(p) in placed:
            continue
        grp = getattr(p, '_idf_group', None)
        members = [q for q in grp[0].params if id(q) in have] if grp is not None else [p]
        for q in members:
            if id(q) not in placed:
                placed.add(id(q))
                out.append(q)
    return out


class GradArena:
    def __init__(self, params):
        params = _grouped_order([p for p in params if p.requires_grad and p.dtype == torch.float32 and _dense(p)])
        self.epoch = 0
        self.flat = None
        if not params:
            return
        offs, total, prev = [], 0, None
        for p in params:
            grp = getattr(p, '_idf_group', (None,))[0]
            if grp is None or grp is not prev:          # group members follow each other without padding
                total = -(-total // _ALIGN) * _ALIGN
            offs.append(total)
            total += p.numel()
            prev = grp
        total = -(-total // _ALIGN) * _ALIGN
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
    """The parameter's slot (or None): looked up at forward time, taken in backward.  For the
    concatenated view of a ParamGroup: the group's adjacent slots as one."""
    if p is None:
        return None
    grp = getattr(p, '_idf_cat_group', None)
    if grp is not None:
        slots = [slot_of(q) for q in grp.params]
        if any(s is None for s in slots) or len({id(s.arena) for s in slots}) != 1:
            return None
        return _GroupSlot(grp, slots)
    s = _SLOTS.get(id(p))
    if s is None:
        return None
    if s.ref() is not p:          # stale id
        del _SLOTS[id(p)]
        return None
    return s
