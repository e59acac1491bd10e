"""Data-parallel gradient exchange: one process per GPU, RCCL over xGMI (torch
backend "nccl"); gloo on CPU for tests.  The reference has no distributed code
(SURVEY.md 2 rows 12-13); this is the only exchange step the path needs (8e):
one sum-all-reduce of the gradients per step, averaged over ranks, then the
*global* grad-norm clip and an identical AdamW step on every rank.

Gradients are packed into a few large persistent flat buckets (xGMI is
point-to-point: fewer, larger collectives) in reverse registration order -- the
order backward produces them.  Packing / unpacking a bucket is ONE multi-tensor
copy each (bucket views carry the gradients' own strides, so channels-last conv
gradients move as dense memory), and each bucket's all-reduce is issued on a side
stream as soon as it is packed, so packing bucket k+1 overlaps the wire time of
bucket k.  Gradients that already live in the optimizer's gradient arena (grad_arena.py: every
conv / GroupNorm gradient the kernels accumulated in place) need no packing at all: the arena's flat
buffer is all-reduced in place as one collective.  Parameters that never receive a gradient (the dead `crossattn.*`
weights, `encoder.fc_mu/fc_var` with kld_weight = 0, the frozen time table) are
skipped.

Overlap with backward (`attach`): the backbone's backward ends before the encoder's begins (the encoder ran first
in the forward pass), so the arena is cut at the backbone / encoder boundary.  The trainer (trainer.GraphedTrainStep)
cuts the backward pass at the latent `a` too: after the backbone's half it calls `reduce_early()` -- the backbone slice
goes onto the exchange stream -- then runs the encoder's half beside it and calls `all_reduce_grads()` for the encoder
slice and the few stand-alone gradients, which also joins the exchange stream.  The collectives are always issued
eagerly (never inside a stream capture: RCCL inside an open capture aborts intermittently on this stack); the three
compute phases around them replay from three hipGraphs.
"""
import torch
import torch.distributed as dist


def _dense(t):
    """Non-overlapping and dense: some permutation of the dims is contiguous."""
    expect = 1
    for size, stride in sorted(((sz, st) for sz, st in zip(t.shape, t.stride()) if sz != 1), key=lambda x: x[1]):
        if stride != expect:
            return False
        expect *= size
    return True


class GradSync:
    def __init__(self, model, world_size=None, bucket_bytes=32 << 20, force=False, arena=None):
        self.arena = arena      # optimizer's GradArena: its resident gradients are reduced in place
        self.force = force      # run the exchange even at world size 1 (single-GPU validation of the DP path)
        self.world = world_size if world_size is not None else dist.get_world_size()
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.model = model
        self.bucket_bytes = bucket_bytes
        self._side = None
        self._plan_key = None
        self._plan = None       # [(flat buffer, [views shaped/strided like the grads], [grad indices])]
        self._cut = None        # arena offset (floats) where the early slice ends; None: no early slice
        self._early_done = False
        # measurement only (bench.py at N > 1 / IDF_FORCE_SYNC): a dict {'early': [], 'late': []} that receives one HIP event pair
        # per collective, recorded on the exchange stream around it
        self.trace = None

    def _timed(self, which, flat):
        if self.trace is None or not flat.is_cuda:
            self._reduce_mean(flat)
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        self._reduce_mean(flat)
        e1.record()
        self.trace[which].append((e0, e1, flat.numel() * flat.element_size()))

    # ---------------------------------------------------------------- overlap with backward
    def attach(self, early_module):
        """Cut the arena after `early_module`'s parameters (the sub-network whose backward pass ends first) and let
        `early_module`'s owner trigger `reduce_early()` when that backward has ended.  Returns True when the arena
        layout allows it (the early parameters occupy a prefix of the arena)."""
        from .grad_arena import slot_of
        if self.arena is None or self.arena.flat is None:
            return False
        early = {id(p) for p in early_module.parameters()}
        hi_early, lo_rest = 0, self.arena.flat.numel()
        for p in self.params:
            sl = slot_of(p)
            if sl is None or sl.arena is not self.arena:
                continue
            off = sl.view.storage_offset()
            if id(p) in early:
                hi_early = max(hi_early, off + p.numel())
            else:
                lo_rest = min(lo_rest, off)
        if hi_early == 0 or hi_early > lo_rest:
            return False
        self._cut = lo_rest
        return True

    def begin_step(self):
        """Start of a forward + backward pass: nothing of the previous step's exchange is pending (an exception between
        `reduce_early` and `all_reduce_grads` must not make the next step skip the early slice)."""
        self._early_done = False

    @torch.no_grad()
    def reduce_early(self, after=None):
        """All-reduce the early slice of the arena on the side stream (everything enqueued so far on the current
        stream has produced it); the rest of backward keeps running on the current stream.  `all_reduce_grads` joins
        the side stream again.  Never call it inside a stream capture."""
        if (self._cut is None or self._early_done
                or (self.world == 1 and not self.force)):
            return
        flat = self.arena.flat
        if not flat.is_cuda:
            self._reduce_mean(flat[:self._cut])
        else:
            cur = torch.cuda.current_stream()
            if self._side is None:
                self._side = torch.cuda.Stream()
            self._side.wait_stream(cur)
            if after is not None:                # a stream that is still producing part of the slice (deferred wgrad)
                self._side.wait_stream(after)
            with torch.cuda.stream(self._side):
                self._timed('early', flat[:self._cut])
        self._early_done = True

    @torch.no_grad()
    def broadcast_parameters(self, src=0):
        """Make every rank start from rank `src`'s parameters and buffers."""
        tensors = [p.data for p in self.model.parameters()] + [b.data for b in self.model.buffers()]
        for group in self._buckets(tensors):
            flat = torch.cat([t.reshape(-1) for t in group])
            dist.broadcast(flat, src)
            off = 0
            for t in group:
                n = t.numel()
                t.copy_(flat[off:off + n].view_as(t))
                off += n

    def _buckets(self, tensors):
        out, cur, size = [], [], 0
        for t in tensors:
            cur.append(t)
            size += t.numel() * t.element_size()
            if size >= self.bucket_bytes:
                out.append(cur)
                cur, size = [], 0
        if cur:
            out.append(cur)
        return out

    def _make_plan(self, grads):
        key = tuple((tuple(g.shape), tuple(g.stride()), g.dtype) for g in grads)
        if key == self._plan_key:
            return self._plan
        plan = []
        for group in self._buckets(grads):
            n = sum(g.numel() for g in group)
            flat = torch.empty((n,), dtype=group[0].dtype, device=group[0].device)
            views, off = [], 0
            for g in group:
                seg = flat[off:off + g.numel()]
                views.append(seg.as_strided(g.shape, g.stride()) if _dense(g) else seg.view(g.shape))
                off += g.numel()
            plan.append((flat, views, len(group)))
        self._plan_key, self._plan = key, plan
        return plan

    def _reduce_mean(self, flat):
        if flat.is_cuda:
            dist.all_reduce(flat, op=dist.ReduceOp.AVG)      # RCCL averages in the collective: no divide pass
        else:
            dist.all_reduce(flat)
            flat.div_(self.world)

    @torch.no_grad()
    def all_reduce_grads(self):
        """Average the gradients over all ranks (in place)."""
        grads = [p.grad for p in reversed(self.params) if p.grad is not None]
        if not grads or (self.world == 1 and not self.force):
            return
        use_side = grads[0].is_cuda
        cur = torch.cuda.current_stream() if use_side else None
        if use_side and self._side is None:
            self._side = torch.cuda.Stream()
        if self.arena is not None and any(self.arena.holds(g) for g in grads):
            # zero-copy part: the arena itself, piecewise (slots nobody wrote this step hold zeros)
            grads = [g for g in grads if not self.arena.holds(g)]
            # ONE collective over the whole arena: nothing is left to overlap with once backward has ended, and a
            # ring over point-to-point xGMI links is per-link bound -- fewer, larger messages
            flat = self.arena.flat
            if self._early_done:                 # the early slice is already on the wire (reduce_early)
                flat = flat[self._cut:]
                self._early_done = False
            if use_side:
                self._side.wait_stream(cur)
                with torch.cuda.stream(self._side):
                    self._timed('late', flat)
            else:
                self._reduce_mean(flat)
        plan = self._make_plan(grads) if grads else []
        i = 0
        for flat, views, n in plan:
            group = grads[i:i + n]
            i += n
            torch._foreach_copy_(views, group)             # pack: one multi-tensor launch
            if use_side:
                self._side.wait_stream(cur)
                with torch.cuda.stream(self._side):
                    self._reduce_mean(flat)
            else:
                self._reduce_mean(flat)
        if use_side:
            cur.wait_stream(self._side)
        i = 0
        for flat, views, n in plan:
            torch._foreach_copy_(grads[i:i + n], views)    # unpack
            i += n


def shard_range(total, rank, world):
    """Contiguous [lo, hi) share of `total` independent units (sampling shards the
    image batch with no collective)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
