"""Data-parallel gradient exchange: one process per GPU, RCCL over xGMI (torch
backend "nccl"); gloo on CPU for tests.  The reference has no distributed code
(SURVEY.md 2 rows 12-13); this is the only exchange step the path needs (8e):
one sum-all-reduce of the gradients per step, averaged over ranks, then the
*global* grad-norm clip and an identical AdamW step on every rank.

Gradients are packed into a few large flat buckets (xGMI is point-to-point:
fewer, larger collectives) in reverse registration order -- the order backward
produces them -- and each bucket's all-reduce is issued on a side stream as soon
as it is packed, so packing bucket k+1 overlaps the wire time of bucket k.
Parameters that never receive a gradient (the dead `crossattn.*` weights,
`encoder.fc_mu/fc_var` with kld_weight = 0, the frozen time table) are skipped.
"""
import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, model, world_size=None, bucket_bytes=32 << 20):
        self.world = world_size if world_size is not None else dist.get_world_size()
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.model = model
        self.bucket_bytes = bucket_bytes
        self._side = None

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

    @torch.no_grad()
    def all_reduce_grads(self):
        """Average the gradients over all ranks (in place)."""
        grads = [p.grad for p in reversed(self.params) if p.grad is not None]
        if not grads or self.world == 1:
            return
        use_side = grads[0].is_cuda
        cur = torch.cuda.current_stream() if use_side else None
        if use_side and self._side is None:
            self._side = torch.cuda.Stream()
        work = []
        for group in self._buckets(grads):
            flat = torch.cat([g.reshape(-1) for g in group])
            if use_side:
                self._side.wait_stream(cur)
                with torch.cuda.stream(self._side):
                    dist.all_reduce(flat)
                    flat.div_(self.world)
                flat.record_stream(self._side)
            else:
                dist.all_reduce(flat)
                flat.div_(self.world)
            work.append((flat, group))
        if use_side:
            cur.wait_stream(self._side)
        for flat, group in work:
            off = 0
            for g in group:
                n = g.numel()
                g.copy_(flat[off:off + n].view_as(g))
                off += n


def shard_range(total, rank, world):
    """Contiguous [lo, hi) share of `total` independent units (sampling shards the
    image batch with no collective)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
