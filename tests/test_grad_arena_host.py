"""Host logic of the gradient arena / parameter groups (no GPU needed: they are plain tensor bookkeeping)."""
import torch

from infodiffusion_amd.grad_arena import GradArena, ParamGroup, slot_of


def _params():
    torch.manual_seed(0)
    q, k, v = (torch.nn.Parameter(torch.randn(4, 6)) for _ in range(3))
    other = torch.nn.Parameter(torch.randn(5))
    conv = torch.nn.Parameter(torch.randn(8, 4, 3, 3).contiguous(memory_format=torch.channels_last))
    return q, k, v, other, conv


def test_slots_mirror_parameter_layout_and_are_handed_out_once_per_zero():
    q, k, v, other, conv = _params()
    arena = GradArena([q, other, conv])
    s = slot_of(conv)
    assert s.view.shape == conv.shape and s.view.stride() == conv.stride()      # channels-last slot
    a = s.take()
    assert a is not None and arena.holds(a) and float(a.abs().sum()) == 0.0
    assert s.take() is None and not s.available()        # second request before zero(): caller's own path
    a.fill_(3.0)
    arena.zero()
    b = s.take()
    assert b is not None and float(b.abs().sum()) == 0.0 and b.data_ptr() == a.data_ptr()
    assert a is not b                                     # a fresh alias every time (AccumulateGrad adopts it)
    assert slot_of(k) is None and not arena.covers(k)


def test_param_group_concatenation_is_a_view_and_gradients_land_in_adjacent_slots():
    from infodiffusion_amd import ops
    q, k, v, other, conv = _params()
    vals = [p.detach().clone() for p in (q, k, v)]
    grp = ParamGroup([q, k, v])
    arena = GradArena([q, other, k, conv, v])             # interleaved registration order
    w = ops.cat_params(grp)
    assert torch.equal(w.detach(), torch.cat(vals))       # values kept, now one buffer
    assert q.data_ptr() + 4 * q.numel() == k.data_ptr() and k.data_ptr() + 4 * k.numel() == v.data_ptr()
    assert w.data_ptr() == q.data_ptr()
    gs = slot_of(w)
    assert gs is not None and gs.adjacent and gs.available()
    # consumer writes the gradient of the concatenation straight into the arena region
    region = gs.take()
    want = torch.arange(region.numel(), dtype=torch.float32).view_as(region)
    region.copy_(want)
    w.backward(region)
    for p, rows in zip((q, k, v), want.split(4)):
        assert arena.holds(p.grad) and torch.equal(p.grad, rows)
    assert not gs.available()
    # no free region (no zero() since): plain row slices of whatever gradient arrives
    for p in (q, k, v):
        p.grad = None
    w2 = ops.cat_params(grp)
    (w2 * 2).sum().backward()
    assert all(torch.equal(p.grad, torch.full_like(p, 2.0)) for p in (q, k, v))
    # module.to()/.float() style re-allocation breaks adjacency; ensure() restores it without changing values
    k.data = k.data.clone()
    w3 = ops.cat_params(grp)
    assert k.data_ptr() == q.data_ptr() + 4 * q.numel() and torch.equal(w3.detach(), torch.cat(vals))
