"""world_size-2 gloo test (CPU) of the data-parallel gradient exchange."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from infodiffusion_amd.dist import GradSync, shard_range
    torch.manual_seed(rank)          # different init per rank on purpose
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.SiLU(), torch.nn.Linear(16, 4))
    # one parameter with permuted (channels-last-like) strides, as the conv weight gradients have
    net[0].weight.data = net[0].weight.data.t().contiguous().t()
    dead = torch.nn.Linear(3, 3)     # never used: grads stay None and must be skipped
    model = torch.nn.ModuleDict({'net': net, 'dead': dead})
    sync = GradSync(model, world, bucket_bytes=256)   # tiny buckets -> several collectives
    sync.broadcast_parameters()
    torch.manual_seed(100 + rank)
    x = torch.randn(5, 8)
    net(x).square().mean().backward()
    net[0].weight.grad = net[0].weight.grad.t().contiguous().t()      # strided like its parameter
    local = [p.grad.clone() for p in net.parameters()]
    sync.all_reduce_grads()
    # plain lists: tensors in an mp.Queue travel by fd and die with the worker
    avg1 = [p.grad.tolist() for p in net.parameters()]
    # second exchange: some gradients live in a gradient arena (reduced in place, no packing), one does not
    from infodiffusion_amd.grad_arena import GradArena, slot_of
    arena = GradArena(list(net.parameters()))
    sync2 = GradSync(model, world, bucket_bytes=256, arena=arena)
    plist = list(net.parameters())
    for p, g in zip(plist, local):
        if p is plist[1]:
            p.grad = g.clone()                   # stand-alone gradient
        else:
            v = slot_of(p).take()
            v.copy_(g)
            p.grad = v
    sync2.all_reduce_grads()
    # third exchange: the arena cut in two slices (early module = net[0]); the early slice goes first, as the
    # hook in InfoDiff.forward triggers it, the rest at the end
    for p, g in zip(plist, local):
        p.grad = None
    arena.zero()
    sync3 = GradSync(model, world, bucket_bytes=256, arena=arena)
    cut_ok = sync3.attach(net[0])
    for p, g in zip(plist, local):
        if p is plist[1]:
            p.grad = g.clone()
        else:
            v = slot_of(p).take()
            v.copy_(g)
            p.grad = v
    sync3.reduce_early()
    early_first = [p.grad.tolist() for p in plist]      # after the early slice only
    sync3.all_reduce_grads()
    res = {'cut_ok': cut_ok, 'early_first': early_first, 'avg_sliced': [p.grad.tolist() for p in plist],
           'params': [p.detach().tolist() for p in net.parameters()],
           'local': [t.tolist() for t in local], 'avg': avg1,
           'avg_arena': [p.grad.tolist() for p in plist],
           'in_arena': [arena.holds(p.grad) for p in plist],
           'dead_none': all(p.grad is None for p in dead.parameters()),
           'shard': shard_range(10, rank, world)}
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_allreduce_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    a, b = got[0], got[1]
    T = torch.tensor
    for pa, pb in zip(a['params'], b['params']):
        assert torch.equal(T(pa), T(pb))                # broadcast made replicas identical
    for la, lb, ga, gb in zip(a['local'], b['local'], a['avg'], b['avg']):
        la, lb, ga, gb = T(la), T(lb), T(ga), T(gb)
        want = (la + lb) / 2
        assert torch.allclose(ga, want, atol=1e-7) and torch.equal(ga, gb)
    for la, lb, ga, gb in zip(a['local'], b['local'], a['avg_arena'], b['avg_arena']):
        la, lb, ga, gb = T(la), T(lb), T(ga), T(gb)
        assert torch.allclose(ga, (la + lb) / 2, atol=1e-7) and torch.equal(ga, gb)
    assert a['in_arena'] == [True, False, True, True]
    # sliced exchange: same averages; after the early slice alone only net[0].weight (its bias is the stand-alone
    # gradient in this set-up) has been averaged
    assert a['cut_ok'] and b['cut_ok']
    for la, lb, ga, gb in zip(a['local'], b['local'], a['avg_sliced'], b['avg_sliced']):
        la, lb, ga, gb = T(la), T(lb), T(ga), T(gb)
        assert torch.allclose(ga, (la + lb) / 2, atol=1e-7) and torch.equal(ga, gb)
    assert torch.allclose(T(a['early_first'][0]), (T(a['local'][0]) + T(b['local'][0])) / 2, atol=1e-7)
    assert torch.equal(T(a['early_first'][2]), T(a['local'][2]))          # late slice untouched so far
    assert a['dead_none'] and b['dead_none']
    assert a['shard'] == (0, 5) and b['shard'] == (5, 10)


def _trainer_worker(rank, world, port, q):
    """GraphedTrainStep's decisions under a capture that fails on ONE rank only (stubbed capture: there is no GPU
    here): every rank must reach the agreement collective, then fall back together to eager steps.  The model has a
    latent cut (an `encoder` whose output enters the `backbone` and the loss as a leaf), so every step -- eager or the
    capture's warm-up pass -- issues the three-phase sequence: backbone backward, early slice, encoder backward, rest."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from infodiffusion_amd.dist import GradSync
    from infodiffusion_amd.grad_arena import GradArena, slot_of
    from infodiffusion_amd.trainer import GraphedTrainStep
    torch.manual_seed(0)

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.backbone = torch.nn.Linear(4, 2)
            self.encoder = torch.nn.Linear(4, 4)
            self.cut_latent = False
            self._latent_cut = None
            self._cut_armed = False
            self.calls = []

        def attach_grad_sync(self, sync):
            self.cut_latent = bool(sync.attach(self.backbone))
            return self.cut_latent

        def arm_latent_cut(self):           # the product's protocol (models.InfoDiff): the cut is per call
            assert self._latent_cut is None
            self._cut_armed = self.cut_latent
            return self._cut_armed

        def pop_latent_cut(self):
            cut, self._latent_cut = self._latent_cut, None
            self._cut_armed = False
            return cut

        def loss_fn(self, args, x, curr_epoch=0):
            lat = self.encoder(x)
            if self._cut_armed:
                self._cut_armed = False
                leaf = lat.detach().requires_grad_(True)
                self._latent_cut = (lat, leaf)
                lat = leaf
            return self.backbone(lat).square().mean() + 0.1 * lat.square().mean()

    model = Model()
    ref = Model()
    ref.load_state_dict(model.state_dict())

    class ArenaSGD(torch.optim.SGD):
        """SGD whose gradients live in a GradArena (what FusedClipAdamW does on the GPU): the slice exchange needs one."""
        def __init__(self, params, lr):
            params = list(params)
            super().__init__(params, lr=lr)
            self.arena = GradArena(params)

        def zero_grad(self, set_to_none=True):
            self.arena.zero()
            super().zero_grad(set_to_none=True)

        def step(self):
            for p in self.param_groups[0]['params']:      # re-home the gradients in their arena slots
                if p.grad is not None and not self.arena.holds(p.grad):
                    v = slot_of(p).take()
                    v.copy_(p.grad)
                    p.grad = v
            super().step()

    opt = ArenaSGD(list(model.backbone.parameters()) + list(model.encoder.parameters()), lr=0.1)
    sync = GradSync(model, world, arena=opt.arena)
    seq = []
    orig_early, orig_all = sync.reduce_early, sync.all_reduce_grads

    def rehome():
        # before an exchange the gradients must sit in the arena for the slices to carry them (CPU autograd hands
        # AccumulateGrad fresh tensors; on the GPU the kernels write the slots themselves)
        for p in opt.param_groups[0]['params']:
            if p.grad is not None and not opt.arena.holds(p.grad):
                v = slot_of(p).take()
                v.copy_(p.grad)
                p.grad = v

    def re(*a, **k):
        rehome()
        seq.append('early')
        return orig_early(*a, **k)

    def ar(*a, **k):
        rehome()
        seq.append('rest')
        return orig_all(*a, **k)
    sync.reduce_early, sync.all_reduce_grads = re, ar
    step = GraphedTrainStep(model, None, opt, sync=sync, use_graph=True, warmup=1)
    split = step.split
    log = []
    fails = {2: 0}             # call number -> the rank whose capture fails in that call

    def fake_try_capture(x, epoch):
        call = step.seen
        step.xbuf = x.clone()
        step.loss = step._eager_step(x, epoch)      # the warm-up pass: a full step, collectives included
        ok = fails.get(call) != rank
        if ok:
            step.graph = 'captured'                  # never replayed below: the other rank's failure drops it
        log.append(('capture', call, ok))
        return ok, True
    step._try_capture = fake_try_capture
    torch.manual_seed(10 + rank)
    xs = [torch.randn(3, 4) for _ in range(4)]
    states = []
    for x in xs:
        step(x, 0)
        states.append((step.graph is not None, step.use_graph))
    # reference: the same four steps with one backward pass per step and the mean of both ranks' gradients
    q.put((rank, {'log': log, 'states': states, 'split': split, 'seq': seq,
                  'w': [p.detach().tolist() for p in model.parameters()],
                  'x': [x.tolist() for x in xs], 'w0': [p.detach().tolist() for p in ref.parameters()]}))
    dist.barrier()
    dist.destroy_process_group()


def test_capture_failure_on_one_rank_takes_every_rank_down_the_same_path():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_trainer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0                         # nobody hung in a mismatched collective
    a, b = got[0], got[1]
    assert a['split'] and b['split']                   # the arena is cut backbone | encoder and the backward pass with it
    assert a['states'] == b['states']
    # call 1 eager; call 2: the capture fails on rank 0 -> both ranks drop it and train eagerly from then on
    assert a['states'] == [(False, True), (False, False), (False, False), (False, False)]
    assert a['log'] == [('capture', 2, False)] and b['log'] == [('capture', 2, True)]
    # every step issued the same collective sequence on both ranks: early slice, then the rest
    assert a['seq'] == b['seq'] == ['early', 'rest'] * 4
    T = torch.tensor
    for wa, wb in zip(a['w'], b['w']):
        assert torch.equal(T(wa), T(wb))               # replicas still identical after 4 steps
    # and the split backward + sliced exchange trained exactly like one backward pass on the averaged gradients
    bb, enc = torch.nn.Linear(4, 2), torch.nn.Linear(4, 4)
    with torch.no_grad():
        for p, w in zip(list(bb.parameters()) + list(enc.parameters()), a['w0']):
            p.copy_(T(w))
    params = list(bb.parameters()) + list(enc.parameters())
    for xa, xb in zip(a['x'], b['x']):
        grads = []
        for x in (T(xa), T(xb)):
            lat = enc(x)
            loss = bb(lat).square().mean() + 0.1 * lat.square().mean()
            grads.append(torch.autograd.grad(loss, params))
        with torch.no_grad():
            for p, g0, g1 in zip(params, *grads):
                p -= 0.1 * (g0 + g1) / 2
    for p, w in zip(params, a['w']):
        assert torch.allclose(p.detach(), T(w), atol=1e-6)


def _health_worker(rank, world, port, q):
    """GraphedTrainStep.check under data parallelism: a synchronised-conv time-out seen by ONE rank must roll EVERY rank back (the
    count is MAX-reduced) -- a rank that rolled back alone would train on from different weights than its peers.  No GPU here: the
    error word is stubbed (ops.rs_sync_timeouts / ops._RS_SYNC_STATE), everything else is the product's code."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from infodiffusion_amd import ops
    from infodiffusion_amd.dist import GradSync
    from infodiffusion_amd.trainer import GraphedTrainStep
    torch.manual_seed(0)

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.net = torch.nn.Linear(4, 2)

        def loss_fn(self, args, x, curr_epoch=0):
            return self.net(x).square().mean()

    model = Model()
    opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9)
    sync = GradSync(model, world)
    fake = {'n': 0}
    ops._RS_SYNC_STATE[0] = torch.zeros(32, dtype=torch.int32)          # "the form has been launched on device 0"
    orig = ops.rs_sync_timeouts
    ops.rs_sync_timeouts = lambda reset=True: fake['n']
    step = GraphedTrainStep(model, None, opt, sync=sync, use_graph=False, warmup=1, health_every=2)
    torch.manual_seed(20 + rank)
    xs = [torch.randn(3, 4) for _ in range(5)]
    trail = []
    try:
        for k, x in enumerate(xs):
            if k == 3 and rank == 1:
                fake['n'] = 3                  # rank 1's launches of steps 3 / 4 timed out; rank 0 saw nothing
            step(x, 0)
            fake['n'] = 0 if ops.sync_convs_retired() else fake['n']      # (retire_sync_convs zeroes the word)
            trail.append([p.detach().clone() for p in model.parameters()])
    finally:
        ops.rs_sync_timeouts = orig
        ops._RS_SYNC_STATE.clear()
        retired = ops.sync_convs_retired()
        ops._RS_SYNC_DEAD[0] = False
    q.put((rank, {'w': [[p.tolist() for p in t] for t in trail], 'timeouts': step.timeouts, 'recoveries': step.recoveries,
                  'retired': retired, 'x': [x.tolist() for x in xs]}))
    dist.barrier()
    dist.destroy_process_group()


def test_a_timeout_on_one_rank_rolls_every_rank_back():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_health_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    a, b = got[0], got[1]
    assert a['recoveries'] == b['recoveries'] == 1 and a['retired'] and b['retired']
    assert a['timeouts'] == b['timeouts'] == 3          # the MAX over the ranks, on both
    T = torch.tensor
    for ta, tb in zip(a['w'], b['w']):                  # replicas identical after every step, the rolled-back one included
        for wa, wb in zip(ta, tb):
            assert torch.equal(T(wa), T(wb))
    # the roll-back went to the last clean check (in front of step 4's predecessor check: call 2 -> the weights after step 1):
    # steps 2 and 3 are gone, step 4 and 5 trained on from there -- replay that on one process with the averaged gradients
    net = torch.nn.Linear(4, 2)
    torch.manual_seed(0)
    ref = torch.nn.Linear(4, 2)
    net.load_state_dict(ref.state_dict())
    opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9)

    def one(k):
        opt.zero_grad()
        (0.5 * (net(T(a['x'][k])).square().mean() + net(T(b['x'][k])).square().mean())).backward()
        opt.step()
    one(0)
    snap = ([p.detach().clone() for p in net.parameters()], {k: {kk: (vv.clone() if torch.is_tensor(vv) else vv) for kk, vv in v.items()}
                                                              for k, v in opt.state_dict()['state'].items()})
    one(1)
    one(2)
    with torch.no_grad():                               # the trainer's roll-back: parameters AND optimizer state (the momentum buffers)
        for p, s in zip(net.parameters(), snap[0]):
            p.copy_(s)
    sd = opt.state_dict()
    sd['state'] = snap[1]
    opt.load_state_dict(sd)
    one(3)
    for p, w in zip(net.parameters(), a['w'][3]):
        assert torch.allclose(p.detach(), T(w), atol=1e-6), (p, w)
    one(4)
    for p, w in zip(net.parameters(), a['w'][4]):
        assert torch.allclose(p.detach(), T(w), atol=1e-6)
