"""The training step of reference run.py:195-200 (loss_fn -> zero_grad -> backward -> [gradient exchange]
-> clip + AdamW) as captured hipGraphs.

Eagerly a CelebA step issues ~500 kernel launches from Python and is host-bound (~30 ms); replayed from a
graph it runs at the GPU's pace (~10 ms).  The step is captured after two eager warm-up steps (they build
the allocator pools, the weight shadows and the optimizer state).

One GPU (no `sync`): the whole step is ONE graph.

Data parallel (`sync`): the step is THREE graphs cut where the design already has its seams, and the RCCL
collectives run eagerly between them -- no collective is ever issued inside an open capture (that aborts
intermittently on this stack), yet the backbone's share of the exchange overlaps the encoder's backward pass:

    G1  forward + backward of the loss down to the latent (the backbone's backward pass and its deferred
        weight gradients; `InfoDiff.cut_latent`: the latent enters the backbone and the loss as a leaf)
        -> eager: all-reduce of the backbone slice of the gradient arena on the exchange stream
    G2  backward of the encoder from the latent's gradient (+ its deferred weight gradients)
        -> eager: all-reduce of the rest (encoder slice + the few stand-alone gradients), join
    G3  clip + AdamW

Every rank issues the same collective sequence by construction (two all-reduces + the packed buckets per step,
eager or replayed alike).  Models without a latent cut (Diff, VAE, InfoDiff with a KL term) run their whole
backward in G1 and G2 is empty.  Should a capture fail on ANY rank, every rank drops it together (`_agree`)
and trains eagerly -- the same three phases, uncaptured.  A batch whose shape differs from the captured one (the
last, short batch of an epoch) runs eagerly, after which the step is captured afresh (the eager pass re-homes
gradients the graph's kernels write).  Objectives whose draws are made on the host every step (--prior 10mix /
roll: numpy samplers, models.py:654-657) are never captured; the KL capacity of --use_C lives in a device scalar
refreshed per call, so its schedule needs no re-capture.

Health check.  The group-synchronised data-gradient conv (ops.conv_dgrad_gn_sync_raw: the workgroups of an image wait for each
other inside the launch) needs its whole grid on the chip at once; if something else holds CUs -- another process on the GPU, a CU
mask, a collective -- a workgroup gives up after a bounded spin, bumps an error word in device memory, and THAT STEP'S GRADIENTS
ARE GARBAGE.  The step therefore reads the word (`check`) after the warm-up steps, then every `health_every` steps, and whenever the
caller asks (run.py: before every check-point and at the end of every epoch).  A clean check copies parameters + optimizer state into
a shadow ("last good", ~3x the parameter bytes); a check that finds time-outs retires the synchronised form for the rest of the
process, rolls the weights back to the last good shadow, drops the graph (it is captured afresh without the form) and says so on
stderr -- or raises SyncTimeoutError when `recover` is off or no clean check has happened yet.  The steps between the last clean check
and the time-out are lost (their batches are not replayed); nothing computed from a poisoned step survives."""
import sys

import torch

from . import ops


class SyncTimeoutError(RuntimeError):
    """A group-synchronised conv launch gave up waiting for its group: the gradients of that step were garbage."""


class GraphedTrainStep:
    def __init__(self, model, args, opt, sync=None, use_graph=True, warmup=2, pre_step=None, health_every=64, recover=True):
        self.health_every, self.recover = int(health_every), bool(recover)
        self.timeouts = 0             # time-outs seen over the life of this trainer
        self.recoveries = 0           # ... and the roll-backs they caused
        self._good = None             # (live tensors, shadow copies) of the last clean check
        self.trace = None             # measurement only (bench.py): a list that receives the five HIP events of each three-graph replay
        self.model, self.args, self.opt, self.sync = model, args, opt, sync
        self.pre_step = pre_step      # e.g. clip_grad_norm_ in front of a stock optimizer (between exchange and step)
        self.use_graph, self.warmup = use_graph, warmup
        self.graph = None             # one graph (no exchange) or the tuple (G1, G2 or None, G3)
        self.xbuf = None
        self.loss = None          # device scalar of the last step
        self.seen = 0
        host_prior = getattr(args, 'prior', 'regular') != 'regular' and getattr(args, 'mmd_weight', 0) != 0
        if host_prior:
            self.use_graph = False
        # data parallel: cut the backward pass at the latent so the backbone's slice of the gradients can travel while
        # the encoder's backward pass runs (models.InfoDiff.cut_latent; models without one keep a single backward pass)
        self.split = False
        if sync is not None and hasattr(model, 'attach_grad_sync'):
            self.split = bool(model.attach_grad_sync(sync))

    # ------------------------------------------------------------------ the phases of one step
    def _phase_a(self, x, epoch):
        """forward + the backward pass down to the latent cut (the whole backward pass when there is no cut)."""
        ops.WgradBatch.reset()          # nothing of an earlier, aborted backward pass may leak into this one
        if self.sync is not None:
            self.sync.begin_step()
        if self.split:
            self.model.arm_latent_cut()         # this call only: nobody else's forward pass is cut at the latent
        loss = self.model.loss_fn(args=self.args, x=x, curr_epoch=epoch)
        cut = self.model.pop_latent_cut() if self.split else None
        self.opt.zero_grad()
        loss.backward()
        return loss.detach(), cut

    def _phase_b(self, cut):
        """the encoder's backward pass from the gradient that arrived at the latent leaf.  With peers it runs beside the
        backbone slice's all-reduce: its data-gradient convs then keep to forms whose workgroups do not wait for each other
        inside a launch (ops.sync_convs) -- an RCCL kernel holding CUs would stall every such wait until the collective ends."""
        if cut is not None:
            lat, leaf = cut
            if leaf.grad is not None:
                with ops.sync_convs(not self._shares_chip()):
                    lat.backward(leaf.grad)
                leaf.grad = None

    def _shares_chip(self):
        if self.sync is None:
            return False
        if getattr(self.sync, 'force', False):      # the exchange path on one rank: RCCL's kernels still run beside phase B
            return True
        return (torch.distributed.is_available() and torch.distributed.is_initialized()
                and torch.distributed.get_world_size() > 1)

    # ------------------------------------------------------------------ health: time-outs of the synchronised convs
    def _live_state(self):
        live = []
        for grp in self.opt.param_groups:
            for p in grp['params']:
                live.append(p.data)
                st = self.opt.state.get(p)
                if st:
                    live.extend(v for v in st.values() if torch.is_tensor(v))
        extra = getattr(self.opt, '_state', None)       # FusedClipAdamW: step count + bias corrections on the device
        if torch.is_tensor(extra):
            live.append(extra)
        return live

    def _snapshot(self):
        live = self._live_state()
        if (self._good is not None and len(self._good[0]) == len(live)
                and all(a.data_ptr() == b.data_ptr() and a.shape == b.shape for a, b in zip(self._good[0], live))):
            torch._foreach_copy_(self._good[1], live)
        else:
            self._good = (live, [t.clone() for t in live])

    def _timeouts_everywhere(self):
        n = ops.rs_sync_timeouts(reset=False)
        if (self.sync is not None and torch.distributed.is_available() and torch.distributed.is_initialized()
                and torch.distributed.get_world_size() > 1):
            dev = next(iter(self.model.parameters())).device
            t = torch.tensor([float(n)], device=dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)     # every rank rolls back, or none
            n = max(n, int(t.item()))
        return n

    def check(self):
        """Read the synchronised convs' error word (synchronises the device).  -> time-outs found by this call (0: healthy, and
        the weights + optimizer state are now the "last good" shadow).  On time-outs: the form is retired, the weights are
        rolled back, the graph is dropped; SyncTimeoutError when that is not possible (`recover` off / no clean check yet)."""
        if not ops._RS_SYNC_STATE:              # no launch of the form so far (fp32 path, small batches, the form switched off)
            return 0
        n = self._timeouts_everywhere()
        if n == 0:
            if self.recover and not ops.sync_convs_retired():
                self._snapshot()
            return 0
        self.timeouts += n
        ops.retire_sync_convs()
        self.graph = None
        if hasattr(self.opt, '_key'):
            self.opt._key = None
        msg = ('%d workgroup(s) of the group-synchronised data-gradient conv timed out waiting for their group (the launch was not '
               'resident at once: is something else running on this GPU?); the gradients since the last clean check are garbage'
               % n)
        if not self.recover or self._good is None:
            raise SyncTimeoutError(msg + '; no clean state to roll back to -- restart from the last check-point with '
                                         'IDF_CONV_RS_SYNC=0')
        live, good = self._good
        with torch.no_grad():
            torch._foreach_copy_(live, good)
        torch.autograd.graph.increment_version([p for grp in self.opt.param_groups for p in grp['params']])
        self.recoveries += 1
        print('WARNING: ' + msg + '; rolled the weights and the optimizer state back to the last clean check (<= %d steps ago), '
              'retired the synchronised form for this process and re-capturing the step without it' % max(self.health_every, 1),
              file=sys.stderr)
        return n

    def _opt_step(self):
        if self.pre_step is not None:
            self.pre_step()
        self.opt.step()

    def forward_backward(self, x, epoch=0):
        """One eager forward + complete backward pass WITHOUT exchange or optimizer (measurement passes)."""
        loss, cut = self._phase_a(x, epoch)
        self._phase_b(cut)
        return loss

    def _eager_step(self, x, epoch):
        loss, cut = self._phase_a(x, epoch)
        if self.sync is not None and cut is not None:
            self.sync.reduce_early()
        self._phase_b(cut)
        if self.sync is not None:
            self.sync.all_reduce_grads()
        self._opt_step()
        return loss

    # ------------------------------------------------------------------ capture
    def _capture(self, x, epoch):
        self.xbuf = x.clone()
        self.loss = torch.zeros((), dtype=torch.float32, device=x.device)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        self._warm_done = False
        with torch.cuda.stream(side):               # one step on a side stream: private-pool warm-up
            self.loss.copy_(self._eager_step(self.xbuf, epoch))
        self._warm_done = True
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.opt.zero_grad()
        g1 = torch.cuda.CUDAGraph()
        if self.sync is None:
            with torch.cuda.graph(g1, capture_error_mode='thread_local'):
                loss, cut = self._phase_a(self.xbuf, epoch)
                self.loss.copy_(loss)
                self._phase_b(cut)
                self._opt_step()
            self.graph = g1
            return
        # thread_local: RCCL's watchdog thread keeps querying events while this thread captures
        with torch.cuda.graph(g1, capture_error_mode='thread_local'):
            loss, cut = self._phase_a(self.xbuf, epoch)
            self.loss.copy_(loss)
        g2 = None
        if cut is not None:
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, pool=g1.pool(), capture_error_mode='thread_local'):
                self._phase_b(cut)
        del cut
        g3 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g3, pool=g1.pool(), capture_error_mode='thread_local'):
            self._opt_step()
        self.graph = (g1, g2, g3)

    def _replay(self):
        if self.sync is None:
            self.graph.replay()
            return
        g1, g2, g3 = self.graph
        tr = self.trace

        def mark():
            if tr is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                tr[-1].append(e)
        if tr is not None:
            tr.append([])
        self.sync.begin_step()
        mark()
        g1.replay()
        mark()
        if g2 is not None:
            self.sync.reduce_early()        # backbone slice: on the wire while the encoder's backward pass replays
            g2.replay()
        mark()
        self.sync.all_reduce_grads()        # the rest; joins the exchange stream
        mark()
        g3.replay()
        mark()

    def _try_capture(self, x, epoch):
        """-> (captured, stepped).  Never raises: under data parallelism every rank must come out of here and meet
        the others in `_agree` before anyone decides how to go on."""
        try:
            self._capture(x, epoch)
            return True, True
        except Exception as e:  # noqa: BLE001
            # _capture's warm-up pass is a full optimisation step: if the failure came after it, this batch has been
            # trained on already
            stepped = bool(self.loss is not None and self.xbuf is not None and self.xbuf.shape == x.shape
                           and getattr(self, '_warm_done', False))
            print('graph capture failed (%s: %s)' % (type(e).__name__, str(e)[:200]), file=sys.stderr)
            self.graph = None
            # a backward pass that raised ran no end-of-backward callbacks: drop what it queued
            ops.WgradBatch.reset()
            if hasattr(self.model, 'pop_latent_cut'):
                self.model.pop_latent_cut()
            if self.sync is not None:
                self.sync.begin_step()              # a half-issued exchange must not leak into the next step
            torch.cuda.synchronize()
            return False, stepped

    def _agree(self, ok, stepped):
        """Under data parallelism every rank must take the same path (a rank that replays and a rank that steps
        eagerly could otherwise drift apart): ONE collective settles (capture worked everywhere, this batch's step ran
        everywhere, ... anywhere)."""
        if self.sync is None or not torch.distributed.is_initialized() or torch.distributed.get_world_size() == 1:
            return ok, stepped, stepped
        dev = self.loss.device if self.loss is not None else ('cuda' if torch.cuda.is_available() else 'cpu')
        flag = torch.tensor([float(ok), float(stepped), -float(stepped)], device=dev)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
        ok_all, st_all, st_any = (float(v) for v in flag.cpu())
        return ok_all > 0.5, st_all > 0.5, st_any < -0.5

    def __call__(self, x, epoch=0):
        """One optimisation step on batch x; returns the loss as a device scalar (no host sync)."""
        self.seen += 1
        if self.health_every > 0 and (self.seen == self.warmup + 1 or self.seen % self.health_every == 0):
            self.check()        # the steps so far (before this one touches the weights)
        if hasattr(self.model, 'set_epoch'):
            self.model.set_epoch(self.args, epoch)        # --use_C: the KL capacity's device scalar (no re-capture)
        if (self.use_graph and self.graph is None and self.seen > self.warmup
                and (self.xbuf is None or x.shape == self.xbuf.shape)):
            ok, stepped = self._try_capture(x, epoch)
            ok, stepped, stepped_any = self._agree(ok, stepped)      # every rank, before any rank retries or returns
            if stepped != stepped_any:
                raise RuntimeError('graph capture: the warm-up step ran on some ranks only; the ranks\' weights '
                                   'have diverged')
            if not ok:
                self.graph = None
                self.use_graph = False
                print('training eagerly', file=sys.stderr)
            if stepped:
                # the capture's warm-up pass WAS this batch's optimisation step (capturing itself executes nothing):
                # replaying / stepping now would train on the batch a second time
                return self.loss
        if self.graph is not None and x.shape == self.xbuf.shape:
            self.xbuf.copy_(x)
            self._replay()
            return self.loss
        had_graph = self.graph is not None
        loss = self._eager_step(x, epoch)
        if had_graph:
            # this eager step dropped the `.grad` tensors that live in the graph's private pool and installed its own
            # (gradients outside the arena), and re-keyed the optimizer's chunk table: the graph would now write
            # buffers nobody reads.  Capture afresh at the next full batch.
            self.graph = None
            if hasattr(self.opt, '_key'):
                self.opt._key = None
        return loss
