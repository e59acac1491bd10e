"""The training step of reference run.py:195-200 (loss_fn -> zero_grad -> backward -> [gradient exchange]
-> clip + AdamW) as a captured hipGraph.

Eagerly a CelebA step issues ~550 kernel launches from Python and is host-bound (~30 ms); replayed from a
graph it runs at the GPU's pace (~10 ms).  The step is captured after two eager warm-up steps (they build
the allocator pools, the weight shadows and the optimizer state).  With a gradient exchange (`sync`) forward +
backward are captured and the all-reduce + optimizer run eagerly after each replay (the overlap of the
backbone's slice with the encoder's backward pass is then off: a collective forked inside a capture must be
joined inside it).  IDF_DP_INGRAPH=1 captures the all-reduces (RCCL on a side stream, event-joined) and the
optimizer too -- the whole data-parallel step as one graph; it works but the process aborts intermittently
while such a capture is open on this stack, so it is opt-in.  Should a capture fail on ANY rank, every rank
drops it together and retries with less in the graph at its next step.  A batch whose shape differs from the captured one (the last, short batch of an
epoch) runs eagerly, after which the step is captured afresh (the eager pass re-homes gradients the graph's
kernels write).  Objectives whose draws are made on the host every step (--prior 10mix / roll: numpy samplers,
models.py:654-657) are never captured; the KL capacity of --use_C lives in a device scalar refreshed per call,
so its schedule needs no re-capture."""
import sys

import torch


class GraphedTrainStep:
    def __init__(self, model, args, opt, sync=None, use_graph=True, warmup=2, pre_step=None):
        self.model, self.args, self.opt, self.sync = model, args, opt, sync
        self.pre_step = pre_step      # e.g. clip_grad_norm_ in front of a stock optimizer (between exchange and step)
        self.use_graph, self.warmup = use_graph, warmup
        self.graph = None
        self.xbuf = None
        self.loss = None          # device scalar of the last step
        self.seen = 0
        host_prior = getattr(args, 'prior', 'regular') != 'regular' and getattr(args, 'mmd_weight', 0) != 0
        if host_prior:
            self.use_graph = False
        # The gradient exchange stays OUTSIDE the captured step by default: forward + backward replay from the graph, the
        # all-reduce and the optimizer follow eagerly (five launches).  Capturing RCCL collectives works on this stack but the
        # process aborts intermittently while such a capture is open (3 of 8 runs of the one-rank test, 1 of 8 with every
        # collective issued from the main thread; none in 8 with the exchange outside) -- IDF_DP_INGRAPH=1 opts in.
        import os
        self.sync_in_graph = sync is not None and os.environ.get('IDF_DP_INGRAPH', '0') == '1'
        if sync is not None and not self.sync_in_graph and self.use_graph:
            sync.early_enabled = False             # a collective forked inside a capture must be joined inside it
        if sync is not None and hasattr(model, 'attach_grad_sync'):
            model.attach_grad_sync(sync)

    def _fwd_bwd(self, x, epoch):
        if self.sync is not None:
            self.sync.begin_step()
        loss = self.model.loss_fn(args=self.args, x=x, curr_epoch=epoch)
        self.opt.zero_grad()
        loss.backward()
        return loss.detach()

    def _tail(self):
        if self.sync is not None:
            self.sync.all_reduce_grads()
        if self.pre_step is not None:
            self.pre_step()
        self.opt.step()

    def _capture(self, x, epoch):
        self.xbuf = x.clone()
        self.loss = torch.zeros((), dtype=torch.float32, device=x.device)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        self._warm_done = False
        with torch.cuda.stream(side):               # one step on a side stream: private-pool warm-up
            self.loss.copy_(self._fwd_bwd(self.xbuf, epoch))
            self._tail()
        self._warm_done = True
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        self.opt.zero_grad()
        # thread_local: RCCL's watchdog thread keeps querying events while this thread captures
        with torch.cuda.graph(g, capture_error_mode='thread_local'):
            self.loss.copy_(self._fwd_bwd(self.xbuf, epoch))
            if self.sync is None or self.sync_in_graph:
                self._tail()
        self.graph = g

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
            print('graph capture %sfailed (%s: %s)' % ('with the gradient exchange ' if self.sync is not None and
                                                       self.sync_in_graph else '', type(e).__name__, str(e)[:200]),
                  file=sys.stderr)
            self.graph = None
            if self.sync is not None:
                self.sync.begin_step()              # a half-issued exchange must not leak into the next step
            torch.cuda.synchronize()
            return False, stepped

    def _agree(self, ok, stepped):
        """Under data parallelism every rank must take the same path (a rank that replays and a rank that steps
        eagerly issue different numbers of collectives): ONE collective settles (capture worked everywhere, this
        batch's step ran everywhere, ... anywhere)."""
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
                if self.sync is not None and self.sync_in_graph:
                    # together, at the next step: forward + backward only in the graph, exchange + optimizer eagerly
                    # after each replay, no collective forked inside the capture
                    self.sync_in_graph = False
                    self.sync.early_enabled = False
                    print('retrying at the next step with forward + backward only in the graph', file=sys.stderr)
                else:
                    self.use_graph = False
                    print('training eagerly', file=sys.stderr)
            if stepped:
                # the capture's warm-up pass WAS this batch's optimisation step (capturing itself executes nothing):
                # replaying / stepping now would train on the batch a second time
                return self.loss
        if self.graph is not None and x.shape == self.xbuf.shape:
            self.xbuf.copy_(x)
            self.graph.replay()
            if self.sync is not None and not self.sync_in_graph:
                self._tail()
            return self.loss
        had_graph = self.graph is not None
        loss = self._fwd_bwd(x, epoch)
        self._tail()
        if had_graph:
            # this eager step dropped the `.grad` tensors that live in the graph's private pool and installed its own
            # (gradients outside the arena), and re-keyed the optimizer's chunk table: the graph would now write
            # buffers nobody reads.  Capture afresh at the next full batch.
            self.graph = None
            if hasattr(self.opt, '_key'):
                self.opt._key = None
        return loss
