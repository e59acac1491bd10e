"""The training step of reference run.py:195-200 (loss_fn -> zero_grad -> backward -> [gradient exchange]
-> clip + AdamW) as a captured hipGraph.

Eagerly a CelebA step issues ~800 kernel launches from Python and is host-bound (~30 ms); replayed from a
graph it runs at the GPU's pace (~12 ms).  The step is captured after two eager warm-up steps (they build
the allocator pools, the weight shadows and the optimizer state).  With a gradient exchange (`sync`) only
forward + backward are captured; the all-reduce and the optimizer then run eagerly, as RCCL wants.
A batch whose shape differs from the captured one (the last, short batch of an epoch) runs eagerly."""
import sys

import torch


class GraphedTrainStep:
    def __init__(self, model, args, opt, sync=None, use_graph=True, warmup=2):
        self.model, self.args, self.opt, self.sync = model, args, opt, sync
        self.use_graph, self.warmup = use_graph, warmup
        self.graph = None
        self.xbuf = None
        self.loss = None          # device scalar of the last step
        self.epoch = None
        self.seen = 0

    def _fwd_bwd(self, x, epoch):
        loss = self.model.loss_fn(args=self.args, x=x, curr_epoch=epoch)
        self.opt.zero_grad()
        loss.backward()
        return loss.detach()

    def _tail(self):
        if self.sync is not None:
            self.sync.all_reduce_grads()
        self.opt.step()

    def _capture(self, x, epoch):
        self.xbuf = x.clone()
        self.loss = torch.zeros((), dtype=torch.float32, device=x.device)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):               # one step on a side stream: private-pool warm-up
            self.loss.copy_(self._fwd_bwd(self.xbuf, epoch))
            self._tail()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        self.opt.zero_grad()
        # thread_local: RCCL's watchdog thread keeps querying events while this thread captures
        with torch.cuda.graph(g, capture_error_mode='thread_local'):
            self.loss.copy_(self._fwd_bwd(self.xbuf, epoch))
            if self.sync is None:
                self._tail()
        self.graph, self.epoch = g, epoch

    def __call__(self, x, epoch=0):
        """One optimisation step on batch x; returns the loss as a device scalar (no host sync)."""
        self.seen += 1
        # the KL capacity schedule (models.py:662-671, --use_C) bakes the epoch into the graph: re-capture
        epoch_baked = getattr(self.args, 'use_C', False) and getattr(self.args, 'kld_weight', 0) != 0
        if self.graph is not None and epoch_baked and epoch != self.epoch:
            self.graph = None
        if (self.use_graph and self.graph is None and self.seen > self.warmup
                and (self.xbuf is None or x.shape == self.xbuf.shape)):
            try:
                self._capture(x, epoch)
                # the capture's warm-up pass WAS this batch's optimisation step (capturing itself executes nothing):
                # replaying now would train on the batch a second time
                return self.loss
            except Exception as e:  # noqa: BLE001
                print('graph capture failed (%s: %s); training eagerly' % (type(e).__name__, str(e)[:200]),
                      file=sys.stderr)
                self.use_graph, self.graph = False, None
                torch.cuda.synchronize()
        if self.graph is not None and x.shape == self.xbuf.shape:
            self.xbuf.copy_(x)
            self.graph.replay()
            if self.sync is not None:
                self._tail()
            return self.loss
        loss = self._fwd_bwd(x, epoch)
        self._tail()
        return loss
