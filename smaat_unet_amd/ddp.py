"""Data-parallel gradient exchange for the SmaAt-UNet training step: one process per GPU,
identical replicas, per-replica BatchNorm statistics (stock DistributedDataParallel semantics; the reference itself
is single-GPU, train_precip_lightning.py:53-55), gradients AVERAGED over ranks through `torch.distributed`
(backend "nccl" = RCCL over xGMI on ROCm, "gloo" on CPU for the unit tests).

Layout: ONE persistent flat fp32 buffer (4,033,537 elements = 16.1 MB for SmaAt_UNet(12, 1)), ordered by REVERSE
registration order (outc, up4, ... , inc: the gradients of the decoder are complete first) and cut into `buckets`
contiguous ranges, each all-reduced with one collective.  What happens per step depends on the mode:
  * default (`overlap=False`): `zero_grad()` sets every `p.grad` to None, autograd hands over fresh gradient tensors,
    `finish()` PACKS them into the flat buffer with one multi-tensor copy (`torch._foreach_copy_`, 16 MB), all-reduces the
    buckets, averages, and re-points every `p.grad` at its view of the reduced buffer (no copy back: the optimizer reads
    the views).  One 16 MB pack per step, nothing overlapped with the backward.
  * `overlap=True`: every `p.grad` IS a view of the flat buffer for the whole run (no pack, no copy back); autograd
    accumulates into the views and post-accumulate hooks launch a bucket's all-reduce as soon as its last gradient has
    been accumulated, overlapping the exchange of the decoder's gradients with the encoder's backward.
With xGMI's point-to-point links a ring all-reduce of 16 MB costs ~0.2 ms against a ~35 ms step, so two buckets are
plenty; more only add launch overhead.  Neither mode has been timed on more than one GPU (the build box has one).

    ddp = FlatGradAllReduce(model)        # module or iterable of parameters
    ddp.broadcast_parameters()            # identical replicas (parameters AND buffers from rank 0)
    for batch in data:
        ddp.zero_grad()                   # replaces optimizer.zero_grad
        loss = f(model(x)); loss.backward()
        ddp.finish()                      # gather -> all-reduce -> average; p.grad are views of the reduced buffer
        optimizer.step()

Two modes (measured on MI355X, profiles/r2): with `.grad` pre-set to the views, autograd ACCUMULATES into them -- one
small add kernel per parameter, 145 launches = +0.7 ms per step -- which is what buys the overlap of the bucket
all-reduces with the backward (`overlap=True`).  The default (`overlap=False`) lets autograd hand over fresh gradient
tensors (no add kernels), packs them into the flat buffer with ONE multi-tensor copy after the backward and issues the
bucket all-reduces then: the 16 MB exchange (~0.2-0.4 ms on 8 GPUs over xGMI) is not overlapped but costs less than
the accumulation kernels it avoids.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class FlatGradAllReduce:
    def __init__(self, module_or_params, world_size=None, group=None, buckets=2, overlap=False, force_collectives=False):
        self.module = module_or_params if isinstance(module_or_params, torch.nn.Module) else None
        params = list(self.module.parameters() if self.module is not None else module_or_params)
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.world = world_size if world_size is not None else (dist.get_world_size(group) if dist.is_initialized()
                                                                else 1)
        self.numel = sum(p.numel() for p in self.params)
        if not self.params:
            raise ValueError("no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        if any(p.device != dev or p.dtype != dt for p in self.params):
            raise ValueError("FlatGradAllReduce needs all parameters on one device with one dtype")
        # persistent flat gradient buffer, reverse registration order
        self.flat = torch.zeros(self.numel, dtype=dt, device=dev)
        self._order = list(reversed(self.params))
        off = 0
        self._range = {}
        for p in self._order:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            self._range[p] = (off, off + n)
            off += n
        # contiguous buckets of ~equal size; bucket 0 holds the gradients that are ready first
        nb = max(1, min(int(buckets), len(self._order)))
        target = (self.numel + nb - 1) // nb
        self._buckets, cur, start = [], [], 0
        for p in self._order:
            cur.append(p)
            end = self._range[p][1]
            if end - start >= target and len(self._buckets) < nb - 1:
                self._buckets.append((start, end, cur))
                cur, start = [], end
        if cur:
            self._buckets.append((start, self.numel, cur))
        self._bucket_of = {p: i for i, (_, _, ps) in enumerate(self._buckets) for p in ps}
        self._pending = [0] * len(self._buckets)
        self._works = []
        self._launched = [False] * len(self._buckets)
        # force_collectives: issue the all-reduces even at world size 1 (RCCL smoke test on a single-GPU box)
        self._coll = self.world > 1 or bool(force_collectives)
        self.overlap = bool(overlap) and self._coll
        self.active = True  # False: no collectives at all (a rank-local profiling step must not talk to its peers)
        self._hooks = []
        if self.overlap:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self._arm()

    # ---------------------------------------------------------------------------------------------
    def _arm(self):
        for i, (_, _, ps) in enumerate(self._buckets):
            self._pending[i] = len(ps)
            self._launched[i] = False
        self._works = []

    def _check_views(self, p):
        a, b = self._range[p]
        if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + a * self.flat.element_size():
            # something (optimizer.zero_grad(set_to_none=True), a user assignment) replaced the view: copy the
            # gradient into its slot and restore the view
            g = p.grad
            view = self.flat[a:b].view_as(p)
            if g is not None:
                view.copy_(g)
            p.grad = view

    def _launch(self, i):
        a, b, _ = self._buckets[i]
        self._launched[i] = True
        if self._coll and self.active:
            self._works.append(dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _on_grad(self, p):
        self._check_views(p)
        i = self._bucket_of[p]
        self._pending[i] -= 1
        if self._pending[i] == 0 and not self._launched[i]:
            self._launch(i)

    # ---------------------------------------------------------------------------------------------
    def broadcast_parameters(self, src=0):
        """identical replicas at start: parameters and (when a module was given) its buffers -- BatchNorm running
        statistics and num_batches_tracked -- from rank `src`."""
        if self.world == 1:
            return
        flat = torch.cat([p.detach().reshape(-1) for p in self.params])
        dist.broadcast(flat, src, group=self.group)
        off = 0
        with torch.no_grad():
            for p in self.params:
                p.copy_(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
        self.sync_buffers(src)

    def sync_buffers(self, src=0):
        """rank `src`'s buffers to every rank (what stock DDP does before each forward with broadcast_buffers=True;
        8,865 elements for SmaAt_UNet: one small broadcast)."""
        if self.world == 1 or self.module is None:
            return
        bufs = [b for b in self.module.buffers()]
        for dtype in sorted({b.dtype for b in bufs}, key=str):
            group = [b for b in bufs if b.dtype == dtype]
            flat = torch.cat([b.detach().reshape(-1) for b in group])
            dist.broadcast(flat, src, group=self.group)
            off = 0
            with torch.no_grad():
                for b in group:
                    b.copy_(flat[off:off + b.numel()].view_as(b))
                    off += b.numel()

    def zero_grad(self):
        """use INSTEAD of optimizer.zero_grad().  overlap mode: one memset, every p.grad stays a view of the flat buffer
        (autograd accumulates into it); default mode: gradients are dropped, autograd will hand over fresh tensors."""
        if self.overlap:
            self.flat.zero_()
            for p in self.params:
                self._check_views(p)
        else:
            for p in self.params:
                p.grad = None
        self._arm()

    def finish(self):
        """wait for the bucket all-reduces (launching those whose hooks did not fire: parameters that received no
        gradient this step keep their zeros), average.  Returns the flat buffer."""
        if self.overlap:
            for p in self.params:
                self._check_views(p)
        else:  # pack the fresh gradients with one multi-tensor copy; parameters without a gradient contribute zeros
            have = [p for p in self.params if p.grad is not None]
            if len(have) != len(self.params):
                self.flat.zero_()
            if have:
                torch._foreach_copy_([self.flat[self._range[p][0]:self._range[p][1]].view_as(p) for p in have],
                                     [p.grad for p in have])
            for p in self.params:
                a, b = self._range[p]
                p.grad = self.flat[a:b].view_as(p)
        if self._coll and self.active:
            for i in range(len(self._buckets)):
                if not self._launched[i]:
                    self._launch(i)
            for w in self._works:
                w.wait()
            self.flat.mul_(1.0 / self.world)
        self._arm()
        return self.flat

    # kept for callers that do not use the hooks: reduce everything now
    def reduce(self):
        return self.finish()

    def close(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
