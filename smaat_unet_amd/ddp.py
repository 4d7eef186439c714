"""Data-parallel gradient exchange for the SmaAt-UNet training step: one process per GPU,
ONE all-reduce per step over a flat fp32 gradient buffer (4,033,537 elements = 16.1 MB for
SmaAt_UNet(12,1)) through torch.distributed -- backend "nccl" is RCCL over xGMI on ROCm,
"gloo" on CPU for the unit tests.  Semantics = stock DistributedDataParallel: per-replica
BatchNorm statistics, gradients averaged (the reference itself is single-GPU,
train_precip_lightning.py:53-55).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class FlatGradAllReduce:
    def __init__(self, params, world_size=None, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.world = world_size if world_size is not None else (dist.get_world_size(group) if dist.is_initialized()
                                                                else 1)
        self.numel = sum(p.numel() for p in self.params)
        self.flat = None

    def broadcast_parameters(self, src=0):
        """identical replicas at start (and BN buffers if given as extra tensors)."""
        if self.world == 1:
            return
        flat = torch.cat([p.detach().reshape(-1) for p in self.params])
        dist.broadcast(flat, src, group=self.group)
        off = 0
        with torch.no_grad():
            for p in self.params:
                p.copy_(flat[off:off + p.numel()].view_as(p))
                off += p.numel()

    def reduce(self):
        """average .grad over ranks; afterwards every p.grad is a view into one flat buffer."""
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in self.params]
        flat = torch.cat([g.reshape(-1) for g in grads])
        if self.world > 1:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            flat.mul_(1.0 / self.world)
        off = 0
        for p in self.params:
            p.grad = flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        self.flat = flat
        return flat
