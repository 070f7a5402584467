"""Optimiser of the reference's training loop (main.py:22-26,31-37) on flat HBM arenas.

The reference builds ``torch.optim.Adam`` with two parameter groups - names without 'bias' get coupled L2
(weight_decay = l2_regularization), names with 'bias' get none - and an ``ExponentialLR`` stepped once per epoch.
Here every trainable parameter of a group is re-pointed into ONE contiguous fp32 arena (same for its gradient and the
two Adam moments), so an optimiser step is one ``umpr_adam_step`` launch per group, ``zero_grad`` is one memset, and
the data-parallel gradient exchange is a handful of large RCCL all-reduces on the arena instead of one per tensor.
"""
from __future__ import annotations

import torch

from ._lib import lib, stream_ptr


class _Group:
    def __init__(self, named_params, weight_decay, device):
        self.names = [n for n, _ in named_params]
        self.params = [p for _, p in named_params]
        self.weight_decay = weight_decay
        n = sum(p.numel() for p in self.params)
        self.numel = n
        self.p = torch.empty(n, dtype=torch.float32, device=device)
        self.g = torch.zeros(n, dtype=torch.float32, device=device)
        self.m = torch.zeros(n, dtype=torch.float32, device=device)
        self.v = torch.zeros(n, dtype=torch.float32, device=device)
        off = 0
        self.offsets = {}
        with torch.no_grad():
            for name, p in zip(self.names, self.params):
                k = p.numel()
                self.p[off:off + k].copy_(p.reshape(-1))
                p.data = self.p[off:off + k].view(p.shape)
                p.grad = self.g[off:off + k].view(p.shape)
                self.offsets[name] = (off, k)
                off += k


class FusedAdam:
    """Adam(betas=(0.9,0.999), eps=1e-8) with the reference's grouping; ``lr_decay`` per ``epoch_end()``."""

    def __init__(self, model, lr, l2_regularization, lr_decay=1.0, betas=(0.9, 0.999), eps=1e-8, order_key=None):
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        if order_key is not None:
            named.sort(key=lambda np_: order_key(np_[0]))
        device = named[0][1].device
        if device.type != "cuda":
            raise RuntimeError("FusedAdam launches HIP kernels: the model must be on a cuda device")
        self.groups = [_Group([(n, p) for n, p in named if 'bias' not in n], l2_regularization, device),
                       _Group([(n, p) for n, p in named if 'bias' in n], 0.0, device)]
        self.base_lr = lr
        self.lr = lr
        self.lr_decay = lr_decay
        self.betas = betas
        self.eps = eps
        self.step_count = 0

    def zero_grad(self):
        for g in self.groups:
            g.g.zero_()

    def step(self, grad_scale=1.0):
        self.step_count += 1
        for g in self.groups:
            if g.numel:
                lib().call("umpr_adam_step", g.p, g.g, g.m, g.v, g.numel, self.lr, self.betas[0], self.betas[1],
                           self.eps, g.weight_decay, self.step_count, grad_scale, stream_ptr())

    def epoch_end(self):
        """ExponentialLR.step() (main.py:54)."""
        self.lr *= self.lr_decay

    def grad_arenas(self):
        return [g.g for g in self.groups if g.numel]
