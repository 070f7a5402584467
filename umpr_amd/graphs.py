"""A training step as ONE hipGraph launch (HIP graphs through torch.cuda.CUDAGraph), for the launch-bound case: UMPR-R
(BASELINE.json configs[0]) is ~70 kernels / 0.9 ms of GPU work per step, and issuing them one by one costs the host about as long
as the GPU needs to run them.  Captured once per batch geometry: forward (one C call for the whole ReviewNet + the head), backward,
Adam - everything between `model(*batch)` and `opt.step()` of main.py:32-37.  What changes from step to step lives in device
memory the graph reads: the batch (static input tensors, refreshed by copies before each replay), the sentence permutations (one
int32 buffer, `UMPR._index_upload`'s host half) and the Adam kernel's bias corrections (`umpr_adam_step_dev`).  Results are
bit-identical to the eager step (tests/test_gpu_e2e.py::test_graphed_umpr_r_step_equals_eager).

Not for the full model: its VGG backward forks onto the library's weight-gradient stream and its early Adam step onto another,
and it is GPU-bound anyway (31 ms of kernels behind 4 ms of host issue)."""
from __future__ import annotations

import torch

from .model import UMPR
from .train import train_step


class GraphedTrainStep:
    def __init__(self, model: UMPR, opt, example_batch):
        assert model.review_net_only, "GraphedTrainStep captures the UMPR-R step (see the module docstring)"
        dev = model.embedding.weight.device
        u, i, ui, ul, il, uil, photos, labels = example_batch
        self.model, self.opt, self.dev = model, opt, dev
        self.shape = tuple(u.shape)
        self.u, self.i, self.ui = (t.to(dev).clone() for t in (u, i, ui))
        self.labels = labels.to(dev).float().clone()
        self.photos = photos.to(dev)
        host, N, _ = UMPR._index_upload(ul, il, None, None)
        self.idx = host.to(dev)
        self.idx_host = [torch.empty_like(host).pin_memory() for _ in range(8)]
        self.idx_ev = [None] * 8
        self.k = 0
        self.static_index = (self.idx, self.idx[:2 * N], self.idx[2 * N:4 * N], None, None)
        self.lengths = (ul, il, uil)           # only their shapes matter inside the capture
        opt.enable_graph_mode()
        # warm-up (allocations, lazy initialisations) and capture must not change the training state: keep and restore it
        keep = [(g.p.clone(), g.m.clone(), g.v.clone()) for g in opt.groups]
        step0 = opt.step_count
        model._static_index = self.static_index
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(2):
                opt.prepare_step(1.0)
                self._body()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        opt.prepare_step(1.0)
        with torch.cuda.graph(self.graph):
            self.pred, self.loss = self._body()
        model._static_index = None
        with torch.no_grad():
            for g, (p, m, v) in zip(opt.groups, keep):
                g.p.copy_(p); g.m.copy_(m); g.v.copy_(v)
        opt.step_count = step0

    def _body(self):
        return train_step(self.model, self.opt, (self.u, self.i, self.ui, *self.lengths, self.photos, self.labels))

    def __call__(self, batch):
        """One training step on `batch` (same geometry as the example).  Returns (pred, loss): static tensors, overwritten by the
        next call."""
        u, i, ui, ul, il, uil, photos, labels = batch
        assert tuple(u.shape) == self.shape, "GraphedTrainStep is captured for one batch geometry"
        self.u.copy_(u, non_blocking=True); self.i.copy_(i, non_blocking=True); self.labels.copy_(labels, non_blocking=True)
        host, _, _ = UMPR._index_upload(ul, il, None, None)
        k = self.k % 8
        self.k += 1
        if self.idx_ev[k] is not None:
            self.idx_ev[k].synchronize()
        self.idx_host[k].copy_(host)
        self.idx.copy_(self.idx_host[k], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.dev))
        self.idx_ev[k] = ev
        self.opt.prepare_step(1.0)
        self.graph.replay()
        self.opt.step_count += 1
        return self.pred, self.loss
