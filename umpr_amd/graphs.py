"""A training step as ONE hipGraph launch (HIP graphs through torch.cuda.CUDAGraph), for the launch-bound case: UMPR-R
(BASELINE.json configs[0]) is ~55 kernels / 0.8 ms of GPU work per step, and issuing them one by one costs the host about as long
as the GPU needs to run them.  Captured once per batch geometry: forward (one C call for the whole ReviewNet + the head), backward,
Adam - everything between `model(*batch)` and `opt.step()` of main.py:32-37.

What changes from step to step lives in ONE device buffer the graph reads:

    [ sentence permutations int32 | Adam scalars float32 [groups][4] | labels float32 [B] | user ids int64 | item ids int64 ]

* a batch that arrives from the HOST is packed into a pinned copy of that buffer and travels in one upload;
* a batch that is already on the device costs one small upload (permutations + Adam scalars: both are host data) and copies of
  its tensors into the buffer - none at all for tensors that ARE the buffer's views (`resident()` hands those out);
* user and item ids sit back to back, which is the [2N][L] tensor the shared GRU reads: no concatenation inside the graph.

Results are bit-identical to the eager step (tests/test_gpu_e2e.py::test_graphed_umpr_r_step_equals_eager).

Not for the full model: its VGG backward forks onto the library's weight-gradient stream and its early Adam step onto another,
and it is GPU-bound anyway (31 ms of kernels behind 4 ms of host issue)."""
from __future__ import annotations

import torch

from .model import UMPR
from .train import train_step

_RING = 8


def _align(n, a=256):
    return (n + a - 1) // a * a


class GraphedTrainStep:
    def __init__(self, model: UMPR, opt, example_batch):
        assert model.review_net_only, "GraphedTrainStep captures the UMPR-R step (see the module docstring)"
        dev = model.embedding.weight.device
        u, i, ui, ul, il, uil, photos, labels = example_batch
        self.model, self.opt, self.dev = model, opt, dev
        self.shape = tuple(u.shape)
        B, S, L = self.shape
        assert tuple(i.shape) == self.shape
        N = B * S
        idx_host, n_pair, _ = UMPR._index_upload(ul, il, None, None)
        assert n_pair == N
        G = len(opt.groups)
        # byte layout of the step buffer
        self.o_idx, n_idx = 0, idx_host.numel()
        self.o_hyp = _align(self.o_idx + 4 * n_idx, 16)
        self.head_bytes = _align(self.o_hyp + 16 * G)          # what a device-resident batch still uploads per step
        self.o_lab = self.head_bytes
        self.o_ids = _align(self.o_lab + 4 * B)
        self.nbytes = self.o_ids + 2 * N * L * 8
        self.blob = torch.zeros(self.nbytes, dtype=torch.uint8, device=dev)
        self.host = [torch.zeros(self.nbytes, dtype=torch.uint8).pin_memory() for _ in range(_RING)]
        self.host_ev = [None] * _RING
        self.k = 0

        def views(b):
            idx = b[self.o_idx:self.o_idx + 4 * n_idx].view(torch.int32)
            hyp = b[self.o_hyp:self.o_hyp + 16 * G].view(torch.float32).view(G, 4)
            lab = b[self.o_lab:self.o_lab + 4 * B].view(torch.float32)
            ids = b[self.o_ids:self.o_ids + 2 * N * L * 8].view(torch.int64).view(2, B, S, L)
            return idx, hyp, lab, ids
        self.idx, self.hyper, self.labels, ids = views(self.blob)
        self.u, self.i = ids[0], ids[1]
        self.host_views = [views(h) for h in self.host]
        self.ui = ui.to(dev)
        self.photos = photos.to(dev)
        self.static_index = (self.idx, self.idx[:2 * N], self.idx[2 * N:4 * N], None, None)
        self.lengths = (ul, il, uil)           # only their shapes matter inside the capture
        self._load(example_batch, 1.0)
        opt.enable_graph_mode(hyper=self.hyper)
        # warm-up (allocations, lazy initialisations) and capture must not change the training state: keep and restore it
        keep = [(g.p.clone(), g.m.clone(), g.v.clone()) for g in opt.groups]
        step0 = opt.step_count
        model._static_index = self.static_index
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(2):
                self._body()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.pred, self.loss = self._body()
        model._static_index = None
        with torch.no_grad():
            for g, (p, m, v) in zip(opt.groups, keep):
                g.p.copy_(p); g.m.copy_(m); g.v.copy_(v)
        opt.step_count = step0

    def _body(self):
        return train_step(self.model, self.opt, (self.u, self.i, self.ui, *self.lengths, self.photos, self.labels))

    def resident(self, batch):
        """The batch with its ids and labels living in the step buffer itself: stepping on the returned tuple copies nothing but
        the permutations and the Adam scalars (a training loop that keeps one batch on the device - bench.py, a test)."""
        u, i, ui, ul, il, uil, photos, labels = batch
        assert tuple(u.shape) == self.shape
        self.u.copy_(u); self.i.copy_(i); self.labels.copy_(labels.to(self.dev).float())
        return (self.u, self.i, ui, ul, il, uil, photos, self.labels)

    def _is(self, t, mine):
        return isinstance(t, torch.Tensor) and t.device == mine.device and t.data_ptr() == mine.data_ptr()

    def _load(self, batch, grad_scale):
        u, i, ui, ul, il, uil, photos, labels = batch
        assert tuple(u.shape) == self.shape, "GraphedTrainStep is captured for one batch geometry"
        k = self.k % _RING
        self.k += 1
        if self.host_ev[k] is not None:
            self.host_ev[k].synchronize()
        h_idx, h_hyp, h_lab, h_ids = self.host_views[k]
        idx_host, _, _ = UMPR._index_upload(ul, il, None, None)
        h_idx.copy_(idx_host)
        h_hyp.copy_(torch.tensor(self.opt.hyper_rows(grad_scale), dtype=torch.float32))
        on_host = not u.is_cuda and not i.is_cuda and not labels.is_cuda
        if on_host:                      # everything in one upload
            h_ids[0].copy_(u); h_ids[1].copy_(i); h_lab.copy_(labels.float())
            self.blob.copy_(self.host[k], non_blocking=True)
        else:
            self.blob[:self.head_bytes].copy_(self.host[k][:self.head_bytes], non_blocking=True)
            if not self._is(u, self.u):
                self.u.copy_(u, non_blocking=True)
            if not self._is(i, self.i):
                self.i.copy_(i, non_blocking=True)
            if not self._is(labels, self.labels):
                self.labels.copy_(labels, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.dev))
        self.host_ev[k] = ev

    def __call__(self, batch, grad_scale=1.0):
        """One training step on `batch` (same geometry as the example).  Returns (pred, loss): static tensors, overwritten by the
        next call."""
        self._load(batch, grad_scale)
        self.graph.replay()
        self.opt.step_count += 1
        return self.pred, self.loss
