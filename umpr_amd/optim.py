"""Optimiser of the reference's training loop (main.py:22-26,31-37) on flat HBM arenas.

The reference builds ``torch.optim.Adam`` with two parameter groups - names without 'bias' get coupled L2
(weight_decay = l2_regularization), names with 'bias' get none - and an ``ExponentialLR`` stepped once per epoch.
Here every trainable parameter of a group is re-pointed into ONE contiguous fp32 arena (same for its gradient and the
two Adam moments), so an optimiser step is one ``umpr_adam_step`` launch per group, ``zero_grad`` is one memset, and
the data-parallel gradient exchange is a handful of large RCCL all-reduces on the arena instead of one per tensor.
"""
from __future__ import annotations

import os
import weakref

import torch

from .streams import wait_for_gradients
from ._lib import lib, stream_ptr


# "1" (default): on; "0": off.
_EARLY_ADAM = os.environ.get("UMPR_EARLY_ADAM", "1")


class WeakCallback:
    """Entry of a module's ``grad_callbacks``: holds its target method weakly, so an optimiser / reducer that has been
    dropped (and its ~2 GB of flat arenas) is not kept alive - or called - by the model it was once bound to."""

    def __init__(self, method):
        self.ref = weakref.WeakMethod(method)

    def __call__(self):
        m = self.ref()
        if m is not None:
            m()

    def targets(self, method):
        m = self.ref()
        return m is not None and m == method


def add_callback(module, method):
    module.grad_callbacks[:] = [cb for cb in module.grad_callbacks if not (isinstance(cb, WeakCallback) and cb.ref() is None)]
    module.grad_callbacks.append(WeakCallback(method))


def remove_callback(module, method):
    module.grad_callbacks[:] = [cb for cb in module.grad_callbacks
                                if not (isinstance(cb, WeakCallback) and (cb.ref() is None or cb.targets(method)))]


def has_callback(module, method):
    return any(isinstance(cb, WeakCallback) and cb.targets(method) for cb in module.grad_callbacks)


def _is_classifier(name):
    return name.startswith("classifier.") or ".classifier." in name


def _parallel_active():
    from . import parallel
    return parallel.active()


_ALIGN = 64      # floats


def _pad(k):
    return (k + _ALIGN - 1) // _ALIGN * _ALIGN


class _Group:
    def __init__(self, named_params, weight_decay, device):
        self.names = [n for n, _ in named_params]
        self.params = [p for _, p in named_params]
        self.weight_decay = weight_decay
        # every parameter starts on a 256-byte boundary of the arenas: the kernels read weights as float4 / whole 128-B lines, and a
        # one-element bias in front of the 411 MB fc1 matrix must not leave it (and every tensor behind it) at an odd offset.  The
        # padding elements are zeros with zero gradients; Adam leaves them at zero, collectives carry them along.
        n = sum(_pad(p.numel()) for p in self.params)
        self.numel = n
        self.p = torch.zeros(n, dtype=torch.float32, device=device)
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
                off += _pad(k)
        # parameters whose backward node writes the gradient in place (model.py::_grad_targets): their arena slices need
        # no zero fill; everything else is zeroed as merged ranges
        self.direct = [p for p in self.params if getattr(p, "_umpr_direct", False)]
        self.zero_ranges = []
        for name, p in zip(self.names, self.params):
            if getattr(p, "_umpr_direct", False):
                continue
            lo, k = self.offsets[name]
            if self.zero_ranges and self.zero_ranges[-1][1] == lo:
                self.zero_ranges[-1][1] = lo + _pad(k)          # (the padding behind a parameter belongs to its range)
            else:
                self.zero_ranges.append([lo, lo + _pad(k)])


class FusedAdam:
    """Adam(betas=(0.9,0.999), eps=1e-8) with the reference's grouping; ``lr_decay`` per ``epoch_end()``."""

    def __init__(self, model, lr, l2_regularization, lr_decay=1.0, betas=(0.9, 0.999), eps=1e-8, order_key=None):
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        if order_key is None:
            # the VGG classifier's gradients are produced first in backward and are 89 % of all gradient bytes: put them
            # at the front of the arena so that one contiguous slice can be all-reduced while the conv backward runs
            order_key = lambda name: 0 if _is_classifier(name) else 1
        named.sort(key=lambda np_: order_key(np_[0]))  # stable: model order inside each class
        device = named[0][1].device
        self.groups = [_Group([(n, p) for n, p in named if 'bias' not in n], l2_regularization, device),
                       _Group([(n, p) for n, p in named if 'bias' in n], 0.0, device)]
        self.model = model
        self.base_lr = lr
        self.lr = lr
        self.lr_decay = lr_decay
        self.betas = betas
        self.eps = eps
        self.step_count = 0
        # Early update of the VGG classifier slice (89 % of all parameters): its gradients are final as soon as the
        # classifier's backward node has run, long before the convolutional backward ends.  When train_step has armed it,
        # the Adam kernel for that slice runs on a side stream underneath the conv backward (behind the slice's
        # all-reduce when data parallel) instead of after everything: 0.75 of the 0.85 ms the optimiser takes at batch 64
        # leaves the critical path.  Elementwise, so the result is bit-identical to the late update.
        self._early = None          # (grad_scale,) while armed
        self._early_done = None     # (lo, hi, event) once the slice has been updated in this step
        self._early_stream = None
        self.reducer = None         # set by parallel.GradReducer: it then calls early_step after launching its all-reduce
        # one optimiser per model: binding a new one (a resumed run, a second test leg) detaches the old one's callbacks
        prev = getattr(model, "_umpr_optimizer", None)
        prev = prev() if prev is not None else None
        if prev is not None and prev is not self:
            prev.close()
        model._umpr_optimizer = weakref.ref(self)
        for m in model.modules():
            if hasattr(m, "grad_callbacks"):
                add_callback(m, self._on_classifier_grads)

    def close(self):
        """Detach from the model: remove this optimiser's callbacks (and its reducer's).  The parameters stay views of the
        arenas (they own them from here on); a new FusedAdam on the same model re-points them into its own."""
        if self.reducer is not None:
            self.reducer.close()
        for m in self.model.modules():
            if hasattr(m, "grad_callbacks"):
                remove_callback(m, self._on_classifier_grads)
        self._early = self._early_done = None

    # ---- hipGraph support: the Adam kernel's per-step scalars live in device memory (umpr_adam_step_dev) --------------------
    def enable_graph_mode(self, hyper=None):
        """From now on step() launches the device-scalar form of the kernel; prepare_step() must run before every step (eager or
        replayed) to refresh the scalars for the step about to be taken.  The update is bit-identical to the plain form.
        `hyper`: a device float32 [groups][4] view the CALLER keeps fresh from hyper_rows() (umpr_amd/graphs.py carries the scalars
        inside its one upload per step); prepare_step() is then not used."""
        dev = self.groups[0].p.device
        if hyper is not None:
            assert hyper.shape == (len(self.groups), 4) and hyper.dtype == torch.float32 and hyper.device == dev
            self._hyper = hyper
            self._hyper_host = None
            return
        self._hyper = torch.zeros(len(self.groups), 4, device=dev, dtype=torch.float32)
        self._hyper_host = [torch.zeros(len(self.groups), 4, dtype=torch.float32).pin_memory() for _ in range(8)]
        self._hyper_ev = [None] * 8
        self._hyper_k = 0

    def hyper_rows(self, grad_scale=1.0):
        """[grad_scale, lr / (1 - b1^t), 1 / sqrt(1 - b2^t), weight_decay] per group for the step about to be taken (float64 on
        the host, as the plain kernel's launcher computes them)."""
        t = self.step_count + 1
        return [[float(grad_scale), float(self.lr / (1.0 - self.betas[0] ** t)), float(1.0 / (1.0 - self.betas[1] ** t) ** 0.5),
                 float(g.weight_decay)] for g in self.groups]

    def prepare_step(self, grad_scale=1.0):
        rows = self.hyper_rows(grad_scale)
        assert self._hyper_host is not None, "prepare_step: this optimiser's scalars live in a caller-owned buffer (hyper_rows())"
        k = self._hyper_k % 8
        self._hyper_k += 1
        if self._hyper_ev[k] is not None:
            self._hyper_ev[k].synchronize()
        h = self._hyper_host[k]
        h.copy_(torch.tensor(rows, dtype=torch.float32))
        self._hyper.copy_(h, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self._hyper.device))
        self._hyper_ev[k] = ev

    def zero_grad(self):
        for g in self.groups:
            for lo, hi in g.zero_ranges:
                g.g[lo:hi].zero_()
            for p in g.direct:
                p._umpr_fresh = True

    def arm_early(self, grad_scale=1.0):
        """train_step: the coming backward belongs to exactly one optimiser step with this gradient scale.
        UMPR_EARLY_ADAM=0 turns it off.  Worth 0.3-0.5 ms per step behind the all-reduce in the RCCL rehearsal (43.16 vs
        43.49 ms fp32, 16.06 vs 16.51 ms bf16); on a single GPU 0.1-0.3 ms in bf16 (9.56 vs 9.71 ms) and nothing in fp32.  Its stream is one more next to main / text / weight-gradient / RCCL:
        umpr_amd/__init__.py raises GPU_MAX_HW_QUEUES so that they do not share hardware queues."""
        if _EARLY_ADAM == "0":
            return
        self._early = (float(grad_scale),)
        self._early_done = None

    def disarm(self):
        self._early = None

    def _on_classifier_grads(self):
        # data parallel without a GradReducer (train_step's allreduce_arenas path): the slice is only summed over the ranks
        # AFTER backward, so an update from here would use this rank's local gradient - never early in that case
        if self._early is not None and self.reducer is None and not _parallel_active():
            self.early_step(())

    def early_step(self, handles, stream=None):
        """Adam on the classifier-weight slice, on a side stream; `handles`: the slice's pending async all-reduces;
        `stream`: run it there (the reducer's exchange stream, already ordered behind the slice's in-stream all-reduce)."""
        if self._early is None or self._early_done is not None:
            return
        eb = self.early_bucket()
        if eb is None or self.groups[0].p.device.type != "cuda":
            return
        (scale,) = self._early
        g = self.groups[0]
        _, lo, hi, _ = eb
        dev = g.p.device
        if stream is not None:
            self._early_stream = stream
        if self._early_stream is None:
            # Without a gradient exchange the update rides on the library's weight-gradient stream (Adam is HBM-bound, the
            # weight-gradient kernels behind it MFMA-bound): a sixth stream of its own - main, text, weight-gradient, copy
            # and this one - cost 1.8 ms (bf16) / 3.7 ms (fp32) per step as soon as per-step uploads ran on a copy stream
            # (bench.py --h2d: 11.6 vs 9.79 ms, 45.3 vs 41.4 ms).  Behind an all-reduce it keeps its own stream: the
            # weight-gradient kernels must not queue behind the collective.
            ws = lib().fn["umpr_vgg16_wgrad_stream"]() if not handles else None
            self._early_stream = torch.cuda.ExternalStream(ws, device=dev) if ws else torch.cuda.Stream(dev)
        main = torch.cuda.current_stream(dev)
        self._early_stream.wait_stream(main)            # the gradients were written on the backward's stream
        with torch.cuda.stream(self._early_stream):
            for h in handles:
                h.wait()                                # orders this stream behind the collective, not the host
            lib().call("umpr_adam_step", g.p[lo:hi], g.g[lo:hi], g.m[lo:hi], g.v[lo:hi], hi - lo, self.lr, self.betas[0],
                       self.betas[1], self.eps, g.weight_decay, self.step_count + 1, scale, stream_ptr())
            ev = torch.cuda.Event()
            ev.record(self._early_stream)
        self._early_done = (lo, hi, ev)

    def step(self, grad_scale=1.0):
        if self.groups[0].p.device.type != "cuda":
            raise RuntimeError("FusedAdam.step launches a HIP kernel: the model must be on a cuda device")
        self.step_count += 1
        wait_for_gradients(self.groups[0].p.device)    # in-place gradients written from a side stream (umpr_amd/streams.py)
        done, self._early_done, self._early = self._early_done, None, None
        for gi, g in enumerate(self.groups):
            for p in g.direct:
                if getattr(p, "_umpr_fresh", False):   # no backward node wrote it since zero_grad: its gradient is zero
                    p.grad.zero_()
                    p._umpr_fresh = False
            if not g.numel:
                continue
            ranges = [(0, g.numel)]
            if gi == 0 and done is not None:           # the classifier slice was updated during backward
                lo, hi, ev = done
                ranges = [(0, lo), (hi, g.numel)]
                torch.cuda.current_stream(g.p.device).wait_event(ev)
            for lo, hi in ranges:
                if hi > lo and getattr(self, "_hyper", None) is not None:
                    lib().call("umpr_adam_step_dev", g.p[lo:hi], g.g[lo:hi], g.m[lo:hi], g.v[lo:hi], hi - lo, self.betas[0],
                               self.betas[1], self.eps, self._hyper[gi], stream_ptr())
                elif hi > lo:
                    lib().call("umpr_adam_step", g.p[lo:hi], g.g[lo:hi], g.m[lo:hi], g.v[lo:hi], hi - lo, self.lr,
                               self.betas[0], self.betas[1], self.eps, g.weight_decay, self.step_count, grad_scale,
                               stream_ptr())

    def epoch_end(self):
        """ExponentialLR.step() (main.py:54)."""
        self.lr *= self.lr_decay

    def state_dict(self):
        """Adam moments per parameter NAME (independent of the arena order), step count and current learning rate."""
        st = {"step": self.step_count, "lr": float(self.lr), "base_lr": float(self.base_lr), "m": {}, "v": {}}
        for g in self.groups:
            for n in g.names:
                off, k = g.offsets[n]
                st["m"][n] = g.m[off:off + k].detach().cpu().clone()
                st["v"][n] = g.v[off:off + k].detach().cpu().clone()
        return st

    def load_state_dict(self, st):
        self.step_count = int(st["step"])
        self.lr = float(st["lr"])
        self.base_lr = float(st.get("base_lr", self.base_lr))
        for g in self.groups:
            for n in g.names:
                off, k = g.offsets[n]
                g.m[off:off + k].copy_(st["m"][n])
                g.v[off:off + k].copy_(st["v"][n])

    def reattach(self):
        """No-op hook: ``model.load_state_dict`` copies INTO the arena views, so parameters stay attached."""
        return None

    def grad_arenas(self):
        return [g.g for g in self.groups if g.numel]

    def early_bucket(self):
        """(arena, lo, hi, parameters) of the classifier-weight slice of the weight arena, or None.  The slice is
        complete once every one of `parameters` has had its gradient accumulated (one backward node returns all of
        them, and autograd does not promise an order among their AccumulateGrad nodes)."""
        g = self.groups[0]
        names = [n for n in g.names if _is_classifier(n)]
        if not names:
            return None
        lo = min(g.offsets[n][0] for n in names)
        hi = max(g.offsets[n][0] + g.offsets[n][1] for n in names)
        params = [p for n, p in zip(g.names, g.params) if n in set(names)]
        return (g.g, lo, hi, params)
