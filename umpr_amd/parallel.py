"""Data parallelism: one process per GPU, gradients summed with RCCL all-reduce over xGMI.

The reference's only multi-GPU mechanism is single-process ``nn.DataParallel`` (main.py:81-82): inputs are chunked
on dim 0, every replica runs the full model on its shard, replica losses are averaged (main.py:34) and gradients are
reduce-added to device 0 - i.e. gradient = mean over replicas of per-shard gradients.  The MI355X-native equivalent
keeps persistent replicas (no per-step parameter broadcast) and all-reduces the flat gradient arenas
(umpr_amd/optim.py); the 1/world scaling is folded into the Adam kernel's ``grad_scale``.

xGMI is a point-to-point mesh (7 links x ~153 GB/s per GPU): few, large collectives are what it wants, so the
554 MB weight-gradient arena is reduced in `n_buckets` chunks issued on a side stream so that they can overlap.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from .streams import wait_for_gradients


def _force():
    """UMPR_REDUCE_AT_WORLD1=1: run the whole exchange path (RCCL init, overlapped all-reduce) with a single rank -
    the rehearsal a one-GPU box allows."""
    return os.environ.get("UMPR_REDUCE_AT_WORLD1", "") == "1"


def active():
    return dist.is_initialized() and (dist.get_world_size() > 1 or _force())


def init_distributed(backend=None, timeout_s=None):
    """Read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run) and join the process group.
    `timeout_s`: finite rendezvous / collective timeout (default UMPR_DIST_TIMEOUT_S or 600 s) - a rank whose peer died
    must not sit in a collective for RCCL's default 10 minutes or the store's 30."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or _force()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        import datetime
        if timeout_s is None:
            timeout_s = float(os.environ.get("UMPR_DIST_TIMEOUT_S", "600"))
        dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=float(timeout_s)))
    return rank, local, world


def shard_bounds(B, rank, world):
    """[lo, hi) of rank's contiguous chunk of a batch of B along dim 0 - torch.chunk semantics, i.e. DataParallel's
    scatter (main.py:82): chunks of ceil(B/world); when B is small the trailing ranks get NOTHING (lo == hi) and the
    reference simply runs fewer replicas.  A rank with an empty shard still joins every collective (train_step,
    evaluate_mse) with zero gradients / (0, 0) sums; the gradient scale is 1 / active_shards(B, world)."""
    per = -(-B // world) if B > 0 else 0
    lo = min(B, rank * per)
    return lo, min(B, lo + per)


def active_shards(B, world):
    """Number of non-empty chunks torch.chunk(B, world) yields = replicas DataParallel would use for this batch."""
    if B <= 0:
        return 1
    per = -(-B // world)
    return -(-B // per)


def shard_batch(batch, rank, world):
    """Contiguous chunks along dim 0, as DataParallel's scatter does (torch.chunk semantics).  May be empty."""
    B = batch[0].shape[0]
    lo, hi = shard_bounds(B, rank, world)
    return tuple(t[lo:hi] if (t.dim() > 0 and t.shape[0] == B) else t for t in batch)


class Shard(tuple):
    """A rank's chunk of a collated batch plus `n_active`, the number of ranks whose chunk of that batch is non-empty
    (= the replicas the reference's DataParallel would have used; the gradient mean runs over those)."""
    n_active = 1


def shard_with_count(batch, rank, world):
    s = Shard(shard_batch(batch, rank, world))
    s.n_active = active_shards(batch[0].shape[0], world)
    return s


def allreduce_arenas(arenas, n_buckets=4):
    """Sum the flat gradient arenas over all ranks (in place).  Returns after the collectives are enqueued on the
    current stream (RCCL orders them with the following Adam kernel)."""
    if not active():
        return
    if arenas:
        wait_for_gradients(arenas[0].device)      # in-place gradients written from a side stream (umpr_amd/streams.py)
    for a in arenas:
        if a.numel() > (1 << 22) and n_buckets > 1:
            for chunk in torch.chunk(a, n_buckets):
                dist.all_reduce(chunk, op=dist.ReduceOp.SUM)
        else:
            dist.all_reduce(a, op=dist.ReduceOp.SUM)


def allreduce_scalars(values, device):
    t = torch.tensor(values, dtype=torch.float64, device=device)
    if active():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.tolist()


class GradReducer:
    """Gradient exchange for one training step, overlapped with backward.

    The weight-gradient arena starts with the VGG classifier slice (FusedAdam's ordering).  Post-accumulate hooks on
    the classifier weights - the first gradients autograd produces - count down; when the last of them has landed the
    all-reduce of that slice is launched asynchronously (RCCL runs on its own stream), so 494 MB of the 554 MB travel
    over xGMI while the convolutional backward is still computing; `finish()` reduces the remainder and waits.
    Sum only: the 1/world scale is folded into the Adam kernel."""

    def __init__(self, opt, n_buckets=4):
        self.opt = opt
        self.n_buckets = n_buckets
        self.handles = []
        self.comm = None               # exchange stream of the in-stream form (early slice + its Adam update)
        if active() and opt.groups[0].g.is_cuda and self._in_stream(opt.groups[0].g):
            self.comm = torch.cuda.Stream(opt.groups[0].g.device)   # created before the text / weight-gradient streams: 9.7 vs 10.2 ms
        self.comm_used = False         # a collective of this step was enqueued on the exchange stream
        self.early = opt.early_bucket() if hasattr(opt, "early_bucket") else None
        self.fired = False
        self.pending = 0
        self.hooks = []
        if active() and hasattr(opt, "early_step"):
            prev = getattr(opt, "reducer", None)
            if prev is not None and prev is not self:
                prev.close()       # one reducer per optimiser: a stale one must not fire collectives on this arena
            opt.reducer = self     # the optimiser's early classifier update must follow this reducer's all-reduce
        # per-VGG-block buckets: the conv weight gradients of block b are exchanged as soon as the feature backward has
        # enqueued that block (model._features_bwd_call), ordered behind the library's weight-gradient stream - 60 MB that used
        # to wait for the end of the whole backward
        self.block_slices = self._find_block_slices() if active() and os.environ.get("UMPR_BLOCK_BUCKETS", "1") != "0" else {}
        self.reduced = []          # (lo, hi) ranges of arena 0 already handed to an all-reduce in this step
        self._hook_param = None
        if self.block_slices:
            import weakref
            g0 = opt.groups[0]
            first = [p for n, p in zip(g0.names, g0.params) if n.endswith(".features.0.weight")]
            if first:
                # the hook lives ON this model's first conv weight (model._features_bwd_call looks it up there), held
                # weakly: it goes away with the parameter or with this reducer, and no id() can be reused for it
                first[0]._umpr_block_hook = weakref.WeakMethod(self._on_block)
                self._hook_param = weakref.ref(first[0])
        if self.early is not None and active():
            self.pending = len(self.early[3])
            self.hooks = [p.register_post_accumulate_grad_hook(self._landed) for p in self.early[3]]
            # gradients written in place (model.py::_grad_targets) pass no AccumulateGrad node: the module says so
            from .optim import add_callback
            for m in getattr(opt, "model", torch.nn.Module()).modules():
                if hasattr(m, "grad_callbacks"):
                    add_callback(m, self._written_in_place)

    def close(self):
        """Detach from the optimiser / model: hooks and callbacks removed, no collective can be fired from here again."""
        from .optim import remove_callback
        for h in self.hooks:
            h.remove()
        self.hooks = []
        for m in getattr(self.opt, "model", torch.nn.Module()).modules():
            if hasattr(m, "grad_callbacks"):
                remove_callback(m, self._written_in_place)
        p = self._hook_param() if getattr(self, "_hook_param", None) is not None else None
        if p is not None and hasattr(p, "_umpr_block_hook"):
            del p._umpr_block_hook
        self._hook_param = None
        if getattr(self.opt, "reducer", None) is self:
            self.opt.reducer = None
        self.early = None
        self.block_slices = {}

    _VGG_BLOCK_CONVS = {0: (0, 2), 1: (5, 7), 2: (10, 12, 14), 3: (17, 19, 21), 4: (24, 26, 28)}

    def _find_block_slices(self):
        g = self.opt.groups[0] if getattr(self.opt, "groups", None) else None
        if g is None:
            return {}
        out = {}
        for b, idxs in self._VGG_BLOCK_CONVS.items():
            names = [n for n in g.names if any(n.endswith(f".features.{i}.weight") for i in idxs)]
            if len(names) != len(idxs):
                return {}
            spans = sorted(g.offsets[n] for n in names)
            lo, hi = spans[0][0], spans[-1][0] + spans[-1][1]
            # contiguous in the arena up to the alignment padding between parameters (optim._ALIGN floats)?  else leave it to finish()
            if any(not (0 <= b0 - (a0 + ak) < 64) for (a0, ak), (b0, _) in zip(spans, spans[1:])):
                return {}
            out[b] = (lo, hi)
        return out

    def _in_stream(self, arena):
        """Collectives as SYNCHRONOUS ops on a stream of our choice (torch runs a sync NCCL op on the current stream; the
        host is not blocked) instead of async ops on the process group's internal stream: the early slice travels on one
        exchange stream that also runs its Adam update and, behind events of the library's weight-gradient stream, the block
        buckets; the remainder runs on the caller's stream.
        Why: every async collective costs three cross-stream event hops (current -> NCCL stream -> waiter) and one more
        hardware queue, and a 9.5 ms bf16 step is sensitive to that - world-1 rehearsal on one MI355X, where the collectives
        themselves are no-ops: 12.8 ms async, 9.7-10.2 ms in-stream, 9.5 ms without any exchange; a step with a SINGLE async
        collective already measured 12.1 ms, with none of the time in a kernel or copy.  The 35 ms fp32 step shows the
        opposite, smaller effect (35.1 none / 35.5-35.9 async / 37.0-37.5 in-stream: the text kernels get CU slots earlier
        and push the weight-gradient stream back), so the default follows the arithmetic mode: in-stream for bf16, async for
        fp32.  UMPR_COMM_ASYNC=1 / 0 forces one form."""
        if not (arena.is_cuda and dist.get_backend() == "nccl"):
            return False
        force = os.environ.get("UMPR_COMM_ASYNC", "")
        if force in ("0", "1"):
            return force == "0"
        return getattr(getattr(self.opt, "model", None), "compute_dtype", "fp32") == "bf16"

    def _wgrad_stream(self, arena):
        from ._lib import lib
        ws = lib().fn["umpr_vgg16_wgrad_stream"]() if arena.is_cuda else None
        return torch.cuda.ExternalStream(ws, device=arena.device) if ws else None

    def _on_block(self, block):
        if not active() or block not in self.block_slices:
            return
        import contextlib
        lo, hi = self.block_slices[block]
        arena = self.opt.groups[0].g
        ws = self._wgrad_stream(arena)
        if self._in_stream(arena):
            # On the exchange stream, behind an event of the weight-gradient stream - NOT on the weight-gradient stream itself:
            # NCCL runs the collectives of one communicator in issue order, so a bucket issued there would hold that stream
            # (and every later weight-gradient kernel) until the 494 MB early slice has finished travelling.
            if self.comm is None:
                self.comm = torch.cuda.Stream(arena.device)
            src = ws if ws is not None else torch.cuda.current_stream(arena.device)
            ev = torch.cuda.Event()
            ev.record(src)
            self.comm.wait_event(ev)
            with torch.cuda.stream(self.comm):
                dist.all_reduce(arena[lo:hi], op=dist.ReduceOp.SUM)
            self.comm_used = True
        else:
            ctx = torch.cuda.stream(ws) if ws is not None else contextlib.nullcontext()
            with ctx:   # the collective is ordered behind the stream the block's weight-gradient kernels were issued on
                self.handles.append(dist.all_reduce(arena[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
        self.reduced.append((lo, hi))

    def _landed(self, _param=None):
        self.pending -= 1
        if self.pending == 0:
            self._fire()

    def _written_in_place(self):
        if self.early is not None and active() and not self.fired:
            self._fire()

    def _fire(self):
        if self.fired:
            return
        arena, lo, hi, _ = self.early
        if self._in_stream(arena):
            if self.comm is None:
                self.comm = torch.cuda.Stream(arena.device)
            self.comm.wait_stream(torch.cuda.current_stream(arena.device))   # the slice was written on the backward's stream
            with torch.cuda.stream(self.comm):
                dist.all_reduce(arena[lo:hi], op=dist.ReduceOp.SUM)
            self.fired = True
            self.comm_used = True
            if hasattr(self.opt, "early_step"):
                self.opt.early_step((), stream=self.comm)   # Adam on the slice, same stream: in order behind its all-reduce
            return
        early = []
        for chunk in torch.chunk(arena[lo:hi], self.n_buckets):
            early.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, async_op=True))
        self.handles += early
        self.fired = True
        if hasattr(self.opt, "early_step"):
            self.opt.early_step(early)   # Adam on the slice, behind its all-reduce, underneath the conv backward

    def skip_backward(self):
        """This rank has an empty shard (fewer samples than ranks in the last batch): no backward runs here, so the
        gradients are set to zero and the early bucket is launched by hand - every rank must issue the same sequence of
        collectives."""
        for a in self.opt.grad_arenas():
            a.zero_()
        for g in self.opt.groups:
            for p in g.direct:
                p._umpr_fresh = False
        if self.early is not None and active() and self.hooks:
            self._fire()
        for b in (4, 3, 2, 1, 0):
            self._on_block(b)

    def finish(self):
        if not active():
            return
        arenas = self.opt.grad_arenas()
        if arenas:
            # the text path wrote its gradients in place from its own stream: the remainder's collectives (enqueued behind the
            # current stream in both forms) must not start before those kernels have run (umpr_amd/streams.py)
            wait_for_gradients(arenas[0].device)
        done = list(self.reduced)
        if self.fired:
            done.append((self.early[1], self.early[2]))
        if done:   # what is left of arena 0 (the early slice and the block buckets are under way) + the other arenas
            arena = self.early[0] if self.early is not None else self.opt.groups[0].g
            rest, pos = [], 0
            for lo, hi in sorted(done):
                if lo > pos:
                    rest.append(arena[pos:lo])
                pos = max(pos, hi)
            if pos < arena.numel():
                rest.append(arena[pos:])
            rest += [a for a in arenas if a.data_ptr() != arena.data_ptr()]
        else:
            rest = arenas
        in_stream = bool(arenas) and self._in_stream(arenas[0])
        for a in rest:
            if a.numel():
                if in_stream:
                    dist.all_reduce(a, op=dist.ReduceOp.SUM)          # on the caller's stream
                else:
                    self.handles.append(dist.all_reduce(a, op=dist.ReduceOp.SUM, async_op=True))
        if in_stream:   # the optimiser step that follows on the caller's stream needs every bucket
            main = torch.cuda.current_stream(arenas[0].device)
            if self.comm is not None and self.comm_used:
                main.wait_stream(self.comm)
                self.comm_used = False
        for h in self.handles:
            h.wait()
        self.handles = []
        self.reduced = []
        self.fired = False
        self.pending = len(self.early[3]) if self.hooks else 0
