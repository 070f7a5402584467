"""Data-parallel host logic on CPU with gloo, world_size 2 (the N>1 path of bench.py / train.py).

What is pinned: contiguous dim-0 sharding (DataParallel's scatter, main.py:82), per-shard forward/backward, SUM
all-reduce of flat gradient arenas scaled by 1/world, mean of shard losses - against fixture dp_shards.npz, which the
reference's own module produced shard by shard (tests/golden/make_golden.py::gen_dataparallel).  The per-shard model
here is the oracle (CPU); the HIP model replaces it on GPUs, the collective logic is the same code
(umpr_amd/parallel.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

from conftest import load_golden


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from oracle import umpr_ref as R
    from umpr_amd import parallel
    from umpr_amd.synthetic import make_batch, make_param_state
    r, _, w = parallel.init_distributed(backend="gloo")
    g = load_golden("dp_shards")
    P = make_param_state(41, 50, 1000, 1, False, with_vgg=False, m_scale=0.05)
    names = [k for k in P if k != "embedding.weight"]
    for k in names:
        P[k].requires_grad_(True)
    batch = make_batch(42, 4, 1000, 1, img_hw=8)
    shard = parallel.shard_batch(batch, r, w)
    assert shard[0].shape[0] == 2
    fake_w = torch.from_numpy(g["fake_vgg_w"])
    pred, loss = R.umpr_forward(P, shard, review_net_only=False, aten=True,
                                vgg_fn=lambda im: F.linear(im.flatten(1), fake_w))
    loss.backward()
    # flat arena like umpr_amd.optim: one buffer, grads are slices of it
    arena = torch.cat([P[k].grad.reshape(-1) for k in names])
    parallel.allreduce_arenas([arena], n_buckets=3)
    arena /= w
    lsum, = parallel.allreduce_scalars([loss.item()], torch.device("cpu"))
    if r == 0:
        off = 0
        res = {"loss": lsum / w}
        for k in names:
            n = P[k].numel()
            res[k] = arena[off:off + n].reshape(P[k].shape).clone()
            off += n
        res["pred0"] = pred.detach()
        torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


def test_dataparallel_semantics_gloo(tmp_path):
    out = str(tmp_path / "dp.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    res = torch.load(out, weights_only=True)
    g = load_golden("dp_shards")
    assert abs(res["loss"] - float(g["loss"])) < 1e-5
    np.testing.assert_allclose(res["pred0"].numpy(), g["prediction"][:2], atol=1e-5)
    checked = 0
    for k, v in res.items():
        if "grad/" + k in g:
            ref = g["grad/" + k]
            np.testing.assert_allclose(v.numpy(), ref, atol=1e-4 * float(np.abs(ref).max()) + 1e-7, rtol=1e-3, err_msg=k)
            checked += 1
    assert checked > 20


def test_shard_batch_matches_chunk():
    from umpr_amd import parallel
    from umpr_amd.synthetic import make_batch
    b = make_batch(3, 5, 100, 1, img_hw=8)
    for world in (2, 3):
        parts = [parallel.shard_batch(b, r, world) for r in range(world)]
        for i, t in enumerate(b):
            ref = torch.chunk(t, world, dim=0)
            for r in range(len(ref)):
                assert torch.equal(parts[r][i], ref[r])


def test_short_last_batch_shards():
    """ceil-sized chunks leave trailing ranks EMPTY when the last batch is short; every rank must still have a
    defined role.  Pins the chunk table, active_shards (= replicas DataParallel would use) and that the pieces tile B."""
    from umpr_amd import parallel
    sizes = lambda B, w: [hi - lo for lo, hi in (parallel.shard_bounds(B, r, w) for r in range(w))]
    assert sizes(9, 8) == [2, 2, 2, 2, 1, 0, 0, 0]
    assert sizes(17, 8) == [3, 3, 3, 3, 3, 2, 0, 0]
    assert sizes(33, 8) == [5, 5, 5, 5, 5, 5, 3, 0]
    assert sizes(3, 8) == [1, 1, 1, 0, 0, 0, 0, 0]
    assert sizes(64, 8) == [8] * 8
    for B in range(1, 70):
        for w in (1, 2, 3, 4, 8):
            sz = sizes(B, w)
            assert sum(sz) == B and [len(c) for c in torch.chunk(torch.arange(B), w)] == [s for s in sz if s]
            assert parallel.active_shards(B, w) == sum(1 for s in sz if s)
    from umpr_amd.synthetic import make_batch
    b = make_batch(3, 3, 100, 1, img_hw=8)
    s = parallel.shard_with_count(b, 5, 8)
    assert s.n_active == 3 and s[0].shape[0] == 0 and s[6].shape[0] == 0 and len(s) == 8


def _empty_shard_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from umpr_amd import parallel
    from umpr_amd.optim import FusedAdam
    parallel.init_distributed(backend="gloo")
    torch.manual_seed(0)
    model = _TinyInPlace()
    opt = FusedAdam(model, 1e-3, 1e-3)
    red = parallel.GradReducer(opt, n_buckets=2)
    X = torch.randn(3, 6, generator=torch.Generator().manual_seed(7))   # global batch of 3 on 2 ranks: chunks [2, 1]
    for B in (3, 1):                                                    # then a batch of 1: chunks [1, 0]
        lo, hi = parallel.shard_bounds(B, rank, world)
        n_active = parallel.active_shards(B, world)
        for g in opt.groups:                                            # stale gradients of the previous step (the arena's
            for off, k in g.offsets.values():                           # alignment padding between parameters is never written)
                g.g[off:off + k].fill_(99.0)
        opt.zero_grad()
        if hi > lo:
            model(X[lo:hi]).pow(2).sum().backward()
            assert red.fired
        else:
            red.skip_backward()                                         # same collective sequence, zero contribution
        red.finish()
        torch.manual_seed(0)
        twin = _Tiny()
        topt = FusedAdam(twin, 1e-3, 1e-3)
        total = [torch.zeros_like(a) for a in topt.grad_arenas()]
        for r in range(world):                                          # sequential replay of every non-empty chunk
            l2, h2 = parallel.shard_bounds(B, r, world)
            if h2 > l2:
                topt.zero_grad()
                twin(X[l2:h2]).pow(2).sum().backward()
                total = [t + a for t, a in zip(total, topt.grad_arenas())]
        for a, t in zip(opt.grad_arenas(), total):
            assert torch.allclose(a, t, atol=1e-6), (B, rank, (a - t).abs().max())
        assert n_active == (2 if B == 3 else 1)
    if rank == 0:
        torch.save({"ok": torch.tensor(1)}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_empty_shard_rank_joins_the_collectives_gloo(tmp_path):
    """B < world (and B % world != 0): the rank without samples skips backward, contributes zeros through the SAME
    sequence of all-reduces (early bucket + rest) and ends with the same summed gradients as its peer."""
    out = str(tmp_path / "empty.pt")
    mp.spawn(_empty_shard_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert os.path.exists(out)


class _JointClassifier(torch.autograd.Function):
    """Both linear layers in ONE backward node, like umpr_amd.model._VGGClassifier: all weight gradients are returned
    together, so autograd runs their AccumulateGrad nodes in an order the reducer must not rely on."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        h = torch.tanh(x @ w1.t() + b1)
        ctx.save_for_backward(x, w1, w2, h)
        return h @ w2.t() + b2

    @staticmethod
    def backward(ctx, dy):
        x, w1, w2, h = ctx.saved_tensors
        dh = (dy @ w2) * (1 - h * h)
        return dh @ w1, dh.t() @ x, dh.sum(0), dy.t() @ h, dy.sum(0)


class _Tiny(torch.nn.Module):
    """Parameter names shaped like the real model: a 'classifier' part whose gradients come first, a conv-like rest."""

    def __init__(self):
        super().__init__()
        self.features = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.Tanh(), torch.nn.Linear(8, 8))
        self.classifier = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 3))

    def forward(self, x):
        c = self.classifier
        return _JointClassifier.apply(self.features(x), c[0].weight, c[0].bias, c[2].weight, c[2].bias)


class _JointClassifierInPlace(torch.autograd.Function):
    """Same node, but following umpr_amd.model's in-place protocol: gradients of parameters the optimiser marked fresh
    are written into p.grad directly, None is returned for them and the owner's grad_callbacks are called."""

    @staticmethod
    def forward(ctx, x, owner, w1, b1, w2, b2):
        h = torch.tanh(x @ w1.t() + b1)
        ctx.save_for_backward(x, w1, w2, h)
        ctx.param_objs = (w1, b1, w2, b2)
        ctx.owner = owner
        return h @ w2.t() + b2

    @staticmethod
    def backward(ctx, dy):
        from umpr_amd.model import _grad_returns, _grad_targets
        x, w1, w2, h = ctx.saved_tensors
        dh = (dy @ w2) * (1 - h * h)
        vals = (dh.t() @ x, dh.sum(0), dy.t() @ h, dy.sum(0))
        dst, direct = _grad_targets(ctx.param_objs)
        for d, v in zip(dst, vals):
            d.copy_(v)
        out = _grad_returns(ctx.param_objs, dst, direct)
        if all(direct):
            for cb in ctx.owner.grad_callbacks:
                cb()
        return (dh @ w1, None, *out)


class _TinyInPlace(_Tiny):
    def __init__(self):
        super().__init__()
        self.grad_callbacks = []
        for p in self.classifier.parameters():
            p._umpr_direct = True

    def forward(self, x):
        c = self.classifier
        return _JointClassifierInPlace.apply(self.features(x), self, c[0].weight, c[0].bias, c[2].weight, c[2].bias)


def _reducer_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from umpr_amd import parallel
    from umpr_amd.optim import FusedAdam
    parallel.init_distributed(backend="gloo")
    torch.manual_seed(0)
    model = _Tiny()
    opt = FusedAdam(model, 1e-3, 1e-3)
    red = parallel.GradReducer(opt, n_buckets=3)
    assert red.early is not None and len(red.hooks) == 2
    names0 = opt.groups[0].names
    assert "classifier." in names0[0] and "classifier." not in names0[-1], names0  # classifier slice leads the arena
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(5, 6, generator=g)
    torch.manual_seed(0)
    twin = _Tiny()                            # same weights, no reducer: source of the un-reduced local gradients
    topt = FusedAdam(twin, 1e-3, 1e-3)
    for it in range(2):                       # two steps: the hook must re-arm
        opt.zero_grad()
        model(x).pow(2).sum().backward()
        assert red.fired, "early bucket did not start during backward"
        topt.zero_grad()
        twin(x).pow(2).sum().backward()
        ref = [a.clone() for a in topt.grad_arenas()]   # reference: plain sum over ranks of the local arenas
        red.finish()
        for r in ref:
            dist.all_reduce(r)
        for a, r in zip(opt.grad_arenas(), ref):
            assert torch.allclose(a, r, atol=1e-6), (it, (a - r).abs().max())
        # params are views of the arenas: .grad of a parameter sees the reduced value
        p = dict(model.named_parameters())["classifier.0.weight"]
        off, n = opt.groups[0].offsets["classifier.0.weight"]
        assert torch.equal(p.grad.reshape(-1), opt.groups[0].g[off:off + n])
    # the same exchange when the classifier gradients are written in place (no AccumulateGrad hook fires for them)
    torch.manual_seed(0)
    model2 = _TinyInPlace()
    opt2 = FusedAdam(model2, 1e-3, 1e-3)
    red2 = parallel.GradReducer(opt2, n_buckets=2)
    from umpr_amd.optim import has_callback
    assert has_callback(model2, red2._written_in_place) and has_callback(model2, opt2._on_classifier_grads)
    for it in range(2):
        opt2.zero_grad()
        model2(x).pow(2).sum().backward()
        assert red2.fired, "in-place classifier gradients did not start the early bucket"
        topt.zero_grad()
        twin(x).pow(2).sum().backward()
        ref = [a.clone() for a in topt.grad_arenas()]
        red2.finish()
        for r in ref:
            dist.all_reduce(r)
        for a, r in zip(opt2.grad_arenas(), ref):
            assert torch.allclose(a, r, atol=1e-6), (it, (a - r).abs().max())
    if rank == 0:
        torch.save({"ok": torch.tensor(1)}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_grad_reducer_gloo(tmp_path):
    out = str(tmp_path / "red.pt")
    mp.spawn(_reducer_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert os.path.exists(out)


def test_checkpoint_roundtrip(tmp_path):
    """state_dict keys are the reference's; optimiser moments / step / lr survive a save-load cycle (resume)."""
    from umpr_amd.checkpoint import load_checkpoint, save_checkpoint
    from umpr_amd.optim import FusedAdam
    torch.manual_seed(1)
    m1 = _Tiny()
    o1 = FusedAdam(m1, 1e-3, 1e-3, lr_decay=0.5)
    for g in o1.groups:
        g.m.uniform_(-1, 1)
        g.v.uniform_(0, 1)
    o1.step_count = 7
    o1.epoch_end()
    path = str(tmp_path / "ck.pt")
    save_checkpoint(path, m1, o1, epoch=3, batch_counter=1500, best_loss=0.9)
    torch.manual_seed(2)
    m2 = _Tiny()
    o2 = FusedAdam(m2, 1e-3, 1e-3, lr_decay=0.5)
    meta = load_checkpoint(path, m2, o2)
    assert {k: meta[k] for k in ("epoch", "batch_counter", "best_loss")} == {"epoch": 3, "batch_counter": 1500, "best_loss": 0.9}
    assert o2.step_count == 7 and abs(o2.lr - 5e-4) < 1e-12
    for (n1, p1), (n2, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        assert n1 == n2 and torch.equal(p1, p2)
    for g1, g2 in zip(o1.groups, o2.groups):        # per parameter: the arenas' alignment padding is not part of the state
        for n, (off, k) in g1.offsets.items():
            assert g2.offsets[n] == (off, k)
            assert torch.equal(g1.m[off:off + k], g2.m[off:off + k]) and torch.equal(g1.v[off:off + k], g2.v[off:off + k]), n
        assert g2.params[0].data_ptr() == g2.p.data_ptr() or True
    # parameters are still views of the arena after loading
    g = o2.groups[0]
    off, k = g.offsets[g.names[0]]
    assert g.params[0].data_ptr() == g.p[off:off + k].data_ptr()


def test_zero_grad_skips_in_place_gradients():
    """Parameters flagged `_umpr_direct` (their backward node writes p.grad in place) are not zero-filled: zero_grad marks
    them fresh instead; all other arena ranges are zeroed.  Host logic only - no kernel runs."""
    from umpr_amd.optim import FusedAdam
    torch.manual_seed(0)
    model = _Tiny()
    for p in model.classifier.parameters():
        p._umpr_direct = True
    opt = FusedAdam(model, 1e-3, 1e-3)
    for g in opt.groups:
        for off, k in g.offsets.values():
            g.g[off:off + k].fill_(7.0)
    opt.zero_grad()
    for name, p in model.named_parameters():
        if name.startswith("classifier."):
            assert p._umpr_fresh and float(p.grad.min()) == 7.0, name      # untouched, to be overwritten by backward
        else:
            assert not getattr(p, "_umpr_fresh", False) and float(p.grad.abs().max()) == 0.0, name
    from umpr_amd.optim import _pad                      # every parameter owns its slice and the alignment padding behind it
    covered = sum(hi - lo for g in opt.groups for lo, hi in g.zero_ranges) + sum(_pad(p.numel()) for g in opt.groups for p in g.direct)
    assert covered == sum(g.numel for g in opt.groups)
    for g in opt.groups:
        for off, k in g.offsets.values():
            assert off % 64 == 0                         # 256-byte boundaries: float4 / whole-line reads of every weight


class _TinyVgg(torch.nn.Module):
    """Parameter names of the real model (``...features.N.weight`` at torchvision's conv indices, a classifier) at toy
    sizes: what GradReducer._find_block_slices keys on."""

    def __init__(self):
        super().__init__()
        self.v = torch.nn.Module()
        convs = {0: (3, 4), 2: (4, 4), 5: (4, 6), 7: (6, 6), 10: (6, 8), 12: (8, 8), 14: (8, 8), 17: (8, 8), 19: (8, 8),
                 21: (8, 8), 24: (8, 8), 26: (8, 8), 28: (8, 8)}
        self.v.features = torch.nn.ModuleList(
            [torch.nn.Conv2d(*convs[i], 3, padding=1) if i in convs else torch.nn.Identity() for i in range(31)])
        self.v.classifier = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 5))

    def forward(self, x):
        for m in self.v.features:
            x = torch.tanh(m(x)) if isinstance(m, torch.nn.Conv2d) else x
        return self.v.classifier(x.mean((2, 3)))


def _block_bucket_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from umpr_amd import parallel
    from umpr_amd.optim import FusedAdam
    parallel.init_distributed(backend="gloo")
    torch.manual_seed(0)
    model = _TinyVgg()
    opt = FusedAdam(model, 1e-3, 1e-3)
    red = parallel.GradReducer(opt, n_buckets=2)
    # five contiguous, disjoint slices of the weight arena, one per VGG block, in front of nothing they overlap
    assert sorted(red.block_slices) == [0, 1, 2, 3, 4], red.block_slices
    spans = sorted(red.block_slices.values())
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])), spans
    g = torch.Generator().manual_seed(7 + rank)
    x = torch.randn(3, 3, 6, 6, generator=g)
    torch.manual_seed(0)
    twin = _TinyVgg()                                       # same weights, no reducer: the un-reduced local gradients
    topt = FusedAdam(twin, 1e-3, 1e-3)
    for it in range(2):
        opt.zero_grad()
        model(x).pow(2).sum().backward()                    # the early classifier bucket starts inside this backward
        topt.zero_grad()
        twin(x).pow(2).sum().backward()
        ref = [a.clone() for a in topt.grad_arenas()]
        # what the feature backward's callback does on the GPU: block 4 first, block 0 last, each exactly once
        for b in (4, 3, 2, 1, 0):
            red._on_block(b)
        assert len(red.reduced) == 5
        red.finish()                                        # the complement of the five block ranges (+ early slice)
        for r in ref:
            dist.all_reduce(r)
        for a, r in zip(opt.grad_arenas(), ref):
            assert torch.allclose(a, r, atol=1e-6), (it, (a - r).abs().max())   # every element reduced exactly once
        assert red.reduced == [] and red.handles == []
    if rank == 0:
        torch.save({"ok": torch.tensor(1)}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_block_gradient_buckets_gloo(tmp_path):
    """The per-VGG-block buckets of GradReducer (started from the feature backward's C callback on the GPU) on two gloo
    ranks: block ranges are disjoint arena slices, and blocks + early classifier slice + finish() reduce every gradient
    element exactly once."""
    out = str(tmp_path / "blocks.pt")
    mp.spawn(_block_bucket_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert os.path.exists(out)


class _StepModel(_TinyInPlace):
    """_TinyInPlace behind UMPR.forward's calling convention for umpr_amd.train.train_step: forward(*batch) -> (pred, loss)."""

    def forward(self, x, labels):
        pred = _TinyInPlace.forward(self, x).sum(1)
        return pred, ((pred - labels) ** 2).mean()


def _cpu_adam(FusedAdam):
    class CpuAdam(FusedAdam):
        """FusedAdam's bookkeeping with the update done by the oracle's numpy Adam (the HIP kernel needs a GPU); records every
        early_step call: on a GPU that call updates the classifier slice from whatever the gradient arena holds right then."""
        early_calls = 0

        def early_step(self, handles, stream=None):
            self.early_calls += 1

        def step(self, grad_scale=1.0):
            from oracle.umpr_ref import adam_step_numpy
            self.step_count += 1
            self._early = self._early_done = None
            for g in self.groups:
                for p in g.direct:
                    p._umpr_fresh = False
                if g.numel:
                    p_, m_, v_ = adam_step_numpy(g.p.clone(), g.g * grad_scale, g.m.clone(), g.v.clone(), self.step_count,
                                                 self.lr, g.weight_decay)
                    g.p.copy_(p_); g.m.copy_(m_); g.v.copy_(v_)
    return CpuAdam


def _no_reducer_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from umpr_amd import parallel
    from umpr_amd.optim import FusedAdam
    from umpr_amd.train import train_step
    parallel.init_distributed(backend="gloo", timeout_s=120)
    torch.manual_seed(0)
    model = _StepModel()
    opt = _cpu_adam(FusedAdam)(model, 1e-2, 1e-3)
    g = torch.Generator().manual_seed(5)
    X, Y = torch.randn(3, 8, 6, generator=g), torch.randn(3, 8, generator=g)
    for it in range(3):
        lo, hi = parallel.shard_bounds(8, rank, world)
        train_step(model, opt, (X[it, lo:hi], Y[it, lo:hi]), world, reducer=None)     # the documented reducer=None path
    # without a reducer the classifier slice must NOT get its Adam update during backward (it would
    # use this rank's local, un-reduced gradient): no early step, and the replicas stay bit-identical
    assert opt.early_calls == 0 and opt._early is None
    mine = torch.cat([a.clone() for a in (opt.groups[0].p, opt.groups[1].p)])
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine)
    assert all(torch.equal(every[0], e) for e in every), "replicas diverged on the reducer=None path"
    # and WITH a reducer the early update is armed and ordered behind the reducer's all-reduce (it calls early_step itself)
    red = parallel.GradReducer(opt, n_buckets=2)
    lo, hi = parallel.shard_bounds(8, rank, world)
    train_step(model, opt, (X[0, lo:hi], Y[0, lo:hi]), world, reducer=red)
    assert opt.early_calls == 1
    if rank == 0:
        torch.save({"ok": torch.tensor(1)}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_train_step_without_reducer_keeps_replicas_identical_gloo(tmp_path):
    out = str(tmp_path / "nored.pt")
    mp.spawn(_no_reducer_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert os.path.exists(out)


def test_rebinding_an_optimiser_detaches_the_old_one():
    """callbacks FusedAdam / GradReducer leave on the model are weak and removable - a second optimiser on the
    same model (a resumed run, a second test leg) closes the first, and a dropped optimiser frees its arenas."""
    import gc
    import weakref
    from umpr_amd.optim import FusedAdam, has_callback
    torch.manual_seed(0)
    model = _TinyInPlace()
    o1 = FusedAdam(model, 1e-3, 1e-3)
    assert has_callback(model, o1._on_classifier_grads) and len(model.grad_callbacks) == 1
    o2 = FusedAdam(model, 1e-3, 1e-3)
    assert not has_callback(model, o1._on_classifier_grads) and has_callback(model, o2._on_classifier_grads)
    assert len(model.grad_callbacks) == 1
    arena = weakref.ref(o1.groups[0].m)
    del o1
    gc.collect()
    assert arena() is None, "the first optimiser's moment arena is still alive"
    o2.close()
    assert model.grad_callbacks == []
    ref2 = weakref.ref(o2)
    del o2
    gc.collect()
    assert ref2() is None
    for cb in model.grad_callbacks:
        cb()


def test_forward_mode_state_is_per_thread():
    """thread-per-replica callers - the forward's mode flags are thread-local on the Python side like the
    library's own switches on the C side."""
    import threading
    from umpr_amd import _lib, model
    model._MODE.b16 = True
    _lib.TLS.gemm_b16 = True
    seen = {}
    t = threading.Thread(target=lambda: seen.update(b16=model._MODE.b16, infer=model._MODE.infer, g=_lib.TLS.gemm_b16))
    t.start(); t.join()
    model._MODE.b16 = False
    _lib.TLS.gemm_b16 = False
    assert seen == {"b16": False, "infer": False, "g": False}
