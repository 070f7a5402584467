"""Training / evaluation drivers mirroring main.py:16-61 and src/evaluate.py:6-14."""
from __future__ import annotations

import time

import torch

from . import parallel
from .checkpoint import load_checkpoint, save_checkpoint
from .optim import FusedAdam


def evaluate_mse(model, dataloader):
    """Sum of squared errors over all samples / N (src/evaluate.py:6-14); summed over ranks when distributed.
    The per-batch sums are accumulated ON THE DEVICE by a library kernel (float64 accumulator, `umpr_sq_err_accumulate`)
    and read back once at the end: the reference's `.item()` per batch is a host sync per batch."""
    from ._lib import lib, stream_ptr
    acc = None
    with torch.no_grad():
        model.eval()
        for batch in dataloader:
            if batch[0].shape[0] == 0:   # this rank's chunk of a short last batch is empty (parallel.shard_bounds)
                continue
            pred, _ = model(*batch)
            lab = batch[-1].to(pred.device, non_blocking=True).float().contiguous()
            if acc is None:
                acc = torch.zeros(2, dtype=torch.float64, device=pred.device)
            lib().call("umpr_sq_err_accumulate", pred.contiguous(), lab, pred.numel(), acc, stream_ptr())
    dev = next(model.parameters()).device
    se, cnt = (acc.tolist() if acc is not None else (0.0, 0.0))
    se, cnt = parallel.allreduce_scalars([se, cnt], dev)
    return se / max(cnt, 1)


_ONES = {}


def _one_like(loss):
    key = (loss.device, loss.dtype)
    t = _ONES.get(key)
    if t is None:
        t = _ONES[key] = torch.ones((), device=loss.device, dtype=loss.dtype)
    return t


def train_step(model, opt: FusedAdam, batch, world=1, reducer=None):
    """model.train(); pred, loss = model(*batch); loss.mean(); zero_grad; backward; step  (main.py:32-37).
    With world > 1 the gradients are summed over ranks (RCCL) - by `reducer` overlapped with backward if given."""
    model.train()
    n_active = getattr(batch, "n_active", world)     # ranks with a non-empty chunk of this batch (parallel.Shard)
    if batch[0].shape[0] == 0:
        # Fewer samples than ranks in a short last batch: DataParallel would run fewer replicas (main.py:82).  This rank
        # contributes zero gradients but joins every collective, then applies the same averaged update as its peers.
        dev = next(model.parameters()).device
        opt.zero_grad()
        if reducer is not None:
            reducer.skip_backward()
            reducer.finish()
        else:
            for a in opt.grad_arenas():
                a.zero_()
            for g in opt.groups:
                for p in g.direct:
                    p._umpr_fresh = False
            parallel.allreduce_arenas(opt.grad_arenas())
        opt.step(grad_scale=1.0 / n_active)
        return torch.empty(0, device=dev), torch.zeros((), device=dev)
    pred, loss = model(*batch)
    if loss.dim() > 0:               # main.py:34 takes the mean over DataParallel's per-replica losses; one process per
        loss = loss.mean()           # GPU returns a scalar, whose mean is itself (and costs five tiny kernels)
    opt.zero_grad()
    if world == 1 or reducer is not None:
        # the classifier slice may be updated as soon as its gradients are final: at once on one GPU, behind the reducer's
        # all-reduce of that slice when data parallel.  Without a reducer the exchange only happens after backward
        # (allreduce_arenas below), so the early update stays off (it would use the local gradient).
        opt.arm_early(1.0 / n_active)
    loss.backward(gradient=_one_like(loss))     # a cached device scalar: autograd's implicit ones_like is a fill kernel per step
    if world > 1 or reducer is not None:
        if reducer is not None:
            reducer.finish()
        else:
            parallel.allreduce_arenas(opt.grad_arenas())
    opt.step(grad_scale=1.0 / n_active)
    return pred, loss


def _shuffle_generator(loader):
    """The torch.Generator that orders a (possibly wrapped) DataLoader's epochs, or None."""
    seen = 0
    while loader is not None and seen < 4:
        g = getattr(loader, "generator", None)
        if isinstance(g, torch.Generator):
            return g
        loader = getattr(loader, "loader", None)
        seen += 1
    return None


def training(train_dataloader, valid_dataloader, model, config, model_path, logger=None, world=1, rank=0):
    """main.py:16-61.  Returns (optimiser, saved): `saved` says whether THIS run wrote `model_path`.
    Not in the reference: `config.resume` continues a run exactly - parameters, Adam moments, learning rate, the
    epoch's shuffle order, the position inside the epoch and the dropout counters all come from the checkpoint, so the
    resumed run visits the batches the uninterrupted run would have visited next."""
    log = logger.info if logger else print
    valid_mse = evaluate_mse(model, valid_dataloader)
    log(f'Initial validation mse is {valid_mse:.6f}')
    start = time.perf_counter()
    opt = FusedAdam(model, config.learning_rate, config.l2_regularization, config.lr_decay)
    reducer = parallel.GradReducer(opt) if parallel.active() else None
    best_loss, batch_counter, first_epoch, skip, saved = 100, 0, 0, 0, False
    valid_every = int(getattr(config, "valid_every", 500))   # the reference hard-codes 500 (main.py:43)
    gen = _shuffle_generator(train_dataloader)
    resume = getattr(config, "resume", "")
    if resume:
        meta = load_checkpoint(resume, model, opt, map_location=next(model.parameters()).device)
        first_epoch, batch_counter = meta.get("epoch", 0), meta.get("batch_counter", 0)
        best_loss = meta.get("best_loss", best_loss)
        skip = meta.get("batch_in_epoch", 0)
        if gen is not None and "epoch_rng" in meta:
            gen.set_state(meta["epoch_rng"].cpu())
        log(f'Resumed from {resume}: epoch {first_epoch}, batch {batch_counter} ({skip} batches into the epoch)')
    for epoch in range(first_epoch, config.train_epochs):
        total_loss, total_samples = 0.0, 0
        t0 = time.perf_counter()
        epoch_rng = gen.get_state() if gen is not None else None
        for bi, batch in enumerate(train_dataloader):
            if epoch == first_epoch and bi < skip:   # consumed before the checkpoint was taken
                continue
            pred, loss = train_step(model, opt, batch, world, reducer)
            total_loss += loss.item() * len(pred)
            total_samples += len(pred)
            batch_counter += 1
            if batch_counter % valid_every == 0:
                valid_mse = evaluate_mse(model, valid_dataloader)
                log(f'Epoch {epoch:2d}; batch {batch_counter:5d}; train loss {total_loss / max(total_samples, 1):.6f}; '
                    f'valid mse {valid_mse:.6f}')
                if best_loss > valid_mse:        # valid_mse is all-reduced: every rank takes the same branch
                    best_loss = valid_mse
                    if rank == 0:
                        save_checkpoint(model_path, model, opt, epoch, batch_counter, best_loss, batch_in_epoch=bi + 1,
                                        epoch_rng=epoch_rng)
                    saved = True
                    if parallel.active():        # nobody runs ahead (and later loads) while rank 0 is still writing
                        torch.distributed.barrier()
        opt.epoch_end()
        dt = time.perf_counter() - t0
        log(f'Epoch {epoch:3d} done; train loss {total_loss / max(total_samples, 1):.6f}; '
            f'{world * total_samples / max(dt, 1e-9):.1f} samples/s')
        if batch_counter > 50000:
            break
    sec = int(time.perf_counter() - start)
    log(f'End of training! Time used {sec // 3600}:{sec % 3600 // 60}:{sec % 60}.')
    return opt, saved
