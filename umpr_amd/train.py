"""Training / evaluation drivers mirroring main.py:16-61 and src/evaluate.py:6-14."""
from __future__ import annotations

import time

import torch

from . import parallel
from .checkpoint import load_checkpoint, save_checkpoint
from .optim import FusedAdam


def evaluate_mse(model, dataloader):
    """Sum of squared errors over all samples / N (src/evaluate.py:6-14); summed over ranks when distributed."""
    se, cnt = 0.0, 0
    with torch.no_grad():
        model.eval()
        for batch in dataloader:
            pred, _ = model(*batch)
            lab = batch[-1].to(pred.device)
            se += torch.nn.functional.mse_loss(pred, lab, reduction='sum').item()
            cnt += len(pred)
    dev = next(model.parameters()).device
    se, cnt = parallel.allreduce_scalars([se, cnt], dev)
    return se / max(cnt, 1)


def train_step(model, opt: FusedAdam, batch, world=1, reducer=None):
    """model.train(); pred, loss = model(*batch); loss.mean(); zero_grad; backward; step  (main.py:32-37).
    With world > 1 the gradients are summed over ranks (RCCL) - by `reducer` overlapped with backward if given."""
    model.train()
    pred, loss = model(*batch)
    loss = loss.mean()
    opt.zero_grad()
    loss.backward()
    if world > 1 or reducer is not None:
        if reducer is not None:
            reducer.finish()
        else:
            parallel.allreduce_arenas(opt.grad_arenas())
    opt.step(grad_scale=1.0 / world)
    return pred, loss


def training(train_dataloader, valid_dataloader, model, config, model_path, logger=None, world=1, rank=0):
    log = logger.info if logger else print
    valid_mse = evaluate_mse(model, valid_dataloader)
    log(f'Initial validation mse is {valid_mse:.6f}')
    start = time.perf_counter()
    opt = FusedAdam(model, config.learning_rate, config.l2_regularization, config.lr_decay)
    reducer = parallel.GradReducer(opt) if parallel.active() else None
    best_loss, batch_counter, first_epoch = 100, 0, 0
    resume = getattr(config, "resume", "")
    if resume:  # not in the reference (it keeps no optimiser / epoch state): exact resume from umpr_amd.checkpoint
        meta = load_checkpoint(resume, model, opt, map_location=next(model.parameters()).device)
        first_epoch, batch_counter = meta.get("epoch", 0), meta.get("batch_counter", 0)
        best_loss = meta.get("best_loss", best_loss)
        log(f'Resumed from {resume}: epoch {first_epoch}, batch {batch_counter}')
    for epoch in range(first_epoch, config.train_epochs):
        total_loss, total_samples = 0.0, 0
        t0 = time.perf_counter()
        for batch in train_dataloader:
            pred, loss = train_step(model, opt, batch, world, reducer)
            total_loss += loss.item() * len(pred)
            total_samples += len(pred)
            batch_counter += 1
            if batch_counter % 500 == 0:
                valid_mse = evaluate_mse(model, valid_dataloader)
                log(f'Epoch {epoch:2d}; batch {batch_counter:5d}; train loss {total_loss / total_samples:.6f}; '
                    f'valid mse {valid_mse:.6f}')
                if best_loss > valid_mse:
                    best_loss = valid_mse
                    if rank == 0:
                        save_checkpoint(model_path, model, opt, epoch, batch_counter, best_loss)
        opt.epoch_end()
        dt = time.perf_counter() - t0
        log(f'Epoch {epoch:3d} done; train loss {total_loss / max(total_samples, 1):.6f}; '
            f'{world * total_samples / max(dt, 1e-9):.1f} samples/s')
        if batch_counter > 50000:
            break
    sec = int(time.perf_counter() - start)
    log(f'End of training! Time used {sec // 3600}:{sec % 3600 // 60}:{sec % 60}.')
    return opt
