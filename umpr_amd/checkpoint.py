"""Checkpoint interchange and resume.

The reference only ever pickles the whole module when validation improves (main.py:43-52) and keeps no optimiser /
scheduler / epoch state.  Here a checkpoint is a plain dict of tensors and numbers (loadable with
``torch.load(..., weights_only=True)``): the model ``state_dict`` under the reference's key names - so the weights
move between the two implementations by ``state_dict`` - plus, optionally, the Adam moments, step count and learning
rate for an exact resume.
"""
from __future__ import annotations

import os

import torch

_META = ("epoch", "batch_counter", "best_loss", "batch_in_epoch", "epoch_rng", "rng_calls")


def save_checkpoint(path, model, opt=None, epoch=0, batch_counter=0, best_loss=None, batch_in_epoch=None,
                    epoch_rng=None):
    """`batch_in_epoch` batches of epoch `epoch` were consumed when the checkpoint was taken and `epoch_rng` is the state
    the loader's shuffle generator had when that epoch began: together they let a resumed run replay the epoch's order
    and skip what was already trained on.  The file appears atomically (written beside the target, then renamed)."""
    ck = {"model": {k: v.detach().cpu() for k, v in model.state_dict().items()},
          "epoch": int(epoch), "batch_counter": int(batch_counter)}
    if best_loss is not None:
        ck["best_loss"] = float(best_loss)
    if batch_in_epoch is not None:
        ck["batch_in_epoch"] = int(batch_in_epoch)
    if epoch_rng is not None:
        ck["epoch_rng"] = epoch_rng.clone()
    # forward-call counters that seed the dropout masks (model.VGG16._calls): part of an exact resume
    ck["rng_calls"] = {n: int(m._calls) for n, m in model.named_modules() if hasattr(m, "_calls")}
    if opt is not None:
        ck["optimizer"] = opt.state_dict()
    tmp = f"{path}.tmp{os.getpid()}"
    torch.save(ck, tmp)
    os.replace(tmp, path)


def load_checkpoint(path, model, opt=None, map_location="cpu"):
    """Returns the metadata dict (epoch, batch_counter, best_loss).  Accepts a checkpoint of save_checkpoint or a bare
    state_dict file (e.g. ``torch.save(reference_model.state_dict(), path)`` written on the reference side)."""
    ck = torch.load(path, map_location=map_location, weights_only=True)
    sd = ck["model"] if isinstance(ck, dict) and "model" in ck and isinstance(ck["model"], dict) else ck
    model.load_state_dict(sd)
    if opt is not None:
        if isinstance(ck, dict) and "optimizer" in ck:
            opt.load_state_dict(ck["optimizer"])
        else:
            opt.reattach()  # parameters were copied into place: nothing to restore, keep fresh moments
    meta = {k: ck[k] for k in _META if isinstance(ck, dict) and k in ck}
    for n, m in model.named_modules():
        if hasattr(m, "_calls") and n in meta.get("rng_calls", {}):
            m._calls = int(meta["rng_calls"][n])
    return meta
