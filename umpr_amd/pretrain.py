"""R-Net pre-training on MI355X (reference pretrain/pretrain_rnet.py:144-205).

``PretrainRNet`` keeps the reference's constructor, ``forward(u, u_length, i, i_length, target) -> (result, loss)``
and state_dict keys (``embedding.weight``, ``r_net.M``, ``r_net.gru.module.*``, ``linear.0.{weight,bias}``); the
arithmetic is the same gfx950 kernels the UMPR model uses (embedding + BiGRU, co-attention) plus the BCE head
kernel, all through the C ABI.  ``save_r_net`` writes the ``r_net`` state_dict (the reference pickles the module
object; a state_dict loads into either implementation without executing anything).  ABAE, which only labels the
sentence pairs (pretrain/abae.py), is out of scope: ``pretrain_r_net`` takes ready (u, len, i, len, label) tensors.
"""
from __future__ import annotations

import torch
from torch import nn

from ._lib import lib, stream_ptr
from .model import D, RNet, UMPR, _EmbedGru, _c, _ws


class _PretrainHead(torch.autograd.Function):
    """co-attention (src/model.py:50-55, S = 1) -> cat -> Linear -> Sigmoid -> BCELoss (pretrain_rnet.py:164-168)."""

    @staticmethod
    def forward(ctx, gru_u, gru_i, M, w, b, target):
        B, SL, _ = gru_u.shape
        dev = gru_u.device
        f = dict(device=dev, dtype=torch.float32)
        gru_u, gru_i = _c(gru_u), _c(gru_i)
        T = torch.empty(B, SL, D, **f)
        soft_u, soft_i = torch.empty(B, SL, **f), torch.empty(B, SL, **f)
        colmax, rowmax = torch.empty(B, SL, **f), torch.empty(B, SL, **f)
        argcol = torch.empty(B, SL, device=dev, dtype=torch.int32)
        argrow = torch.empty(B, SL, device=dev, dtype=torch.int32)
        att = torch.empty(B, 2 * D, **f)
        st = stream_ptr()
        ws, wsb = _ws(lib().size("umpr_coattention_fwd_ws_bytes", B, SL), dev)
        lib().call("umpr_coattention_fwd", gru_u, gru_i, M, B, SL, T, soft_u, soft_i, att, 2 * D,
                   att.data_ptr() + D * 4, 2 * D, colmax, argcol, rowmax, argrow, ws, wsb, st)
        result = torch.empty(B, **f)
        loss = torch.empty((), **f)
        ws, wsb = _ws(B * 4, dev)
        lib().call("umpr_bce_head_fwd", att, 2 * D, _c(w), _c(b), target, B, 2 * D, result, loss, ws, wsb, st)
        ctx.save_for_backward(gru_u, gru_i, M, w, target, T, soft_u, soft_i, colmax, argcol, rowmax, argrow, att, result)
        return result, loss

    @staticmethod
    def backward(ctx, d_result, d_loss):
        gru_u, gru_i, M, w, target, T, soft_u, soft_i, colmax, argcol, rowmax, argrow, att, result = ctx.saved_tensors
        B, SL, _ = gru_u.shape
        dev = gru_u.device
        f = dict(device=dev, dtype=torch.float32)
        st = stream_ptr()
        d_att = torch.empty(B, 2 * D, **f)
        dw, db = torch.empty_like(w), torch.empty(1, **f)
        if d_loss is None:
            d_loss = torch.zeros((), **f)
        ws, wsb = _ws(B * 4, dev)
        lib().call("umpr_bce_head_bwd", att, 2 * D, _c(w), result, target, None if d_result is None else _c(d_result),
                   _c(d_loss), B, 2 * D, d_att, 2 * D, dw, db, ws, wsb, st)
        dGu, dGi = torch.empty(B, SL, D, **f), torch.empty(B, SL, D, **f)
        dM = torch.empty_like(M)
        ws, wsb = _ws(lib().size("umpr_coattention_bwd_ws_bytes", B, SL), dev)
        lib().call("umpr_coattention_bwd", gru_u, gru_i, M, T, soft_u, soft_i, colmax, argcol, rowmax, argrow,
                   d_att, 2 * D, d_att.data_ptr() + D * 4, 2 * D, None, None, B, SL, dGu, dGi, dM, 0, ws, wsb, st)
        return dGu, dGi, dM, dw, db, None


class PretrainRNet(nn.Module):
    def __init__(self, word2vec, gru_hidden):
        super().__init__()
        self.embedding = nn.Embedding.from_pretrained(torch.Tensor(word2vec.embedding))
        self.r_net = RNet(word2vec.word_dim, gru_hidden)
        self.linear = nn.Sequential(nn.Linear(gru_hidden * 4, 1), nn.Sigmoid())

    def forward(self, u, u_length, i, i_length, target):
        device = self.embedding.weight.device
        if device.type != "cuda":
            raise RuntimeError("umpr_amd.PretrainRNet runs on an MI355X only (no CPU fallback)")
        u, i = _c(u.to(device)), _c(i.to(device))
        target = _c(target.to(device).float())
        B, L = u.shape
        lu, ou = UMPR._host_perm(u_length.view(B, 1), device)
        li, oi = UMPR._host_perm(i_length.view(B, 1), device)
        emb = self.embedding.weight
        gru_u = self.r_net.gru(u, lu, ou, emb).view(B, L, D)
        gru_i = self.r_net.gru(i, li, oi, emb).view(B, L, D)
        lin = self.linear[0]
        return _PretrainHead.apply(gru_u, gru_i, self.r_net.M, lin.weight, lin.bias, target)

    def save_r_net(self, save_path):
        torch.save({k: v.detach().cpu() for k, v in self.r_net.state_dict().items()}, save_path)


def pretrain_r_net(word2vec, batches, save_r_net_path=None, *, gru_size=64, learning_rate=0.01, lr_decay=0.99,
                   l2_regularization=1e-3, train_epochs=10, device="cuda", log=print):
    """Training loop of pretrain_rnet.py:172-205.  ``batches``: a re-iterable of (u, u_length, i, i_length, target)."""
    from .optim import FusedAdam
    model = PretrainRNet(word2vec, gru_hidden=gru_size).to(device)
    opt = FusedAdam(model, learning_rate, l2_regularization, lr_decay=lr_decay)
    for epoch in range(train_epochs):
        model.train()
        total_loss, total_samples = 0.0, 0
        for batch in batches:
            result, loss = model(*batch)
            opt.zero_grad()
            loss.backward()
            opt.step()
            total_loss += loss.item() * len(result)
            total_samples += len(result)
        opt.epoch_end()  # ExponentialLR, stepped per epoch (pretrain_rnet.py:200)
        log(f"Epoch {epoch:3d}; train loss {total_loss / max(total_samples, 1):.6f}")
    if save_r_net_path:
        model.save_r_net(save_r_net_path)
    return model
