"""ORACLE - TEST INFRASTRUCTURE ONLY.  Never imported by the product path (umpr_amd/).

CPU restatement (PyTorch CPU ops, fp32) of the reference's UMPR hot path, written from the
reference's source as a *functional* program over a flat parameter dict whose keys are the
reference's state_dict names.  Only tests/, __graft_entry__.smoke() and bench.py's
``cpu_baseline`` leg may import this file.

Parity status: the reference has no tests or golden vectors of its own, so
this restatement is pinned against outputs of the reference itself, generated in the build
container by tests/golden/make_golden.py (which imports /root/reference/src/model.py) and
committed under tests/golden/.  The VGG16 arithmetic lives in torchvision, which is absent from
this image: its restatement (``vgg16_forward``) follows the published VGG16-D layer list with
torchvision's layer indices and is "parity unpinned" against torchvision itself.

Every function cites the reference lines it restates (paths relative to /root/reference).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

VGG16_CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M")
# torchvision's features.<idx> numbering of the 13 conv layers of configuration "D"
VGG16_CONV_IDX = (0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28)
VGG16_FC_IDX = (0, 3, 6)


# ----------------------------------------------------------------------------------------------
# ImprovedRnn  (src/model.py:6-21)
# ----------------------------------------------------------------------------------------------
def gru_sort_indices(lengths_cpu: Tensor):
    """The (sorted_indices, unsorted_indices) pair pack_padded_sequence(enforce_sorted=False)
    builds (src/model.py:18; torch/nn/utils/rnn.py: ``torch.sort(lengths, descending=True)`` on the
    CPU lengths - NOT a stable sort, ties keep whatever order that call produces)."""
    lengths_cpu = lengths_cpu.cpu()
    _, sorted_indices = torch.sort(lengths_cpu, descending=True)
    unsorted = torch.empty_like(sorted_indices)
    unsorted[sorted_indices] = torch.arange(sorted_indices.numel(), dtype=sorted_indices.dtype)
    return sorted_indices, unsorted


def gru_cell_seq(x: Tensor, lengths: Tensor, w_ih, w_hh, b_ih, b_hh, reverse: bool) -> Tensor:
    """One direction of nn.GRU over zero-padded variable-length rows, explicit time loop.
    Gate order (r, z, n); n = tanh(W_in x + b_in + r*(W_hn h + b_hn)); h' = (1-z)*n + z*h; h0 = 0.
    Steps t >= len emit zeros and leave h untouched (what pack/pad does, src/model.py:18-20)."""
    N, L, _ = x.shape
    H = w_hh.shape[1]
    h = x.new_zeros(N, H)
    out = []
    steps = range(L - 1, -1, -1) if reverse else range(L)
    gx_all = x @ w_ih.t() + b_ih  # [N, L, 3H]
    for t in steps:
        active = (lengths > t).unsqueeze(1)
        gx = gx_all[:, t]
        gh = h @ w_hh.t() + b_hh
        r = torch.sigmoid(gx[:, :H] + gh[:, :H])
        z = torch.sigmoid(gx[:, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gx[:, 2 * H:] + r * gh[:, 2 * H:])
        h_new = (1 - z) * n + z * h
        h = torch.where(active, h_new, h)
        out.append((t, torch.where(active, h_new, torch.zeros_like(h_new))))
    out.sort(key=lambda p: p[0])
    return torch.stack([o for _, o in out], dim=1)


def improved_rnn(x: Tensor, lengths: Tensor, P: Dict[str, Tensor], prefix: str, aten: bool = False) -> Tensor:
    """ImprovedRnn.forward, src/model.py:12-21, including the EXTRA ``result[unsorted_indices]``
    gather after pad_packed_sequence: out[n] = BiGRU(x[unsorted[n]]) (SURVEY.md header fact 1).

    aten=False: explicit loop (independent statement of the semantics).
    aten=True : same ATen calls as the reference (pack -> _VF.gru -> pad); used for CPU timing."""
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]
    fw = [P[prefix + n] for n in names]
    bw = [P[prefix + n + "_reverse"] for n in names]
    lengths_cpu = lengths.cpu()
    if aten:
        packed = torch.nn.utils.rnn.pack_padded_sequence(x, lengths_cpu, batch_first=True, enforce_sorted=False)
        H = fw[1].shape[1]
        hx = x.new_zeros(2, x.shape[0], H)
        data, _ = torch._VF.gru(packed.data, packed.batch_sizes, hx, fw + bw, True, 1, 0.0, False, True)
        res = torch.nn.utils.rnn.PackedSequence(data, packed.batch_sizes, packed.sorted_indices,
                                                packed.unsorted_indices)
        res, _ = torch.nn.utils.rnn.pad_packed_sequence(res, batch_first=True, total_length=x.shape[1])
        return res[packed.unsorted_indices]
    _, unsorted = gru_sort_indices(lengths_cpu)
    lens = lengths.to(x.device)
    out = torch.cat([gru_cell_seq(x, lens, *fw, reverse=False), gru_cell_seq(x, lens, *bw, reverse=True)], dim=-1)
    return out[unsorted.to(x.device)]


# ----------------------------------------------------------------------------------------------
# RNet (src/model.py:36-56)
# ----------------------------------------------------------------------------------------------
def r_net(user_emb, item_emb, u_lengths, i_lengths, P, prefix="review_net.r_net.", aten=False):
    B, S, L, E = user_emb.shape
    gru_u = improved_rnn(user_emb.reshape(B * S, L, E), u_lengths.reshape(-1), P, prefix + "gru.module.", aten)
    gru_i = improved_rnn(item_emb.reshape(B * S, L, E), i_lengths.reshape(-1), P, prefix + "gru.module.", aten)
    gru_u = gru_u.reshape(B, S * L, -1)
    gru_i = gru_i.reshape(B, S * L, -1)
    A = torch.tanh(gru_i @ P[prefix + "M"] @ gru_u.transpose(-1, -2))          # model.py:50-51
    soft_u = torch.softmax(torch.max(A, dim=-2).values, dim=-1)                 # model.py:52
    soft_i = torch.softmax(torch.max(A, dim=-1).values, dim=-1)                 # model.py:53
    atte_u = (gru_u.transpose(-1, -2) @ soft_u.unsqueeze(-1)).squeeze(-1)       # model.py:54
    atte_i = (gru_i.transpose(-1, -2) @ soft_i.unsqueeze(-1)).squeeze(-1)       # model.py:55
    return gru_u, gru_i, soft_u, soft_i, atte_u, atte_i


# ----------------------------------------------------------------------------------------------
# SNet (src/model.py:71-81)
# ----------------------------------------------------------------------------------------------
def s_net(gru_repr, word_soft, sent_length, P, prefix):
    B = gru_repr.shape[0]
    S = gru_repr.shape[1] // sent_length
    X = gru_repr.reshape(B * S, sent_length, -1).transpose(-1, -2)             # [BS, 2u, L]
    sent_soft = torch.softmax(P[prefix + "Ws"] @ torch.tanh(P[prefix + "Ms"] @ X), dim=-1)  # [BS,1,L]
    self_atte = X @ sent_soft.transpose(-1, -2)                                 # [BS, 2u, 1]
    senti = word_soft.reshape(B * S, -1).sum(dim=-1, keepdim=True) * self_atte.squeeze(-1)
    senti = senti.view(B, S, -1).sum(dim=-2)
    return self_atte.view(B, S, -1), senti


# ----------------------------------------------------------------------------------------------
# CNet (src/model.py:110-126)
# ----------------------------------------------------------------------------------------------
def c_net(review_emb, lengths, P, threshold, prefix="control_net.c_net.", aten=False):
    B, S, L, E = review_emb.shape
    gru_repr = improved_rnn(review_emb.reshape(B * S, L, E), lengths.reshape(-1), P, prefix + "gru.module.", aten)
    gru_repr = gru_repr.reshape(B, S * L, -1)
    cnn_in = gru_repr.reshape(B * S, L, -1).transpose(-1, -2)
    k = P[prefix + "cnn.0.weight"].shape[-1]
    cnn_out = F.relu(F.conv1d(cnn_in, P[prefix + "cnn.0.weight"], P[prefix + "cnn.0.bias"], padding=(k - 1) // 2))
    cnn_out = cnn_out.max(dim=-1)[0].view(B, S, -1)
    view_p = torch.sigmoid(F.linear(cnn_out, P[prefix + "linear.0.weight"], P[prefix + "linear.0.bias"]))
    view_p = torch.where(view_p < threshold, torch.zeros_like(view_p), view_p)  # model.py:124
    final = torch.sum(view_p ** 2, dim=-2)                                      # model.py:125
    return gru_repr, view_p, final


# ----------------------------------------------------------------------------------------------
# ControlNet (src/model.py:179-198) incl. SSNet (src/model.py:142-143)
# ----------------------------------------------------------------------------------------------
def control_net(user_emb, item_emb, ui_emb, u_lengths, i_lengths, ui_lengths, P, threshold, aten=False):
    pre = "control_net."
    L_ui = ui_emb.shape[-2]
    gru_repr, view_p, c_out = c_net(ui_emb, ui_lengths, P, threshold, pre + "c_net.", aten)
    _, _, c_u = c_net(user_emb, u_lengths, P, threshold, pre + "c_net.", aten)
    _, _, c_i = c_net(item_emb, i_lengths, P, threshold, pre + "c_net.", aten)
    s, _ = s_net(gru_repr, view_p, L_ui, P, pre + "s_net.")
    senti = torch.sigmoid(F.linear(s, P[pre + "ss_net.linear.0.weight"], P[pre + "ss_net.linear.0.bias"]))
    senti = senti.expand(-1, -1, view_p.shape[-1])
    view_score = torch.sum(senti * view_p ** 2, dim=-2).div(torch.sum(view_p ** 2, dim=-2) + 1e-4)
    q_p = (view_score > 0.5).to(view_score.dtype)                               # model.py:189,192
    q_pos = torch.where(view_score < 0.5, torch.zeros_like(view_score), 4 * (view_score - 0.5) ** 2)
    q_neg = torch.where(view_score > 0.5, torch.zeros_like(view_score), 4 * (0.5 - view_score) ** 2)
    prefer_pos = c_out * q_p * q_pos
    prefer_neg = c_out * (1 - q_p) * q_neg
    return c_u, c_i, prefer_pos, prefer_neg, dict(view_p=view_p, view_score=view_score, c_out=c_out)


# ----------------------------------------------------------------------------------------------
# VGG16-D (torchvision.models.vgg16, call site src/model.py:204-207,217) - see module docstring
# ----------------------------------------------------------------------------------------------
def vgg16_forward(images, P, prefix="visual_net.vgg16.0.", train=False,
                  dropout_masks: Optional[Sequence[Tensor]] = None, p_drop=0.5):
    x = images
    ci = 0
    for v in VGG16_CFG:
        if v == "M":
            x = F.max_pool2d(x, 2, 2)
        else:
            idx = VGG16_CONV_IDX[ci]
            x = F.relu(F.conv2d(x, P[f"{prefix}features.{idx}.weight"], P[f"{prefix}features.{idx}.bias"], padding=1))
            ci += 1
    x = F.adaptive_avg_pool2d(x, 7).flatten(1)
    for j, idx in enumerate(VGG16_FC_IDX):
        x = F.linear(x, P[f"{prefix}classifier.{idx}.weight"], P[f"{prefix}classifier.{idx}.bias"])
        if j < 2:
            x = F.relu(x)
            if dropout_masks is not None:            # injected keep-masks (1/0), scaled like nn.Dropout
                x = x * dropout_masks[j] / (1.0 - p_drop)
            elif train:
                x = F.dropout(x, p_drop, True)
    return x


# ----------------------------------------------------------------------------------------------
# VisualNet (src/model.py:212-229)
# ----------------------------------------------------------------------------------------------
def visual_net(images, c_u, c_i, P, train=False, dropout_masks=None, vgg_fn=None):
    pre = "visual_net."
    B, V, Pc = images.shape[:3]
    flat = images.reshape(B * V * Pc, *images.shape[3:])
    raw = vgg_fn(flat) if vgg_fn is not None else vgg16_forward(flat, P, pre + "vgg16.0.", train, dropout_masks)
    img = raw.view(B, V, Pc, -1).mean(dim=-2)
    w, b = P[pre + "linear.weight"], P[pre + "linear.bias"]
    img_emb = F.linear(img, w, b).squeeze(-1)
    pos_emb = F.linear(P[pre + "pos_v_emb"], w, b).squeeze(-1)
    neg_emb = F.linear(P[pre + "neg_v_emb"], w, b).squeeze(-1)
    pos_match = torch.tanh(torch.abs(pos_emb - img_emb))
    neg_match = torch.tanh(torch.abs(neg_emb - img_emb))
    final_pos = c_u * c_i * (1 - pos_match)
    final_neg = c_u * c_i * (1 - neg_match)
    return pos_match, neg_match, final_pos, final_neg, raw


# ----------------------------------------------------------------------------------------------
# ReviewNet (src/model.py:157-169) and UMPR.forward (src/model.py:257-278)
# ----------------------------------------------------------------------------------------------
def review_net(user_emb, item_emb, u_lengths, i_lengths, P, aten=False, keep=None):
    pre = "review_net."
    L = user_emb.shape[-2]
    gru_u, gru_i, soft_u, soft_i, atte_u, atte_i = r_net(user_emb, item_emb, u_lengths, i_lengths, P, pre + "r_net.", aten)
    _, senti_u = s_net(gru_u, soft_u, L, P, pre + "s_net_u.")
    _, senti_i = s_net(gru_i, soft_i, L, P, pre + "s_net_i.")
    repr_u = torch.cat([atte_u, senti_u], dim=-1)
    repr_i = torch.cat([atte_i, senti_i], dim=-1)
    out = torch.tanh(F.linear(repr_u, P[pre + "linear_u.weight"]) + F.linear(repr_i, P[pre + "linear_i.weight"]))
    if keep is not None:
        keep.update(gru_u=gru_u, gru_i=gru_i, soft_u=soft_u, soft_i=soft_i, atte_u=atte_u, atte_i=atte_i,
                    senti_u=senti_u, senti_i=senti_i, review_repr=out)
    return out


def umpr_forward(P: Dict[str, Tensor], batch, *, review_net_only: bool, threshold=0.35, loss_v_rate=0.1,
                 train=False, dropout_masks=None, aten=False, keep: Optional[dict] = None, vgg_fn=None):
    """UMPR.forward, src/model.py:257-278.  ``batch`` is the 8-tuple src/dataset.py:173-182 builds."""
    user_reviews, item_reviews, ui_reviews, u_len, i_len, ui_len, photos, labels = batch
    emb = P["embedding.weight"]
    user_emb, item_emb, ui_emb = F.embedding(user_reviews, emb), F.embedding(item_reviews, emb), F.embedding(ui_reviews, emb)
    rr = review_net(user_emb, item_emb, u_len, i_len, P, aten, keep)
    w, b = P["linear_fusion.0.weight"], P["linear_fusion.0.bias"]
    if review_net_only:
        pred = F.relu(F.linear(rr, w, b)).squeeze(-1)
        loss = F.mse_loss(pred, labels, reduction="mean")
        return pred, loss
    c_u, c_i, prefer_pos, prefer_neg, kc = control_net(user_emb, item_emb, ui_emb, u_len, i_len, ui_len, P, threshold, aten)
    pos_match, neg_match, final_pos, final_neg, img = visual_net(photos, c_u, c_i, P, train, dropout_masks, vgg_fn)
    pred = F.relu(F.linear(torch.cat([rr, final_pos, final_neg], dim=-1), w, b)).squeeze(-1)
    loss_r = F.mse_loss(pred, labels, reduction="mean")
    loss_v = torch.mean(prefer_pos.transpose(-1, -2) @ pos_match + prefer_neg.transpose(-1, -2) @ neg_match)
    loss = loss_r + loss_v * loss_v_rate
    if keep is not None:
        keep.update(c_u=c_u, c_i=c_i, prefer_pos=prefer_pos, prefer_neg=prefer_neg, pos_match=pos_match,
                    neg_match=neg_match, final_pos=final_pos, final_neg=final_neg, vgg_out=img, loss_r=loss_r,
                    loss_v=loss_v, **kc)
    return pred, loss


# ----------------------------------------------------------------------------------------------
# Optimiser step (main.py:22-26,33-37) and evaluate_mse (src/evaluate.py:6-14)
# ----------------------------------------------------------------------------------------------
def adam_reference(P: Dict[str, Tensor], lr=1e-6, l2=1e-3):
    """torch.optim.Adam with the reference's two parameter groups (weight decay on names without 'bias')."""
    train = [(n, p) for n, p in P.items() if p.requires_grad]
    return torch.optim.Adam([
        {"params": [p for n, p in train if "bias" not in n]},
        {"params": [p for n, p in train if "bias" in n], "weight_decay": 0.0},
    ], lr, weight_decay=l2)


def adam_step_numpy(p, g, m, v, step, lr, wd, b1=0.9, b2=0.999, eps=1e-8):
    """Single-tensor statement of torch.optim.Adam's update (coupled L2), for kernel tests."""
    g = g + wd * p
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * m / denom
    return p, m, v


def evaluate_mse(P, batches, **kw):
    tot, cnt = 0.0, 0
    with torch.no_grad():
        for b in batches:
            pred, _ = umpr_forward(P, b, **kw)
            tot += F.mse_loss(pred, b[-1], reduction="sum").item()
            cnt += len(pred)
    return tot / cnt


def pretrain_rnet_forward(P: Dict[str, Tensor], u, u_length, i, i_length, target, aten=False):
    """PretrainRNet.forward, pretrain/pretrain_rnet.py:155-169: one sentence per side (S = 1), R-Net co-attention,
    ``sigmoid(Linear(4u -> 1)([atte_u; atte_i]))`` and ``nn.BCELoss`` (mean; ATen clamps the logs at -100)."""
    emb = P["embedding.weight"]
    B, L = u.shape
    ue = F.embedding(u.view(B, 1, L), emb)
    ie = F.embedding(i.view(B, 1, L), emb)
    _, _, _, _, att_u, att_i = r_net(ue, ie, u_length.view(B, 1), i_length.view(B, 1), P, "r_net.", aten)
    att = torch.cat([att_u, att_i], dim=-1)
    result = torch.sigmoid(att @ P["linear.0.weight"].t() + P["linear.0.bias"]).squeeze(-1)
    t = target.float()
    terms = -(t * torch.clamp(torch.log(result), min=-100.0) + (1 - t) * torch.clamp(torch.log1p(-result), min=-100.0))
    return result, terms.mean()
