"""Deterministic synthetic parameters and batches (no datasets / checkpoints exist offline).

* ``make_param_state`` - a seed-reproducible parameter dict under the reference's state_dict names
  (SURVEY.md section 8(a) key list; shapes from src/model.py:26-29,61-64,86-108,131-136,148-155,202-210,233-255).
  Generated with the CPU torch generator, so the build container and the GPU box produce identical bits.
* ``make_batch`` - the 8-tensor batch src/dataset.py:173-182 (``batch_loader``) hands to ``UMPR.forward``:
  ids int64 padded with 0, lengths int64 floored at 1 (src/dataset.py:122-131), photos f32 in [0,1) CHW
  (src/dataset.py:134-143), labels f32.
"""
from __future__ import annotations

import math
from typing import Dict, Sequence

import torch

VGG16_CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M")
VGG16_CONV_IDX = (0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28)
VGG16_FC = ((0, 25088, 4096), (3, 4096, 4096), (6, 4096, 1000))


def _uniform(g, shape, bound):
    return (torch.rand(shape, generator=g) * 2 - 1) * bound


def make_param_state(seed: int, emb_dim: int = 50, vocab: int = 1000, n_views: int = 1,
                     review_net_only: bool = False, gru_size: int = 64, atte_size: int = 64,
                     kernel_count: int = 120, kernel_size: int = 3, with_vgg: bool = True,
                     m_scale: float = 1.0, emb_std: float = 0.4) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    H, H2 = gru_size, 2 * gru_size
    P: Dict[str, torch.Tensor] = {}
    emb = torch.randn(vocab, emb_dim, generator=g) * emb_std
    emb[:3] = 0  # <PAD>/<UNK>/<NUM> are zero vectors (src/word2vec.py:19-20)
    P["embedding.weight"] = emb

    def gru(prefix):
        k = 1.0 / math.sqrt(H)
        for suf in ("", "_reverse"):
            P[f"{prefix}weight_ih_l0{suf}"] = _uniform(g, (3 * H, emb_dim), k)
            P[f"{prefix}weight_hh_l0{suf}"] = _uniform(g, (3 * H, H), k)
            P[f"{prefix}bias_ih_l0{suf}"] = _uniform(g, (3 * H,), k)
            P[f"{prefix}bias_hh_l0{suf}"] = _uniform(g, (3 * H,), k)

    P["review_net.r_net.M"] = torch.randn(H2, H2, generator=g) * m_scale
    gru("review_net.r_net.gru.module.")
    for s in ("s_net_u", "s_net_i"):
        P[f"review_net.{s}.Ms"] = torch.randn(atte_size, H2, generator=g) * m_scale
        P[f"review_net.{s}.Ws"] = torch.randn(1, atte_size, generator=g) * m_scale
    k = 1.0 / math.sqrt(2 * H2)
    P["review_net.linear_u.weight"] = _uniform(g, (H2, 2 * H2), k)
    P["review_net.linear_i.weight"] = _uniform(g, (H2, 2 * H2), k)
    if not review_net_only:
        V = n_views
        gru("control_net.c_net.gru.module.")
        k = 1.0 / math.sqrt(H2 * kernel_size)
        P["control_net.c_net.cnn.0.weight"] = _uniform(g, (kernel_count, H2, kernel_size), k)
        P["control_net.c_net.cnn.0.bias"] = _uniform(g, (kernel_count,), k)
        k = 1.0 / math.sqrt(kernel_count)
        P["control_net.c_net.linear.0.weight"] = _uniform(g, (V, kernel_count), k)
        P["control_net.c_net.linear.0.bias"] = _uniform(g, (V,), k)
        P["control_net.s_net.Ms"] = torch.randn(atte_size, H2, generator=g) * m_scale
        P["control_net.s_net.Ws"] = torch.randn(1, atte_size, generator=g) * m_scale
        k = 1.0 / math.sqrt(H2)
        P["control_net.ss_net.linear.0.weight"] = _uniform(g, (1, H2), k)
        P["control_net.ss_net.linear.0.bias"] = _uniform(g, (1,), k)
        if with_vgg:
            cin = 3
            ci = 0
            for v in VGG16_CFG:
                if v == "M":
                    continue
                idx = VGG16_CONV_IDX[ci]
                std = math.sqrt(2.0 / (cin * 9))
                P[f"visual_net.vgg16.0.features.{idx}.weight"] = torch.randn(v, cin, 3, 3, generator=g) * std
                P[f"visual_net.vgg16.0.features.{idx}.bias"] = _uniform(g, (v,), 0.05)
                cin = v
                ci += 1
            for idx, fin, fout in VGG16_FC:
                P[f"visual_net.vgg16.0.classifier.{idx}.weight"] = torch.randn(fout, fin, generator=g) * math.sqrt(2.0 / fin)
                P[f"visual_net.vgg16.0.classifier.{idx}.bias"] = _uniform(g, (fout,), 0.05)
        P["visual_net.pos_v_emb"] = torch.randn(V, 1000, generator=g)
        P["visual_net.neg_v_emb"] = torch.randn(V, 1000, generator=g)
        k = 1.0 / math.sqrt(1000)
        P["visual_net.linear.weight"] = _uniform(g, (1, 1000), k)
        P["visual_net.linear.bias"] = _uniform(g, (1,), k)
        fin = H2 + 2 * V
    else:
        fin = H2
    k = 1.0 / math.sqrt(fin)
    P["linear_fusion.0.weight"] = _uniform(g, (1, fin), k)
    # a positive bias keeps ReLU(prediction) alive so gradients flow in parity tests
    P["linear_fusion.0.bias"] = torch.full((1,), 0.5) + _uniform(g, (1,), k)
    return P


def _zipf_ids(g, n, vocab):
    # Zipf(1.0)-like over [3, vocab): inverse-CDF of 1/x on a continuous support, 5% forced to UNK/NUM
    u = torch.rand(n, generator=g)
    ids = (3 + (torch.exp(u * math.log(max(vocab - 3, 2))) - 1)).long().clamp_(3, vocab - 1)
    special = torch.rand(n, generator=g) < 0.05
    which = (torch.rand(n, generator=g) < 0.5).long() + 1
    return torch.where(special, which, ids)


def _reviews(g, B, max_count, min_count, max_len, vocab, full_pad):
    counts = torch.randint(min_count, max_count + 1, (B,), generator=g)
    if full_pad:
        counts[:] = max_count
    sents = []
    for b in range(B):
        lens = torch.randint(6, max_len + 1, (int(counts[b]),), generator=g)
        if full_pad:
            lens[:] = max_len
        lens, _ = torch.sort(lens, descending=True, stable=True)
        sents.append([_zipf_ids(g, int(l), vocab) for l in lens])
    return sents


def _pad(sents, S, L):
    B = len(sents)
    ids = torch.zeros(B, S, L, dtype=torch.int64)
    lengths = torch.ones(B, S, dtype=torch.int64)  # empty sentences have length 1 (dataset.py:127)
    for b, ss in enumerate(sents):
        for s, t in enumerate(ss):
            ids[b, s, : len(t)] = t
            lengths[b, s] = max(1, len(t))
    return ids, lengths


def make_batch(seed: int, B: int, vocab: int = 1000, n_views: int = 1, photo_count: int = 1,
               max_sent_count: int = 20, min_sent_count: int = 5, max_ui_sent_count: int = 5,
               max_sent_length: int = 20, review_net_only: bool = False, full_pad: bool = False,
               img_hw: int = 224):
    g = torch.Generator().manual_seed(seed)
    u = _reviews(g, B, max_sent_count, min_sent_count, max_sent_length, vocab, full_pad)
    i = _reviews(g, B, max_sent_count, min_sent_count, max_sent_length, vocab, full_pad)
    ui = _reviews(g, B, max_ui_sent_count, 1, max_sent_length, vocab, full_pad)
    # user & item share the batch-wide (max_count, max_len) (dataset.py:164-170); ui padded on its own (:171)
    S = max(max(len(x) for x in u), max(len(x) for x in i))
    L = max(max(len(t) for x in u for t in x), max(len(t) for x in i for t in x))
    S_ui = max(len(x) for x in ui)
    L_ui = max(len(t) for x in ui for t in x)
    u_ids, u_len = _pad(u, S, L)
    i_ids, i_len = _pad(i, S, L)
    ui_ids, ui_len = _pad(ui, S_ui, L_ui)
    if review_net_only:
        photos = torch.zeros(0)  # torch.Tensor([]) when photos are ignored (dataset.py:158,180)
    else:
        photos = torch.rand(B, n_views, photo_count, 3, img_hw, img_hw, generator=g)
        missing = torch.rand(B, n_views, photo_count, generator=g) < 0.02
        photos[missing] = 0  # unreadable photo -> zeros (dataset.py:142-143)
    labels = torch.randint(1, 6, (B,), generator=g).float()
    return u_ids, i_ids, ui_ids, u_len, i_len, ui_len, photos, labels


def make_pretrain_state(seed: int, emb_dim: int = 50, vocab: int = 1000, gru_size: int = 64,
                        m_scale: float = 0.05, emb_std: float = 0.4) -> Dict[str, torch.Tensor]:
    """state_dict of the reference's PretrainRNet (pretrain/pretrain_rnet.py:144-153) from a seed."""
    g = torch.Generator().manual_seed(seed)
    H, H2 = gru_size, 2 * gru_size
    P: Dict[str, torch.Tensor] = {}
    emb = torch.randn(vocab, emb_dim, generator=g) * emb_std
    emb[:3] = 0
    P["embedding.weight"] = emb
    P["r_net.M"] = torch.randn(H2, H2, generator=g) * m_scale
    k = 1.0 / math.sqrt(H)
    for suf in ("", "_reverse"):
        P[f"r_net.gru.module.weight_ih_l0{suf}"] = _uniform(g, (3 * H, emb_dim), k)
        P[f"r_net.gru.module.weight_hh_l0{suf}"] = _uniform(g, (3 * H, H), k)
        P[f"r_net.gru.module.bias_ih_l0{suf}"] = _uniform(g, (3 * H,), k)
        P[f"r_net.gru.module.bias_hh_l0{suf}"] = _uniform(g, (3 * H,), k)
    k = 1.0 / math.sqrt(2 * H2)
    P["linear.0.weight"] = _uniform(g, (1, 2 * H2), k)
    P["linear.0.bias"] = _uniform(g, (1,), k)
    return P


def make_pretrain_batch(seed: int, B: int, L: int = 20, vocab: int = 1000, ragged: bool = True):
    """(u, u_length, i, i_length, target) as PretrainRNetDataset yields them (pretrain_rnet.py:129-135): padded id
    rows [B, L] int64, lengths [B] int64, labels [B] float in {0, 1}.  The reference's dataset passes len(padded row)
    = L as the length; ``ragged`` also exercises shorter lengths (the model accepts any 1 <= length <= L)."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(2):
        ids = torch.randint(3, vocab, (B, L), generator=g)
        ln = torch.randint(1, L + 1, (B,), generator=g) if ragged else torch.full((B,), L, dtype=torch.int64)
        ids = ids * (torch.arange(L)[None, :] < ln[:, None])
        out += [ids, ln]
    out.append(torch.randint(0, 2, (B,), generator=g).float())
    return tuple(out)
