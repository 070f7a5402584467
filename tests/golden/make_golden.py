#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE ITSELF in the build container.

Run:  python tests/golden/make_golden.py        (needs /root/reference; never runs on the GPU box)

What is imported: /root/reference/src/model.py, and for the input-pipeline fixtures (gen_dataset) src/dataset.py and
src/word2vec.py with empty module objects named cv2 / gensim (never called).  model.py's top-level
``import torchvision`` cannot be satisfied in this image (torchvision is not installed), so a module object
named ``torchvision`` is placed in sys.modules first.  For every text-path fixture (ImprovedRnn, RNet, SNet,
CNet, ControlNet, ReviewNet, UMPR-R) that object is never called - all arithmetic is the reference's code on
torch CPU ops.  For full-UMPR fixtures the reference calls ``torchvision.models.vgg16(pretrained=True,
num_classes=1000)`` (src/model.py:204-207): the ImageNet checkpoint is a network fetch and torchvision is
absent, so ``vgg16`` here returns the published VGG16-D layer list built from plain torch.nn layers with
torchvision's layer indices and random weights.  Consequently the VGG16 part of those fixtures pins the
build against torch.nn Conv2d/MaxPool2d/Linear composed per configuration D, NOT against torchvision itself
("parity unpinned" for torchvision's own code; everything downstream of the 1000-d VGG output is the
reference's code).

Fixtures hold inputs-by-seed and expected outputs only (data); parameters and batches are regenerated from
seeds by umpr_amd/synthetic.py, so no weights and no reference source are stored.
"""
import os
import sys
import types

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from umpr_amd.synthetic import VGG16_CFG, make_batch, make_param_state  # noqa: E402


class MaskDropout(nn.Module):
    """nn.Dropout(0.5) stand-in that can take an injected keep-mask (train-mode parity, SURVEY 7 hard parts)."""

    def __init__(self, p=0.5):
        super().__init__()
        self.p = p
        self.mask = None

    def forward(self, x):
        if self.mask is not None:
            return x * self.mask / (1 - self.p)
        return nn.functional.dropout(x, self.p, self.training)


class _VGG16D(nn.Module):
    def __init__(self, num_classes=1000):
        super().__init__()
        layers, cin = [], 3
        for v in VGG16_CFG:
            if v == "M":
                layers.append(nn.MaxPool2d(2, 2))
            else:
                layers += [nn.Conv2d(cin, v, 3, padding=1), nn.ReLU(inplace=True)]
                cin = v
        self.features = nn.Sequential(*layers)
        self.avgpool = nn.AdaptiveAvgPool2d((7, 7))
        self.classifier = nn.Sequential(
            nn.Linear(512 * 7 * 7, 4096), nn.ReLU(True), MaskDropout(),
            nn.Linear(4096, 4096), nn.ReLU(True), MaskDropout(),
            nn.Linear(4096, num_classes))

    def forward(self, x):
        x = self.avgpool(self.features(x))
        return self.classifier(torch.flatten(x, 1))


def _install_import_shims():
    tv = types.ModuleType("torchvision")
    tv.models = types.ModuleType("torchvision.models")
    tv.models.vgg16 = lambda pretrained=False, num_classes=1000, **kw: _VGG16D(num_classes)
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = tv.models
    sys.path.insert(0, REF)


class Cfg:
    def __init__(self, review_net_only, views):
        self.review_net_only = review_net_only
        self.loss_v_rate = 0.1
        self.gru_size = 64
        self.self_atte_size = 64
        self.views = views
        self.kernel_count = 120
        self.kernel_size = 3
        self.threshold = 0.35


def build_reference(refmodel, P, review_net_only, n_views):
    cfg = Cfg(review_net_only, ["v%d" % i for i in range(n_views)])
    m = refmodel.UMPR(cfg, P["embedding.weight"].numpy())
    missing = m.load_state_dict({k: v.clone() for k, v in P.items()}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return m


def grads_of(model):
    return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}


def pack_grads(out, grads, full_limit=400_000):
    """Small tensors in full; big ones (VGG) as sum / abs-sum / L2 and a strided sample."""
    for n, g in grads.items():
        g = g.detach().float()
        if g.numel() <= full_limit:
            out["grad/" + n] = g.numpy()
        else:
            flat = g.reshape(-1)
            stride = max(1, flat.numel() // 4096)
            out["gradsample/" + n] = flat[::stride].numpy().copy()
            out["gradstat/" + n] = np.array([flat.double().sum().item(), flat.double().abs().sum().item(),
                                             flat.double().pow(2).sum().sqrt().item(), stride], dtype=np.float64)


def save(name, out):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in out.items()})
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


def gen_improved_rnn(refmodel):
    for tag, (N, L, E, H, seed) in {"toy": (7, 5, 6, 4, 11), "true": (40, 20, 50, 64, 12)}.items():
        g = torch.Generator().manual_seed(seed)
        rnn = refmodel.ImprovedRnn(nn.GRU, input_size=E, hidden_size=H, batch_first=True, bidirectional=True)
        with torch.no_grad():
            for p in rnn.parameters():
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) / H ** 0.5)
        lengths = torch.randint(1, L + 1, (N,), generator=g)
        lengths[0] = L
        x = torch.randn(N, L, E, generator=g)
        for n in range(N):
            x[n, lengths[n]:] = 0
        x.requires_grad_(True)
        out, _ = rnn(x, lengths)
        gout = torch.randn(out.shape, generator=g)
        out.backward(gout)
        pk = nn.utils.rnn.pack_padded_sequence(x.detach(), lengths, batch_first=True, enforce_sorted=False)
        d = dict(x=x.detach().numpy(), lengths=lengths.numpy(), out=out.detach().numpy(), gout=gout.numpy(),
                 sorted_indices=pk.sorted_indices.numpy(), unsorted_indices=pk.unsorted_indices.numpy(),
                 gx=x.grad.numpy())
        for n, p in rnn.named_parameters():
            d["param/" + n] = p.detach().numpy()
            d["grad/" + n] = p.grad.numpy()
        save("improved_rnn_" + tag, d)
    # tie order of the (non-stable) sort that defines the sentence permutation
    g = torch.Generator().manual_seed(13)
    lengths = torch.randint(1, 21, (1280,), generator=g)
    pk = nn.utils.rnn.pack_padded_sequence(torch.zeros(1280, 20, 1), lengths, batch_first=True, enforce_sorted=False)
    save("sort_ties", dict(lengths=lengths.numpy(), sorted_indices=pk.sorted_indices.numpy(),
                           unsorted_indices=pk.unsorted_indices.numpy()))


def capture(model, store):
    hooks = []

    def hk(name):
        def f(mod, inp, out):
            store[name] = out
        return f
    hooks.append(model.review_net.r_net.register_forward_hook(hk("r_net")))
    hooks.append(model.review_net.s_net_u.register_forward_hook(hk("s_net_u")))
    hooks.append(model.review_net.s_net_i.register_forward_hook(hk("s_net_i")))
    hooks.append(model.review_net.register_forward_hook(hk("review_net")))
    if not model.review_net_only:
        hooks.append(model.control_net.register_forward_hook(hk("control_net")))
        hooks.append(model.control_net.s_net.register_forward_hook(hk("ctrl_s_net")))
        hooks.append(model.visual_net.register_forward_hook(hk("visual_net")))
        hooks.append(model.visual_net.vgg16.register_forward_hook(hk("vgg16")))
    return hooks


def gen_umpr(refmodel, name, *, B, n_views, review_net_only, m_scale, pseed, bseed, full_pad=False,
             drop_masks=False, vocab=1000, photo_count=1):
    P = make_param_state(pseed, 50, vocab, n_views, review_net_only, m_scale=m_scale)
    batch = make_batch(bseed, B, vocab, n_views, photo_count, review_net_only=review_net_only, full_pad=full_pad)
    model = build_reference(refmodel, P, review_net_only, n_views)
    out = dict(meta=np.array([B, n_views, int(review_net_only), pseed, bseed, int(full_pad), vocab], dtype=np.int64),
               m_scale=np.array(m_scale), photo_count=np.array(photo_count))
    if drop_masks:
        model.train()
        g = torch.Generator().manual_seed(bseed + 1000)
        masks = [(torch.rand(B * n_views, 4096, generator=g) < 0.5).float() for _ in range(2)]
        cls = model.visual_net.vgg16[0].classifier
        cls[2].mask, cls[5].mask = masks
        out["drop_mask0"] = masks[0].numpy().astype(np.uint8)
        out["drop_mask1"] = masks[1].numpy().astype(np.uint8)
    else:
        model.eval()
    store = {}
    hooks = capture(model, store)
    pred, loss = model(*batch)
    loss.mean().backward()
    for h in hooks:
        h.remove()
    out["prediction"] = pred.detach().numpy()
    out["loss"] = loss.detach().numpy()
    gru_u, gru_i, soft_u, soft_i, atte_u, atte_i = store["r_net"]
    out.update(gru_u=gru_u.detach().numpy(), gru_i=gru_i.detach().numpy(), soft_u=soft_u.detach().numpy(),
               soft_i=soft_i.detach().numpy(), atte_u=atte_u.detach().numpy(), atte_i=atte_i.detach().numpy(),
               senti_u=store["s_net_u"][1].detach().numpy(), senti_i=store["s_net_i"][1].detach().numpy(),
               review_repr=store["review_net"].detach().numpy())
    if not review_net_only:
        c_u, c_i, pp, pn = store["control_net"]
        pm, nm, fp, fn = store["visual_net"]
        out.update(c_u=c_u.detach().numpy(), c_i=c_i.detach().numpy(), prefer_pos=pp.detach().numpy(),
                   prefer_neg=pn.detach().numpy(), pos_match=pm.detach().numpy(), neg_match=nm.detach().numpy(),
                   final_pos=fp.detach().numpy(), final_neg=fn.detach().numpy(),
                   vgg_out=store["vgg16"].detach().numpy(), ctrl_self_atte=store["ctrl_s_net"][0].detach().numpy())
    pack_grads(out, grads_of(model))
    save(name, out)


def gen_adam(refmodel):
    """3 optimiser steps exactly as main.py:22-26,31-37 drives them (Adam, two groups, loss.mean())."""
    P = make_param_state(31, 50, 1000, 1, True, m_scale=0.05)
    model = build_reference(refmodel, P, True, 1)
    lr, l2 = 1e-3, 1e-3  # larger lr than the 1e-6 default so three steps move the weights measurably
    opt = torch.optim.Adam([
        {'params': (p for name, p in model.named_parameters() if 'bias' not in name)},
        {'params': (p for name, p in model.named_parameters() if 'bias' in name), 'weight_decay': 0.}
    ], lr, weight_decay=l2)
    lr_sch = torch.optim.lr_scheduler.ExponentialLR(opt, 0.99)
    out = dict(lr=np.array(lr), l2=np.array(l2))
    losses = []
    for step in range(3):
        batch = make_batch(500 + step, 4, 1000, review_net_only=True)
        model.train()
        pred, loss = model(*batch)
        loss = loss.mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
        if step == 0:
            lr_sch.step()  # exercise the per-epoch decay once (main.py:54)
    out["losses"] = np.array(losses, dtype=np.float64)
    for n, p in model.named_parameters():
        if p.requires_grad:
            out["param/" + n] = p.detach().numpy()
    save("adam_umpr_r", out)


def gen_dataparallel(refmodel):
    """DataParallel semantics (main.py:81-82,34): contiguous chunks, per-replica forward, loss = mean of replica
    losses; gradient = mean over replicas.  Emulated replica-by-replica on CPU with the reference module."""
    B, R = 4, 2
    P = make_param_state(41, 50, 1000, 1, False, with_vgg=False, m_scale=0.05)
    # VGG is not needed to pin the sharding semantics; feed the visual head a fixed fake VGG output instead
    batch = make_batch(42, B, 1000, 1, img_hw=8)
    cfg = Cfg(False, ["v0"])

    class FakeVGG(nn.Module):
        def __init__(self):
            super().__init__()
            self.proj = nn.Linear(3 * 8 * 8, 1000)
            g = torch.Generator().manual_seed(43)
            with torch.no_grad():
                self.proj.weight.copy_(torch.randn(1000, 192, generator=g) * 0.05)
                self.proj.bias.zero_()

        def forward(self, x):
            return self.proj(x.flatten(1))

    sys.modules["torchvision"].models.vgg16 = lambda **kw: FakeVGG()
    model = refmodel.UMPR(cfg, P["embedding.weight"].numpy())
    sd = model.state_dict()
    for k, v in P.items():
        sd[k] = v.clone()
    model.load_state_dict(sd)
    model.eval()
    sys.modules["torchvision"].models.vgg16 = lambda pretrained=False, num_classes=1000, **kw: _VGG16D(num_classes)
    losses, preds = [], []
    for r in range(R):
        sl = slice(r * B // R, (r + 1) * B // R)
        shard = tuple(t[sl] for t in batch)
        pred, loss = model(*shard)
        losses.append(loss)
        preds.append(pred)
    loss = torch.stack(losses).mean()
    loss.backward()
    out = dict(prediction=torch.cat(preds).detach().numpy(), shard_losses=torch.stack(losses).detach().numpy(),
               loss=loss.detach().numpy(), fake_vgg_w=model.visual_net.vgg16[0].proj.weight.detach().numpy())
    pack_grads(out, {n: g for n, g in grads_of(model).items() if "vgg16" not in n})
    save("dp_shards", out)


def write_tiny_corpus(root, seed=7):
    """A small CSV / GloVe / photos.json set that exercises every filter of src/dataset.py (short sentences, empty
    reviews, too few sentences, truncation to max counts, items without photos, fewer photos than photo_count)."""
    import json
    import random
    rnd = random.Random(seed)
    os.makedirs(root, exist_ok=True)
    words = ["w%d" % i for i in range(60)]
    with open(os.path.join(root, "glove.txt"), "w") as f:
        for i, w in enumerate(words[:50]):      # w50..w59 are out of vocabulary -> <UNK>
            f.write(w + " " + " ".join("%.4f" % rnd.uniform(-1, 1) for _ in range(8)) + "\n")
    rows = []
    n_users, n_items = 16, 8
    for u in range(n_users):
        for it in rnd.sample(range(n_items), rnd.randint(3, 6)):
            n_sent = rnd.choice([0, 1, 2, 3, 3, 4, 4, 6, 8])
            sents = []
            for _ in range(n_sent):
                L = rnd.choice([2, 4, 6, 6, 7, 8, 9, 12, 25])
                toks = [rnd.choice(words) if rnd.random() > 0.1 else str(rnd.randint(0, 99)) for _ in range(L)]
                sents.append(" ".join(toks))
            review = " . ".join(sents) + (" ." if sents else "")
            rows.append(dict(userID="U%d" % u, itemID="I%d" % it, review=review, rating=float(rnd.randint(1, 5)),
                             user_num=u, item_num=it))
    import pandas as pd
    pd.DataFrame(rows).to_csv(os.path.join(root, "train.csv"), index=False)
    with open(os.path.join(root, "photos.json"), "w") as f:
        for it in range(n_items):
            if it == 3:
                continue                         # item without photos: its samples are dropped
            for k in range(rnd.randint(1, 3)):
                f.write(json.dumps(dict(business_id="I%d" % it, photo_id="p%d_%d" % (it, k),
                                        label=rnd.choice(["food", "inside"]))) + "\n")
    return root


class DataCfg:
    max_sent_count = 6
    min_sent_count = 2
    max_ui_sent_count = 3
    max_sent_length = 10
    photo_count = 2
    views = ["food"]
    review_level = "sentence"


def gen_dataset():
    """Dataset / Word2vec / collate fixtures from the reference's own src/dataset.py and src/word2vec.py (imported with
    empty module objects named cv2 and gensim - neither is called: photos are ignored in the collate fixture)."""
    import json
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    sys.modules.setdefault("gensim", types.ModuleType("gensim"))
    from src.dataset import Dataset, batch_loader, pad_reviews
    from src.word2vec import Word2vec
    root = write_tiny_corpus(os.path.join(HERE, "tiny_corpus"))
    w2v = Word2vec(os.path.join(root, "glove.txt"))
    out = {}
    probe = ["w1 w2. w3 77 w55", "  w9 w9 w9 ", "12 w0 w59 w3 w4 w5 w6 w7"]
    out["sent2indices"] = [w2v.sent2indices(p) for p in probe] + [w2v.sent2indices(probe[2], 5), w2v.sent2indices(probe[1], 6)]
    out["embedding_rows"] = [list(map(float, w2v.embedding[i])) for i in (0, 1, 2, 3, 52)]
    out["vocab_len"] = len(w2v)
    for name, views, level in (("amazon", ["food"], "sentence"), ("two_views", ["food", "inside"], "sentence"),
                               ("review_level", ["food"], "review")):
        cfg = DataCfg()
        cfg.views = views
        cfg.review_level = level
        ds = Dataset(os.path.join(root, "train.csv"), os.path.join(root, "photos.json"), os.path.join(root, "photos"), w2v, cfg)
        out["dataset/" + name] = [list(x) for x in ds.data]
        out["retain/" + name] = [bool(b) for b in ds.retain_idx]
        if len(ds) >= 3:
            b = batch_loader([ds[i] for i in range(min(4, len(ds)))], ignore_photos=True)
            out["batch/" + name] = [t.tolist() for t in b]
    out["pad_reviews"] = pad_reviews([[[1, 2, 3], []], [[4]]])
    with open(os.path.join(HERE, "dataset_golden.json"), "w") as f:
        json.dump(out, f)
    print("wrote dataset_golden.json", os.path.getsize(os.path.join(HERE, "dataset_golden.json")) // 1024, "KB")


def gen_pretrain_rnet():
    """PretrainRNet (pretrain/pretrain_rnet.py:144-169) forward / grads / 3 Adam steps as pretrain_rnet.py:177-198
    drives them.  The module imports gensim and sklearn at the top; gensim gets an empty module object (never called)."""
    sys.modules.setdefault("gensim", types.ModuleType("gensim"))
    from pretrain.pretrain_rnet import PretrainRNet
    from umpr_amd.synthetic import make_pretrain_batch, make_pretrain_state

    class W2V:
        pass

    for tag, (B, L, ragged, pseed, bseed) in {"ragged": (6, 9, True, 41, 42), "full": (16, 20, False, 43, 44)}.items():
        P = make_pretrain_state(pseed, 50, 300)
        w2v = W2V()
        w2v.embedding = P["embedding.weight"].numpy()
        w2v.word_dim = 50
        m = PretrainRNet(w2v, 64)
        m.load_state_dict(P)
        batch = make_pretrain_batch(bseed, B, L, 300, ragged)
        result, loss = m(*batch)
        loss.mean().backward()
        out = {"meta": np.array([B, L, int(ragged), pseed, bseed]), "result": result.detach(), "loss": loss.detach()}
        for n, g in grads_of(m).items():
            out["grad/" + n] = g
        # 3 optimiser steps, pretrain_rnet.py:177-198 (lr 0.01, l2 1e-3 on names without 'bias')
        m.load_state_dict(P)
        opt = torch.optim.Adam([
            {'params': (p for name, p in m.named_parameters() if 'bias' not in name)},
            {'params': (p for name, p in m.named_parameters() if 'bias' in name), 'weight_decay': 0.}
        ], 0.01, weight_decay=1e-3)
        losses = []
        for step in range(3):
            _, l = m(*batch)
            l = l.mean()
            opt.zero_grad()
            l.backward()
            opt.step()
            losses.append(float(l))
        out["traj_loss"] = np.array(losses)
        for n, p_ in m.named_parameters():
            out["traj_param/" + n] = p_.detach()
        save("pretrain_rnet_" + tag, out)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    _install_import_shims()
    from src import model as refmodel  # the reference's own code
    if len(sys.argv) > 1 and sys.argv[1] == "dataset":
        gen_dataset()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "photos":
        gen_umpr(refmodel, "umpr_full_V2_P2_B2", B=2, n_views=2, review_net_only=False, m_scale=0.05, pseed=59,
                 bseed=60, photo_count=2)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "pretrain":
        gen_pretrain_rnet()
        return
    gen_dataset()
    gen_improved_rnn(refmodel)
    gen_umpr(refmodel, "umpr_r_B4", B=4, n_views=1, review_net_only=True, m_scale=1.0, pseed=21, bseed=22)
    gen_umpr(refmodel, "umpr_r_B4_soft", B=4, n_views=1, review_net_only=True, m_scale=0.05, pseed=23, bseed=24)
    gen_umpr(refmodel, "umpr_r_B3_fullpad", B=3, n_views=1, review_net_only=True, m_scale=0.05, pseed=25, bseed=26,
             full_pad=True)
    gen_adam(refmodel)
    gen_dataparallel(refmodel)
    gen_umpr(refmodel, "umpr_full_V1_B2", B=2, n_views=1, review_net_only=False, m_scale=0.05, pseed=51, bseed=52)
    gen_umpr(refmodel, "umpr_full_V1_B2_randnM", B=2, n_views=1, review_net_only=False, m_scale=1.0, pseed=53, bseed=54)
    gen_umpr(refmodel, "umpr_full_V4_B2", B=2, n_views=4, review_net_only=False, m_scale=0.05, pseed=55, bseed=56)
    gen_umpr(refmodel, "umpr_full_V1_B2_drop", B=2, n_views=1, review_net_only=False, m_scale=0.05, pseed=57, bseed=58,
             drop_masks=True)
    gen_pretrain_rnet()
    gen_umpr(refmodel, "umpr_full_V2_P2_B2", B=2, n_views=2, review_net_only=False, m_scale=0.05, pseed=59, bseed=60,
             photo_count=2)


if __name__ == "__main__":
    main()
