#!/usr/bin/env python3
"""Real-data path throughput: a synthetic corpus in the reference's on-disk layout
(train.csv, photos.json, photos/*.jpg, GloVe text) -> umpr_amd.data.Dataset -> DataLoader(batch_loader) -> UMPR.
Prints loader-only batches/s for several worker counts, then end-to-end training samples/s with the best one.

    python tools/bench_loader.py [--items 64] [--users 400] [--batch 64] [--workers 0,4,8,16] [--steps 20]
"""
import argparse
import json
import os
import random
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def write_corpus(root, n_users, n_items, seed=3):
    from PIL import Image
    rnd = random.Random(seed)
    words = ["w%d" % i for i in range(2000)]
    with open(os.path.join(root, "glove.txt"), "w") as f:
        for w in words:
            f.write(w + " " + " ".join("%.4f" % rnd.uniform(-1, 1) for _ in range(50)) + "\n")
    rows = []
    for u in range(n_users):
        for it in rnd.sample(range(n_items), 8):
            sents = [" ".join(rnd.choice(words) for _ in range(rnd.randint(7, 18))) for _ in range(rnd.randint(2, 4))]
            rows.append(dict(userID="U%d" % u, itemID="I%d" % it, review=" . ".join(sents) + " .",
                             rating=float(rnd.randint(1, 5)), user_num=u, item_num=it))
    import pandas as pd
    pd.DataFrame(rows).to_csv(os.path.join(root, "train.csv"), index=False)
    os.makedirs(os.path.join(root, "photos"))
    g = np.random.default_rng(seed)
    with open(os.path.join(root, "photos.json"), "w") as f:
        for it in range(n_items):
            f.write(json.dumps(dict(business_id="I%d" % it, photo_id="p%d" % it, label="food")) + "\n")
            # smooth random field at a typical photo size: JPEG decode cost close to a real picture's
            low = g.random((24, 32, 3))
            img = Image.fromarray((low * 255).astype(np.uint8)).resize((500, 375), Image.BICUBIC)
            img.save(os.path.join(root, "photos", "p%d.jpg" % it), quality=90)
    return len(rows)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--items", type=int, default=64)
    ap.add_argument("--users", type=int, default=200)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--workers", default="0,4,8,16")
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    from torch.utils.data import DataLoader
    from main import _Collate
    from umpr_amd.config import Config
    from umpr_amd.data import Dataset, Word2vec
    with tempfile.TemporaryDirectory() as d:
        n = write_corpus(d, a.users, a.items)
        cfg = Config(argv=[])
        cfg.views = ["food"]
        w2v = Word2vec(os.path.join(d, "glove.txt"))
        t0 = time.perf_counter()
        ds = Dataset(os.path.join(d, "train.csv"), os.path.join(d, "photos.json"), os.path.join(d, "photos"), w2v, cfg)
        print(f"corpus: {n} reviews -> {len(ds)} samples, Dataset built in {time.perf_counter() - t0:.2f} s", flush=True)
        best = (0.0, 0)
        for w in [int(x) for x in a.workers.split(",")]:
            kw = dict(collate_fn=_Collate(False), num_workers=w, pin_memory=torch.cuda.is_available())
            if w:
                kw.update(prefetch_factor=2, persistent_workers=True)
            dl = DataLoader(ds, batch_size=a.batch, shuffle=True, **kw)
            it = iter(dl)
            next(it)  # workers started, first batch decoded
            t0 = time.perf_counter()
            k = 0
            for b in it:
                k += 1
                if k >= a.steps:
                    break
            dt = time.perf_counter() - t0
            rate = k * a.batch / dt
            print(f"loader only, {w:2d} workers: {rate:8.1f} samples/s ({1e3 * dt / k:.1f} ms per batch of {a.batch})", flush=True)
            if rate > best[0]:
                best = (rate, w)
            del it, dl
        if not torch.cuda.is_available():
            return
        from umpr_amd.model import UMPR
        from umpr_amd.optim import FusedAdam
        from umpr_amd.train import train_step
        dev = torch.device("cuda:0")
        model = UMPR(cfg, w2v.embedding).to(dev)
        opt = FusedAdam(model, cfg.learning_rate, cfg.l2_regularization, cfg.lr_decay)
        w = best[1]
        kw = dict(collate_fn=_Collate(False), num_workers=w, pin_memory=True)
        if w:
            kw.update(prefetch_factor=2, persistent_workers=True)
        dl = DataLoader(ds, batch_size=a.batch, shuffle=True, drop_last=True, **kw)
        it = iter(dl)
        for _ in range(3):
            train_step(model, opt, next(it))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        k = 0
        for b in it:
            train_step(model, opt, b)
            k += 1
            if k >= a.steps:
                break
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"end to end (CSV + JPEG decode + collate + H2D + train step), {w} workers: {k * a.batch / dt:.1f} samples/s "
              f"({1e3 * dt / k:.1f} ms per step)", flush=True)


if __name__ == "__main__":
    main()
