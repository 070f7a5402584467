#!/usr/bin/env python3
"""Entry point mirroring the reference's main.py (config surface: config.py, run recipes: readme.md:70-92).

    python main.py --data_dir synthetic --batch_size 64                       # single MI355X
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 main.py --data_dir synthetic

The model / optimiser / train / evaluate drivers are the MI355X-native ones (umpr_amd).  Data: the reference's CSV +
GloVe + JPEG pipeline (src/dataset.py, src/word2vec.py) is the next scope row; until it lands,
`--data_dir synthetic` (or a data_dir without train.csv) trains on synthetic batches of exactly the layout
`batch_loader` produces (src/dataset.py:173-182), which is what the throughput numbers are quoted on.
"""
import os
import sys
import time

import torch

from umpr_amd import parallel
from umpr_amd.config import Config
from umpr_amd.model import UMPR
from umpr_amd.synthetic import make_batch
from umpr_amd.train import evaluate_mse, training


class SyntheticLoader:
    """Iterable of collated batches (the 8-tuple of src/dataset.py:173-182); lengths stay on the host."""

    def __init__(self, n_batches, config, vocab, seed, device):
        self.batches = []
        for i in range(n_batches):
            b = make_batch(seed + i, config.batch_size, vocab, len(config.views), config.photo_count,
                           config.max_sent_count, config.min_sent_count, config.max_ui_sent_count,
                           config.max_sent_length, review_net_only=config.review_net_only)
            self.batches.append(b)

    def __iter__(self):
        return iter(self.batches)

    def __len__(self):
        return len(self.batches)


def main():
    extra = {"synthetic_batches": 20, "synthetic_vocab": 400003, "synthetic_emb": 50, "vgg_weights": ""}
    for k, v in extra.items():
        setattr(Config, k, v)
    config = Config()
    rank, local, world = parallel.init_distributed()
    if not torch.cuda.is_available():
        sys.exit("main.py needs an MI355X: the UMPR hot path has no CPU fallback (use the reference for CPU runs)")
    config.device = torch.device("cuda", local)
    torch.cuda.set_device(config.device)
    log = (lambda m: print(time.strftime('%Y-%m-%d %H:%M:%S'), m, flush=True)) if rank == 0 else (lambda m: None)
    log(str(config))
    have_csv = os.path.exists(os.path.join(config.data_dir, 'train.csv'))
    if have_csv:
        sys.exit("CSV/GloVe/JPEG loading (src/dataset.py, src/word2vec.py) is not built yet - see DESIGN.md section 6; "
                 "run with --data_dir synthetic")
    g = torch.Generator().manual_seed(0)
    emb = torch.randn(config.synthetic_vocab, config.synthetic_emb, generator=g) * 0.4
    emb[:3] = 0
    model = UMPR(config, emb.numpy()).to(config.device)
    per_rank = max(1, config.synthetic_batches // world)
    train_dlr = SyntheticLoader(per_rank, config, config.synthetic_vocab, 1000 + 7919 * rank, config.device)
    valid_dlr = SyntheticLoader(max(1, per_rank // 4), config, config.synthetic_vocab, 5000 + 7919 * rank, config.device)
    model_path = config.model_path or './model/umpr_synthetic.pt'
    os.makedirs(os.path.dirname(model_path) or '.', exist_ok=True)
    if not config.test_only:
        training(train_dlr, valid_dlr, model, config, model_path, logger=None if rank else type('L', (), {'info': staticmethod(log)}),
                 world=world, rank=rank)
    mse = evaluate_mse(model, valid_dlr)
    log(f"Test end, test mse is {mse:.6f}")


if __name__ == '__main__':
    main()
