#!/usr/bin/env python3
"""Entry point mirroring the reference's main.py (config surface: config.py, run recipes: readme.md:70-92).

    python main.py --data_dir synthetic --batch_size 64                       # single MI355X
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 main.py --data_dir synthetic

The model / optimiser / train / evaluate drivers are the MI355X-native ones (umpr_amd).  With a data_dir that holds
train.csv / valid.csv / test.csv / photos.json (the reference's layout, readme.md:41-60) the CSV + GloVe + JPEG
pipeline of umpr_amd/data.py feeds them; `--data_dir synthetic` trains on synthetic batches of exactly the layout
`batch_loader` produces (src/dataset.py:173-182), which is what the throughput numbers are quoted on.  Multi-GPU: every
rank collates the same global batch and keeps its contiguous chunk (DataParallel's scatter, main.py:82).
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


class ShardedLoader:
    """DataLoader whose collate already cut every batch to this rank's contiguous chunk (DataParallel's scatter,
    main.py:82) - see umpr_amd.data.batch_loader(shard=...): indices are sharded BEFORE the photos are decoded.  Yields
    parallel.Shard tuples: the 8 tensors plus `.n_active`; a chunk may be empty when the last batch is short."""

    def __init__(self, loader, rank, world):
        self.loader, self.rank, self.world = loader, rank, world

    def __iter__(self):
        for b in self.loader:
            if self.world <= 1:
                yield b
                continue
            s = parallel.Shard(b[:8])
            s.n_active = int(b[8])
            yield s

    def __len__(self):
        return len(self.loader)


class _Collate:
    """Picklable collate (worker processes): src/dataset.py:173-182 via umpr_amd.data.batch_loader."""

    def __init__(self, ignore_photos, rank=0, world=1):
        self.ignore_photos = ignore_photos
        self.shard = (rank, world) if world > 1 else None

    def __call__(self, samples):
        from umpr_amd.data import batch_loader
        return batch_loader(samples, self.ignore_photos, shard=self.shard)


def _agree(value, rank):
    """Rank 0's value on every rank (a timestamped default path must be the same file everywhere)."""
    if parallel.active():
        box = [value if rank == 0 else None]
        torch.distributed.broadcast_object_list(box, src=0)
        return box[0]
    return value


def run_real(config, rank, world, log):
    """main.py:64-99 of the reference: Word2vec -> Dataset -> DataLoader(collate) -> training -> test."""
    from torch.utils.data import DataLoader
    from umpr_amd.checkpoint import load_checkpoint
    from umpr_amd.data import Dataset, Word2vec
    d = config.data_dir
    photo_path, photo_json = os.path.join(d, 'photos'), os.path.join(d, 'photos.json')
    # like the reference (main.py:116-119) the default checkpoint name carries the start time, so a run that never
    # saves cannot pick up an older run's file in its test phase
    model_path = config.model_path or _agree(
        f"./model/{os.path.basename(d.strip('/'))}{time.strftime('%Y%m%d_%H%M%S')}.pt", rank)
    if config.test_only and not os.path.exists(model_path):
        log(f'{model_path} is not exist! (--test_only needs --model_path <trained checkpoint>)')
        sys.exit(-1)                                   # the reference exits here too (main.py:89-91)
    w2v = Word2vec(config.word2vec_file)
    collate = _Collate(config.review_net_only, rank, world)
    workers = max(0, int(getattr(config, "loader_workers", 0)))
    # decode + resize + collate in worker processes, pinned staging buffers, batches prefetched ahead of the GPU; with
    # loader_workers = 0 everything runs on the training thread, as in the reference (main.py:70-73)
    dl = dict(collate_fn=collate, num_workers=workers, pin_memory=True)
    if workers:
        dl.update(prefetch_factor=2, persistent_workers=True)
    model = UMPR(config, w2v.embedding).to(config.device)
    os.makedirs(os.path.dirname(model_path) or '.', exist_ok=True)
    logger = None if rank else type('L', (), {'info': staticmethod(log)})
    saved = False
    if not config.test_only:
        train_data = Dataset(os.path.join(d, 'train.csv'), photo_json, photo_path, w2v, config)
        valid_data = Dataset(os.path.join(d, 'valid.csv'), photo_json, photo_path, w2v, config)
        log(f'Training dataset contains {len(train_data)} samples.')
        g = torch.Generator().manual_seed(0)  # same shuffle on every rank
        train_dlr = ShardedLoader(DataLoader(train_data, batch_size=config.batch_size, shuffle=True, generator=g, **dl),
                                  rank, world)
        valid_dlr = ShardedLoader(DataLoader(valid_data, batch_size=config.batch_size, **dl), rank, world)
        _, saved = training(train_dlr, valid_dlr, model, config, model_path, logger=logger, world=world, rank=rank)
    if config.test_only or saved:
        load_checkpoint(model_path, model, map_location=config.device)
        log(f'Testing the checkpoint {model_path}')
    else:
        log('No checkpoint was written in this run (validation never improved at a multiple of the validation '
            'interval): testing the model as it stands at the end of training')
    test_data = Dataset(os.path.join(d, 'test.csv'), photo_json, photo_path, w2v, config)
    test_dlr = ShardedLoader(DataLoader(test_data, batch_size=config.batch_size, **dl), rank, world)
    log(f"Test end, test mse is {evaluate_mse(model, test_dlr):.6f}")


def main():
    try:
        _main()
    except SystemExit:
        raise
    except BaseException:
        # A rank that dies inside a step leaves its peers blocked in the next RCCL collective: report, then leave with a
        # non-zero status at once (no orderly teardown that would itself wait for the peers) so that the launcher
        # (torch.distributed.run) tears the job down.
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        if parallel.active():
            os._exit(1)
        raise


def _main():
    extra = {"synthetic_batches": 20, "synthetic_vocab": 400003, "synthetic_emb": 50, "vgg_weights": "", "resume": "",
             "loader_workers": 0, "valid_every": 500, "dtype": "fp32"}
    Config.extend(extra)
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
        return run_real(config, rank, world, log)
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
