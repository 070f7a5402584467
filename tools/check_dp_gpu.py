#!/usr/bin/env python3
"""Two ranks sharing the one GPU of a test box (gloo transport, CUDA tensors) drive the real model through
umpr_amd.train.train_step with the overlapped GradReducer; rank 0 then replays both shards on a single model copy and
checks that the data-parallel parameters equal that reference.  Exercises shard_batch, the early gradient bucket with
in-place VGG gradients, finish(), and the 1/world scale inside the Adam kernel on real kernels.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 tools/check_dp_gpu.py
"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from umpr_amd import parallel  # noqa: E402
from umpr_amd.config import Config  # noqa: E402
from umpr_amd.model import UMPR  # noqa: E402
from umpr_amd.optim import FusedAdam  # noqa: E402
from umpr_amd.synthetic import make_batch, make_param_state  # noqa: E402
from umpr_amd.train import train_step  # noqa: E402


def build(P, cfg, dev):
    m = UMPR(cfg, P["embedding.weight"].numpy())
    m.load_state_dict(P)
    return m.to(dev)


def main():
    rank, local, world = parallel.init_distributed(backend="gloo")
    assert world == 2
    dev = torch.device("cuda", 0)           # both ranks on the box's single GPU
    torch.cuda.set_device(dev)
    torch.manual_seed(0)
    cfg = Config(argv=[])
    cfg.views = ["unknown"]
    P = make_param_state(201, 50, 600, 1, False, m_scale=0.05)
    # two even batches, then a short "last batch" of 3 (chunks [2, 1]) and one of a single sample (chunks [1, 0]:
    # rank 1 has nothing to do but must join the collectives - ADVICE r1)
    steps = [make_batch(210 + i, b, 600, 1) for i, b in enumerate((4, 4, 3, 1))]
    model = build(P, cfg, dev).eval()       # eval: no dropout, the reference replay sees the same function
    opt = FusedAdam(model, 1e-3, 1e-3)
    red = parallel.GradReducer(opt)
    from umpr_amd.optim import has_callback
    assert red.early is not None and has_callback(model.visual_net.vgg16[0], red._written_in_place)
    fired = []
    for b in steps:
        mine = parallel.shard_with_count(b, rank, world)
        if mine[0].shape[0] == 0:
            opt.zero_grad()
            red.skip_backward()
            fired.append(red.fired)
            red.finish()
            opt.step(grad_scale=1.0 / mine.n_active)
            continue
        model.eval()
        pred, loss = model(*mine)
        opt.zero_grad()
        opt.arm_early(1.0 / mine.n_active)      # as train_step does: the classifier slice is updated behind its all-reduce
        loss.mean().backward()
        fired.append(red.fired)
        red.finish()
        opt.step(grad_scale=1.0 / mine.n_active)
    assert all(fired), "the early bucket did not start during backward"
    if rank == 0:
        ref = build(P, cfg, dev).eval()
        ropt = FusedAdam(ref, 1e-3, 1e-3)
        for b in steps:
            total = None
            n_active = parallel.active_shards(b[0].shape[0], world)
            for r in range(n_active):
                shard = parallel.shard_batch(b, r, world)
                ropt.zero_grad()
                ref(*shard)[1].mean().backward()
                gs = [a.clone() for a in ropt.grad_arenas()]
                total = gs if total is None else [t + g for t, g in zip(total, gs)]
            for a, t in zip(ropt.grad_arenas(), total):
                a.copy_(t)
            for g in ropt.groups:
                for p in g.direct:
                    p._umpr_fresh = False
            ropt.step(grad_scale=1.0 / n_active)
        worst = 0.0
        for (k, a), (_, c) in zip(model.state_dict().items(), ref.state_dict().items()):
            d = float((a - c).abs().max())
            worst = max(worst, d / (float(c.abs().max()) + 1e-30))
            assert torch.allclose(a, c, rtol=1e-6, atol=1e-9), (k, d)
        print(f"data-parallel parameters equal the sequential replay (worst relative difference {worst:.2e})", flush=True)
        print("short last batch (3 samples -> chunks [2,1]; 1 sample -> chunks [1,0], rank 1 idle but in every collective): ok", flush=True)
    # evaluation: every rank scores its shard, (sum of squared errors, count) are all-reduced (src/evaluate.py:6-14)
    from umpr_amd.train import evaluate_mse
    mse = evaluate_mse(model, [parallel.shard_with_count(b, rank, world) for b in steps])
    if rank == 0:
        se, n = 0.0, 0
        with torch.no_grad():   # shard by shard: the reference's sentence permutation couples the samples of a batch
            for b in steps:     # (SURVEY.md header fact 1), so a shard's predictions are not the full batch's
                for r in range(parallel.active_shards(b[0].shape[0], world)):
                    sh = parallel.shard_batch(b, r, world)
                    pred, _ = model(*sh)
                    se += float(((pred.cpu() - sh[-1]) ** 2).sum())
                    n += len(pred)
        assert abs(mse - se / n) < 1e-5, (mse, se / n)
        print(f"two-rank evaluate_mse {mse:.6f} == single-process {se / n:.6f}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
