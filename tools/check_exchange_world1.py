#!/usr/bin/env python3
"""One rank, RCCL backend (UMPR_REDUCE_AT_WORLD1=1): the whole gradient-exchange path - early classifier slice, per-block
buckets, remainder, early Adam step - runs with real NCCL calls whose result at world size 1 is the identity.  Three
training steps in train_step's order with the GradReducer must therefore leave the parameters BIT-IDENTICAL to three steps
without any exchange, in both forms of the exchange (async collectives on the process group's stream; in-stream sync
collectives, parallel.GradReducer._in_stream) and both arithmetic modes.  A missing stream dependency (Adam reading a
slice before its collective, the optimiser step overtaking a bucket) shows up as a difference or a NaN.

    UMPR_REDUCE_AT_WORLD1=1 python tools/check_exchange_world1.py
"""
import os
import sys

os.environ["UMPR_REDUCE_AT_WORLD1"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")

import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from umpr_amd import parallel  # noqa: E402
from umpr_amd.config import Config  # noqa: E402
from umpr_amd.model import UMPR  # noqa: E402
from umpr_amd.optim import FusedAdam  # noqa: E402
from umpr_amd.synthetic import make_batch, make_param_state  # noqa: E402


SYNC = os.environ.get("UMPR_CHECK_SYNC", "")     # diagnostic: device-wide synchronisation after "fwd" and / or "bwd"


def run(P, cfg, dev, batches, exchange):
    m = UMPR(cfg, P["embedding.weight"].numpy())
    m.load_state_dict(P)
    m = m.to(dev)
    opt = FusedAdam(m, 1e-3, 1e-3)
    red = parallel.GradReducer(opt) if exchange else None
    if exchange:
        assert red.early is not None and red.block_slices, "early slice / block buckets not found"
    losses = []
    m.eval()                                                # no dropout: both runs see the same function
    for b in batches:                                       # train_step's sequence (train.py:52-63)
        pred, loss = m(*b)
        if "fwd" in SYNC: torch.cuda.synchronize()
        opt.zero_grad()
        opt.arm_early(1.0)
        loss.backward()
        if "bwd" in SYNC: torch.cuda.synchronize()
        if red is not None:
            red.finish()
        opt.step(grad_scale=1.0)
        losses.append(float(loss))
    torch.cuda.synchronize()
    names = [(gi, n, off, k) for gi, g in enumerate(opt.groups) for n, (off, k) in g.offsets.items()]
    return [a.clone() for g in opt.groups for a in (g.p, g.m, g.v)], losses, (red, names)


def main():
    rank, local, world = parallel.init_distributed(backend="nccl")
    assert world == 1 and parallel.active()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    for dtype in ("fp32", "bf16"):
        Config.extend({"dtype": "fp32"})
        cfg = Config(argv=[])
        cfg.views = ["unknown"]
        cfg.dtype = dtype
        P = make_param_state(301, 50, 600, 1, False, m_scale=0.05)
        batches = [make_batch(310 + i, 4, 600, 1) for i in range(3)]
        os.environ.pop("UMPR_COMM_ASYNC", None)
        ref, lref, _ = run(P, cfg, dev, batches, exchange=False)
        # the run WITHOUT exchange must itself be reproducible (a consumer of the gradient arena overtaking a side stream that
        # wrote gradients in place shows up here first: umpr_amd/streams.py)
        ref2, lref2, (_, names) = run(P, cfg, dev, batches, exchange=False)
        for (gi, n, off, k) in names:
            d = float((ref[3 * gi][off:off + k] - ref2[3 * gi][off:off + k]).abs().max())
            assert d == 0.0, f"{dtype}: two runs without exchange differ in {n} by {d:.3e}"
        assert lref2 == lref
        print(f"{dtype}: three steps without exchange are reproducible bit for bit")
        for form in ("1", "0"):
            os.environ["UMPR_COMM_ASYNC"] = form
            got, lgot, (red, names) = run(P, cfg, dev, batches, exchange=True)
            for (gi, n, off, k) in names:                 # which parameters, if any
                d = float((got[3 * gi][off:off + k] - ref[3 * gi][off:off + k]).abs().max())
                if d > 0:
                    print(f"{dtype}/{form}: {n} differs by {d:.3e} (max |p| {float(ref[3 * gi][off:off + k].abs().max()):.3e})")
            instream = red._in_stream(red.opt.groups[0].g)
            assert instream == (form == "0")
            worst = max(float((a - b).abs().max()) for a, b in zip(got, ref))
            assert all(torch.isfinite(a).all() for a in got)
            assert worst == 0.0 and lgot == lref, (dtype, form, worst, lgot, lref)
            print(f"{dtype}: exchange {'in-stream' if instream else 'async'}: parameters and Adam moments after 3 steps "
                  f"bit-identical to the run without exchange (losses {lgot})")
    os.environ.pop("UMPR_COMM_ASYNC", None)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
    print("world-1 RCCL exchange check passed")


if __name__ == "__main__":
    main()
