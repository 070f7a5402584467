#!/usr/bin/env python3
"""bench.py - training samples/sec of the UMPR hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one resident batch: forward + backward + (RCCL gradient all-reduce) + Adam,
exactly what main.py:32-37 does per batch.  Workload (BASELINE.json configs[1]): full UMPR, 1 view, 1 photo,
GloVe-50d-shaped table (400 003 x 50), batch 64 per GPU, fully padded S=L=L_ui=20, S_ui=5, fp32, synthetic data and
random weights (no datasets / checkpoints offline).  Weak scaling: every rank has its own batch of 64.

Rank 0 prints ONE JSON line; `roofline` is the conv3x3 implicit-GEMM kernel (forward+dgrad launches) timed live with
HIP events on its launch stream inside the timed region (libumpr_hip's umpr_profile_*), algorithmic FLOPs
2*N*H*W*Cout*Cin*9 per launch; `cpu_baseline` times the oracle (oracle/umpr_ref.py, the CPU restatement pinned to
the reference) on this box's host cores for a bounded sample.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
GFLOP_PER_SAMPLE_TRAIN = 93.73  # SURVEY.md 8(d), cfg2 fully padded


def pmc_traffic(eval_mode):
    """HBM-side bytes per launch of the conv forward family from the committed rocprofv3 --pmc passes
    (profiles/README.md: `tools/pmc_traffic.sh --eval`, forward only so that every conv dispatch is a forward launch;
    FETCH_SIZE doubled per the guide's gfx950 correction), averaged over the 13 forward layer calls of a step.
    bench.py cannot run the counter passes itself, so the figure is the stored measurement, or None when absent."""
    path = os.path.join(ROOT, "profiles", "r01_i_pmc_traffic_fwd.json")
    try:
        d = json.load(open(path))
        f = d["families"]["igemm_family"]
        per_step = (2.0 * f["FETCH_SIZE"] + f["WRITE_SIZE"]) * 1024.0 / d["steps"]
        return per_step / 13.0, "bytes/launch from profiles/r01_i_pmc_traffic_fwd.json (batch 64, 13 forward launches/step)"
    except Exception:
        return None, "no stored PMC pass"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="samples per GPU")
    ap.add_argument("--views", type=int, default=1)
    ap.add_argument("--emb", type=int, default=50)
    ap.add_argument("--vocab", type=int, default=400003)
    ap.add_argument("--review_net_only", action="store_true")
    ap.add_argument("--realistic", action="store_true", help="ragged lengths instead of fully padded")
    ap.add_argument("--eval", action="store_true", help="forward only (evaluate.py's path): inference samples/s")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=16)
    ap.add_argument("--cpu-steps", type=int, default=2)
    return ap.parse_args()


def host_cores():
    """CPU share of this process: scheduler affinity capped by the cgroup quota (os.cpu_count() reports the whole host)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 64))


def note(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def to_device(batch, dev):
    u, i, ui, ul, il, uil, photos, labels = batch
    # lengths stay on the host, like the reference (src/model.py:18)
    return (u.to(dev), i.to(dev), ui.to(dev), ul, il, uil, photos.to(dev), labels.to(dev))


def cpu_baseline(args, P, rank):
    from oracle import umpr_ref as R  # the checker, timed as the CPU baseline ("port")
    from umpr_amd.synthetic import make_batch
    cores = host_cores()
    torch.set_num_threads(cores)
    note(f"cpu baseline on {cores} host threads")
    Pc = {k: v.clone() for k, v in P.items()}
    for k, p in Pc.items():
        if k != "embedding.weight":
            p.requires_grad_(True)
    opt = R.adam_reference(Pc, 1e-6, 1e-3)
    batch = make_batch(1234 + rank, args.cpu_batch, args.vocab, args.views, review_net_only=args.review_net_only,
                       full_pad=not args.realistic)

    def step():
        _, loss = R.umpr_forward(Pc, batch, review_net_only=args.review_net_only, train=True, aten=True)
        opt.zero_grad()
        loss.backward()
        opt.step()
    step()
    t0 = time.perf_counter()
    for _ in range(args.cpu_steps):
        step()
    dt = time.perf_counter() - t0
    return {"value": args.cpu_batch * args.cpu_steps / dt, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"{args.cpu_steps} timed train steps (fwd+bwd+Adam, after 1 warm-up) of the same workload at "
                      f"batch {args.cpu_batch}, oracle/umpr_ref.py with the reference's ATen calls, "
                      f"torch.set_num_threads({cores})"}


def main():
    args = parse()
    # The contract is ONE JSON line on stdout.  RCCL prints a version banner to the C-level stdout when the first
    # communicator is created, so everything but the result line is sent to stderr: fd 1 is pointed at fd 2 and the
    # JSON goes to a private duplicate of the original stdout.
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    from umpr_amd import parallel
    from umpr_amd._lib import lib
    from umpr_amd.config import Config
    from umpr_amd.model import UMPR
    from umpr_amd.optim import FusedAdam
    from umpr_amd.synthetic import make_batch, make_param_state
    from umpr_amd.train import train_step

    torch.set_num_threads(host_cores())
    torch.manual_seed(0)  # dropout masks derive from torch.initial_seed()
    rank, local, world = parallel.init_distributed()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    L = lib()

    cfg = Config(argv=[])
    cfg.review_net_only = args.review_net_only
    cfg.views = ["v%d" % i for i in range(args.views)]
    P = make_param_state(0, args.emb, args.vocab, args.views, args.review_net_only)
    model = UMPR(cfg, P["embedding.weight"].numpy())
    model.load_state_dict(P)
    model = model.to(dev)
    opt = FusedAdam(model, cfg.learning_rate, cfg.l2_regularization, cfg.lr_decay)
    reducer = parallel.GradReducer(opt) if parallel.active() else None
    batch = to_device(make_batch(1234 + rank, args.batch, args.vocab, args.views, review_net_only=args.review_net_only,
                                 full_pad=not args.realistic), dev)
    loss_sum = torch.zeros((), device=dev)

    def barrier():
        if parallel.active():
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def step():
        if not args.eval:
            return train_step(model, opt, batch, world, reducer)
        with torch.no_grad():
            model.eval()
            return model(*batch)

    note("model and batch resident; warm-up")
    for _ in range(args.warmup):
        step()
    barrier()
    note("timed region")
    L.fn["umpr_profile_reset"]()
    L.fn["umpr_profile_enable"](1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        _, loss = step()
        loss_sum += loss.detach()
    barrier()
    dt = time.perf_counter() - t0
    L.fn["umpr_profile_enable"](0)
    note(f"{args.steps} steps in {dt:.3f} s")
    if parallel.active():
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    fam = {}
    for name, idx in (("conv3x3_fwd", 0), ("conv3x3_dgrad", 5), ("conv3x3_wgrad", 1), ("gemm_f32", 2), ("wino_gemm", 4),
                      ("wino_wgrad_gemm", 6)):
        ms, work, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_long()
        L.fn["umpr_profile_read"](idx, ctypes.byref(ms), ctypes.byref(work), ctypes.byref(n))
        fam[name] = (ms.value, work.value, n.value)

    if rank == 0:
        value = world * args.batch * args.steps / dt
        ms, work, n = fam["conv3x3_fwd"]
        achieved = work / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        # Winograd layers execute 1/2.25 of the direct-convolution FLOPs `achieved` counts for them;
        # the same nine layers take the Winograd path in forward and in dgrad, so half of family 4's work is forward
        executed = (work - 1.25 * 0.5 * fam["wino_gemm"][1]) / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        traffic, traffic_note = (pmc_traffic(args.eval) if args.batch == 64 and args.views == 1 and not args.review_net_only
                                 else (None, "stored PMC pass is for the batch-64, 1-view workload"))
        out = {
            "metric": "inference samples/sec" if args.eval else "training samples/sec", "value": value, "unit": "samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": ("UMPR-R (review_net_only)" if args.review_net_only else f"full UMPR, {args.views} view(s), 1 photo/view")
                            + f", GloVe-{args.emb}d-shaped table {args.vocab}x{args.emb}, batch {args.batch}/GPU, "
                            + ("ragged lengths" if args.realistic else "fully padded S=L=L_ui=20 S_ui=5")
                            + ", fp32, fwd+bwd+Adam(+RCCL all-reduce), random-init weights",
                "global_batch": world * args.batch, "parallelism": f"dp{world}"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": traffic,
                         "traffic_note": traffic_note, "executed_tflops": executed,
                         "kernel": "conv3x3 forward family (conv3x3_igemm_v2_kernel on 224/112 maps; wino_input / "
                                   "wino_gemm / wino_output kernels, Winograd F(2x2,3x3), on 56/28/14 maps); its dgrad "
                                   "and wgrad launches overlap on two streams and are listed under kernels", "launches": n,
                         "avg_launch_ms": ms / max(n, 1), "algorithmic_gflop_per_launch": work / max(n, 1) / 1e9},
            "kernels": {k: {"ms_per_step": v[0] / args.steps, "tflops": (v[1] / (v[0] * 1e-3) / 1e12 if v[0] > 0 else 0.0),
                            "launches_per_step": v[2] / args.steps} for k, v in fam.items()},
            "loss_mean": float(loss_sum.item()) / args.steps,
        }
        if not args.review_net_only:
            out["model_tflops"] = value * (GFLOP_PER_SAMPLE_TRAIN / 3.0 if args.eval else GFLOP_PER_SAMPLE_TRAIN) / 1e3 / world
        if world == 1 and not args.no_cpu_baseline and not args.eval:
            out["cpu_baseline"] = cpu_baseline(args, P, rank)
        result_out.write(json.dumps(out) + "\n")
        result_out.flush()
    if parallel.active():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
