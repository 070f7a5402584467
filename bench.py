#!/usr/bin/env python3
"""bench.py - training samples/sec of the UMPR hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          # N > 1: starts N worker processes itself (one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one resident batch: forward + backward + (RCCL gradient all-reduce) + Adam,
exactly what main.py:32-37 does per batch.  Default workload (BASELINE.json configs[1]): full UMPR, 1 view, 1 photo,
GloVe-50d-shaped table (400 003 x 50), batch 64 per GPU, fully padded S=L=L_ui=20, S_ui=5, fp32, synthetic data and
random weights (no datasets / checkpoints offline).  Weak scaling: every rank has its own batch of 64.
`--dtype bf16 --emb 300` is BASELINE.json configs[4] per GPU (bf16 MFMA conv stack + attention scores, fp32 masters).
`--review_net_only --batch 32` is configs[0] (UMPR-R).

Rank 0 prints ONE JSON line.  `roofline` describes the DOMINANT kernel of the step, timed live with HIP events on its
launch stream inside the timed region (libumpr_hip's umpr_profile_*):
  fp32 full model : the Winograd batched GEMM (wino_gemm_dma_kernel) - `achieved` = the MFMA FLOPs it EXECUTES / its time;
                    `algorithmic_frac` prices the conv forward family by direct-convolution FLOPs (Winograd layers
                    execute 1/2.25 of those) and `model_frac` the whole step by SURVEY 8(d)'s 93.73 GFLOP per sample.
                    Both are ALGORITHMIC rates and may exceed 1: the forward pass of the 56/28/14 layers executes 1/2.25
                    of its direct-convolution FLOPs (F(2x2,3x3)), the backward pass of the 112/56/28 layers 1/4
                    (F(4x4,3x3) / F(3x3,4x4)).  Only `frac` is an MFMA utilisation.
  bf16 full model : the bf16 implicit-GEMM convolution (forward launches), executed = algorithmic FLOPs.
  UMPR-R          : the recurrent GRU kernels (HBM / latency bound), algorithmic bytes / time against 8 TB/s.
`cpu_baseline` times the oracle (oracle/umpr_ref.py, the CPU restatement pinned to the reference) on this box's host
cores for a bounded sample.
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (the sparse figure is never used)
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec peak (6.3 TB/s achievable)
TEXT_MMAC_FWD = {50: 150.6, 300: 314.9}   # text path MMAC per sample forward, fully padded (cfg2 / cfg5)
VGG_MMAC_FWD = 15470.1                    # per image; train = 3 x forward; cfg2 93.73, cfg4 372.19, cfg5 94.70 GFLOP/sample
KB_PER_SAMPLE_UMPR_R = 187.0   # compulsory forward traffic of UMPR-R per sample (E = 50)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="samples per GPU")
    ap.add_argument("--views", type=int, default=1)
    ap.add_argument("--emb", type=int, default=50)
    ap.add_argument("--vocab", type=int, default=400003)
    ap.add_argument("--dtype", choices=["fp32", "bf16"], default="fp32")
    ap.add_argument("--review_net_only", action="store_true")
    ap.add_argument("--realistic", action="store_true", help="ragged lengths instead of fully padded")
    ap.add_argument("--eval", action="store_true", help="forward only (evaluate.py's path): inference samples/s")
    ap.add_argument("--h2d", action="store_true",
                    help="also time a loop whose batches arrive from pinned host memory (upload double-buffered "
                         "under the previous step), reported as h2d_inclusive; `value` stays the resident figure")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=0, help="batch of the CPU baseline sample (0: the GPU batch)")
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


def spawn_workers(args):
    """`python bench.py --gpus N` without a launcher: start N worker processes (one per GPU, fresh interpreters, this
    process never touches the GPU), give them the torch.distributed.run environment, relay rank 0's JSON line."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), LOCAL_WORLD_SIZE=str(args.gpus), UMPR_BENCH_WORKER="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = procs[0].stdout.read().decode()
    rcs = []
    deadline = time.time() + 1800
    for p in procs:
        try:
            rcs.append(p.wait(timeout=max(1, deadline - time.time())))
        except subprocess.TimeoutExpired:
            p.kill()
            rcs.append(-9)
    if any(rcs):
        for p in procs:
            if p.poll() is None:
                p.kill()
        note(f"worker exit codes {rcs}")
        sys.exit(1)
    sys.stdout.write(out0)
    sys.stdout.flush()


def pmc_traffic(bf16=False, infer=False):
    """HBM-side bytes per launch of the conv forward family from the committed rocprofv3 --pmc passes
    (profiles/README.md: `tools/pmc_traffic.sh --eval [--dtype bf16 --emb 300]`, forward only so that every conv dispatch
    is a forward launch; FETCH_SIZE doubled per the guide's gfx950 correction), averaged over the 13 forward layer calls of
    a step.  bench.py cannot run the counter passes itself, so the figure is the newest stored measurement, or None."""
    prof = os.path.join(ROOT, "profiles")
    suffix, family = ("_pmc_traffic_bf16_fwd.json", "b16_conv_family") if bf16 else ("_pmc_traffic_fwd.json", "igemm_family")
    if infer and not bf16:      # the fp32 inference forward runs the 4x4 Winograd tile: its own stored pass
        suffix = "_pmc_traffic_infer.json"
    # newest first: record tags run r02_a .. r02_z, r02_aa .. (longer tag = later)
    for name in sorted((f for f in os.listdir(prof) if f.endswith(suffix) and ("bf16" in f) == bf16),
                       key=lambda f: (len(f[:-len(suffix)]), f), reverse=True):
        try:
            d = json.load(open(os.path.join(prof, name)))
            f = d["families"][family]
            per_step = (2.0 * f["FETCH_SIZE"] + f["WRITE_SIZE"]) * 1024.0 / d["steps"]
            return per_step / 13.0, f"bytes/launch from profiles/{name} (batch 64, 13 forward launches/step)"
        except Exception:
            continue
    return None, "no stored PMC pass"


def cpu_baseline(args, P, rank):
    import torch
    from oracle import umpr_ref as R  # the checker, timed as the CPU baseline ("port")
    from umpr_amd.synthetic import make_batch
    cores = host_cores()
    torch.set_num_threads(cores)
    if args.cpu_batch <= 0:
        args.cpu_batch = args.batch
    note(f"cpu baseline on {cores} host threads, batch {args.cpu_batch}")
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
                      f"batch {args.cpu_batch} in fp32, oracle/umpr_ref.py with the reference's ATen calls, "
                      f"torch.set_num_threads({cores})"}


class HostFeeder:
    """The reference's forward owns the upload (src/model.py:259-260).  Two pinned host copies of the batch alternate;
    while step k computes, batch k+1 travels on a copy stream into the other device slot; the compute stream waits for
    the copy event, the copy stream waits until the slot's previous consumer has finished."""

    def __init__(self, batch, dev):
        import torch
        self.dev = dev
        self.copy = torch.cuda.Stream(dev)
        self.host = [tuple(t.pin_memory() if isinstance(t, torch.Tensor) and t.numel() else t for t in batch) for _ in range(2)]
        self.slot = [None, None]
        self.ready = [torch.cuda.Event(), torch.cuda.Event()]
        self.free = [torch.cuda.Event(), torch.cuda.Event()]
        self.k = 0
        self._issue(0)

    def _issue(self, i):
        import torch
        u, it, ui, ul, il, uil, photos, labels = self.host[i]
        with torch.cuda.stream(self.copy):
            self.copy.wait_event(self.free[i])
            self.slot[i] = (u.to(self.dev, non_blocking=True), it.to(self.dev, non_blocking=True),
                            ui.to(self.dev, non_blocking=True), ul, il, uil, photos.to(self.dev, non_blocking=True),
                            labels.to(self.dev, non_blocking=True))
            self.ready[i].record(self.copy)

    def next(self):
        import torch
        i = self.k & 1
        self.k += 1
        torch.cuda.current_stream().wait_event(self.ready[i])
        b = self.slot[i]
        self._issue(i ^ 1)
        return b, i

    def done(self, i):
        import torch
        self.free[i].record(torch.cuda.current_stream())


def main():
    args = parse()
    # UMPR_BENCH_FORCE_SPAWN=1 rehearses the self-launch path with a single worker on a one-GPU box
    if (args.gpus > 1 or os.environ.get("UMPR_BENCH_FORCE_SPAWN") == "1") and "WORLD_SIZE" not in os.environ:
        return spawn_workers(args)           # before anything here imports or touches the GPU runtime
    import torch
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

    Config.extend({"dtype": "fp32"})
    cfg = Config(argv=[])
    cfg.review_net_only = args.review_net_only
    cfg.views = ["v%d" % i for i in range(args.views)]
    cfg.dtype = args.dtype
    P = make_param_state(0, args.emb, args.vocab, args.views, args.review_net_only)
    model = UMPR(cfg, P["embedding.weight"].numpy())
    model.load_state_dict(P)
    model = model.to(dev)
    opt = FusedAdam(model, cfg.learning_rate, cfg.l2_regularization, cfg.lr_decay)
    reducer = parallel.GradReducer(opt) if parallel.active() else None
    host_batch = make_batch(1234 + rank, args.batch, args.vocab, args.views, review_net_only=args.review_net_only,
                            full_pad=not args.realistic)
    u, i_, ui, ul, il, uil, photos, labels = host_batch
    # lengths stay on the host, like the reference (src/model.py:18)
    batch = (u.to(dev), i_.to(dev), ui.to(dev), ul, il, uil, photos.to(dev), labels.to(dev))
    loss_sum = torch.zeros((), device=dev)

    def barrier():
        if parallel.active():
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def step(b=batch):
        if not args.eval:
            return train_step(model, opt, b, world, reducer)
        with torch.no_grad():
            model.eval()
            return model(*b)

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
    t_issue = time.perf_counter() - t0        # host time to ENQUEUE the steps (no wait): close to dt => host-bound
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
                      ("wino_wgrad_gemm", 6), ("gru", 3), ("conv_bf16_fwd", 7), ("conv_bf16_dgrad", 8),
                      ("conv_bf16_wgrad", 9)):
        ms, work, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_long()
        if L.fn["umpr_profile_read"](idx, ctypes.byref(ms), ctypes.byref(work), ctypes.byref(n)) == 0:
            fam[name] = (ms.value, work.value, n.value)

    h2d = None
    if args.h2d and not args.eval:
        feeder = HostFeeder(host_batch, dev)
        for _ in range(2):
            b, slot = feeder.next()
            step(b)
            feeder.done(slot)
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            b, slot = feeder.next()
            step(b)
            feeder.done(slot)
        barrier()
        dth = time.perf_counter() - t1
        if parallel.active():
            t = torch.tensor([dth], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dth = float(t.item())
        nbytes = sum(t.numel() * t.element_size() for t in (u, i_, ui, photos, labels))
        h2d = {"value": world * args.batch * args.steps / dth, "unit": "samples/s", "ms_per_step": 1e3 * dth / args.steps,
               "host_bytes_per_step": nbytes,
               "note": "every step's ids/photos/labels are uploaded from pinned host memory on a copy stream, "
                       "double-buffered under the previous step (the reference uploads inside forward, src/model.py:259-260)"}

    if rank == 0:
        value = world * args.batch * args.steps / dt
        per_s = lambda v: (v[1] / (v[0] * 1e-3) / 1e12) if v[0] > 0 else 0.0   # work per second / 1e12
        full = not args.review_net_only
        if full and args.dtype == "fp32":
            ms, work, n = fam["wino_gemm"]          # dominant kernel: executed MFMA FLOPs (16 GEMMs of M x C x tiles)
            achieved = per_s(fam["wino_gemm"])
            fwd = fam["conv3x3_fwd"]
            traffic, traffic_note = (pmc_traffic(infer=args.eval) if args.batch == 64 and args.views == 1
                                     else (None, "stored PMC pass is for the batch-64, 1-view workload"))
            roof = {"bound": "mfma", "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": traffic, "traffic_note": traffic_note
                    + "; the figure is per conv-forward layer call (transforms + GEMM), of which this kernel is the GEMM",
                    "kernel": "wino_gemm_dma_kernel (Winograd batched GEMM: 16 planes of F(2x2,3x3) for the forward pass of the "
                              "56/28/14 layers, 36 planes of F(4x4,3x3) for the data gradient of the 112/56/28 layers; "
                              "largest share of GPU time).  achieved = MFMA FLOPs it executes / its "
                              "HIP-event time on its launch stream; in the timed region it shares the chip with the "
                              "weight-gradient stream and the text stream",
                    "launches": n, "avg_launch_ms": ms / max(n, 1),
                    "executed_gflop_per_launch": work / max(n, 1) / 1e9,
                    "algorithmic_frac": per_s(fwd) / PEAK_F32_MFMA_TFLOPS,
                    "algorithmic_note": "conv forward family (13 layer calls/step) priced by direct-convolution FLOPs "
                                        "2*N*H*W*Cout*Cin*9; 9 of the 13 run Winograd and execute 1/2.25 of that"}
        elif full:
            ms, work, n = fam.get("conv_bf16_fwd", (0.0, 0.0, 0))
            achieved = per_s((ms, work, n))
            traffic, traffic_note = (pmc_traffic(True) if args.batch == 64 and args.views == 1
                                     else (None, "stored PMC pass is for the batch-64, 1-view workload"))
            roof = {"bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / PEAK_BF16_MFMA_TFLOPS, "traffic": traffic,
                    "traffic_note": traffic_note + "; at the measured launch time this is < 2 TB/s of HBM traffic, "
                                    "close to the compulsory activation bytes: the kernel is bound by the matrix pipe",
                    "kernel": "conv3x3_bf16_kernel forward launches (implicit GEMM on v_mfma_f32_16x16x32_bf16, padded "
                              "NHWC bf16 activations); executed FLOPs = algorithmic FLOPs over the padded pixel grid",
                    "launches": n, "avg_launch_ms": ms / max(n, 1), "executed_gflop_per_launch": work / max(n, 1) / 1e9}
        else:
            ms, work, n = fam["gru"]
            achieved = (work / (ms * 1e-3) / 1e9) if ms > 0 else 0.0
            roof = {"bound": "hbm", "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": achieved / PEAK_HBM_GBS, "traffic": None,
                    "kernel": "gru_fwd16_kernel / gru_bwd16_kernel (recurrent part of the packed BiGRU, 16-sequence tiles): algorithmic bytes "
                              "(gx, out, saved gates, dgx) / HIP-event time.  The kernel is latency-bound - a chain of "
                              "<= 20 dependent steps per sequence tile - not bandwidth-bound, so the fraction is small "
                              "by construction", "launches": n, "avg_launch_ms": ms / max(n, 1),
                    "step_bytes_frac": value * KB_PER_SAMPLE_UMPR_R * 1e3 * (1 if args.eval else 3) / 1e9 / PEAK_HBM_GBS,
                    "step_bytes_note": "whole step: SURVEY 8(d) compulsory traffic 187 KB/sample forward (x3 for a "
                                       "training step) x samples/s against 8 TB/s"}
        out = {
            "metric": "inference samples/sec" if args.eval else "training samples/sec", "value": value, "unit": "samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.dtype == "fp32" else "bf16", "data": "synthetic",
            "config": {
                "workload": ("UMPR-R (review_net_only)" if args.review_net_only else f"full UMPR, {args.views} view(s), 1 photo/view")
                            + f", GloVe-{args.emb}d-shaped table {args.vocab}x{args.emb}, batch {args.batch}/GPU, "
                            + ("ragged lengths" if args.realistic else "fully padded S=L=L_ui=20 S_ui=5")
                            + (", fp32" if args.dtype == "fp32" else ", bf16 MFMA conv stack + attention scores with fp32 "
                               "accumulation, fp32 master weights / classifier / GRU gates / Adam")
                            + ", fwd+bwd+Adam(+RCCL all-reduce), random-init weights",
                "global_batch": world * args.batch, "parallelism": f"dp{world}"},
            "roofline": roof,
            "kernels": {k: {"ms_per_step": v[0] / args.steps, ("gbytes_per_s" if k == "gru" else "tflops"):
                            ((v[1] / (v[0] * 1e-3) / 1e9 if v[0] > 0 else 0.0) if k == "gru" else per_s(v)),
                            "launches_per_step": v[2] / args.steps} for k, v in fam.items() if v[2]},
            "loss_mean": float(loss_sum.item()) / args.steps,
            "host_issue_ms_per_step": 1e3 * t_issue / args.steps,
        }
        if full:
            gf = 3 * 2 * (TEXT_MMAC_FWD.get(args.emb, 150.6) + args.views * VGG_MMAC_FWD) / 1e3   # GFLOP per sample, training
            out["model_tflops"] = value * (gf / 3.0 if args.eval else gf) / 1e3 / world
            out["roofline"]["model_frac"] = out["model_tflops"] / (PEAK_F32_MFMA_TFLOPS if args.dtype == "fp32" else PEAK_BF16_MFMA_TFLOPS)
            if args.dtype == "fp32":
                out["roofline"]["model_note"] = ("algorithmic (direct-convolution) FLOPs of the whole step / time / peak; it may "
                                                 "exceed 1 because the Winograd layers execute 1/2.25 (forward) or 1/4 (backward) "
                                                 "of them - `frac` is the utilisation figure")
        if h2d is not None:
            out["h2d_inclusive"] = h2d
        if world == 1 and not args.no_cpu_baseline and not args.eval:
            out["cpu_baseline"] = cpu_baseline(args, P, rank)
        result_out.write(json.dumps(out) + "\n")
        result_out.flush()
    if parallel.active():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
