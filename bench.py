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
A default invocation (no workload flag, one GPU) also runs the other single-GPU workloads BASELINE.json names and reports them
under `other_configs` (bf16 GloVe-300d batch 64 with its own roofline, UMPR-R batch 32 eager and as one hipGraph launch per step,
4 views x 32 samples in fp32 and bf16, inference fp32 / bf16): 8 timed steps each; `--no-other-configs` skips them.
`python bench.py --gpus N` without a launcher starts N workers, ends all of them when one dies and retries once with the async
form of the gradient exchange if the in-stream form failed (spawn_workers).
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
    ap.add_argument("--fwd-train", action="store_true",
                    help="measurement aid (tools/pmc_traffic.sh): a step is the TRAINING forward alone (grad enabled, no backward), so "
                         "that every convolution dispatch of a counter pass is a forward launch of the algorithm training uses")
    ap.add_argument("--h2d", action="store_true",
                    help="also time a loop whose batches arrive from pinned host memory (upload double-buffered "
                         "under the previous step), reported as h2d_inclusive; `value` stays the resident figure")
    ap.add_argument("--graph", action="store_true",
                    help="UMPR-R only: the training step as one captured hipGraph launch (umpr_amd/graphs.py)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="default invocation only: skip the `other_configs` sub-lines (bf16 GloVe-300d, UMPR-R, 4 views, inference)")
    ap.add_argument("--other-steps", type=int, default=12, help="timed steps of each `other_configs` sub-line")
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


def _tail(path, n=30):
    try:
        with open(path, "rb") as f:
            return b"\n".join(f.read().splitlines()[-n:]).decode(errors="replace")
    except OSError:
        return ""


def _run_worker_set(n, argv, cmd, extra_env, deadline_s, poll_s=0.2):
    """Start n workers (rank r gets RANK/LOCAL_RANK=r ...), poll ALL of them; the first non-zero exit (or the deadline) kills
    the rest at once.  Returns (ok, rank-0 stdout, {"rank", "rc", "stderr_tail", "timed_out"} of the first failure or None)."""
    import shutil
    import socket
    import tempfile
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    tmp = tempfile.mkdtemp(prefix="umpr_bench_")
    procs, errs, outs = [], [], []
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), LOCAL_WORLD_SIZE=str(n), UMPR_BENCH_WORKER="1")
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            env.update(extra_env)
            errs.append(os.path.join(tmp, f"rank{r}.err"))
            outs.append(os.path.join(tmp, f"rank{r}.out"))
            procs.append(subprocess.Popen(cmd + argv, env=env, stdout=open(outs[-1], "wb"), stderr=open(errs[-1], "wb")))
        t0 = time.time()
        failure, last_note, shown = None, t0, 0
        while failure is None:
            rcs = [p.poll() for p in procs]
            bad = [r for r, rc in enumerate(rcs) if rc not in (None, 0)]
            if bad:
                failure = {"rank": bad[0], "rc": rcs[bad[0]], "timed_out": False}
                break
            if all(rc == 0 for rc in rcs):
                break
            now = time.time()
            if now - t0 > deadline_s:
                failure = {"rank": next(r for r, rc in enumerate(rcs) if rc is None), "rc": None, "timed_out": True}
                break
            if now - last_note > 30:     # a line now and then: a silent launcher looks hung from outside
                last_note = now
                note(f"{sum(rc is None for rc in rcs)} of {n} workers running, {now - t0:.0f} s")
            # relay rank 0's progress notes as they appear
            try:
                with open(errs[0], "rb") as f:
                    f.seek(shown)
                    chunk = f.read()
                if chunk:
                    shown += len(chunk)
                    sys.stderr.write(chunk.decode(errors="replace"))
                    sys.stderr.flush()
            except OSError:
                pass
            time.sleep(poll_s)
        if failure is not None:
            for p in procs:            # peers of a dead rank sit in a collective until the RCCL timeout: end them now
                if p.poll() is None:
                    p.kill()
            for p in procs:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    pass
            failure["stderr_tail"] = _tail(errs[failure["rank"]])
            return False, "", failure
        with open(errs[0], "rb") as f:
            f.seek(shown)
            sys.stderr.write(f.read().decode(errors="replace"))
        return True, open(outs[0]).read(), None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def spawn_workers(args, argv=None, cmd=None, deadline_s=None):
    """`python bench.py --gpus N` without a launcher: start N worker processes (one per GPU, fresh interpreters, this
    process never touches the GPU), give them the torch.distributed.run environment, relay rank 0's JSON line.
    Fail fast: all workers are polled; when one exits non-zero the others are killed at once and the launcher exits
    non-zero with that rank's stderr tail (a rank blocked in an RCCL collective would otherwise sit out the NCCL timeout).
    Fallback: if the first attempt ran the in-stream form of the gradient exchange (parallel.GradReducer._in_stream) and a
    worker ended with an ordinary error status - a Python exception, not a signal and not the deadline - ONE fresh set of
    workers is started with UMPR_COMM_ASYNC=1; the JSON line then carries `comm_fallback`.  A worker killed by a signal (GPU
    fault) or by the deadline is never retried."""
    argv = list(sys.argv[1:] if argv is None else argv)
    cmd = list(cmd) if cmd is not None else [sys.executable, os.path.abspath(__file__)]
    deadline_s = float(os.environ.get("UMPR_BENCH_DEADLINE_S", "900")) if deadline_s is None else deadline_s
    ok, out0, fail = _run_worker_set(args.gpus, argv, cmd, {}, deadline_s)
    if not ok:
        note(f"rank {fail['rank']} " + ("hit the deadline" if fail["timed_out"] else f"exited with status {fail['rc']}")
             + "; its stderr tail:\n" + fail["stderr_tail"])
        forced = os.environ.get("UMPR_COMM_ASYNC", "")
        in_stream = forced == "0" or (forced == "" and getattr(args, "dtype", "fp32") == "bf16")   # GradReducer._in_stream
        retry = (in_stream and not fail["timed_out"] and fail["rc"] is not None and fail["rc"] > 0
                 and os.environ.get("UMPR_BENCH_NO_FALLBACK", "") != "1")
        if not retry:
            sys.exit(1)
        why = f"first attempt: rank {fail['rank']} exited {fail['rc']}"
        note("starting a fresh set of workers with UMPR_COMM_ASYNC=1 (async form of the gradient exchange)")
        ok, out0, fail2 = _run_worker_set(args.gpus, argv, cmd, {"UMPR_COMM_ASYNC": "1", "UMPR_BENCH_FALLBACK": why},
                                          deadline_s)
        if not ok:
            note(f"fallback attempt: rank {fail2['rank']} " + ("hit the deadline" if fail2["timed_out"] else
                 f"exited with status {fail2['rc']}") + "; its stderr tail:\n" + fail2["stderr_tail"])
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
    # newest first: record names are r<round>_<tag>_...; within a round tags run a .. z, aa .. (longer tag = later)
    def age(f):
        head = f[:-len(suffix)].split("_")
        rnd = int(head[0][1:]) if head[0][1:].isdigit() else 0
        tag = head[1] if len(head) > 1 else ""
        return (rnd, len(tag), tag)
    for name in sorted((f for f in os.listdir(prof) if f.endswith(suffix) and ("bf16" in f) == bf16), key=age, reverse=True):
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


def _roofline(w, fam, value, traffic_ok=True):
    """`roofline` of one workload from the HIP-event family timings of its timed region (see the module docstring)."""
    per_s = lambda v: (v[1] / (v[0] * 1e-3) / 1e12) if v[0] > 0 else 0.0   # work per second / 1e12
    full = not w.review_net_only
    if full and w.dtype == "fp32":
        ms, work, n = fam.get("wino_gemm", (0.0, 0.0, 0))   # dominant kernel: executed MFMA FLOPs (planes of M x C x tiles)
        achieved = per_s((ms, work, n))
        fwd = fam.get("conv3x3_fwd", (0.0, 0.0, 0))
        traffic, traffic_note = (pmc_traffic(infer=w.eval) if w.batch == 64 and w.views == 1 and traffic_ok
                                 else (None, "stored PMC pass is for the batch-64, 1-view workload"))
        return {"bound": "mfma", "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": traffic, "traffic_note": traffic_note
                + "; the figure is per conv-forward layer call (transforms + GEMM), of which this kernel is the GEMM",
                "kernel": "wino_gemm_dma_kernel (Winograd batched GEMM over the planes of the F(2x2,3x3) / F(4x4,3x3) tiles of "
                          "the 112/56/28/14 layers, forward and data gradient; largest share of GPU time).  achieved = MFMA "
                          "FLOPs it executes / its HIP-event time on its launch stream; in the timed region it shares the "
                          "chip with the weight-gradient stream and the text stream",
                "launches": n, "avg_launch_ms": ms / max(n, 1),
                "executed_gflop_per_launch": work / max(n, 1) / 1e9,
                "algorithmic_frac": per_s(fwd) / PEAK_F32_MFMA_TFLOPS,
                "algorithmic_note": "conv forward family (13 layer calls/step) priced by direct-convolution FLOPs "
                                    "2*N*H*W*Cout*Cin*9; the Winograd layers execute 1/2.25 or 1/4 of that"}
    if full:
        ms, work, n = fam.get("conv_bf16_fwd", (0.0, 0.0, 0))
        achieved = per_s((ms, work, n))
        traffic, traffic_note = (pmc_traffic(True) if w.batch == 64 and w.views == 1 and traffic_ok
                                 else (None, "stored PMC pass is for the batch-64, 1-view workload"))
        return {"bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_BF16_MFMA_TFLOPS, "traffic": traffic,
                "traffic_note": traffic_note + "; at the measured launch time this is < 2 TB/s of HBM traffic, "
                                "close to the compulsory activation bytes: the kernel is bound by the matrix pipe",
                "kernel": "conv3x3_bf16_kernel forward launches (implicit GEMM on v_mfma_f32_16x16x32_bf16, padded "
                          "NHWC bf16 activations); executed FLOPs = algorithmic FLOPs over the padded pixel grid",
                "launches": n, "avg_launch_ms": ms / max(n, 1), "executed_gflop_per_launch": work / max(n, 1) / 1e9}
    ms, work, n = fam.get("gru", (0.0, 0.0, 0))
    achieved = (work / (ms * 1e-3) / 1e9) if ms > 0 else 0.0
    return {"bound": "hbm", "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": achieved / PEAK_HBM_GBS, "traffic": None,
            "kernel": "gru_fwd16_kernel / gru_bwd16_kernel (recurrent part of the packed BiGRU, 16-sequence tiles): algorithmic bytes "
                      "(gx, out, saved gates, dgx) / HIP-event time.  The kernel is latency-bound - a chain of "
                      "<= 20 dependent steps per sequence tile - not bandwidth-bound, so the fraction is small "
                      "by construction", "launches": n, "avg_launch_ms": ms / max(n, 1),
            "step_bytes_frac": value * KB_PER_SAMPLE_UMPR_R * 1e3 * (1 if w.eval else 3) / 1e9 / PEAK_HBM_GBS,
            "step_bytes_note": "whole step: SURVEY 8(d) compulsory traffic 187 KB/sample forward (x3 for a "
                               "training step) x samples/s against 8 TB/s"}


def workload_name(w):
    return (("UMPR-R (review_net_only)" if w.review_net_only else f"full UMPR, {w.views} view(s), 1 photo/view")
            + f", GloVe-{w.emb}d-shaped table {w.vocab}x{w.emb}, batch {w.batch}/GPU, "
            + ("ragged lengths" if w.realistic else "fully padded S=L=L_ui=20 S_ui=5")
            + (", fp32" if w.dtype == "fp32" else ", bf16 MFMA conv stack + attention scores with fp32 "
               "accumulation, fp32 master weights / classifier / GRU gates / Adam")
            + (", forward only (evaluate.py)" if w.eval else ", fwd+bwd+Adam(+RCCL all-reduce)") + ", random-init weights")


def run_workload(w, env):
    """Build the model of workload `w` (an argparse-like namespace), run w.warmup untimed and w.steps timed steps, return
    (result dict without the driver-contract envelope, parameter state for the CPU baseline)."""
    import gc
    import torch
    from umpr_amd import parallel
    from umpr_amd.config import Config
    from umpr_amd.model import UMPR
    from umpr_amd.optim import FusedAdam
    from umpr_amd.synthetic import make_batch, make_param_state
    from umpr_amd.train import train_step
    L, dev, rank, world = env["L"], env["dev"], env["rank"], env["world"]
    torch.manual_seed(0)  # dropout masks derive from torch.initial_seed()
    Config.extend({"dtype": "fp32"})
    cfg = Config(argv=[])
    cfg.review_net_only = w.review_net_only
    cfg.views = ["v%d" % i for i in range(w.views)]
    cfg.dtype = w.dtype
    P = make_param_state(0, w.emb, w.vocab, w.views, w.review_net_only)
    model = UMPR(cfg, P["embedding.weight"].numpy())
    model.load_state_dict(P)
    model = model.to(dev)
    opt = FusedAdam(model, cfg.learning_rate, cfg.l2_regularization, cfg.lr_decay) if not w.eval else None
    reducer = parallel.GradReducer(opt) if (parallel.active() and opt is not None) else None
    host_batch = make_batch(1234 + rank, w.batch, w.vocab, w.views, review_net_only=w.review_net_only,
                            full_pad=not w.realistic)
    u, i_, ui, ul, il, uil, photos, labels = host_batch
    # lengths stay on the host, like the reference (src/model.py:18)
    batch = (u.to(dev), i_.to(dev), ui.to(dev), ul, il, uil, photos.to(dev), labels.to(dev))
    losses = []          # device scalars, summed once after the timed region (no per-step torch kernel inside it)

    def barrier():
        if parallel.active():
            torch.distributed.barrier()
        torch.cuda.synchronize()

    graphed = None
    if getattr(w, "graph", False) and w.review_net_only and not w.eval and not parallel.active():
        from umpr_amd.graphs import GraphedTrainStep
        graphed = GraphedTrainStep(model, opt, batch)
        batch = graphed.resident(batch)      # "batch resident in HBM": its ids / labels live where the graph reads them

    def step(b=batch):
        if graphed is not None:
            return graphed(b)
        if getattr(w, "fwd_train", False):
            model.train()
            return model(*b)
        if not w.eval:
            return train_step(model, opt, b, world, reducer)
        with torch.no_grad():
            model.eval()
            return model(*b)

    note(f"{workload_name(w)}: model and batch resident; warm-up")
    for _ in range(w.warmup):
        step()
    barrier()
    note("timed region")
    L.fn["umpr_profile_reset"]()
    L.fn["umpr_profile_enable"](1)
    t0 = time.perf_counter()
    for _ in range(w.steps):
        _, loss = step()
        losses.append(loss.detach().clone() if graphed is not None else loss.detach())   # a replayed graph rewrites its outputs
    t_issue = time.perf_counter() - t0        # host time to ENQUEUE the steps (no wait): close to dt => host-bound
    barrier()
    dt_local = dt = time.perf_counter() - t0
    L.fn["umpr_profile_enable"](0)
    note(f"{w.steps} steps in {dt:.3f} s")
    per_rank_ms = [1e3 * dt / w.steps]
    rccl_ranks = 1
    if parallel.active():
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(t) for _ in range(torch.distributed.get_world_size())]
        torch.distributed.all_gather(every, t)
        per_rank_ms = [1e3 * float(x.item()) / w.steps for x in every]
        dt = max(float(x.item()) for x in every)          # MAX over ranks
        rccl_ranks = torch.distributed.get_world_size()   # what the process group itself reports after init

    fam = {}
    for name, idx in (("conv3x3_fwd", 0), ("conv3x3_dgrad", 5), ("conv3x3_wgrad", 1), ("gemm_f32", 2), ("wino_gemm", 4),
                      ("wino_wgrad_gemm", 6), ("gru", 3), ("conv_bf16_fwd", 7), ("conv_bf16_dgrad", 8),
                      ("conv_bf16_wgrad", 9)):
        ms, work, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_long()
        if L.fn["umpr_profile_read"](idx, ctypes.byref(ms), ctypes.byref(work), ctypes.byref(n)) == 0:
            fam[name] = (ms.value, work.value, n.value)

    h2d = None
    if getattr(w, "h2d", False) and not w.eval:
        feeder = HostFeeder(host_batch, dev)
        for _ in range(2):
            b, slot = feeder.next()
            step(b)
            feeder.done(slot)
        barrier()
        t1 = time.perf_counter()
        for _ in range(w.steps):
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
        h2d = {"value": world * w.batch * w.steps / dth, "unit": "samples/s", "ms_per_step": 1e3 * dth / w.steps,
               "host_bytes_per_step": nbytes,
               "note": "every step's ids/photos/labels are uploaded from pinned host memory on a copy stream, "
                       "double-buffered under the previous step (the reference uploads inside forward, src/model.py:259-260)"}

    value = world * w.batch * w.steps / dt
    per_s = lambda v: (v[1] / (v[0] * 1e-3) / 1e12) if v[0] > 0 else 0.0
    full = not w.review_net_only
    out = {
        "metric": "inference samples/sec" if w.eval else "training samples/sec", "value": value, "unit": "samples/s",
        "n_gpus": world, "steps": w.steps, "warmup": w.warmup, "ms_per_step": 1e3 * dt / w.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if w.dtype == "fp32" else "bf16", "data": "synthetic",
        "config": {"workload": workload_name(w), "global_batch": world * w.batch, "parallelism": f"dp{world}"},
        "roofline": _roofline(w, fam, value),
        "kernels": {k: {"ms_per_step": v[0] / w.steps, ("gbytes_per_s" if k == "gru" else "tflops"):
                        ((v[1] / (v[0] * 1e-3) / 1e9 if v[0] > 0 else 0.0) if k == "gru" else per_s(v)),
                        "launches_per_step": v[2] / w.steps} for k, v in fam.items() if v[2]},
        "loss_mean": float(torch.stack(losses).sum().item()) / w.steps,
        "host_issue_ms_per_step": 1e3 * t_issue / w.steps,
        "rccl_ranks": rccl_ranks, "ms_per_step_per_rank": per_rank_ms,
    }
    if graphed is not None:
        out["hip_graph"] = "the whole training step (forward, backward, Adam) is one captured hipGraph launch per step"
    if parallel.active():
        out["comm"] = {"backend": torch.distributed.get_backend(),
                       "exchange": ("in-stream" if (reducer is not None and reducer.comm is not None) else "async")}
        if os.environ.get("UMPR_BENCH_FALLBACK"):
            out["comm_fallback"] = "UMPR_COMM_ASYNC=1 in a fresh set of workers; " + os.environ["UMPR_BENCH_FALLBACK"]
    if full:
        gf = 3 * 2 * (TEXT_MMAC_FWD.get(w.emb, 150.6) + w.views * VGG_MMAC_FWD) / 1e3   # GFLOP per sample, training
        out["model_tflops"] = value * (gf / 3.0 if w.eval else gf) / 1e3 / world
        out["roofline"]["model_frac"] = out["model_tflops"] / (PEAK_F32_MFMA_TFLOPS if w.dtype == "fp32" else PEAK_BF16_MFMA_TFLOPS)
        if w.dtype == "fp32":
            out["roofline"]["model_note"] = ("algorithmic (direct-convolution) FLOPs of the whole step / time / peak; it may "
                                             "exceed 1 because the Winograd layers execute 1/2.25 or 1/4 "
                                             "of them - `frac` is the utilisation figure")
    if h2d is not None:
        out["h2d_inclusive"] = h2d
    # release this workload's 2-6 GB of arenas before the next one is built
    if reducer is not None:
        reducer.close()
    if opt is not None:
        opt.close()
    del model, opt, reducer, batch, step
    gc.collect()
    torch.cuda.empty_cache()
    return out, P


# The other single-GPU workloads BASELINE.json names, run after the headline by a default invocation so that the
# driver-timed record holds them.  Keys: overrides of the headline's argparse namespace.
OTHER_CONFIGS = (
    ("configs[4]_per_gpu_bf16_glove300_b64", dict(dtype="bf16", emb=300, batch=64)),
    ("configs[0]_umpr_r_b32", dict(review_net_only=True, batch=32)),
    ("configs[0]_umpr_r_b32_hipgraph", dict(review_net_only=True, batch=32, graph=True)),
    ("configs[3]_per_gpu_4views_b32", dict(views=4, batch=32)),
    ("configs[3]_per_gpu_4views_b32_bf16", dict(views=4, batch=32, dtype="bf16")),
    ("inference_fp32_b64", dict(eval=True)),
    ("inference_bf16_glove300_b64", dict(eval=True, dtype="bf16", emb=300)),
)


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

    torch.set_num_threads(host_cores())
    try:
        rank, local, world = parallel.init_distributed(timeout_s=float(os.environ.get("UMPR_DIST_TIMEOUT_S", "180")))
        assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
        assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
        dev = torch.device("cuda", local)
        torch.cuda.set_device(dev)
        env = {"L": lib(), "dev": dev, "rank": rank, "world": world}
        out, P = run_workload(args, env)
        default_line = (world == 1 and not args.no_other_configs and not args.eval and not args.fwd_train and not args.review_net_only
                        and not args.realistic and not args.h2d
                        and (args.batch, args.views, args.emb, args.dtype) == (64, 1, 50, "fp32"))
        if default_line:
            others = {}
            for name, over in OTHER_CONFIGS:
                w = argparse.Namespace(**{**vars(args), **over, "steps": args.other_steps, "warmup": 5, "h2d": False})
                try:
                    o, _ = run_workload(w, env)
                except Exception as e:    # a sub-line must never cost the headline; say what happened
                    note(f"other config {name} failed: {e!r}")
                    others[name] = {"error": repr(e)}
                    continue
                keep = ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "roofline", "kernels",
                        "host_issue_ms_per_step", "model_tflops", "loss_mean")
                others[name] = {k: o[k] for k in keep if k in o}
                others[name]["workload"] = o["config"]["workload"]
                r = others[name]["roofline"]      # keep the sub-lines compact: the prose is in the headline / docstring
                others[name]["roofline"] = {k: r[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic",
                                                              "launches", "avg_launch_ms") if k in r}
            out["other_configs"] = others
        if rank == 0:
            if world == 1 and not args.no_cpu_baseline and not args.eval and not args.fwd_train:
                out["cpu_baseline"] = cpu_baseline(args, P, rank)
            result_out.write(json.dumps(out) + "\n")
            result_out.flush()
        if parallel.active():
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
    except BaseException:
        # a rank that dies leaves its peers blocked in the next RCCL collective: report and leave with a non-zero status at
        # once (no teardown that would itself wait for the peers); the launcher then ends the other ranks
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        if os.environ.get("WORLD_SIZE", "1") != "1":
            os._exit(1)
        raise


if __name__ == "__main__":
    main()
