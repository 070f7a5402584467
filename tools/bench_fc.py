#!/usr/bin/env python3
"""The VGG classifier's three fc1 GEMMs at batch n through umpr_gemm_f32 (forward, dx, dW), HIP-event timed.
usage: UMPR_GEMM_SMALL_M_WGS=<target> python tools/bench_fc.py [--n 64]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from umpr_amd._lib import lib


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


ap = argparse.ArgumentParser(); ap.add_argument("--n", type=int, default=64); a = ap.parse_args()
L = lib(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
n = a.n
for K, N in ((25088, 4096), (4096, 4096)):
    x = torch.randn(n, K, device=dev); W = torch.randn(N, K, device=dev) * 0.01; b = torch.randn(N, device=dev)
    y = torch.empty(n, N, device=dev); g = torch.randn(n, N, device=dev); dx = torch.empty(n, K, device=dev); dW = torch.empty(N, K, device=dev)
    wsb = 512 << 20; ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    t_f = timed(lambda: L.call("umpr_gemm_f32", x, K, 0, W, K, 1, y, N, n, N, K, b, 1, 1, 0, 1.0, ws, wsb, st))
    t_x = timed(lambda: L.call("umpr_gemm_f32", g, N, 0, W, K, 0, dx, K, n, K, N, None, 0, 0, 0, 1.0, ws, wsb, st))
    t_w = timed(lambda: L.call("umpr_gemm_f32", g, N, 1, x, K, 0, dW, K, N, K, n, None, 0, 0, 0, 1.0, None, 0, st))
    fl = 2.0 * n * K * N
    print(f"fc {K}->{N} n={n} target={os.environ.get('UMPR_GEMM_SMALL_M_WGS','512')}: fwd {t_f:7.1f} us ({fl/t_f/1e6:5.1f} TF, {N*K*4/t_f/1e6:4.2f} TB/s)  dx {t_x:7.1f} us  dW {t_w:7.1f} us", flush=True)
