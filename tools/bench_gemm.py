#!/usr/bin/env python3
"""Timing of umpr_gemm_f32 at the text path's shapes, fp32 and bf16-operand mode (HIP events).
usage: python tools/bench_gemm.py [M,N,K,ta,tb,splitk ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from umpr_amd._lib import lib

SHAPES = [  # (M, N, K, ta, tb, splitk)
    (51200, 384, 300, 0, 1, 0), (51200, 384, 304, 0, 1, 0), (51200, 384, 50, 0, 1, 0), (384, 300, 51200, 1, 0, 1),
    (25600, 128, 128, 0, 0, 0), (51200, 64, 128, 0, 1, 0), (51200, 100, 384, 0, 1, 0), (100, 384, 51200, 1, 0, 1),
    (51200, 128, 300, 0, 0, 0)]


def main():
    global SHAPES
    if len(sys.argv) > 1:      # M,N,K,ta,tb,splitk ...
        SHAPES = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
    L = lib()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    for M, N, K, ta, tb, sk in SHAPES:
        A = torch.randn((K, M) if ta else (M, K), device=dev)
        B = torch.randn((N, K) if tb else (K, N), device=dev)
        C = torch.empty(M, N, device=dev)
        wsb = 64 * M * N * 4 if sk else 0
        ws = torch.empty(max(wsb // 4, 1), device=dev)
        row = f"M{M:6d} N{N:4d} K{K:6d} ta{ta} tb{tb}:"
        for mode in (0, 1):
            L.call("umpr_set_gemm_bf16", mode)
            def run():
                L.call("umpr_gemm_f32", A, A.shape[1], ta, B, B.shape[1], tb, C, N, M, N, K, None, 0, 0, 0, 1.0,
                       ws if sk else None, wsb, st)
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100
            row += f"  {'bf16' if mode else 'fp32'} {us:7.1f} us {2.0 * M * N * K / us / 1e6:6.1f} TF"
        L.call("umpr_set_gemm_bf16", 0)
        print(row)


if __name__ == "__main__":
    main()
