#!/usr/bin/env python3
"""Per-layer timing of the bf16 VGG16 conv kernels through the C ABI (forward / dgrad / wgrad), HIP-event timed.
usage: python tools/bench_conv_bf16.py [--n 64] [--layers 1,3,5] [--reps 5] [--only fwd|dgrad|wgrad]
FLOPs are the real-pixel count 2*N*H*W*Cout*Cin*9 (the kernels also compute the zero pads: +1.8 % at 224 ... +15 % at 14)."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from umpr_amd._lib import lib

CFG = [(3, 64, 224), (64, 64, 224), (64, 128, 112), (128, 128, 112), (128, 256, 56), (256, 256, 56), (256, 256, 56),
       (256, 512, 28), (512, 512, 28), (512, 512, 28), (512, 512, 14), (512, 512, 14), (512, 512, 14)]


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=64)
    ap.add_argument("--layers", default="1,2,3,4,5,7,8,10")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    L = lib()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    tot = {"fwd": [0, 0], "dgrad": [0, 0], "wgrad": [0, 0]}
    for li in [int(x) for x in a.layers.split(",")]:
        ci, co, hw = CFG[li]
        n = a.n
        g = torch.Generator(device="cpu").manual_seed(li)
        mk = lambda c: torch.empty(L.size("umpr_bf16_tensor_bytes", n, c, hw, hw), dtype=torch.uint8, device=dev)
        x, y, gy, dx = mk(ci), mk(co), mk(co), mk(ci)
        # random bf16 payloads (random data: a zero-filled operand lets the chip clock higher and reads too fast)
        for t, c in ((x, ci), (gy, co)):
            src = torch.randn(n, c, hw, hw, device=dev)
            L.call("umpr_bf16_from_nchw_f32", src, t, n, c, hw, hw, st)
            del src
        w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
        b = torch.randn(co, device=dev)
        dw, db = torch.empty_like(w), torch.empty_like(b)
        wsb = L.size("umpr_conv3x3_bf16_ws_bytes", n, ci, co, hw, hw)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        fl = 2.0 * n * hw * hw * co * ci * 9
        res = []
        for name, fn in (("fwd", lambda: L.call("umpr_conv3x3_bf16_fwd", x, w, b, y, n, ci, hw, hw, co, 1, ws, wsb, st)),
                         ("dgrad", lambda: L.call("umpr_conv3x3_bf16_bwd_data", gy, w, x, dx, n, ci, hw, hw, co, ws, wsb, st)),
                         ("wgrad", lambda: L.call("umpr_conv3x3_bf16_bwd_weight", gy, x, dw, db, n, ci, hw, hw, co, ws, wsb, st))):
            if a.only and a.only != name:
                continue
            ms = timed(fn, a.reps)
            tot[name][0] += ms; tot[name][1] += fl
            res.append(f"{name} {ms:7.3f} ms {fl / ms / 1e9:6.1f} TF")
        print(f"layer {li:2d} {ci:3d}->{co:3d} @{hw:3d}: " + " | ".join(res), flush=True)
    for k, (ms, fl) in tot.items():
        if ms:
            print(f"total {k}: {ms:.3f} ms, {fl / ms / 1e9:.1f} TF")


if __name__ == "__main__":
    main()
