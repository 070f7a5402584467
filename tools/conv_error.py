#!/usr/bin/env python3
"""Rounding error of the library's fp32 convolution forward (whatever algorithm the environment selects: direct, F(2x2,3x3),
F(4x4,3x3) with either point set) against a float64 convolution of the same inputs, per VGG16 layer shape, through the C ABI.
The GPU-side counterpart of tools/wino43_error.py / tools/wino_points.py.
usage: [UMPR_WINO_F4=0|1|2] [UMPR_WINO_POINTS=0|1] [UMPR_CONV_WINO=0] python tools/conv_error.py [--n 2] [--layers 5,8,11]"""
import argparse, os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from umpr_amd._lib import lib

CFG = [(3, 64, 224), (64, 64, 224), (64, 128, 112), (128, 128, 112), (128, 256, 56), (256, 256, 56), (256, 256, 56),
       (256, 512, 28), (512, 512, 28), (512, 512, 28), (512, 512, 14), (512, 512, 14), (512, 512, 14)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2)
    ap.add_argument("--layers", default="4,5,7,8,10,11")
    ap.add_argument("--inference", type=int, default=0)
    a = ap.parse_args()
    L = lib()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    tag = " ".join(f"{k}={os.environ[k]}" for k in ("UMPR_WINO_F4", "UMPR_WINO_POINTS", "UMPR_CONV_WINO") if k in os.environ) or "default"
    if a.inference:
        L.fn["umpr_set_conv_inference"](1)
    for li in [int(x) for x in a.layers.split(",")]:
        ci, co, hw = CFG[li]
        g = torch.Generator().manual_seed(li)
        x = torch.relu(torch.randn(a.n, ci, hw, hw, generator=g))
        w = torch.randn(co, ci, 3, 3, generator=g) * (2.0 / (9 * ci)) ** 0.5
        b = torch.randn(co, generator=g) * 0.05
        ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
        y = torch.empty(a.n, co, hw, hw, device=dev)
        wt = torch.empty(L.size("umpr_conv3x3_pack_bytes", a.n, ci, co, hw, hw) // 4, device=dev)
        L.call("umpr_conv3x3_fwd", x.to(dev), w.to(dev), b.to(dev), y, a.n, ci, hw, hw, co, 0, wt, wt.numel() * 4, st)
        e = (y.cpu().double() - ref)
        cpu = (F.conv2d(x, w, b, padding=1).double() - ref)
        print(f"[{tag}] layer {li:2d} {ci:3d}->{co:3d} @{hw:3d}: max err / max|y| {float(e.abs().max() / ref.abs().max()):.2e}  "
              f"rms err / rms y {float(e.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()):.2e}   (torch CPU fp32: "
              f"{float(cpu.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()):.2e})", flush=True)


if __name__ == "__main__":
    main()
