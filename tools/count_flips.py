#!/usr/bin/env python3
"""Counts the ReLU / max-pool DECISIONS of the library's fp32 VGG16 feature forward (layer by layer through the C ABI, training
mode = what the backward pass will replay) that differ from a float64 forward of the same network on the CPU, per layer, for
whatever convolution algorithm the environment selects.  The measure behind the decision fix-up of winograd.hip: the
golden-fixture gradient distances are driven by a handful of such flips.

    [UMPR_WINO_F4=0|1|2] [UMPR_WINO_POINTS=0] [UMPR_WINO_FIX_KAPPA=0] [UMPR_CONV_WINO=0] python tools/count_flips.py [--n 8]

Each layer is fed the float64 forward's own input (rounded to fp32), so a layer's count is that layer's kernel alone and the
counts of different algorithms are comparable.  Also prints the torch CPU fp32 convolution's count (the reference's arithmetic)."""
import argparse, os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from umpr_amd._lib import lib
from umpr_amd.synthetic import VGG16_CFG, VGG16_CONV_IDX, make_param_state


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=8)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--fixture", default=None, help="pseed,bseed,B of a tests/golden umpr_full fixture: its parameters and photos")
    ap.add_argument("--chained", action="store_true",
                    help="feed each layer the LIBRARY's own previous output (what a real forward does) instead of the float64 one")
    a = ap.parse_args()
    L = lib()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    tag = " ".join(f"{k}={os.environ[k]}" for k in ("UMPR_WINO_F4", "UMPR_WINO_POINTS", "UMPR_WINO_FIX_KAPPA", "UMPR_CONV_WINO")
                   if k in os.environ) or "default"
    if a.fixture:
        from umpr_amd.synthetic import make_batch
        pseed, bseed, B = (int(v) for v in a.fixture.split(","))
        P = make_param_state(pseed, 50, 1000, 1, False, m_scale=0.05)      # VGG weights do not depend on m_scale
        x64 = make_batch(bseed, B, 1000, 1, 1)[6].reshape(-1, 3, 224, 224).double()
        tag += f" fixture={a.fixture}" + (" chained" if a.chained else "")
    else:
        P = make_param_state(a.seed, 8, 16, 1, False)
        g = torch.Generator().manual_seed(a.seed + 100)
        x64 = torch.rand(a.n, 3, 224, 224, generator=g).double()
    xh = xc = x64.float()                                                 # the library's / torch CPU fp32's own chain (--chained)
    ci = 0
    tot = {"hip_relu": 0, "cpu_relu": 0, "hip_pool": 0, "cpu_pool": 0, "relu_n": 0, "pool_n": 0}
    last_hip = last_cpu = None
    for v in VGG16_CFG:
        if v == "M":
            # pool decisions: argmax of the window on each path's own activation of the layer before
            i64 = F.max_pool2d(x64, 2, 2, return_indices=True)[1]
            ih = F.max_pool2d(last_hip, 2, 2, return_indices=True)[1]
            ic = F.max_pool2d(last_cpu, 2, 2, return_indices=True)[1]
            live = F.max_pool2d(x64, 2, 2) > 0            # a window of zeros routes no gradient: its argmax is immaterial
            fh, fc = int(((ih != i64) & live).sum()), int(((ic != i64) & live).sum())
            tot["hip_pool"] += fh; tot["cpu_pool"] += fc; tot["pool_n"] += int(live.sum())
            print(f"[{tag}] pool after features.{VGG16_CONV_IDX[ci - 1]:2d}: argmax differs from float64 in {fh:4d} (library) / {fc:4d} "
                  f"(torch CPU fp32) of {int(live.sum())} live windows", flush=True)
            x64 = F.max_pool2d(x64, 2, 2)
            xh, xc = F.max_pool2d(last_hip, 2, 2), F.max_pool2d(last_cpu, 2, 2)
            continue
        idx = VGG16_CONV_IDX[ci]
        w, b = P[f"visual_net.vgg16.0.features.{idx}.weight"], P[f"visual_net.vgg16.0.features.{idx}.bias"]
        n, cin, hw = x64.shape[0], x64.shape[1], x64.shape[-1]
        cout = w.shape[0]
        x32 = xh if a.chained else x64.float()
        y64 = F.conv2d(x64, w.double(), b.double(), padding=1)
        yc = F.relu(F.conv2d(xc if a.chained else x32, w, b, padding=1))
        y = torch.empty(n, cout, hw, hw, device=dev)
        wt = torch.empty(L.size("umpr_conv3x3_pack_bytes", n, cin, cout, hw, hw) // 4, device=dev)
        L.call("umpr_set_conv_pool_follows", int(idx in (2, 7, 14, 21, 28)))      # as umpr_vgg16_features_fwd does
        L.call("umpr_conv3x3_fwd", x32.to(dev), w.to(dev), b.to(dev), y, n, cin, hw, hw, cout, 1, wt, wt.numel() * 4, st)
        L.call("umpr_set_conv_pool_follows", 0)
        yh = y.cpu()
        fh, fc = int(((yh > 0) != (y64 > 0)).sum()), int(((yc > 0) != (y64 > 0)).sum())
        tot["hip_relu"] += fh; tot["cpu_relu"] += fc; tot["relu_n"] += y64.numel()
        print(f"[{tag}] relu features.{idx:2d} {cin:3d}->{cout:3d} @{hw:3d}: sign differs from float64 in {fh:4d} (library) / {fc:4d} "
              f"(torch CPU fp32) of {y64.numel()}", flush=True)
        last_hip, last_cpu = yh, yc
        xh, xc = yh, yc
        x64 = F.relu(y64)
        ci += 1
    print(f"[{tag}] TOTAL relu flips {tot['hip_relu']} (library) / {tot['cpu_relu']} (torch CPU fp32) of {tot['relu_n']}; "
          f"pool flips {tot['hip_pool']} / {tot['cpu_pool']} of {tot['pool_n']}", flush=True)


if __name__ == "__main__":
    main()
