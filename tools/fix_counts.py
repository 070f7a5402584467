#!/usr/bin/env python3
"""How many outputs the decision fix-up of the F(4x4,3x3) training forward lists per VGG16 layer (umpr_debug_wino_fix_count), for
four kinds of input image: i.i.d. uniform noise (the benchmark's synthetic photos), an all-zero image (a missing photo), a
constant image, and a smooth low-frequency image.  Layers are chained through the C ABI like umpr_vgg16_features_fwd does."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from umpr_amd._lib import lib
from umpr_amd.synthetic import VGG16_CFG, VGG16_CONV_IDX, make_param_state


def main():
    L = lib()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    P = make_param_state(0, 8, 16, 1, False)
    g = torch.Generator().manual_seed(5)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, 224), torch.linspace(0, 1, 224), indexing="ij")
    smooth = torch.stack([0.5 + 0.4 * torch.sin(6 * xx + k) * torch.cos(4 * yy + 2 * k) for k in range(3)])
    kinds = {"noise": torch.rand(4, 3, 224, 224, generator=g), "zero": torch.zeros(4, 3, 224, 224),
             "constant": torch.full((4, 3, 224, 224), 0.7), "smooth": smooth.expand(4, 3, 224, 224).contiguous()}
    for kind, img in kinds.items():
        x = img.to(dev)
        ci = 0
        out = []
        for v in VGG16_CFG:
            if v == "M":
                n, c, hw = x.shape[0], x.shape[1], x.shape[-1]
                y = torch.empty(n, c, hw // 2, hw // 2, device=dev)
                L.call("umpr_maxpool2_fwd", x, y, n * c, hw, hw, st)
                x = y
                continue
            idx = VGG16_CONV_IDX[ci]
            w, b = P[f"visual_net.vgg16.0.features.{idx}.weight"].to(dev), P[f"visual_net.vgg16.0.features.{idx}.bias"].to(dev)
            n, cin, hw, cout = x.shape[0], x.shape[1], x.shape[-1], w.shape[0]
            y = torch.empty(n, cout, hw, hw, device=dev)
            wt = torch.empty(L.size("umpr_conv3x3_pack_bytes", n, cin, cout, hw, hw) // 4, device=dev)
            pooled = idx in (2, 7, 14, 21, 28)
            L.call("umpr_set_conv_pool_follows", int(pooled))
            L.call("umpr_conv3x3_fwd", x, w, b, y, n, cin, hw, hw, cout, 1, wt, wt.numel() * 4, st)
            L.call("umpr_set_conv_pool_follows", 0)
            if hw <= 56:
                nf = L.fn["umpr_debug_wino_fix_count"]()
                out.append(f"f{idx}{'p' if pooled else ''}:{nf}/{y.numel()}={nf / y.numel():.1e}")
            x = y
            ci += 1
        print(f"{kind:9s} " + "  ".join(out), flush=True)


if __name__ == "__main__":
    main()
