#!/usr/bin/env python3
"""Which neighbour makes the first merge-backward kernel (UMPR_MERGE_DX=1) return wrong sums?  umpr_review_merge_bwd runs on one
stream with fixed inputs while ONE library entry point is issued again and again on a second stream; every d_repr is compared bit
for bit with the quiet-device result.  (tools/hazard_repro.hip does the same with synthetic co-runners: VALU loop, bf16 MFMA loop,
streaming copy - none of them triggers it.)

    UMPR_MERGE_DX=1 python tools/hazard_repro.py [--launches 300]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from umpr_amd._lib import lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--launches", type=int, default=300)
    a = ap.parse_args()
    L = lib()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(7)
    B = 4
    ru, ri = (torch.randn(B, 256, generator=g).to(dev) for _ in range(2))
    Wu, Wi = (torch.randn(128, 256, generator=g).to(dev) / 16 for _ in range(2))
    out = torch.tanh(torch.randn(B, 128, generator=g)).to(dev)
    dout = (torch.randn(B, 128, generator=g) * 0.2).to(dev)
    ws = torch.empty(L.size("umpr_review_merge_bwd_ws_bytes", B) // 4 + 64, device=dev)
    outs = [torch.empty(B, 256, device=dev), torch.empty(B, 256, device=dev), torch.empty(128, 256, device=dev), torch.empty(128, 256, device=dev)]
    sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

    def dx():
        L.call("umpr_review_merge_bwd", ru, ri, Wu, Wi, out, dout, B, *outs, ws, ws.numel() * 4, sa.cuda_stream)
    dx(); torch.cuda.synchronize()
    ref = torch.cat([outs[0], outs[1]]).clone()

    N, C, HW = 4, 256, 56
    nb = L.size("umpr_bf16_tensor_bytes", N, C, HW, HW)
    xb = torch.zeros(nb, dtype=torch.uint8, device=dev); yb = torch.zeros(nb, dtype=torch.uint8, device=dev)
    L.call("umpr_bf16_from_nchw_f32", torch.randn(N, C, HW, HW, device=dev), xb, N, C, HW, HW, 0)
    L.call("umpr_bf16_from_nchw_f32", torch.randn(N, C, HW, HW, device=dev), yb, N, C, HW, HW, 0)
    wsb = L.size("umpr_conv3x3_bf16_ws_bytes", N, C, C, HW, HW)
    cws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    w = torch.randn(C, C, 3, 3, device=dev) / 48
    dw, db = torch.empty_like(w), torch.empty(C, device=dev)
    xf, yf = torch.randn(N, C, HW, HW, device=dev), torch.empty(N, C, HW, HW, device=dev)
    fwt = torch.empty(L.size("umpr_conv3x3_pack_bytes", N, C, C, HW, HW) // 4, device=dev)
    fws = torch.empty(L.size("umpr_conv3x3_bwd_weight_ws_bytes", N, C, C, HW, HW) // 4 + 64, device=dev)
    co = {
        "nothing": lambda: None,
        "bf16 weight gradient 256->256 @56": lambda: L.call("umpr_conv3x3_bf16_bwd_weight", yb, xb, dw, db, N, C, HW, HW, C, cws, wsb, sb.cuda_stream),
        "bf16 forward conv 256->256 @56": lambda: L.call("umpr_conv3x3_bf16_fwd", xb, w, db, yb, N, C, HW, HW, C, 1, cws, wsb, sb.cuda_stream),
        "bf16 data gradient 256->256 @56": lambda: L.call("umpr_conv3x3_bf16_bwd_data", yb, w, None, xb, N, C, HW, HW, C, cws, wsb, sb.cuda_stream),
        "fp32 Winograd forward 256->256 @56": lambda: L.call("umpr_conv3x3_fwd", xf, w, db, yf, N, C, HW, HW, C, 1, fwt, fwt.numel() * 4, sb.cuda_stream),
        "fp32 Winograd weight gradient": lambda: L.call("umpr_conv3x3_bwd_weight", yf, xf, dw, db, N, C, HW, HW, C, fws, fws.numel() * 4, sb.cuda_stream),
    }
    # the bf16 weight gradient at every VGG layer shape (different template instances / LDS footprints)
    keep = []
    for cin, cout, hw in ((64, 64, 224), (64, 128, 112), (128, 128, 112), (128, 256, 56), (256, 512, 28), (512, 512, 28), (512, 512, 14)):
        nbx, nby = L.size("umpr_bf16_tensor_bytes", N, cin, hw, hw), L.size("umpr_bf16_tensor_bytes", N, cout, hw, hw)
        bx, by = torch.zeros(nbx, dtype=torch.uint8, device=dev), torch.zeros(nby, dtype=torch.uint8, device=dev)
        L.call("umpr_bf16_from_nchw_f32", torch.randn(N, cin, hw, hw, device=dev), bx, N, cin, hw, hw, 0)
        L.call("umpr_bf16_from_nchw_f32", torch.randn(N, cout, hw, hw, device=dev), by, N, cout, hw, hw, 0)
        wb = L.size("umpr_conv3x3_bf16_ws_bytes", N, cin, cout, hw, hw)
        cw = torch.empty(wb, dtype=torch.uint8, device=dev)
        gw, gb = torch.empty(cout, cin, 3, 3, device=dev), torch.empty(cout, device=dev)
        keep.append((bx, by, cw, gw, gb))
        co[f"bf16 weight gradient {cin}->{cout} @{hw}"] = (lambda by=by, bx=bx, gw=gw, gb=gb, cin=cin, cout=cout, hw=hw, cw=cw, wb=wb:
                                                          L.call("umpr_conv3x3_bf16_bwd_weight", by, bx, gw, gb, N, cin, hw, hw, cout, cw, wb, sb.cuda_stream))
    for n2 in (1, 16, 64):   # the 14x14 layer at other batch sizes: the main kernel's work scales with n2, the reduce kernel's does not
        cin = cout = 512; hw = 14
        nbx = L.size("umpr_bf16_tensor_bytes", n2, cin, hw, hw)
        bx, by = torch.zeros(nbx, dtype=torch.uint8, device=dev), torch.zeros(nbx, dtype=torch.uint8, device=dev)
        L.call("umpr_bf16_from_nchw_f32", torch.randn(n2, cin, hw, hw, device=dev), bx, n2, cin, hw, hw, 0)
        L.call("umpr_bf16_from_nchw_f32", torch.randn(n2, cout, hw, hw, device=dev), by, n2, cout, hw, hw, 0)
        wb = L.size("umpr_conv3x3_bf16_ws_bytes", n2, cin, cout, hw, hw)
        cw = torch.empty(wb, dtype=torch.uint8, device=dev)
        gw, gb = torch.empty(cout, cin, 3, 3, device=dev), torch.empty(cout, device=dev)
        keep.append((bx, by, cw, gw, gb))
        co[f"bf16 weight gradient 512->512 @14, {n2} images"] = (lambda by=by, bx=bx, gw=gw, gb=gb, cw=cw, wb=wb, n2=n2:
                                                                 L.call("umpr_conv3x3_bf16_bwd_weight", by, bx, gw, gb, n2, 512, 14, 14, 512, cw, wb, sb.cuda_stream))
    tag = "UMPR_MERGE_DX=" + os.environ.get("UMPR_MERGE_DX", "default")
    for name, fn in co.items():
        bad = 0
        for it in range(a.launches):
            for _ in range(3):
                fn()
            dx()
            sa.synchronize()
            got = torch.cat([outs[0], outs[1]])
            if not torch.equal(got, ref):
                bad += 1
        torch.cuda.synchronize()
        print(f"[{tag}] beside {name:36s}: {bad} of {a.launches} launches differ from the quiet-device result", flush=True)


if __name__ == "__main__":
    main()
