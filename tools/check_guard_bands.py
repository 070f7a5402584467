#!/usr/bin/env python3
"""Do the bf16 convolution-stack kernels stay inside the buffers they are given?  Every output / scratch buffer of each call is a
slice in the middle of a larger allocation filled with a sentinel; after the call the bands on both sides must be untouched.
(An out-of-bounds WRITE does not show in any parity test - it lands in whatever the allocator placed next door - but makes an
unrelated result depend on timing: the text path's gradients, computed on another stream, were what made us look.)

    python tools/check_guard_bands.py [--n 4]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from umpr_amd._lib import lib

GUARD = 1 << 20
LAYERS = [(64, 64, 224), (64, 128, 112), (128, 128, 112), (128, 256, 56), (256, 256, 56), (256, 512, 28), (512, 512, 28),
          (512, 512, 14)]


class Banded:
    def __init__(self, nbytes, dev, fill=0xA5):
        self.n = (int(nbytes) + 255) // 256 * 256
        self.big = torch.full((self.n + 2 * GUARD,), fill, dtype=torch.uint8, device=dev)
        self.fill = fill

    def u8(self):
        return self.big[GUARD:GUARD + self.n]

    def f32(self, shape):
        return self.u8()[:4 * int(torch.tensor(shape).prod())].view(torch.float32).view(*shape)

    def intact(self):
        lo, hi = self.big[:GUARD], self.big[GUARD + self.n:]
        return bool((lo == self.fill).all()) and bool((hi == self.fill).all())

    def where(self):
        out = []
        for name, band, base in (("below", self.big[:GUARD], -GUARD), ("above", self.big[GUARD + self.n:], self.n)):
            nz = (band != self.fill).nonzero().reshape(-1)
            if nz.numel():
                out.append(f"{name}: {nz.numel()} bytes, offsets {base + int(nz.min())}..{base + int(nz.max())} relative to the buffer")
        return "; ".join(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=4)
    a = ap.parse_args()
    L = lib()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    N, bad = a.n, 0

    def report(what, bufs):
        nonlocal bad
        torch.cuda.synchronize()
        for name, b in bufs:
            if not b.intact():
                bad += 1
                print(f"OUT OF BOUNDS  {what}: {name}: {b.where()}", flush=True)

    for Cin, Cout, HW in LAYERS:
        what = f"N{N} {Cin}->{Cout} @{HW}"
        nbx = L.size("umpr_bf16_tensor_bytes", N, Cin, HW, HW)
        nby = L.size("umpr_bf16_tensor_bytes", N, Cout, HW, HW)
        wsb = L.size("umpr_conv3x3_bf16_ws_bytes", N, Cin, Cout, HW, HW)
        x = torch.zeros(nbx, dtype=torch.uint8, device=dev)
        L.call("umpr_bf16_from_nchw_f32", torch.randn(N, Cin, HW, HW, device=dev), x, N, Cin, HW, HW, st)
        w = torch.randn(Cout, Cin, 3, 3, device=dev) / (Cin * 9) ** 0.5
        b = torch.randn(Cout, device=dev)
        ws, y, dx = Banded(wsb, dev), Banded(nby, dev), Banded(nbx, dev)
        dw, db = Banded(w.numel() * 4, dev), Banded(Cout * 4, dev)
        L.call("umpr_conv3x3_bf16_fwd", x, w, b, y.u8(), N, Cin, HW, HW, Cout, 1, ws.u8(), wsb, st)
        report(what + " fwd", [("ws", ws), ("y", y)])
        L.call("umpr_conv3x3_bf16_bwd_data", y.u8(), w, None, dx.u8(), N, Cin, HW, HW, Cout, ws.u8(), wsb, st)
        report(what + " dgrad", [("ws", ws), ("dx", dx)])
        L.call("umpr_conv3x3_bf16_bwd_data", y.u8(), w, x, dx.u8(), N, Cin, HW, HW, Cout, ws.u8(), wsb, st)
        report(what + " dgrad+mask", [("ws", ws), ("dx", dx)])
        L.call("umpr_conv3x3_bf16_bwd_weight", y.u8(), x, dw.f32(w.shape), db.f32(b.shape), N, Cin, HW, HW, Cout, ws.u8(), wsb, st)
        report(what + " wgrad", [("ws", ws), ("dw", dw), ("db", db)])
        if HW > 14 or True:
            p = Banded(L.size("umpr_bf16_tensor_bytes", N, Cout, HW // 2, HW // 2), dev)
            gx = Banded(nby, dev)
            L.call("umpr_maxpool2_bf16_fwd", y.u8(), p.u8(), N, Cout, HW, HW, st)
            report(what + " pool fwd", [("pooled", p)])
            L.call("umpr_maxpool2_bf16_bwd_relu", y.u8(), p.u8(), gx.u8(), N, Cout, HW, HW, st)
            report(what + " pool bwd", [("gx", gx)])
        print(f"checked {what}", flush=True)
    print("guard bands intact" if not bad else f"{bad} buffers written out of bounds")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
