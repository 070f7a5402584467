"""bf16 mixed-precision path (BASELINE.json configs[4]: "MFMA bf16 conv + attention; MSE within 1e-3 of fp32").

Per-layer: the bf16 implicit-GEMM convolution (forward, data gradient, weight gradient) at the PRODUCTION VGG16 shapes
against torch's fp32 F.conv2d.  Two references per case:
  * "q": F.conv2d in fp32 on the bf16-ROUNDED inputs - isolates the kernel (products of bf16 values are exact in fp32, so
    only the summation order and the final rounding to bf16 differ): tight elementwise bound;
  * "f": F.conv2d on the unrounded fp32 inputs - the stated bf16 bound: relative L2 error <= 2^-7 (one bf16 rounding of
    each operand, 2^-9 relative each, plus the output rounding; errors add in quadrature over K).
End to end: eval MSE / predictions of the bf16 model against the committed fp32 golden fixtures (reference outputs)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import ROOT, load_golden

pytestmark = pytest.mark.gpu
LOG = os.path.join(ROOT, "gpurun_out", "parity_bf16.log")
BF16_REL_L2 = 2.0 ** -7


def log(msg):
    os.makedirs(os.path.dirname(LOG), exist_ok=True)
    with open(LOG, "a") as f:
        f.write(msg + "\n")


@pytest.fixture(scope="module")
def L():
    from umpr_amd._lib import lib
    return lib()


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def st():
    return torch.cuda.current_stream().cuda_stream


@pytest.fixture(autouse=True)
def poison_lds(L, dev):
    sink = torch.zeros(1, dtype=torch.int32, device=dev)
    L.call("umpr_debug_poison_lds", sink, st())
    yield


def to_cb8(L, x, dev, poison=True):
    """fp32 NCHW host tensor -> device CB8-PF bf16 tensor (bytes)."""
    N, C, H, W = x.shape
    nb = L.size("umpr_bf16_tensor_bytes", N, C, H, W)
    y = torch.full((nb,), 0xFF if poison else 0, dtype=torch.uint8, device=dev)   # 0xFFFF bf16 = NaN: unwritten pads show
    L.call("umpr_bf16_from_nchw_f32", x.to(dev).contiguous(), y, N, C, H, W, st())
    return y


def from_cb8(L, y, shape, dev):
    N, C, H, W = shape
    out = torch.full(shape, float("nan"), device=dev)
    L.call("umpr_bf16_to_nchw_f32", y, out, N, C, H, W, st())
    return out.cpu()


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def q(t):
    return t.bfloat16().float()


def test_layout_roundtrip_and_pads(L, dev):
    """fp32 -> CB8-PF -> fp32 is the bf16 rounding of the input; every pad / guard pixel of the tensor is zero."""
    g = torch.Generator().manual_seed(1)
    for shape in [(2, 64, 14, 14), (3, 8, 28, 28), (1, 24, 56, 56)]:
        x = torch.randn(shape, generator=g)
        y = to_cb8(L, x, dev)
        back = from_cb8(L, y, shape, dev)
        assert torch.equal(back, q(x))
        N, C, H, W = shape
        planes = (C + 7) // 8
        img = y.view(torch.bfloat16).view(planes, -1, 8).float().cpu()          # [plane][pixel incl. guards][8]
        assert torch.isfinite(img).all()
        total = float(img.abs().sum())
        assert abs(total - float(q(x).abs().sum())) <= 1e-3 * total              # nothing but the real pixels is non-zero


# the 12 bf16 conv layers of VGG16 (the first, 3 -> 64, runs the fp32 first-layer kernel) + batches that make tiles
# straddle images / rows, + the 512-pixel tile of the 64-channel layers on every map width
SHAPES = [(1, 64, 64, 224), (2, 64, 64, 224), (1, 64, 128, 112), (2, 128, 128, 112), (1, 128, 256, 56), (2, 256, 256, 56),
          (1, 256, 512, 28), (3, 512, 512, 28), (5, 512, 512, 14), (3, 64, 64, 56), (2, 128, 64, 28), (7, 64, 128, 14)]


@pytest.mark.parametrize("N,Cin,Cout,HW", SHAPES)
def test_conv3x3_bf16(L, dev, N, Cin, Cout, HW):
    g = torch.Generator().manual_seed(N + Cin + Cout + HW)
    x = torch.randn(N, Cin, HW, HW, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g)
    gy = torch.randn(N, Cout, HW, HW, generator=g)
    # references
    y_f = F.relu(F.conv2d(x, w, b, padding=1))
    xq, wq = q(x).requires_grad_(True), q(w).requires_grad_(True)
    y_q = F.relu(F.conv2d(xq, wq, b, padding=1))
    gz_q = q(gy * (y_q > 0))                         # gradient w.r.t. the pre-activation, as the bf16 path stores it
    pre = F.conv2d(xq, wq, b, padding=1)
    pre.backward(gz_q)
    xd = to_cb8(L, x, dev)
    wsb = L.size("umpr_conv3x3_bf16_ws_bytes", N, Cin, Cout, HW, HW)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    nb = L.size("umpr_bf16_tensor_bytes", N, Cout, HW, HW)
    yd = torch.full((nb,), 0xFF, dtype=torch.uint8, device=dev)
    wd, bd = w.to(dev), b.to(dev)
    L.call("umpr_conv3x3_bf16_fwd", xd, wd, bd, yd, N, Cin, HW, HW, Cout, 1, ws, wsb, st())
    y = from_cb8(L, yd, (N, Cout, HW, HW), dev)
    assert torch.isfinite(y).all()
    eq, ef = rel_l2(y, y_q), rel_l2(y, y_f)
    err = (y - y_q.detach()).abs()
    tol = 2.0 ** -8 * y_q.detach().abs() + 1e-3
    log(f"bf16 conv fwd {N},{Cin},{Cout},{HW}: relL2 vs q {eq:.2e} vs f32 {ef:.2e} max|err| {float(err.max()):.2e} bad {int((err > tol).sum())}")
    assert int((err > tol).sum()) == 0 and ef <= BF16_REL_L2
    # the zero pads of the output must really be zero (the next layer's taps read them): every element of the CB8-PF
    # tensor that is not a real pixel
    img = yd.view(torch.bfloat16).float()
    assert torch.isfinite(img).all()
    assert abs(float(img.abs().sum()) - float(y.abs().sum())) <= 2e-3 * float(y.abs().sum()) + 1e-3
    # data gradient, plain and with the ReLU mask of the layer below fused
    gzd = to_cb8(L, gz_q, dev)
    nbx = L.size("umpr_bf16_tensor_bytes", N, Cin, HW, HW)
    dxd = torch.full((nbx,), 0xFF, dtype=torch.uint8, device=dev)
    L.call("umpr_conv3x3_bf16_bwd_data", gzd, wd, None, dxd, N, Cin, HW, HW, Cout, ws, wsb, st())
    dx = from_cb8(L, dxd, (N, Cin, HW, HW), dev)
    err = (dx - xq.grad).abs()
    tol = 2.0 ** -8 * xq.grad.abs() + 1e-3 * float(xq.grad.abs().max())
    log(f"bf16 conv dgrad {N},{Cin},{Cout},{HW}: relL2 {rel_l2(dx, xq.grad):.2e} max|err| {float(err.max()):.2e} bad {int((err > tol).sum())}")
    assert int((err > tol).sum()) == 0
    mask_src = torch.randn(N, Cin, HW, HW, generator=g)
    md = to_cb8(L, mask_src, dev)
    L.call("umpr_conv3x3_bf16_bwd_data", gzd, wd, md, dxd, N, Cin, HW, HW, Cout, ws, wsb, st())
    dxm = from_cb8(L, dxd, (N, Cin, HW, HW), dev)
    assert torch.equal(dxm, dx * (q(mask_src) > 0)), "masked data gradient differs from mask x plain"
    # weight / bias gradient: fp32 outputs, bf16 products are exact in fp32 - only the summation order differs
    dw = torch.full(w.shape, float("nan"), device=dev)
    db = torch.full(b.shape, float("nan"), device=dev)
    L.call("umpr_conv3x3_bf16_bwd_weight", gzd, xd, dw, db, N, Cin, HW, HW, Cout, ws, wsb, st())
    dwc, dbc = dw.cpu(), db.cpu()
    sw, sb = float(wq.grad.abs().max()), float(gz_q.sum((0, 2, 3)).abs().max())
    ew, eb = float((dwc - wq.grad).abs().max()), float((dbc - gz_q.sum((0, 2, 3))).abs().max())
    log(f"bf16 conv wgrad {N},{Cin},{Cout},{HW}: max|err| {ew:.2e} of {sw:.2e}; bias {eb:.2e} of {sb:.2e}")
    assert ew <= 1e-4 * sw + 1e-6 and eb <= 1e-4 * sb + 1e-5


def test_maxpool_bf16(L, dev):
    g = torch.Generator().manual_seed(5)
    N, C, H, W = 3, 16, 28, 28
    x = q(torch.relu(torch.randn(N, C, H, W, generator=g) + 0.3)).requires_grad_(True)
    y_ref = F.max_pool2d(x, 2, 2)
    gy = q(torch.randn(y_ref.shape, generator=g))
    y_ref.backward(gy)
    xd = to_cb8(L, x.detach(), dev)
    yd = torch.full((L.size("umpr_bf16_tensor_bytes", N, C, H // 2, W // 2),), 0xFF, dtype=torch.uint8, device=dev)
    L.call("umpr_maxpool2_bf16_fwd", xd, yd, N, C, H, W, st())
    assert torch.equal(from_cb8(L, yd, (N, C, H // 2, W // 2), dev), y_ref.detach())
    gyd = to_cb8(L, gy, dev)
    gxd = torch.full((L.size("umpr_bf16_tensor_bytes", N, C, H, W),), 0xFF, dtype=torch.uint8, device=dev)
    L.call("umpr_maxpool2_bf16_bwd_relu", xd, gyd, gxd, N, C, H, W, st())
    gx = from_cb8(L, gxd, (N, C, H, W), dev)
    # ties (several equal maxima in a window, likely among the ReLU zeros) route to the FIRST maximum on both sides only if
    # torch does the same; zeros never receive gradient through the fused ReLU mask, so compare where x > 0
    assert torch.equal(gx, x.grad * (x.detach() > 0))
    assert torch.isfinite(gxd.view(torch.bfloat16).float()).all()


@pytest.mark.parametrize("M,N,K,ta,tb,ws", [(300, 384, 300, 0, 1, 0), (1000, 128, 128, 0, 0, 0), (384, 300, 5000, 1, 0, 1),
                                            (64, 64, 40, 0, 1, 0), (130, 70, 52, 1, 1, 0)])
def test_gemm_bf16_operands(L, dev, M, N, K, ta, tb, ws):
    """umpr_gemm_f32 under umpr_set_gemm_bf16(1) (the text path's products in bf16 mode) against the float64 product of the
    bf16-rounded operands, all four operand layouts, ragged edges, split-K: only the fp32 summation differs."""
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn((K, M) if ta else (M, K), generator=g).to(dev)
    B = torch.randn((N, K) if tb else (K, N), generator=g).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    C = torch.full((M, N), float("nan"), device=dev)
    wsb = 64 * M * N * 4 if ws else 0
    wst = torch.empty(max(wsb // 4, 1), device=dev)
    L.call("umpr_set_gemm_bf16", 1)
    try:
        L.call("umpr_gemm_f32", A, A.shape[1], ta, B, B.shape[1], tb, C, N, M, N, K, bias, 1, 0, 0, 1.0,
               wst if ws else None, wsb, st())
    finally:
        L.call("umpr_set_gemm_bf16", 0)
    opA = q(A).double().t() if ta else q(A).double()
    opB = q(B).double().t() if tb else q(B).double()
    ref = (opA @ opB + bias.double()).float()
    e = rel_l2(C.cpu(), ref.cpu())
    log(f"gemm bf16 operands M{M} N{N} K{K} ta{ta} tb{tb} splitk{ws}: relL2 {e:.2e}")
    assert torch.isfinite(C).all() and e <= 1e-5
    # and the switch is off again: the same call now gives the fp32 product
    L.call("umpr_gemm_f32", A, A.shape[1], ta, B, B.shape[1], tb, C, N, M, N, K, bias, 1, 0, 0, 1.0,
           wst if ws else None, wsb, st())
    opA = A.double().t() if ta else A.double()
    opB = B.double().t() if tb else B.double()
    e32 = rel_l2(C.cpu(), (opA @ opB + bias.double()).float().cpu())
    assert e32 <= 1e-5, e32


class _QLinear(torch.autograd.Function):
    """nn.Linear with every matrix-product operand rounded to bf16 and fp32 accumulation: what the bf16 classifier
    kernels compute (bias and bias gradient stay fp32).  The products run in float64 so that the reference does not
    depend on which fp32 GEMM algorithm the BLAS library picks for a shape."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return (q(x).double() @ q(w).double().t() + b.double()).float()

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        gq = q(g).double()
        return (gq @ q(w).double()).float(), (gq.t() @ q(x).double()).float(), g.double().sum(0).float()


@pytest.mark.parametrize("n", [4, 64])
def test_classifier_bf16_through_c_abi(L, dev, n):
    """umpr_vgg16_classifier_{fwd,bwd}_compact_bf16 (eval mode: no dropout) against the same three layers in torch with
    bf16-rounded operands.  The first layer's output is a single product of rounded operands: only the fp32 summation
    order differs (<= 2e-5 relative L2).  Behind it the ~1e-5 summation noise of a K = 25088 dot product moves a few per
    mille of the activations across a bf16 rounding boundary when the NEXT product rounds them (one bf16 ulp = 4e-3
    each), so outputs and gradients further down agree to ~2e-4 (measured); the bound there is 1e-3, a quarter of one bf16 rounding -
    a wrong index anywhere in a kernel gives O(1).  n = 64 is the bench batch."""
    from umpr_amd.model import _ptr_array
    g = torch.Generator().manual_seed(11 + n)
    dims = [(25088, 4096), (4096, 4096), (4096, 1000)]
    params = []
    for fin, fout in dims:
        params += [(torch.randn(fout, fin, generator=g) * (2.0 / fin) ** 0.5).to(dev),
                   (torch.randn(fout, generator=g) * 0.1).to(dev)]
    x = torch.rand(n, 25088, generator=g).to(dev)
    gout = torch.randn(n, 1000, generator=g).to(dev)
    # reference
    ps = [p.clone().requires_grad_(True) for p in params]
    xr = x.clone().requires_grad_(True)
    h = torch.relu(_QLinear.apply(xr, ps[0], ps[1]))
    h = torch.relu(_QLinear.apply(h, ps[2], ps[3]))
    ref = _QLinear.apply(h, ps[4], ps[5])
    ref.backward(gout)
    # C ABI
    arena = torch.empty(L.size("umpr_vgg16_cls_arena_bytes", n) // 4, device=dev, dtype=torch.float32)
    arena[: n * 25088] = x.reshape(-1)
    masks = torch.empty(2, n, 4096, device=dev, dtype=torch.uint8)
    out = torch.empty(n, 1000, device=dev)
    wsb = L.size("umpr_vgg16_fwd_ws_bytes", n)
    ws = torch.empty(wsb // 4 + 1, device=dev)
    keep, parr = _ptr_array([params[0]] * 26 + params)
    L.call("umpr_vgg16_classifier_fwd_compact_bf16", parr, n, 0, 0, 0, arena, masks, out, ws, wsb, st())
    h1 = torch.relu(_QLinear.apply(x, params[0], params[1]))
    e1 = rel_l2(arena[n * 25088: n * 25088 + n * 4096].reshape(n, 4096).cpu(), h1.cpu())
    e = rel_l2(out.cpu(), ref.detach().cpu())
    log(f"classifier bf16 n{n}: fc1 relL2 {e1:.2e} out relL2 {e:.2e}")
    assert torch.isfinite(out).all() and e1 <= 2e-5 and e <= 1e-3
    grads = [torch.full_like(p, float("nan")) for p in params]
    d_pool5 = torch.full((n, 25088), float("nan"), device=dev)
    wsb2 = L.size("umpr_vgg16_classifier_bwd_ws_bytes", n)
    ws2 = torch.empty(wsb2 // 4 + 1, device=dev)
    keep_g, garr = _ptr_array([grads[0]] * 26 + grads)
    L.call("umpr_vgg16_classifier_bwd_compact_bf16", parr, n, 0, arena, masks, gout, garr, d_pool5, ws2, wsb2, st())
    torch.cuda.synchronize()
    bad = {}
    for i, (gr, pr) in enumerate(zip(grads, ps)):
        e = rel_l2(gr.cpu(), pr.grad.cpu())
        log(f"classifier bf16 n{n}: d param {i} relL2 {e:.2e}")
        if not (torch.isfinite(gr).all() and e <= 1e-3):
            bad[i] = e
    e = rel_l2(d_pool5.cpu(), xr.grad.cpu())
    log(f"classifier bf16 n{n}: d pool5 relL2 {e:.2e}")
    assert torch.isfinite(d_pool5).all() and e <= 1e-3 and not bad, (e, bad)


def _vgg_pair(dev, seed):
    from umpr_amd.model import VGG16
    torch.manual_seed(seed)
    ref = VGG16().to(dev).eval()
    low = VGG16(dtype="bf16").to(dev).eval()
    low.load_state_dict(ref.state_dict())
    return ref, low


def test_vgg16_bf16_vs_fp32_path(dev):
    """The whole VGG16 in bf16 against the fp32 HIP path (itself pinned to torch).  Forward: 13 bf16 layers, each adding
    ~2e-3 relative rounding noise (inputs, weights, output) in quadrature -> ~7e-3 relative L2 at the 1000-d output.
    Gradients: that forward noise flips the ReLU / max-pool decision of the ~0.5 % of units whose pre-activation sits
    within it of zero (measured: classifier.3 already differs by 6e-2 with the classifier itself in fp32), and every layer
    further down adds its own flips - an intrinsic property of bf16 activations, not of the kernels (those are pinned
    elementwise by test_conv3x3_bf16).  So the bound is on direction: cosine >= 0.9 and relative L2 <= 0.5 for every
    parameter, tightening towards the output."""
    ref, low = _vgg_pair(dev, 3)
    g = torch.Generator().manual_seed(4)
    x = torch.rand(4, 3, 224, 224, generator=g).to(dev)
    gout = torch.randn(4, 1000, generator=g).to(dev)
    y0 = ref(x)
    y1 = low(x)
    e = rel_l2(y1.detach().cpu(), y0.detach().cpu())
    log(f"vgg16 bf16 fwd: relL2 {e:.2e}")
    assert torch.isfinite(y1).all() and e <= 2e-2
    y0.backward(gout)
    y1.backward(gout)
    bad = {}
    for (k, p0), (_, p1) in zip(ref.named_parameters(), low.named_parameters()):
        assert torch.isfinite(p1.grad).all(), k
        e = rel_l2(p1.grad.cpu(), p0.grad.cpu())
        cos = float((p1.grad * p0.grad).sum() / (p1.grad.norm() * p0.grad.norm() + 1e-30))
        log(f"vgg16 bf16 d{k}: relL2 {e:.2e} cosine {cos:.4f} |g| {float(p0.grad.norm()):.3e}")
        if e > 0.5 or cos < 0.9 or (k.startswith("classifier.6") and e > 2e-2):
            bad[k] = (e, cos)
    assert not bad, bad


class _QConvRelu(torch.autograd.Function):
    """conv3x3 + bias + ReLU as the bf16 path computes it: both MFMA operands bf16 (activations are stored in bf16, weights are
    rounded when packed), fp32 accumulation from the bias, ReLU, output ROUNDED to bf16; backward: the incoming gradient is a
    bf16 tensor, the data gradient is rounded to bf16 on store, weight / bias gradients stay fp32.  CPU, torch fp32 convolutions
    on the rounded operands (products of bf16 values are exact in fp32; only the summation order differs from the kernels)."""

    @staticmethod
    def forward(ctx, x, w, b):
        xq, wq = q(x), q(w)
        y = q(torch.relu(F.conv2d(xq, wq, b, padding=1)))
        ctx.save_for_backward(xq, wq, y)
        return y

    @staticmethod
    def backward(ctx, gy):
        xq, wq, y = ctx.saved_tensors
        gz = q(gy) * (y > 0)
        gx = q(torch.nn.grad.conv2d_input(xq.shape, wq, gz, padding=1))
        gw = torch.nn.grad.conv2d_weight(xq, wq.shape, gz, padding=1)
        return gx, gw, gz.sum((0, 2, 3))


def test_vgg16_bf16_vs_rounded_operand_oracle(L, dev):
    """The bf16 VGG16 feature stack against a CPU network that rounds in exactly the same places (_QConvRelu), on the network's
    own activations (a quantised-operand oracle, not only the direction against the fp32 path).
    (a) Layer by layer, each bf16 layer fed the ORACLE's input: relative L2 <= 1e-4 and at most 5e-4 of the outputs one bf16
        step away (measured 2.4-4.9e-5 and 0.4-1.5e-4: fp32 summation order moves a few outputs across a rounding boundary).
    (b) The whole stack through the model's own Function: pool5 within 1e-2 relative L2 (measured 4.9e-3).  That is the FLOOR
        for any two implementations of the same bf16 network, not kernel error: an output that rounds the other way perturbs the
        next layer's sums, which moves sqrt(eps x 2^-8) of ITS outputs across a boundary - the recursion settles at a fraction
        of a bf16 step (2^-8 = 3.9e-3) whatever the starting difference; 43 % of the pool5 values differ by one step.  So
        whole-network bf16 results can only be bounded at this level, and the per-layer form (a) is the pin."""
    from umpr_amd.model import VGG16, _VGGFeaturesBF16
    from umpr_amd.synthetic import VGG16_CFG
    torch.manual_seed(7)
    m = VGG16(dtype="bf16").eval()
    g = torch.Generator().manual_seed(8)
    x = torch.rand(2, 3, 224, 224, generator=g)
    P = {k: v.detach() for k, v in m.named_parameters()}
    conv_idx = [i for i, mm in enumerate(m.features) if isinstance(mm, torch.nn.Conv2d)]
    h, ci = x, 0
    for v in VGG16_CFG:
        if v == "M":
            h = F.max_pool2d(h, 2, 2)
            continue
        i = conv_idx[ci]
        w, b = P[f"features.{i}.weight"], P[f"features.{i}.bias"]
        N, Cin, HW, Cout = h.shape[0], h.shape[1], h.shape[-1], w.shape[0]
        y_ref = _QConvRelu.apply(h, w, b)
        if Cin >= 64:          # (the 3-channel first layer has no per-layer bf16 entry point: it is covered by (b))
            xd = to_cb8(L, h, dev)
            wsb = L.size("umpr_conv3x3_bf16_ws_bytes", N, Cin, Cout, HW, HW)
            ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            yd = torch.full((L.size("umpr_bf16_tensor_bytes", N, Cout, HW, HW),), 0xFF, dtype=torch.uint8, device=dev)
            L.call("umpr_conv3x3_bf16_fwd", xd, w.to(dev), b.to(dev), yd, N, Cin, HW, HW, Cout, 1, ws, wsb, st())
            y = from_cb8(L, yd, (N, Cout, HW, HW), dev)
            e, frac = rel_l2(y, y_ref), float((y != y_ref).float().mean())
            log(f"vgg16 bf16 features.{i} on the oracle's input: relL2 {e:.2e}, outputs one bf16 step away {frac:.1e}")
            assert torch.isfinite(y).all() and e <= 1e-4 and frac <= 5e-4, (i, e, frac)
        h = y_ref
        ci += 1
    ref5 = h.flatten(1)          # VGG16_CFG ends with the fifth pool: h is [n, 512, 7, 7]
    m = m.to(dev)
    with torch.no_grad():
        pool5, _ = _VGGFeaturesBF16.apply(x.to(dev), *m.param_list()[:26])
    e = rel_l2(pool5.cpu(), ref5)
    log(f"vgg16 bf16 pool5, model vs rounded-operand oracle: relL2 {e:.2e}, differing {float((pool5.cpu() != ref5).float().mean()):.2f}")
    assert torch.isfinite(pool5).all() and e <= 1e-2


def _bf16_model(cfg_views, P, dev, dtype, review_net_only=False):
    from umpr_amd.config import Config
    from umpr_amd.model import UMPR
    Config.extend({"dtype": "fp32"})
    cfg = Config(argv=[])
    cfg.review_net_only = review_net_only
    cfg.views = cfg_views
    cfg.dtype = dtype
    model = UMPR(cfg, P["embedding.weight"].numpy())
    model.load_state_dict(P)
    return model.to(dev).eval()


@pytest.mark.parametrize("name", ["umpr_full_V1_B2", "umpr_full_V4_B2", "umpr_full_V2_P2_B2"])
def test_umpr_bf16_predictions_vs_fp32_golden(dev, name):
    """Per-sample predictions of the bf16 model (conv stack + attention scores in bf16) against the reference's own fp32
    outputs in the golden fixtures: the stated bf16 bound is 5e-2 absolute per prediction (7e-3 relative noise at the VGG
    output, see test_vgg16_bf16_vs_fp32_path).  These fixtures hold TWO samples of a random-initialised model whose
    predictions are 3-5 away from the labels, so their MSE moves by 2 * |pred - label| * |dpred| ~ 1e-2: the MSE criterion
    of configs[4] is a statement about an evaluation SET and is tested on one below."""
    from umpr_amd.synthetic import make_batch, make_param_state
    g = load_golden(name)
    B, V, ronly, pseed, bseed, full_pad, vocab = [int(v) for v in g["meta"]]
    P = make_param_state(pseed, 50, vocab, V, bool(ronly), m_scale=float(g["m_scale"]))
    batch = make_batch(bseed, B, vocab, V, int(g["photo_count"]) if "photo_count" in g else 1,
                       review_net_only=bool(ronly), full_pad=bool(full_pad))
    model = _bf16_model(["v%d" % i for i in range(V)], P, dev, "bf16", bool(ronly))
    with torch.no_grad():
        pred, loss = model(*batch)
    labels = batch[-1]
    mse_bf16 = float(((pred.cpu() - labels) ** 2).mean())
    mse_f32 = float(((torch.from_numpy(g["prediction"]) - labels) ** 2).mean())
    dp = float((pred.cpu() - torch.from_numpy(g["prediction"])).abs().max())
    log(f"{name} bf16: mse {mse_bf16:.6f} vs fp32 {mse_f32:.6f} (diff {abs(mse_bf16 - mse_f32):.2e}); max|dpred| {dp:.2e}; "
        f"loss {float(loss):.6f} vs {float(g['loss']):.6f}")
    assert dp <= 5e-2
    assert abs(float(loss) - float(g["loss"])) <= 5e-2


def test_bf16_eval_mse_within_1e3_of_fp32(dev):
    """north_star for configs[4]: "MSE within 1e-3 of fp32".  evaluate_mse (src/evaluate.py:6-14) over an evaluation set of
    1024 synthetic samples (16 batches of 64, the per-GPU batch of configs[4]) with the SAME weights in fp32 and in bf16
    mixed precision.  The output bias is first calibrated so that the mean prediction equals the mean label - what the
    first steps of training do - which puts the residuals at the scale of the reference's published test MSEs (1 - 2)
    instead of the 3 - 5 of an uncalibrated random initialisation."""
    from umpr_amd.synthetic import make_batch, make_param_state
    from umpr_amd.train import evaluate_mse
    P = make_param_state(401, 50, 2000, 1, False, m_scale=0.05)
    batches = [make_batch(410 + i, 64, 2000, 1) for i in range(16)]
    m32 = _bf16_model(["unknown"], P, dev, "fp32")
    with torch.no_grad():
        mean_pred = float(torch.cat([m32(*b)[0] for b in batches[:4]]).mean())
    mean_label = float(torch.cat([b[-1] for b in batches]).mean())
    P["linear_fusion.0.bias"] = P["linear_fusion.0.bias"] + (mean_label - mean_pred)
    m32 = _bf16_model(["unknown"], P, dev, "fp32")
    m16 = _bf16_model(["unknown"], P, dev, "bf16")
    mse32 = evaluate_mse(m32, batches)
    mse16 = evaluate_mse(m16, batches)
    with torch.no_grad():
        d = torch.cat([m16(*b)[0] - m32(*b)[0] for b in batches[:4]]).cpu()
    log(f"bf16 eval set: MSE fp32 {mse32:.6f} bf16 {mse16:.6f} diff {abs(mse16 - mse32):.2e}; per-sample dpred mean "
        f"{float(d.mean()):.2e} std {float(d.std()):.2e} max {float(d.abs().max()):.2e}")
    assert abs(mse16 - mse32) <= 1e-3


def test_bf16_training_step_runs_and_tracks_fp32(dev):
    """Two optimiser steps of the full model in bf16 mixed precision next to the fp32 path from the same start: finite,
    the losses agree to 1e-3 and the fp32 master weights move the same way (Adam's sign-like first steps make the
    parameter deltas robust to bf16 gradient noise)."""
    from umpr_amd.config import Config
    from umpr_amd.model import UMPR
    from umpr_amd.optim import FusedAdam
    from umpr_amd.synthetic import make_batch, make_param_state
    from umpr_amd.train import train_step
    Config.extend({"dtype": "fp32"})
    P = make_param_state(301, 50, 500, 1, False, m_scale=0.05)
    batches = [make_batch(310 + i, 3, 500, 1) for i in range(2)]
    out = {}
    for dt in ("fp32", "bf16"):
        cfg = Config(argv=[])
        cfg.views = ["unknown"]
        cfg.dtype = dt
        torch.manual_seed(5)
        m = UMPR(cfg, P["embedding.weight"].numpy())
        m.load_state_dict(P)
        m = m.to(dev)
        opt = FusedAdam(m, 1e-4, 1e-3)
        m.visual_net.vgg16[0].dropout_masks = torch.ones(2, 3, 4096, dtype=torch.uint8, device=dev)   # same (no) dropout
        losses = [float(train_step(m, opt, b)[1]) for b in batches]
        out[dt] = (losses, {k: v.detach().clone() for k, v in m.state_dict().items()})
    log(f"bf16 train losses {out['bf16'][0]} vs fp32 {out['fp32'][0]}")
    for a, b in zip(out["bf16"][0], out["fp32"][0]):
        assert np.isfinite(a) and abs(a - b) <= 1e-3 * max(1.0, abs(b)) + 1e-3
    # (features.5.weight sat at 0.819 against this 0.8 bound - ADVICE r2; the criterion that matters for training is
    # test_bf16_trained_model_evaluates_within_1e3_of_fp32_trained, this test only guards the direction of the first steps)
    for k in ("visual_net.vgg16.0.features.28.weight", "visual_net.vgg16.0.features.17.weight", "review_net.r_net.M"):
        d0 = out["fp32"][1][k] - P[k].to(dev)
        d1 = out["bf16"][1][k] - P[k].to(dev)
        cos = float((d0 * d1).sum() / (d0.norm() * d1.norm() + 1e-30))
        log(f"bf16 train delta {k}: cosine {cos:.4f}")
        assert cos > 0.8, (k, cos)


# ------------------------------------------------------------------------------------------------ configs[4] at its own width
# BASELINE.json configs[4] uses GloVe-300d.  The reference has no bf16 path, so every bound below is against the repo's own
# fp32 arithmetic / the fp32 oracle: "parity unpinned" for the bf16 mode.  Two references per text test:
#   "q" - the fp32 oracle fed the bf16-ROUNDED operands of the products the kernels run on the bf16 pipe: isolates the kernels
#         (what differs is the fp32 summation order and the rounding of intermediate GRADIENT operands): tight;
#   "f" - the fp32 oracle on the unrounded operands: the stated bf16 bound.
class _b16_mode:
    """What UMPR.forward does in bf16 mode: publish the mode to the text Functions of this thread."""

    def __enter__(self):
        from umpr_amd import model
        self.m = model
        self.prev = model._MODE.b16
        model._MODE.b16 = True

    def __exit__(self, *exc):
        self.m._MODE.b16 = self.prev


def test_embed_gru_bf16_glove300(L, dev):
    """_EmbedGru with the gather-projection GEMM [sum T x 300] x [300 x 384] on the bf16 pipe (umpr_set_gemm_bf16) at
    E = 300, N = 130 ragged sequences: forward within 1e-4 of the oracle run on bf16-rounded embedding rows and W_ih (the
    products of bf16 values are exact in fp32), <= 2e-2 absolute of the unrounded fp32 oracle; weight gradients within 1e-2
    relative L2 of the rounded-operand oracle (their products round dgx to bf16 as well) and 3e-2 of the fp32 one."""
    from oracle import umpr_ref as R
    from umpr_amd.model import UMPR, _EmbedGru
    N, Lmax, E = 130, 20, 300
    g = torch.Generator().manual_seed(77)
    vocab = 400
    emb = torch.randn(vocab, E, generator=g) * 0.4
    emb[:3] = 0
    lengths = torch.randint(1, Lmax + 1, (N,), generator=g)
    lengths[0] = Lmax
    ids = torch.randint(3, vocab, (N, Lmax), generator=g)
    for n in range(N):
        ids[n, lengths[n]:] = 0
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]
    shapes = [(192, E), (192, 64), (192,), (192,)]
    W = {nm + suf: (torch.rand(sh, generator=g) * 2 - 1) / 8 for suf in ("", "_reverse") for nm, sh in zip(names, shapes)}
    gout = torch.randn(N, Lmax, 128, generator=g)
    refs = {}
    for tag in ("f", "q"):
        P = {"g." + k: (q(v) if (tag == "q" and k.startswith("weight_ih")) else v.clone()).requires_grad_(True) for k, v in W.items()}
        x = F.embedding(ids, q(emb) if tag == "q" else emb)
        out = R.improved_rnn(x, lengths, P, "g.", aten=False)
        out.backward(gout)
        refs[tag] = (out.detach(), {k: P["g." + k].grad for k in W})
    lens, order = UMPR._host_perm(lengths, dev)
    w = [W[n + s].to(dev).requires_grad_(True) for s in ("", "_reverse") for n in names]
    with _b16_mode():
        out = _EmbedGru.apply(ids.to(dev), lens, order, emb.to(dev), 0, *w)
    out.backward(gout.to(dev))
    eq = float((out.detach().cpu() - refs["q"][0]).abs().max())
    ef = float((out.detach().cpu() - refs["f"][0]).abs().max())
    log(f"embed-gru bf16 E300: fwd max|err| vs rounded-operand oracle {eq:.2e}, vs fp32 oracle {ef:.2e}")
    assert torch.isfinite(out).all() and eq <= 1e-4 and ef <= 2e-2
    assert ef > 10 * eq, "the bf16 switch did not take effect (the result equals the fp32 product)"
    for t, k in zip(w, [n + s for s in ("", "_reverse") for n in names]):
        e_q, e_f = rel_l2(t.grad.cpu(), refs["q"][1][k]), rel_l2(t.grad.cpu(), refs["f"][1][k])
        log(f"embed-gru bf16 E300 d{k}: relL2 vs rounded-operand oracle {e_q:.2e}, vs fp32 oracle {e_f:.2e}")
        assert torch.isfinite(t.grad).all() and e_q <= 1e-2 and e_f <= 3e-2, (k, e_q, e_f)


def qs(x):
    """bf16 rounding with a straight-through gradient: the forward value is the rounded operand a bf16-pipe product sees, the
    backward treats the rounding as the identity (the kernels differentiate the fp32 function at the rounded forward values)."""
    return x + (q(x.detach()) - x.detach())


def _s_net_q(gru_repr, word_soft, sent_length, Ms, Ws, rounded):
    """oracle.umpr_ref.s_net (src/model.py:75-80) with the operands of its one GEMM-shaped product, tanh(Ms X), rounded."""
    r = qs if rounded else (lambda t: t)
    B = gru_repr.shape[0]
    S = gru_repr.shape[1] // sent_length
    X = gru_repr.reshape(B * S, sent_length, -1).transpose(-1, -2)
    sent_soft = torch.softmax(Ws @ torch.tanh(r(Ms) @ r(X)), dim=-1)
    self_atte = X @ sent_soft.transpose(-1, -2)
    senti = word_soft.reshape(B * S, -1).sum(dim=-1, keepdim=True) * self_atte.squeeze(-1)
    return self_atte.view(B, S, -1), senti.view(B, S, -1).sum(dim=-2)


@pytest.mark.parametrize("B,S,Lm,m_scale", [(3, 20, 20, 0.05), (4, 20, 20, 1.0), (2, 7, 9, 0.3)])
def test_review_head_bf16_mode_vs_oracle(L, dev, B, S, Lm, m_scale):
    """_ReviewHead in full bf16 mode - T = G_i M, the score contraction tanh(T G_u^T), the S-Net projections and the gradient
    products of all three on the bf16 pipe - against the oracle (src/model.py:50-55,71-81,166-168) in two forms:
      q: the oracle with exactly those operands rounded to bf16 (G_i, M, T, G_u, Ms, X): isolates the kernels.  Forward 2e-5
         like the fp32 test; the argmax routing then agrees, and the gradients are within 5e-3 relative L2 (the kernels also
         round the GRADIENT operands of their products to bf16, 2^-9 relative each);
      f: the unrounded fp32 oracle: outputs within 3e-2 absolute (the stated bf16 bound); gradients are logged only - a bf16
         score can pick another near-tied maximum, which re-routes the gradient."""
    from umpr_amd.model import _ReviewHead
    from umpr_amd.synthetic import make_param_state
    P0 = make_param_state(11, 50, 500, 1, False, with_vgg=False, m_scale=m_scale)
    g = torch.Generator().manual_seed(B * 100 + S + 1)
    gu0 = torch.randn(B, S * Lm, 128, generator=g) * 0.5
    gi0 = torch.randn(B, S * Lm, 128, generator=g) * 0.5
    pre = "review_net."
    keys = [pre + "r_net.M", pre + "s_net_u.Ms", pre + "s_net_u.Ws", pre + "s_net_i.Ms", pre + "s_net_i.Ws",
            pre + "linear_u.weight", pre + "linear_i.weight"]
    gout = torch.randn(B, 128, generator=g)
    refs = {}
    for tag in ("q", "f"):
        r = qs if tag == "q" else (lambda t: t)
        gru_u, gru_i = gu0.clone().requires_grad_(True), gi0.clone().requires_grad_(True)
        W = [P0[k].clone().requires_grad_(True) for k in keys]
        T = r(gru_i) @ r(W[0])
        A = torch.tanh(r(T) @ r(gru_u).transpose(-1, -2))
        soft_u = torch.softmax(A.max(dim=-2).values, -1)
        soft_i = torch.softmax(A.max(dim=-1).values, -1)
        atte_u = (gru_u.transpose(-1, -2) @ soft_u.unsqueeze(-1)).squeeze(-1)
        atte_i = (gru_i.transpose(-1, -2) @ soft_i.unsqueeze(-1)).squeeze(-1)
        _, su = _s_net_q(gru_u, soft_u, Lm, W[1], W[2], tag == "q")
        _, si = _s_net_q(gru_i, soft_i, Lm, W[3], W[4], tag == "q")
        ref = torch.tanh(F.linear(torch.cat([atte_u, su], -1), W[5]) + F.linear(torch.cat([atte_i, si], -1), W[6]))
        ref.backward(gout)
        refs[tag] = (ref.detach(), gru_u.grad, gru_i.grad, [w.grad for w in W])
    du, di = gu0.to(dev).requires_grad_(True), gi0.to(dev).requires_grad_(True)
    wd = [P0[k].to(dev).requires_grad_(True) for k in keys]
    with _b16_mode():
        out = _ReviewHead.apply(du, di, S, Lm, *wd, True)
    out.backward(gout.to(dev))
    eq, ef = float((out.detach().cpu() - refs["q"][0]).abs().max()), float((out.detach().cpu() - refs["f"][0]).abs().max())
    log(f"review head bf16 mode B{B} S{S} m{m_scale}: max|dout| vs rounded-operand oracle {eq:.2e}, vs fp32 oracle {ef:.2e}")
    assert torch.isfinite(out).all() and eq <= 2e-5 and ef <= 3e-2
    got = (du.grad.cpu(), di.grad.cpu(), [w.grad.cpu() for w in wd])
    for nm, gq, gf, gg in [("dGu", refs["q"][1], refs["f"][1], got[0]), ("dGi", refs["q"][2], refs["f"][2], got[1])] + \
            [("d" + k, a, b_, c) for k, a, b_, c in zip(keys, refs["q"][3], refs["f"][3], got[2])]:
        assert torch.isfinite(gg).all(), nm
        e_q, e_f = rel_l2(gg, gq), rel_l2(gg, gf)
        log(f"  {nm}: relL2 vs rounded-operand oracle {e_q:.2e}, vs fp32 oracle {e_f:.2e} (|g| {float(gq.norm()):.2e})")
        if float(gq.norm()) > 1e-6:
            assert e_q <= 5e-3, (nm, e_q)      # measured <= 2.4e-3


@pytest.mark.parametrize("B,S_ui,L_ui,S,Lm,V", [(3, 5, 20, 20, 20, 1), (4, 3, 11, 6, 9, 4)])
def test_control_bf16_mode_vs_oracle(L, dev, B, S_ui, L_ui, S, Lm, V):
    """_Control in bf16 mode - the C-Net sliding-window GEMM (Conv1d), the control S-Net projection and their gradient products
    on the bf16 pipe - against the oracle pieces of tests/test_gpu_parity.py::test_control in the same two forms as above:
    q (Conv1d input / weight and the S-Net's Ms, X rounded to bf16): forward 2e-5 + 1e-5 relative, like the fp32 test, gradients
    5e-3 relative L2; f (unrounded): logged - the hard gates (view_p < 0.35, view_score vs 0.5, src/model.py:124,192-194) and the
    max over positions make outputs and gradients discontinuous in the bf16 noise."""
    from umpr_amd.model import _Control
    from umpr_amd.synthetic import make_param_state
    P0 = make_param_state(13, 50, 500, V, False, with_vgg=False, m_scale=0.3)
    g = torch.Generator().manual_seed(B * 10 + V + 3)
    gs0 = [torch.randn(B, s * l, 128, generator=g) * 0.7 for s, l in ((S_ui, L_ui), (S, Lm), (S, Lm))]
    pre = "control_net."
    keys = [pre + "c_net.cnn.0.weight", pre + "c_net.cnn.0.bias", pre + "c_net.linear.0.weight", pre + "c_net.linear.0.bias",
            pre + "s_net.Ms", pre + "s_net.Ws", pre + "ss_net.linear.0.weight", pre + "ss_net.linear.0.bias"]
    gouts = [torch.randn(B, V, generator=g) for _ in range(4)]
    refs = {}
    for tag in ("q", "f"):
        r = qs if tag == "q" else (lambda t: t)
        gs = [t.clone().requires_grad_(True) for t in gs0]
        W = [P0[k].clone().requires_grad_(True) for k in keys]

        def head(x, s, l):
            cnn_in = x.reshape(B * s, l, -1).transpose(-1, -2)
            y = F.relu(F.conv1d(r(cnn_in), r(W[0]), W[1], padding=1)).max(dim=-1)[0].view(B, s, -1)
            vp = torch.sigmoid(F.linear(y, W[2], W[3]))
            vp = torch.where(vp < 0.35, torch.zeros_like(vp), vp)
            return vp, (vp ** 2).sum(-2)
        vp, c_out = head(gs[0], S_ui, L_ui)
        _, c_u = head(gs[1], S, Lm)
        _, c_i = head(gs[2], S, Lm)
        s_, _ = _s_net_q(gs[0], vp, L_ui, W[4], W[5], tag == "q")
        senti = torch.sigmoid(F.linear(s_, W[6], W[7])).expand(-1, -1, V)
        vs = (senti * vp ** 2).sum(-2) / ((vp ** 2).sum(-2) + 1e-4)
        q_p = (vs > 0.5).float()
        q_pos = torch.where(vs < 0.5, torch.zeros_like(vs), 4 * (vs - 0.5) ** 2)
        q_neg = torch.where(vs > 0.5, torch.zeros_like(vs), 4 * (0.5 - vs) ** 2)
        outs = [c_u, c_i, c_out * q_p * q_pos, c_out * (1 - q_p) * q_neg]
        torch.autograd.backward(outs, gouts)
        refs[tag] = ([o.detach() for o in outs], [t.grad for t in gs], [w.grad for w in W])
    gd = [t.to(dev).requires_grad_(True) for t in gs0]
    wd = [P0[k].to(dev).requires_grad_(True) for k in keys]
    with _b16_mode():
        outs = _Control.apply(gd[0], gd[1], gd[2], (B, S_ui, L_ui, S, Lm), 0.35, *wd)
    torch.autograd.backward(outs, [t.to(dev) for t in gouts])
    for i, nm in enumerate(("c_u", "c_i", "prefer_pos", "prefer_neg")):
        o = outs[i].detach().cpu()
        eq, ef = float((o - refs["q"][0][i]).abs().max()), float((o - refs["f"][0][i]).abs().max())
        log(f"control bf16 mode V{V} {nm}: max|err| vs rounded-operand oracle {eq:.2e}, vs fp32 oracle {ef:.2e} "
            f"of {float(refs['f'][0][i].abs().max()):.2e}")
        assert torch.isfinite(o).all() and eq <= 2e-5 + 1e-5 * float(refs["q"][0][i].abs().max()), (nm, eq)
    pairs = [(f"dG{i}", refs["q"][1][i], refs["f"][1][i], gd[i].grad.cpu()) for i in range(3)] + \
            [("d" + k, a, b_, w.grad.cpu()) for k, a, b_, w in zip(keys, refs["q"][2], refs["f"][2], wd)]
    for nm, gq, gf, gg in pairs:
        assert torch.isfinite(gg).all(), nm
        e_q, e_f = rel_l2(gg, gq), rel_l2(gg, gf)
        log(f"  {nm}: relL2 vs rounded-operand oracle {e_q:.2e}, vs fp32 oracle {e_f:.2e} (|g| {float(gq.norm()):.2e})")
        if float(gq.norm()) > 1e-6:
            assert e_q <= 5e-3, (nm, e_q)      # measured <= 2.4e-3


def test_bf16_eval_mse_within_1e3_of_fp32_glove300_b64(dev):
    """The MSE criterion of configs[4] at ITS width and per-GPU batch: GloVe-300d-shaped embedding, batches of 64, 1024 samples,
    same weights in fp32 and bf16 mixed precision (text products, co-attention scores, VGG conv stack and classifier in bf16)."""
    from umpr_amd.synthetic import make_batch, make_param_state
    from umpr_amd.train import evaluate_mse
    P = make_param_state(402, 300, 3000, 1, False, m_scale=0.05)
    batches = [make_batch(430 + i, 64, 3000, 1) for i in range(16)]
    # calibrate the output bias until the fp32 model's mean prediction over the WHOLE set equals the mean label (prediction =
    # ReLU(linear), so one shift is not enough when part of the set sits on the clipped side): what training does first.
    # Without it the residuals carry a bias of ~1 and the criterion measures 2 x bias x (mean bf16 shift of 8e-3) instead
    mean_label = float(torch.cat([b[-1] for b in batches]).mean())
    for it in range(6):
        m32 = _bf16_model(["unknown"], P, dev, "fp32")
        with torch.no_grad():
            mean_pred = float(torch.cat([m32(*b)[0] for b in batches]).mean())
        del m32
        log(f"bf16 eval set E300 B64: calibration pass {it}: mean pred {mean_pred:.4f} mean label {mean_label:.4f}")
        if abs(mean_pred - mean_label) < 1e-3:
            break
        P["linear_fusion.0.bias"] = P["linear_fusion.0.bias"] + (mean_label - mean_pred)
    m32 = _bf16_model(["unknown"], P, dev, "fp32")
    m16 = _bf16_model(["unknown"], P, dev, "bf16")
    assert m16.embedding.weight.shape[1] == 300 and m16.compute_dtype == "bf16"
    mse32 = evaluate_mse(m32, batches)
    mse16 = evaluate_mse(m16, batches)
    with torch.no_grad():
        d = torch.cat([m16(*b)[0] - m32(*b)[0] for b in batches[:4]]).cpu()
    log(f"bf16 eval set E300 B64: MSE fp32 {mse32:.6f} bf16 {mse16:.6f} diff {abs(mse16 - mse32):.2e}; per-sample dpred mean "
        f"{float(d.mean()):.2e} std {float(d.std()):.2e} max {float(d.abs().max()):.2e}")
    assert abs(mse16 - mse32) <= 1e-3


def test_bf16_trained_model_evaluates_within_1e3_of_fp32_trained(dev):
    """north_star's "MSE within 1e-3 of fp32" for TRAINING: the same initial weights, the same 64 batches
    of 16 samples, the same injected dropout masks, 64 Adam steps (main.py:22-37's optimiser; learning rate 2e-4 so that the
    loss moves: held-out MSE 10.83 -> 2.06; measured difference 4.7e-5, and 3.9e-4 / 1.3e-4 / 3.0e-5 with 64 steps at 1e-4 / 96 at
    2e-4 / 64 at 5e-4 - the less converged the run, the larger the residual bias that multiplies bf16's mean prediction shift) once in fp32 and once in bf16 mixed precision, GloVe-300d-shaped embedding; then evaluate_mse
    (src/evaluate.py:6-14) of both TRAINED models on a held-out synthetic set of 1024 samples, each in its own arithmetic."""
    from umpr_amd.config import Config
    from umpr_amd.model import UMPR
    from umpr_amd.optim import FusedAdam
    from umpr_amd.synthetic import make_batch, make_param_state
    from umpr_amd.train import evaluate_mse, train_step
    Config.extend({"dtype": "fp32"})
    E, vocab, steps, Bt = 300, 3000, 64, 16
    lr = 2e-4
    if os.environ.get("UMPR_TEST_TRAIN"):      # exploration only: "steps,lr"
        steps, lr = int(os.environ["UMPR_TEST_TRAIN"].split(",")[0]), float(os.environ["UMPR_TEST_TRAIN"].split(",")[1])
    P = make_param_state(501, E, vocab, 1, False, m_scale=0.05)
    train = [make_batch(510 + i, Bt, vocab, 1) for i in range(steps)]
    held = [make_batch(560 + i, 64, vocab, 1) for i in range(16)]
    gm = torch.Generator().manual_seed(9)
    masks = [(torch.rand(2, Bt, 4096, generator=gm) < 0.5).to(torch.uint8) for _ in range(steps)]
    res = {}
    for dt in ("fp32", "bf16"):
        cfg = Config(argv=[])
        cfg.views = ["unknown"]
        cfg.dtype = dt
        torch.manual_seed(5)
        m = UMPR(cfg, P["embedding.weight"].numpy())
        m.load_state_dict(P)
        m = m.to(dev)
        mse0 = evaluate_mse(m, held[:4])
        opt = FusedAdam(m, lr, 1e-3)
        losses = []
        for b, mk in zip(train, masks):
            m.visual_net.vgg16[0].dropout_masks = mk.to(dev)      # both modes see the same masks
            losses.append(float(train_step(m, opt, b)[1]))
        m.visual_net.vgg16[0].dropout_masks = None
        res[dt] = (mse0, losses, evaluate_mse(m, held))
        opt.close()
        del m, opt
        torch.cuda.empty_cache()
    (a0, la, a1), (b0, lb, b1) = res["fp32"], res["bf16"]
    log(f"bf16-trained vs fp32-trained: held-out MSE before {a0:.6f} / {b0:.6f}; train loss first {la[0]:.4f} / {lb[0]:.4f} last "
        f"{la[-1]:.4f} / {lb[-1]:.4f}; held-out MSE after {steps} steps: fp32 {a1:.6f} bf16 {b1:.6f} diff {abs(a1 - b1):.2e}")
    assert all(np.isfinite(lb)) and all(np.isfinite(la))
    assert a1 < a0 - 0.05, "the fp32 run did not learn (the criterion needs a learning rate that moves the loss)"
    assert abs(a1 - b1) <= 1e-3, (a1, b1)


def test_bf16_kernels_stay_inside_their_buffers(dev):
    """tools/check_guard_bands.py at 3 images: every output / scratch buffer of the bf16 conv, data-gradient, weight-gradient and pool
    entry points is a slice inside a sentinel-filled allocation; the 1 MiB bands on both sides are untouched after each call."""
    import importlib.util
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("check_guard_bands", os.path.join(root, "tools", "check_guard_bands.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    old = sys.argv
    sys.argv = ["check_guard_bands.py", "--n", "3"]
    try:
        assert mod.main() == 0
    finally:
        sys.argv = old
