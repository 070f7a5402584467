"""GPU parity tests: every HIP kernel is called through the C ABI (umpr_amd._lib -> libumpr_hip.so) and compared
with the oracle (oracle/umpr_ref.py, torch CPU) on the same seeded inputs, and with the committed golden fixtures
(outputs of the reference itself, tests/golden/).  Tolerances: forward / predictions / loss 1e-4 absolute fp32
(BASELINE.json north_star), gradients 1e-3 relative to the tensor's max; index work bit-exact."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import ROOT, load_golden

pytestmark = pytest.mark.gpu

# the child runs tools/run_gpu_children.py starts (UMPR_WINO_F4 = 0 / 2) log to files of their own: parity.log holds the
# DEFAULT mode's per-tensor margins only (tools/final_record.sh copies it into profiles/)
LOG = os.path.join(ROOT, "gpurun_out", "parity.log" if not os.environ.get("UMPR_TEST_CHILD")
                   else "parity_child_wino_f4_mode%s.log" % os.environ.get("UMPR_WINO_F4", "default"))


def log(msg):
    os.makedirs(os.path.dirname(LOG), exist_ok=True)
    with open(LOG, "a") as f:
        f.write(msg + "\n")


def check(name, got, ref, atol=1e-4, rtol=0.0, rel_to_max=None, max_bad_frac=0.0, rel_l2=None, max_bad=0):
    got = got.detach().float().cpu() if isinstance(got, torch.Tensor) else torch.as_tensor(got)
    ref = ref.detach().float().cpu() if isinstance(ref, torch.Tensor) else torch.as_tensor(np.asarray(ref)).float()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    if got.numel() == 0:
        return
    err = (got - ref).abs()
    scale = float(ref.abs().max())
    tol = atol + rtol * ref.abs()
    if rel_to_max is not None:
        tol = tol + rel_to_max * scale
    bad = int((err > tol).sum())
    log(f"{name}: max_err={float(err.max()):.3e} ref_max={scale:.3e} bad={bad}/{got.numel()} nan={int(torch.isnan(got).sum())}")
    assert not torch.isnan(got).any(), f"{name}: NaN in output"
    if rel_l2 is not None:
        l2 = float((got - ref).double().norm() / (ref.double().norm() + 1e-30))
        assert l2 <= rel_l2, f"{name}: relative L2 error {l2:.3e} > {rel_l2:.1e}"
    assert bad <= max(max_bad, max_bad_frac * got.numel()), \
        f"{name}: {bad}/{got.numel()} elements off, max err {float(err.max()):.3e} (ref max {scale:.3e})"


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
    """LDS keeps the previous kernel's bytes: poison it with NaN before every test so that a read of never-written
    LDS (0 * NaN) fails deterministically instead of once in a while."""
    sink = torch.zeros(1, dtype=torch.int32, device=dev)
    L.call("umpr_debug_poison_lds", sink, st())
    yield


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K,ta,tb", [
    (128, 128, 64, 0, 0), (200, 150, 50, 0, 1), (64, 384, 27, 1, 0), (130, 70, 1000, 1, 1), (32, 32, 2, 0, 0),
    (1, 1000, 4096, 0, 1), (64, 4096, 300, 0, 1), (192, 50, 5000, 1, 0), (257, 129, 17, 0, 0)])
def test_gemm(L, dev, M, N, K, ta, tb):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randn((K, M) if ta else (M, K), generator=g)
    B = torch.randn((N, K) if tb else (K, N), generator=g)
    bias = torch.randn(N, generator=g)
    ref = (A.t() if ta else A).double() @ (B.t() if tb else B).double()
    Ad, Bd, bd = A.to(dev), B.to(dev), bias.to(dev)
    for split in (False, True):
        C = torch.full((M, N), float("nan"), device=dev)
        ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev) if split else None
        L.call("umpr_gemm_f32", Ad, A.shape[1], ta, Bd, B.shape[1], tb, C, N, M, N, K, bd, 1, 1, 0, 1.0, ws,
               (64 << 20) if split else 0, st())
        torch.cuda.synchronize()
        check(f"gemm {M}x{N}x{K} ta{ta} tb{tb} split{split}", C, torch.relu(ref + bias.double()).float(),
              atol=1e-5 * max(1, K) ** 0.5, rtol=1e-5)
    # accumulate + alpha + row bias, identity check with an asymmetric operand (catches transposed C maps)
    C0 = torch.randn(M, N, generator=g)
    C = C0.to(dev)
    rb = torch.randn(M, generator=g)
    L.call("umpr_gemm_f32", Ad, A.shape[1], ta, Bd, B.shape[1], tb, C, N, M, N, K, rb.to(dev), 2, 0, 1, 0.5, None, 0, st())
    check(f"gemm-acc {M}x{N}x{K}", C, (0.5 * ref + rb.double()[:, None] + C0.double()).float(), atol=1e-5 * max(1, K) ** 0.5, rtol=1e-5)


# ------------------------------------------------------------------------------------------------ conv / pool
@pytest.mark.parametrize("N,Cin,Cout,HW", [(2, 3, 64, 16), (1, 64, 64, 28), (3, 5, 70, 14), (2, 64, 128, 14),
                                          (1, 130, 40, 7), (2, 8, 8, 36), (1, 16, 200, 9),
                                          # the VGG map widths take the LDS-patch kernel (v2); N>1 makes tiles straddle images
                                          (1, 3, 64, 224), (2, 8, 16, 112), (3, 16, 32, 56), (5, 12, 130, 28),
                                          (7, 6, 64, 14), (3, 130, 70, 14),
                                          # >= 32 reduction channels on 56/28/14 maps: Winograd F(2x2,3x3) path
                                          (2, 64, 96, 56), (3, 40, 200, 28), (5, 256, 256, 14), (1, 512, 512, 14),
                                          # channel counts off the 32 / 128 tile grid, odd batch sizes, one image
                                          (4, 33, 65, 28), (7, 100, 36, 14), (2, 129, 257, 14), (1, 32, 32, 56),
                                          (3, 48, 160, 56),
                                          # the PRODUCTION channel counts of VGG16 on their own map sizes:
                                          # conv3x3_igemm_v2<64,256,224> / <128,128,112> (+ <128,64,112> tail), wgrad_v2<2,16>,
                                          # and the Winograd forward / dgrad / wgrad kernels at 56 / 28
                                          (1, 64, 64, 224), (1, 64, 128, 112), (1, 128, 128, 112), (1, 128, 256, 56),
                                          (2, 256, 256, 56), (1, 256, 512, 28), (2, 512, 512, 28)])
def test_conv3x3(L, dev, N, Cin, Cout, HW):
    g = torch.Generator().manual_seed(N + Cin + Cout + HW)
    x = torch.randn(N, Cin, HW, HW, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g)
    x.requires_grad_(True); w.requires_grad_(True); b.requires_grad_(True)
    y_ref = F.relu(F.conv2d(x, w, b, padding=1))
    gy = torch.randn(y_ref.shape, generator=g)
    gz_ref = gy * (y_ref > 0)
    y_ref.backward(gy)
    xd, wd, bd = x.detach().to(dev), w.detach().to(dev), b.detach().to(dev)
    y = torch.full(y_ref.shape, float("nan"), device=dev)
    wt = torch.empty(L.size("umpr_conv3x3_pack_bytes", N, Cin, Cout, HW, HW) // 4, device=dev)
    wtb = wt.numel() * 4
    L.call("umpr_conv3x3_fwd", xd, wd, bd, y, N, Cin, HW, HW, Cout, 1, wt, wtb, st())
    # Layers on the F(4x4,3x3) Winograd tile (56 / 28 / 14 maps - on 14x14 with tiles that hang over the border -, >= 32 reduction
    # channels; in backward also 128 -> 128 at 112).  With the interpolation points (0, +-3/4, +-3/2, inf) of round 3 the fp32
    # result is 1-2e-6 of max|y| from the float64 convolution (textbook points: 4-8e-6; direct kernels and F(2x2,3x3): 2-4e-7 -
    # tools/conv_error.py on the GPU, tools/wino43_error.py on the CPU).  The stated bound there is 5e-6 of max|y| on top of the
    # 2e-5 absolute every other layer keeps.
    mode = int(os.environ.get("UMPR_WINO_F4", "2"))   # 0: F(2x2,3x3) only, 1: F(4x4,3x3) in backward, 2 (default): forward as well
    f4_fwd = mode >= 2 and ((HW in (56, 28, 14) and Cin >= 32) or (HW == 112 and Cin >= 128 and Cout >= 128))
    f4_bwd = mode >= 1 and ((HW in (56, 28, 14) and Cout >= 32) or (HW == 112 and Cin >= 128 and Cout >= 128))
    check(f"conv fwd {N},{Cin},{Cout},{HW}", y, y_ref, atol=2e-5, rtol=1e-5, rel_to_max=5e-6 if f4_fwd else None)
    gz = gz_ref.to(dev)
    dx = torch.full(x.shape, float("nan"), device=dev)
    L.call("umpr_conv3x3_bwd_data", gz, wd, None, dx, N, Cin, HW, HW, Cout, wt, wtb, st())
    check(f"conv dgrad {N},{Cin},{Cout},{HW}", dx, x.grad, atol=2e-5, rtol=1e-4, rel_to_max=5e-6 if f4_bwd else None)
    # masked dgrad (ReLU of the previous layer fused)
    mask_src = torch.randn(x.shape, generator=g)
    L.call("umpr_conv3x3_bwd_data", gz, wd, mask_src.to(dev), dx, N, Cin, HW, HW, Cout, wt, wtb, st())
    check(f"conv dgrad+mask {N},{Cin},{Cout},{HW}", dx, x.grad * (mask_src > 0), atol=2e-5, rtol=1e-4,
          rel_to_max=5e-6 if f4_bwd else None)
    dw = torch.full(w.shape, float("nan"), device=dev)
    db = torch.full(b.shape, float("nan"), device=dev)
    wsb = L.size("umpr_conv3x3_bwd_weight_ws_bytes", N, Cin, Cout, HW, HW)
    ws = torch.empty(wsb // 4 + 64, device=dev)
    L.call("umpr_conv3x3_bwd_weight", gz, xd, dw, db, N, Cin, HW, HW, Cout, ws, ws.numel() * 4, st())
    # (the weight gradient of the 112 / 56 / 28 layers runs F(3x3,4x4) in modes >= 1; it passes the unchanged bound)
    check(f"conv wgrad {N},{Cin},{Cout},{HW}", dw, w.grad, atol=1e-4, rtol=1e-4, rel_to_max=1e-5)
    check(f"conv bgrad {N},{Cin},{Cout},{HW}", db, b.grad, atol=1e-4, rtol=1e-4, rel_to_max=1e-5)


@pytest.mark.parametrize("N,Cin,Cout,HW", [(1, 64, 128, 112), (2, 128, 128, 112), (2, 256, 256, 56), (3, 512, 512, 14)])
def test_conv3x3_forward_under_inference(L, dev, N, Cin, Cout, HW):
    """Under umpr_set_conv_inference(1) (what VGG16.forward sets under torch.no_grad()) the forward takes the F(4x4,3x3) tile
    wherever the map allows, conv2_1 (64 -> 128 @112) included - a layer the training forward keeps on the direct kernel
    (conv3x3.hip: wino_fwd_c21).  Bound: the 4x4 tile's 5e-6 of max|y| on top of 2e-5 absolute, against torch CPU fp32."""
    g = torch.Generator().manual_seed(N + Cin + Cout + HW + 1)
    x = torch.randn(N, Cin, HW, HW, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g)
    y_ref = F.relu(F.conv2d(x, w, b, padding=1))
    y = torch.full(y_ref.shape, float("nan"), device=dev)
    wt = torch.empty(L.size("umpr_conv3x3_pack_bytes", N, Cin, Cout, HW, HW) // 4, device=dev)
    L.call("umpr_set_conv_inference", 1)
    try:
        L.call("umpr_conv3x3_fwd", x.to(dev), w.to(dev), b.to(dev), y, N, Cin, HW, HW, Cout, 1, wt, wt.numel() * 4, st())
    finally:
        L.call("umpr_set_conv_inference", 0)
    check(f"conv fwd (inference) {N},{Cin},{Cout},{HW}", y, y_ref, atol=2e-5, rtol=1e-5, rel_to_max=5e-6)


@pytest.mark.parametrize("mode", ["0", "1"])
def test_conv3x3_winograd_modes(mode):
    """UMPR_WINO_F4=0 keeps every Winograd layer on F(2x2,3x3) (tight 2e-5 absolute bound everywhere); =1 is the round-2
    arrangement (F(4x4,3x3) in backward only); the default, 2, runs in this process.  The switch is read when the library loads,
    so each setting runs six test_conv3x3 cases in a child test run (tools/run_gpu_children.py, started by conftest before
    this process touched the GPU)."""
    from conftest import child_result
    rc, out = child_result(f"wino_f4_mode{mode}_check")
    log(f"UMPR_WINO_F4={mode} child: " + (out.strip().splitlines() or ["<no output>"])[-1])
    assert rc == 0, out[-3000:]
    assert "6 passed" in out, out[-3000:]


def test_text_path_reads_no_uninitialised_memory():
    """UMPR_POISON_WS=1 (text_path.hip: every fused entry point first fills its workspace - the forward ones also their arena - with
    0xFF bytes = NaN): ten model-level comparisons against the oracle still pass, i.e. nothing the calls read was left over from
    whoever had the memory before.  Runs in a child (the switch is read when the library loads; tools/run_gpu_children.py)."""
    from conftest import child_result
    rc, out = child_result("poison_ws_check")
    log("UMPR_POISON_WS=1 child: " + (out.strip().splitlines() or ["<no output>"])[-1])
    assert rc == 0, out[-3000:]
    assert "10 passed" in out, out[-3000:]


@pytest.mark.parametrize("N,Cin,Cout,HW", [(2, 256, 256, 56), (3, 512, 512, 14), (2, 128, 256, 28)])
def test_winograd_forward_decisions_are_taken_at_direct_accuracy(L, dev, N, Cin, Cout, HW):
    """The decision fix-up of the training forward (winograd.hip: wino4_output_kernel<.., FIX> + wino_fixup_kernel): every output
    whose ReLU decision the 4x4 tile's rounding (1-2e-6 of max|y|) could change is recomputed as a plain dot product.  The bias of
    every output channel is chosen so that one output of that channel sits within ~1e-7 of zero; together with the naturally small
    ones that gives a few hundred outputs inside |y| < 3e-6 max|y|.  Stated bounds: those outputs are within 6e-7 max|y| of the
    float64 convolution (3x the direct kernel's typical error; un-fixed tile outputs are 2-3x further), and NO output whose float64
    value is at least 4e-7 max|y| away from zero has the wrong sign.  Runs in the default mode only (UMPR_WINO_F4=2)."""
    if os.environ.get("UMPR_WINO_F4", "2") != "2" or os.environ.get("UMPR_WINO_FIX_KAPPA", "8") == "0":
        pytest.skip("the training forward is not on the 4x4 tile in this mode")
    g = torch.Generator().manual_seed(N + Cin + HW)
    x = torch.relu(torch.randn(N, Cin, HW, HW, generator=g))
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5
    y0 = F.conv2d(x.double(), w.double(), padding=1)
    pick = torch.randint(0, N * HW * HW, (Cout,), generator=g)
    b = torch.stack([-y0[:, m].reshape(-1)[pick[m]] for m in range(Cout)]).float()      # one output per channel lands at ~0
    y64 = y0 + b.double().view(1, -1, 1, 1)
    scale = float(y64.abs().max())
    y = torch.full(y64.shape, float("nan"), device=dev)
    wt = torch.empty(L.size("umpr_conv3x3_pack_bytes", N, Cin, Cout, HW, HW) // 4, device=dev)
    L.call("umpr_conv3x3_fwd", x.to(dev), w.to(dev), b.to(dev), y, N, Cin, HW, HW, Cout, 0, wt, wt.numel() * 4, st())
    yl = y.cpu().double()
    assert torch.isfinite(yl).all()
    err = (yl - y64).abs()
    near = y64.abs() < 3e-6 * scale
    e_near, e_all = float(err[near].max()) / scale, float(err.max()) / scale
    wrong = ((yl > 0) != (y64 > 0)) & (y64.abs() >= 4e-7 * scale)
    log(f"wino fix-up {N},{Cin},{Cout},{HW}: {int(near.sum())} outputs within 3e-6 max|y| of zero: max err {e_near:.2e} of max|y| "
        f"(all outputs: {e_all:.2e}); wrong signs beyond 4e-7 max|y|: {int(wrong.sum())}")
    assert int(near.sum()) >= Cout
    assert e_near <= 6e-7 and e_all <= 5e-6 and int(wrong.sum()) == 0
    n_fix = L.fn["umpr_debug_wino_fix_count"]()
    log(f"wino fix-up {N},{Cin},{Cout},{HW}: {n_fix} of {y64.numel()} outputs recomputed")
    assert Cout <= n_fix <= 2e-3 * y64.numel()


def test_winograd_pool_decisions_and_constant_images(L, dev):
    """A max-pool follows (umpr_set_conv_pool_follows, as umpr_vgg16_features_fwd sets it around conv3_3 / conv4_3 / conv5_3): the
    fix-up also covers the windows' argmax.  Image 0 is random: no pool window whose float64 leader is at least 4e-7 max|y| ahead
    picks another argmax.  Image 1 is CONSTANT per channel (a flat photo region all the way to the image border), image 2 all
    zeros (a missing photo, src/dataset.py:142-143): their outputs tie exactly in exact arithmetic, which is benign, and must NOT
    be listed for recomputation - stated: fewer than 2e-3 of the batch's outputs are recomputed although two of the three images
    are nothing but ties."""
    if os.environ.get("UMPR_WINO_F4", "2") != "2" or os.environ.get("UMPR_WINO_FIX_KAPPA", "8") == "0":
        pytest.skip("the training forward is not on the 4x4 tile in this mode")
    N, Cin, Cout, HW = 3, 256, 256, 56
    g = torch.Generator().manual_seed(77)
    x = torch.relu(torch.randn(N, Cin, HW, HW, generator=g))
    x[1] = torch.rand(Cin, 1, 1, generator=g).expand(Cin, HW, HW)
    x[2] = 0
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    y64 = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    scale = float(y64[0].abs().max())
    y = torch.full(y64.shape, float("nan"), device=dev)
    wt = torch.empty(L.size("umpr_conv3x3_pack_bytes", N, Cin, Cout, HW, HW) // 4, device=dev)
    L.call("umpr_set_conv_pool_follows", 1)
    try:
        L.call("umpr_conv3x3_fwd", x.to(dev), w.to(dev), b.to(dev), y, N, Cin, HW, HW, Cout, 1, wt, wt.numel() * 4, st())
    finally:
        L.call("umpr_set_conv_pool_follows", 0)
    n_fix = L.fn["umpr_debug_wino_fix_count"]()
    yl = y.cpu()
    check("wino pooled fwd", yl, torch.relu(y64).float(), atol=2e-5, rtol=1e-5, rel_to_max=5e-6)
    r64 = torch.relu(y64[0:1])
    top2 = r64.reshape(1, Cout, HW // 2, 2, HW // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(1, Cout, HW // 2, HW // 2, 4).topk(2, -1).values
    clear = (top2[..., 0] - top2[..., 1]) >= 4e-7 * scale
    i64 = F.max_pool2d(r64, 2, 2, return_indices=True)[1]
    il = F.max_pool2d(yl[0:1].double(), 2, 2, return_indices=True)[1]
    wrong = int(((i64 != il) & clear & (top2[..., 0] > 0)).sum())
    log(f"wino pool rule: {n_fix} of {y64.numel()} outputs recomputed (two of three images constant); random image: "
        f"{int(clear.sum())} clear windows, wrong argmax in {wrong}")
    assert wrong == 0
    assert 0 <= n_fix <= 2e-3 * y64.numel(), n_fix


def test_maxpool(L, dev):
    g = torch.Generator().manual_seed(5)
    x = torch.relu(torch.randn(3, 7, 12, 20, generator=g)).requires_grad_(True)
    y_ref = F.max_pool2d(x, 2, 2)
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    xd = x.detach().to(dev)
    y = torch.empty(y_ref.shape, device=dev)
    L.call("umpr_maxpool2_fwd", xd, y, 21, 12, 20, st())
    check("maxpool fwd", y, y_ref, atol=0)
    gx = torch.empty(x.shape, device=dev)
    L.call("umpr_maxpool2_bwd_relu", xd, gy.to(dev), gx, 21, 12, 20, st())
    # reference: pool backward then ReLU mask of the pooled input (zeros never receive gradient)
    check("maxpool bwd+relu", gx, x.grad * (x.detach() > 0), atol=0)


# ------------------------------------------------------------------------------------------------ GRU
def _gru_case(N, Lmax, E, seed, dev):
    from oracle import umpr_ref as R
    g = torch.Generator().manual_seed(seed)
    vocab = 200
    emb = torch.randn(vocab, E, generator=g) * 0.4
    emb[:3] = 0
    lengths = torch.randint(1, Lmax + 1, (N,), generator=g)
    lengths[0] = Lmax
    ids = torch.randint(3, vocab, (N, Lmax), generator=g)
    for n in range(N):
        ids[n, lengths[n]:] = 0
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]
    P = {}
    for suf in ("", "_reverse"):
        for nm, shape in zip(names, [(192, E), (192, 64), (192,), (192,)]):
            P["g." + nm + suf] = ((torch.rand(shape, generator=g) * 2 - 1) / 8).requires_grad_(True)
    x = F.embedding(ids, emb)
    out_ref = R.improved_rnn(x, lengths, P, "g.", aten=False)
    gout = torch.randn(out_ref.shape, generator=g)
    out_ref.backward(gout)
    return ids, lengths, emb, P, out_ref, gout


@pytest.mark.parametrize("N,Lmax,E,seed", [(40, 20, 50, 1), (130, 20, 50, 2), (7, 5, 50, 3), (64, 12, 300, 4), (65, 20, 52, 5)])
def test_embed_gru(L, dev, N, Lmax, E, seed):
    from umpr_amd.model import UMPR, _EmbedGru
    ids, lengths, emb, P, out_ref, gout = _gru_case(N, Lmax, E, seed, dev)
    lens, order = UMPR._host_perm(lengths, dev)
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]
    w = [P["g." + n + s].detach().to(dev).requires_grad_(True) for s in ("", "_reverse") for n in names]
    out = _EmbedGru.apply(ids.to(dev), lens, order, emb.to(dev), 0, *w)
    check(f"gru fwd N{N} L{Lmax} E{E}", out, out_ref, atol=2e-5)
    out.backward(gout.to(dev))
    for t, (s, n) in zip(w, [(s, n) for s in ("", "_reverse") for n in names]):
        check(f"gru d{n}{s} N{N}", t.grad, P["g." + n + s].grad, atol=1e-5, rel_to_max=1e-3)


def test_embed_gru_golden(L, dev):
    """Fixture generated by the reference's own ImprovedRnn (double un-sort included)."""
    from umpr_amd.model import UMPR
    g = load_golden("improved_rnn_true")
    x = torch.from_numpy(g["x"])  # already embedded inputs: use an identity "embedding" = the rows themselves
    N, Lm, E = x.shape
    emb = x.reshape(N * Lm, E).contiguous()
    ids = torch.arange(N * Lm).reshape(N, Lm)
    lengths = torch.from_numpy(g["lengths"])
    lens, order = UMPR._host_perm(lengths, dev)
    assert np.array_equal(order.cpu().numpy(), g["sorted_indices"].astype(np.int32))
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]
    w = [torch.from_numpy(g["param/module." + n + s]).to(dev) for s in ("", "_reverse") for n in names]
    out = torch.empty(N, Lm, 128, device=dev)
    wsb = L.size("umpr_embed_gru_bidir_ws_bytes", N, Lm, E)
    ws = torch.empty(wsb // 4 + 64, device=dev)
    L.call("umpr_embed_gru_bidir_fwd", ids.to(dev), emb.to(dev), E, *w, lens, order, order, N, Lm, out, None, ws,
           ws.numel() * 4, st())
    check("gru golden fwd", out, g["out"], atol=2e-5)


# ------------------------------------------------------------------------------------------------ fused text regions
def _text_params(seed, V, m_scale):
    from umpr_amd.synthetic import make_param_state
    return make_param_state(seed, 50, 500, V, False, with_vgg=False, m_scale=m_scale)


@pytest.mark.parametrize("B,S,Lm,m_scale", [(3, 20, 20, 1.0), (2, 7, 9, 0.05), (5, 20, 20, 0.05),
                                            # sentences longer than one wave (review_level='review', src/dataset.py:24)
                                            (2, 3, 100, 0.05), (1, 2, 200, 0.3)])
def test_review_head(L, dev, B, S, Lm, m_scale):
    from oracle import umpr_ref as R
    from umpr_amd.model import _ReviewHead
    P = _text_params(11, 1, m_scale)
    g = torch.Generator().manual_seed(B * 100 + S)
    gru_u = (torch.randn(B, S * Lm, 128, generator=g) * 0.5).requires_grad_(True)
    gru_i = (torch.randn(B, S * Lm, 128, generator=g) * 0.5).requires_grad_(True)
    pre = "review_net."
    keys = [pre + "r_net.M", pre + "s_net_u.Ms", pre + "s_net_u.Ws", pre + "s_net_i.Ms", pre + "s_net_i.Ws",
            pre + "linear_u.weight", pre + "linear_i.weight"]
    for k in keys:
        P[k].requires_grad_(True)
    # oracle pieces (src/model.py:50-55,71-81,166-168)
    A = torch.tanh(gru_i @ P[keys[0]] @ gru_u.transpose(-1, -2))
    soft_u = torch.softmax(A.max(dim=-2).values, -1)
    soft_i = torch.softmax(A.max(dim=-1).values, -1)
    atte_u = (gru_u.transpose(-1, -2) @ soft_u.unsqueeze(-1)).squeeze(-1)
    atte_i = (gru_i.transpose(-1, -2) @ soft_i.unsqueeze(-1)).squeeze(-1)
    _, su = R.s_net(gru_u, soft_u, Lm, P, pre + "s_net_u.")
    _, si = R.s_net(gru_i, soft_i, Lm, P, pre + "s_net_i.")
    ref = torch.tanh(F.linear(torch.cat([atte_u, su], -1), P[keys[5]]) + F.linear(torch.cat([atte_i, si], -1), P[keys[6]]))
    gout = torch.randn(ref.shape, generator=g)
    ref.backward(gout)
    du = gru_u.detach().to(dev).requires_grad_(True)
    di = gru_i.detach().to(dev).requires_grad_(True)
    wd = [P[k].detach().to(dev).requires_grad_(True) for k in keys]
    out = _ReviewHead.apply(du, di, S, Lm, *wd)
    check(f"review_head fwd B{B} S{S} m{m_scale}", out, ref, atol=2e-5)
    out.backward(gout.to(dev))
    check(f"review_head dGu m{m_scale}", du.grad, gru_u.grad, atol=1e-6, rel_to_max=1e-3)
    check(f"review_head dGi m{m_scale}", di.grad, gru_i.grad, atol=1e-6, rel_to_max=1e-3)
    for k, t in zip(keys, wd):
        check(f"review_head d{k} m{m_scale}", t.grad, P[k].grad, atol=1e-6, rel_to_max=1e-3)


@pytest.mark.parametrize("B", [1, 5, 32, 33, 64, 130, 256, 300])
def test_review_merge(L, dev, B):
    """tanh(linear_u(repr_u) + linear_i(repr_i)) (src/model.py:150-158) through umpr_review_merge_fwd / _bwd against torch fp32 on
    the CPU: the dedicated batch-sized kernels up to B = 256 (row / workgroup boundaries at 4, 32, 128), the GEMM route above."""
    g = torch.Generator().manual_seed(B)
    ru = torch.randn(B, 256, generator=g).requires_grad_(True)
    ri = torch.randn(B, 256, generator=g).requires_grad_(True)
    Wu = (torch.randn(128, 256, generator=g) / 16).requires_grad_(True)
    Wi = (torch.randn(128, 256, generator=g) / 16).requires_grad_(True)
    ref = torch.tanh(F.linear(ru, Wu) + F.linear(ri, Wi))
    gout = torch.randn(ref.shape, generator=g)
    ref.backward(gout)
    d = [t.detach().to(dev) for t in (ru, ri, Wu, Wi)]
    out = torch.full((B, 128), float("nan"), device=dev)
    L.call("umpr_review_merge_fwd", *d, B, out, st())
    check(f"merge fwd B{B}", out, ref, atol=2e-6)
    wsb = L.size("umpr_review_merge_bwd_ws_bytes", B)
    ws = torch.empty(wsb // 4 + 64, device=dev)
    grads = [torch.full(t.shape, float("nan"), device=dev) for t in d]
    L.call("umpr_review_merge_bwd", *d, out, gout.to(dev), B, *grads, ws, ws.numel() * 4, st())
    for name, got, want in zip(("d_repr_u", "d_repr_i", "dW_u", "dW_i"), grads, (ru.grad, ri.grad, Wu.grad, Wi.grad)):
        check(f"merge {name} B{B}", got, want, atol=1e-6, rel_to_max=2e-6)


@pytest.mark.parametrize("B,S_ui,L_ui,S,Lm,V,KS", [(3, 5, 20, 20, 20, 1, 3), (2, 3, 11, 6, 9, 4, 3), (4, 1, 6, 5, 20, 4, 3),
                                                   # config.py:37's comment lists kernel sizes 1, 2, 3: an even width yields
                                                   # L - 1 conv positions (src/model.py:93); plus sentences of 70 tokens
                                                   (2, 3, 11, 6, 9, 2, 2), (2, 2, 7, 4, 12, 1, 1), (2, 2, 9, 3, 10, 3, 4),
                                                   (2, 2, 70, 3, 66, 2, 3)])
def test_control(L, dev, B, S_ui, L_ui, S, Lm, V, KS):
    from oracle import umpr_ref as R
    from umpr_amd.model import _Control
    from umpr_amd.synthetic import make_param_state
    P = make_param_state(13, 50, 500, V, False, with_vgg=False, m_scale=0.3, kernel_size=KS)
    g = torch.Generator().manual_seed(B * 10 + V)
    gs = [(torch.randn(B, s * l, 128, generator=g) * 0.7).requires_grad_(True) for s, l in ((S_ui, L_ui), (S, Lm), (S, Lm))]
    pre = "control_net."
    keys = [pre + "c_net.cnn.0.weight", pre + "c_net.cnn.0.bias", pre + "c_net.linear.0.weight", pre + "c_net.linear.0.bias",
            pre + "s_net.Ms", pre + "s_net.Ws", pre + "ss_net.linear.0.weight", pre + "ss_net.linear.0.bias"]
    for k in keys:
        P[k].requires_grad_(True)

    def head(x, s, l):
        cnn_in = x.reshape(B * s, l, -1).transpose(-1, -2)
        y = F.relu(F.conv1d(cnn_in, P[keys[0]], P[keys[1]], padding=(KS - 1) // 2)).max(dim=-1)[0].view(B, s, -1)
        vp = torch.sigmoid(F.linear(y, P[keys[2]], P[keys[3]]))
        vp = torch.where(vp < 0.35, torch.zeros_like(vp), vp)
        return vp, (vp ** 2).sum(-2)
    vp, c_out = head(gs[0], S_ui, L_ui)
    _, c_u = head(gs[1], S, Lm)
    _, c_i = head(gs[2], S, Lm)
    s_, _ = R.s_net(gs[0], vp, L_ui, P, pre + "s_net.")
    senti = torch.sigmoid(F.linear(s_, P[keys[6]], P[keys[7]])).expand(-1, -1, V)
    vs = (senti * vp ** 2).sum(-2) / ((vp ** 2).sum(-2) + 1e-4)
    q_p = (vs > 0.5).float()
    q_pos = torch.where(vs < 0.5, torch.zeros_like(vs), 4 * (vs - 0.5) ** 2)
    q_neg = torch.where(vs > 0.5, torch.zeros_like(vs), 4 * (0.5 - vs) ** 2)
    refs = [c_u, c_i, c_out * q_p * q_pos, c_out * (1 - q_p) * q_neg]
    gouts = [torch.randn(r.shape, generator=g) for r in refs]
    torch.autograd.backward(refs, gouts)
    near = int(((vs - 0.5).abs() < 1e-5).sum())
    log(f"control: view_score within 1e-5 of the 0.5 gate: {near}")
    gd = [t.detach().to(dev).requires_grad_(True) for t in gs]
    wd = [P[k].detach().to(dev).requires_grad_(True) for k in keys]
    outs = _Control.apply(gd[0], gd[1], gd[2], (B, S_ui, L_ui, S, Lm), 0.35, *wd)
    for nm, o, r in zip(("c_u", "c_i", "prefer_pos", "prefer_neg"), outs, refs):
        check(f"control fwd {nm} V{V} KS{KS}", o, r, atol=2e-5, rtol=1e-5)
    torch.autograd.backward(outs, [t.to(dev) for t in gouts])
    for i in range(3):
        check(f"control dG{i} V{V} KS{KS}", gd[i].grad, gs[i].grad, atol=1e-6, rel_to_max=1e-3)
    for k, t in zip(keys, wd):
        check(f"control d{k} V{V} KS{KS}", t.grad, P[k].grad, atol=1e-6, rel_to_max=1e-3)


@pytest.mark.parametrize("B,V,Pc", [(5, 1, 1), (3, 4, 2), (6, 0, 0)])
def test_head(L, dev, B, V, Pc):
    from umpr_amd.model import _Head
    g = torch.Generator().manual_seed(B + V)
    rn = lambda *s: torch.randn(*s, generator=g)
    rr = rn(B, 128).requires_grad_(True)
    fw = (rn(1, 128 + 2 * V) * 0.1).requires_grad_(True)
    fb = torch.tensor([0.7]).requires_grad_(True)
    labels = torch.randint(1, 6, (B,), generator=g).float()
    if V:
        c_u, c_i, pp, pn = [(torch.rand(B, V, generator=g)).requires_grad_(True) for _ in range(4)]
        vgg = rn(B * V * Pc, 1000).requires_grad_(True)
        pos_v, neg_v = rn(V, 1000).requires_grad_(True), rn(V, 1000).requires_grad_(True)
        lw, lb = (rn(1, 1000) * 0.03).requires_grad_(True), torch.tensor([0.1]).requires_grad_(True)
        img = vgg.view(B, V, Pc, -1).mean(-2)
        ie = F.linear(img, lw, lb).squeeze(-1)
        pm = torch.tanh((F.linear(pos_v, lw, lb).squeeze(-1) - ie).abs())
        nm = torch.tanh((F.linear(neg_v, lw, lb).squeeze(-1) - ie).abs())
        feat = torch.cat([rr, c_u * c_i * (1 - pm), c_u * c_i * (1 - nm)], -1)
        pred = F.relu(F.linear(feat, fw, fb)).squeeze(-1)
        loss = F.mse_loss(pred, labels) + 0.1 * torch.mean(pp.transpose(-1, -2) @ pm + pn.transpose(-1, -2) @ nm)
        ins = [rr, c_u, c_i, pp, pn, vgg, pos_v, neg_v, lw, lb, fw, fb]
    else:
        pred = F.relu(F.linear(rr, fw, fb)).squeeze(-1)
        loss = F.mse_loss(pred, labels)
        ins = [rr, None, None, None, None, None, None, None, None, None, fw, fb]
    loss.backward()
    ind = [t.detach().to(dev).requires_grad_(True) if t is not None else None for t in ins]
    p2, l2, _ = _Head.apply(*ind, labels.to(dev), 0.1, V, Pc)
    check(f"head pred V{V}", p2, pred, atol=2e-5)
    check(f"head loss V{V}", l2, loss, atol=2e-5)
    l2.backward()
    for i, (a, b) in enumerate(zip(ind, ins)):
        if a is not None:
            check(f"head grad[{i}] V{V}", a.grad, b.grad, atol=1e-6, rel_to_max=1e-3)


# ------------------------------------------------------------------------------------------------ VGG16
def test_vgg16_small(L, dev):
    from oracle import umpr_ref as R
    from umpr_amd.model import VGG16
    from umpr_amd.synthetic import make_param_state
    P = make_param_state(71, 50, 10, 1, False)
    pre = "visual_net.vgg16.0."
    vp = {k: v.requires_grad_(True) for k, v in P.items() if k.startswith(pre)}
    g = torch.Generator().manual_seed(3)
    n = 2
    x = torch.rand(n, 3, 224, 224, generator=g)
    masks = [(torch.rand(n, 4096, generator=g) < 0.5).float() for _ in range(2)]
    ref = R.vgg16_forward(x, vp, pre, dropout_masks=masks)
    gout = torch.randn(ref.shape, generator=g)
    ref.backward(gout)
    m = VGG16()
    m.load_state_dict({k[len(pre):]: v.detach() for k, v in vp.items()})
    m = m.to(dev)
    m.dropout_masks = torch.stack(masks).to(torch.uint8).to(dev)
    out = m(x.to(dev))
    check("vgg16 fwd", out, ref, atol=1e-4, rtol=1e-4)
    out.backward(gout.to(dev))
    # fp64 run of the same network: the yardstick for summation-order noise.  The first conv's weight gradient is a
    # sum over n*224*224 heavily cancelling terms, where two fp32 summation orders (oneDNN's and ours) differ by
    # ~0.5% of the tensor max; the HIP result must be as close to the fp64 truth as the fp32 CPU path is.
    vp64 = {k: v.detach().double().requires_grad_(True) for k, v in vp.items()}
    ref64 = R.vgg16_forward(x.double(), vp64, pre, dropout_masks=[mk.double() for mk in masks])
    ref64.backward(gout.double())
    worst = []
    for k, p in m.named_parameters():
        g64 = vp64[pre + k].grad
        # L2 norms: a single ReLU / max-pool decision that flips between two fp32 summation orders moves isolated
        # elements by O(1e-2) of the max in either path, so the max-norm is not a usable yardstick here
        e_gpu = float((p.grad.detach().cpu().double() - g64).norm())
        e_cpu = float((vp[pre + k].grad.double() - g64).norm())
        scale = float(g64.norm())
        log(f"vgg16 d{k}: L2 |hip-f64|={e_gpu:.3e} |cpu32-f64|={e_cpu:.3e} |g|={scale:.3e} ratio={e_gpu / max(e_cpu, 1e-30):.2f}")
        worst.append((e_gpu / max(e_cpu, 1e-30), e_gpu / scale, k))
    log("vgg16 yardstick worst: " + str(sorted(worst)[-3:]))
    # Measured (profiles/README.md, "yardstick"): every fp32 path jumps to 1-2e-3 relative L2 the first time a ReLU /
    # max-pool decision lands on the other side of the fp64 one, and stays there for all earlier layers.  Where the
    # first flip happens is chance (oneDNN: features.12; direct HIP conv: features.2; Winograd HIP: pool4), so the
    # bound is "as close as the CPU fp32 path, or within the flip plateau"; kernel accuracy proper is pinned by
    # test_conv3x3 / test_gemm against tight absolute tolerances.
    for ratio, rel, k in worst:
        assert ratio <= 3.0 or rel <= 4e-3, (k, ratio, rel)
    for k, p in m.named_parameters():
        check(f"vgg16 d{k}", p.grad, vp[pre + k].grad, atol=1e-7, rel_to_max=5e-3, max_bad_frac=1e-3, max_bad=2,
              rel_l2=1e-2)


@pytest.mark.parametrize("n", [33, 64, 100])
def test_vgg16_classifier_at_batch_sized_rows(dev, n):
    """The classifier's three products at 33 / 64 / 100 images (fc_small.hip: TM = 2 and 4 row tiles, the 32-deep forward kernel
    fc_fwd_k32, dx and dW) against torch on the same pool5 features: forward 1e-4 of the output scale, gradients 1e-3 of theirs.
    (The golden fixtures run 2-8 images = one row tile.)"""
    from umpr_amd.model import VGG16, _VGGClassifier, _VGGFeatures
    torch.manual_seed(5)
    m = VGG16().to(dev).eval()                        # eval: no dropout
    with torch.no_grad():
        for mod in m.classifier:
            if isinstance(mod, torch.nn.Linear):
                mod.bias.uniform_(-0.05, 0.05)
    images = torch.rand(n, 3, 224, 224, device=dev)
    ps = m.param_list()
    pool5, acts = _VGGFeatures.apply(images, *ps[:26])
    p5 = pool5.detach().clone().requires_grad_(True)
    out = _VGGClassifier.apply(p5, acts, False, None, 0, m, False, *ps[26:])
    gout = torch.randn(out.shape, device=dev)
    out.backward(gout)
    # reference: float64 matmuls of the same features
    x = pool5.detach().double().requires_grad_(True)
    W = [p.detach().double().requires_grad_(True) for p in ps[26:]]
    h = torch.relu(x @ W[0].t() + W[1])
    h = torch.relu(h @ W[2].t() + W[3])
    ref = h @ W[4].t() + W[5]
    ref.backward(gout.double())
    check(f"classifier fwd n={n}", out, ref.float(), atol=1e-6, rel_to_max=1e-4)
    check(f"classifier d_pool5 n={n}", p5.grad, x.grad.float(), atol=1e-9, rel_to_max=1e-3)
    for k, (p, w) in enumerate(zip(ps[26:], W)):
        check(f"classifier grad {k} n={n}", p.grad, w.grad.float(), atol=1e-9, rel_to_max=1e-3)


# ------------------------------------------------------------------------------------------------ end to end vs golden
def _build(gname, dev):
    from umpr_amd.config import Config
    from umpr_amd.model import UMPR
    from umpr_amd.synthetic import make_batch, make_param_state
    g = load_golden(gname)
    B, V, ronly, pseed, bseed, full_pad, vocab = [int(v) for v in g["meta"]]
    P = make_param_state(pseed, 50, vocab, V, bool(ronly), m_scale=float(g["m_scale"]))
    batch = make_batch(bseed, B, vocab, V, int(g["photo_count"]) if "photo_count" in g else 1,
                       review_net_only=bool(ronly), full_pad=bool(full_pad))
    cfg = Config(argv=[])
    cfg.review_net_only = bool(ronly)
    cfg.views = ["v%d" % i for i in range(V)]
    model = UMPR(cfg, P["embedding.weight"].numpy())
    model.load_state_dict(P)
    model = model.to(dev)
    return g, model, batch


def _fp64_yardstick(gname, g):
    """Gradients of the same fixture from an fp64 run of the oracle on the CPU (5 s): the truth both fp32 paths - the
    reference's CPU run stored in the fixture and the HIP run - are measured against."""
    from oracle import umpr_ref as R
    from umpr_amd.synthetic import make_batch, make_param_state
    B, V, ronly, pseed, bseed, full_pad, vocab = [int(v) for v in g["meta"]]
    P = make_param_state(pseed, 50, vocab, V, bool(ronly), m_scale=float(g["m_scale"]))
    batch = make_batch(bseed, B, vocab, V, int(g["photo_count"]) if "photo_count" in g else 1,
                       review_net_only=bool(ronly), full_pad=bool(full_pad))
    P64 = {k: (v.double() if v.is_floating_point() else v) for k, v in P.items()}
    for k, p in P64.items():
        if k != "embedding.weight":
            p.requires_grad_(True)
    b64 = tuple(t.double() if t.is_floating_point() else t for t in batch)
    masks = [torch.from_numpy(g["drop_mask0"]).double(), torch.from_numpy(g["drop_mask1"]).double()] if "drop_mask0" in g else None
    _, loss = R.umpr_forward(P64, b64, review_net_only=bool(ronly), aten=True, train=masks is not None, dropout_masks=masks)
    loss.backward()
    return {k: p.grad for k, p in P64.items() if k != "embedding.weight" and p.grad is not None}


def _compare_golden(g, model, pred, loss, gname=None):
    check("pred", pred, g["prediction"], atol=1e-4)
    check("loss", loss, g["loss"], atol=1e-4)
    has_vgg = any("vgg16" in k for k, _ in model.named_parameters())
    g64 = _fp64_yardstick(gname, g) if (has_vgg and gname) else None
    outside = []
    for k, p in model.named_parameters():
        vggp = "vgg16" in k
        if "grad/" + k in g:
            ref, got, stride = torch.from_numpy(g["grad/" + k]), p.grad.detach().cpu(), 1
        elif "gradstat/" + k in g:
            stride = int(g["gradstat/" + k][3])
            ref, got = torch.from_numpy(g["gradsample/" + k]), p.grad.detach().cpu().reshape(-1)[::stride]
            l2 = float(p.grad.double().pow(2).sum().sqrt())
            log(f"gradnorm {k}: got {l2:.6e} ref {g['gradstat/' + k][2]:.6e}")
            assert abs(l2 - g["gradstat/" + k][2]) <= 2e-3 * g["gradstat/" + k][2] + 1e-12, k
        else:
            continue
        if not vggp:
            check("grad " + k, got.reshape(ref.shape), ref, atol=1e-6, rel_to_max=2e-3)
            continue
        # VGG gradients: a 13-layer ReLU / max-pool network is not smooth - the first decision that lands on the other
        # side between two fp32 summation orders moves every earlier layer's gradient by 1-3e-3 relative L2
        # (tools/relu_flip_experiment.py; 1e-7 relative noise moves them by exactly 0).  So the yardstick is the fp64 run:
        # the HIP gradient must be as close to it as the reference's own fp32 CPU gradient is (factor 3), or - when the
        # two fp32 paths flipped in different layers - within the flip plateau (4e-3 relative L2).  Kernel accuracy
        # proper is pinned per layer at the production shapes by test_conv3x3 (2e-5 / 1e-4 absolute).
        t64 = g64[k].reshape(-1)[::stride] if stride > 1 else g64[k].reshape(-1)
        e_gpu = float((got.reshape(-1).double() - t64).norm())
        e_ref = float((ref.reshape(-1).double() - t64).norm())
        scale = float(t64.norm()) + 1e-300
        log(f"grad {k}: L2 |hip-f64|={e_gpu:.3e} |ref32-f64|={e_ref:.3e} |g|={scale:.3e} ratio={e_gpu / max(e_ref, 1e-300):.2f} rel={e_gpu / scale:.2e}")
        assert not torch.isnan(got).any(), k
        if not (e_gpu <= 3.0 * e_ref or e_gpu / scale <= 4e-3):      # every parameter is logged before the test fails
            outside.append((k, e_gpu, e_ref, e_gpu / scale))
        check("grad " + k, got.reshape(ref.shape), ref, atol=1e-6, rel_to_max=5e-3, max_bad_frac=1e-3, max_bad=2, rel_l2=6e-3)
    assert not outside, outside


@pytest.mark.parametrize("name", ["umpr_full_V1_B2", "umpr_full_V4_B2", "umpr_full_V2_P2_B2"])
def test_umpr_full_golden_inference(dev, name):
    """The same fixtures under torch.no_grad() (evaluate.py:8-13): VGG16.forward then lets the library use the F(4x4,3x3)
    tile in the forward pass as well (umpr_set_conv_inference).  Predictions and loss stay within north_star's 1e-4 of the
    REFERENCE's outputs; the distance to the training-mode forward (2x2 tile) is logged - it is the larger tile's rounding,
    a few 1e-6."""
    g, model, batch = _build(name, dev)
    model.eval()
    pred_t, loss_t = model(*batch)
    with torch.no_grad():
        pred_i, loss_i = model(*batch)
    torch.cuda.synchronize()
    log(f"== {name} inference: |pred(no_grad) - pred(grad)| = {float((pred_i - pred_t).abs().max()):.3e}")
    check("pred (inference)", pred_i, g["prediction"], atol=1e-4)
    check("loss (inference)", loss_i, g["loss"], atol=1e-4)
    assert float((pred_i - pred_t).abs().max()) < 5e-5


@pytest.mark.parametrize("name", ["umpr_r_B4", "umpr_r_B4_soft", "umpr_r_B3_fullpad"])
def test_umpr_r_golden(dev, name):
    g, model, batch = _build(name, dev)
    model.eval()
    log(f"== {name}")
    pred, loss = model(*batch)
    loss.backward()
    torch.cuda.synchronize()
    _compare_golden(g, model, pred, loss, name)


@pytest.mark.parametrize("name", ["umpr_full_V1_B2", "umpr_full_V1_B2_randnM", "umpr_full_V4_B2", "umpr_full_V1_B2_drop",
                                  "umpr_full_V2_P2_B2"])
def test_umpr_full_golden(dev, name):
    g, model, batch = _build(name, dev)
    log(f"== {name}")
    if "drop_mask0" in g:
        model.train()
        model.visual_net.vgg16[0].dropout_masks = torch.from_numpy(np.stack([g["drop_mask0"], g["drop_mask1"]])).to(dev)
    else:
        model.eval()
    pred, loss = model(*batch)
    loss.backward()
    torch.cuda.synchronize()
    _compare_golden(g, model, pred, loss, name)


def test_adam_kernel(L, dev):
    from oracle.umpr_ref import adam_step_numpy
    g = torch.Generator().manual_seed(9)
    n = 100003
    p, gr = torch.randn(n, generator=g), torch.randn(n, generator=g) * 0.1
    m, v = torch.zeros(n), torch.zeros(n)
    pd, gd, md, vd = p.to(dev), gr.to(dev), m.to(dev), v.to(dev)
    pr, mr, vr = p.double(), m.double(), v.double()
    for step in (1, 2, 3):
        L.call("umpr_adam_step", pd, gd, md, vd, n, 1e-3, 0.9, 0.999, 1e-8, 1e-3, step, 1.0, st())
        pr, mr, vr = adam_step_numpy(pr, gr.double(), mr, vr, step, 1e-3, 1e-3)
    check("adam p", pd, pr.float(), atol=1e-6)
    check("adam m", md, mr.float(), atol=1e-7)
    check("adam v", vd, vr.float(), atol=1e-8)


def test_train_trajectory_golden(dev):
    """Three optimiser steps driven like main.py:22-37 (Adam, two groups, coupled L2, ExponentialLR after the first
    epoch) against the trajectory the reference produced (fixture adam_umpr_r): loss per step and final parameters."""
    from umpr_amd.config import Config
    from umpr_amd.model import UMPR
    from umpr_amd.optim import FusedAdam
    from umpr_amd.synthetic import make_batch, make_param_state
    from umpr_amd.train import train_step
    g = load_golden("adam_umpr_r")
    P = make_param_state(31, 50, 1000, 1, True, m_scale=0.05)
    cfg = Config(argv=[])
    cfg.review_net_only = True
    model = UMPR(cfg, P["embedding.weight"].numpy())
    model.load_state_dict(P)
    model = model.to(dev)
    opt = FusedAdam(model, float(g["lr"]), float(g["l2"]), lr_decay=0.99)
    for step in range(3):
        batch = make_batch(500 + step, 4, 1000, review_net_only=True)
        _, loss = train_step(model, opt, batch)
        log(f"train step {step}: loss {loss.item():.6f} ref {g['losses'][step]:.6f}")
        assert abs(loss.item() - g["losses"][step]) < 1e-4
        if step == 0:
            opt.epoch_end()
    for k, p in model.named_parameters():
        if "param/" + k in g:
            check("trained " + k, p, g["param/" + k], atol=2e-5, rtol=1e-4)


@pytest.mark.parametrize("B,seed", [(1, 3), (3, 4)])
def test_umpr_r_small_batches_vs_oracle(dev, B, seed):
    """Edge cases the fixtures do not hold: a single sample (fewer sequences than one 64-row GRU tile), ragged lengths,
    and a batch where every sentence but one is an empty (length-1, id 0) pad."""
    from oracle import umpr_ref as R
    from umpr_amd.config import Config
    from umpr_amd.model import UMPR
    from umpr_amd.synthetic import make_batch, make_param_state
    P = make_param_state(61, 50, 800, 1, True, m_scale=0.05)
    batch = list(make_batch(seed, B, 800, review_net_only=True))
    if B == 3:  # blank out sample 1: all sentences empty except the first
        for ids, lens in ((batch[0], batch[3]), (batch[1], batch[4])):
            ids[1, 1:] = 0
            lens[1, 1:] = 1
    cfg = Config(argv=[])
    cfg.review_net_only = True
    model = UMPR(cfg, P["embedding.weight"].numpy())
    model.load_state_dict(P)
    model = model.to(dev).eval()
    pred, loss = model(*batch)
    loss.backward()
    for k, p in P.items():
        if k != "embedding.weight":
            p.requires_grad_(True)
    rp, rl = R.umpr_forward(P, tuple(batch), review_net_only=True, aten=True)
    rl.backward()
    check(f"small-batch pred B{B}", pred, rp, atol=1e-4)
    check(f"small-batch loss B{B}", loss, rl, atol=1e-4)
    for k, p in model.named_parameters():
        if p.requires_grad:
            check(f"small-batch grad {k} B{B}", p.grad, P[k].grad, atol=1e-6, rel_to_max=2e-3)


@pytest.mark.parametrize("KS,max_len", [(2, 20), (3, 90), (4, 70)])
def test_kernel_size_and_sentence_length_surface_vs_oracle(dev, KS, max_len):
    """config.py:29,37: `--kernel_size` 1 / 2 / 3 / 4 and a `--max_sent_length` beyond one wave, through the whole model (fused
    text path, VGG16, head) on one sample pair against the oracle: predictions and loss 1e-4, text-path gradients 2e-3 of the
    tensor maximum."""
    from oracle import umpr_ref as R
    from umpr_amd.config import Config
    from umpr_amd.model import UMPR
    from umpr_amd.synthetic import make_batch, make_param_state
    P = make_param_state(67, 50, 700, 1, False, m_scale=0.05, kernel_size=KS)
    batch = make_batch(68, 2, 700, 1, max_sent_count=4, min_sent_count=2, max_ui_sent_count=2, max_sent_length=max_len)
    cfg = Config(argv=[])
    cfg.views = ["unknown"]
    cfg.kernel_size = KS
    cfg.max_sent_length = max_len
    model = UMPR(cfg, P["embedding.weight"].numpy())
    model.load_state_dict(P)
    model = model.to(dev).eval()
    pred, loss = model(*batch)
    loss.backward()
    for k, p in P.items():
        if k != "embedding.weight":
            p.requires_grad_(True)
    rp, rl = R.umpr_forward(P, batch, review_net_only=False, aten=True)
    rl.backward()
    check(f"KS{KS} L{max_len} pred", pred, rp, atol=1e-4)
    check(f"KS{KS} L{max_len} loss", loss, rl, atol=1e-4)
    for k, p in model.named_parameters():
        if p.requires_grad and "vgg16" not in k:
            check(f"KS{KS} L{max_len} grad {k}", p.grad, P[k].grad, atol=1e-6, rel_to_max=2e-3)


@pytest.mark.parametrize("h,at,ronly", [(32, 64, True), (64, 32, True), (24, 40, False), (32, 32, False)])
def test_hidden_sizes_below_the_kernel_width_vs_oracle(dev, h, at, ronly):
    """config.py:34-35: `--gru_size` / `--self_atte_size` below 64 run EXACTLY embedded in the 64-wide kernels (units with
    all-zero parameters stay at zero and add zero to every sum - umpr_amd/model.py::_ZeroPad); the state_dict keeps the
    reference's shapes.  Predictions / loss 1e-4 and every text-path gradient 2e-3 of its maximum against the oracle run natively
    at the configured sizes; one training step moves the parameters like the oracle's Adam step."""
    from oracle import umpr_ref as R
    from umpr_amd.config import Config
    from umpr_amd.model import UMPR
    from umpr_amd.optim import FusedAdam
    from umpr_amd.synthetic import make_batch, make_param_state
    from umpr_amd.train import train_step
    P = make_param_state(71, 50, 700, 1, ronly, gru_size=h, atte_size=at, m_scale=0.05)
    batch = make_batch(72, 3, 700, 1, review_net_only=ronly)
    cfg = Config(argv=[])
    cfg.review_net_only = ronly
    cfg.views = ["unknown"]
    cfg.gru_size, cfg.self_atte_size = h, at
    model = UMPR(cfg, P["embedding.weight"].numpy())
    assert {k: tuple(v.shape) for k, v in model.state_dict().items()} == {k: tuple(v.shape) for k, v in P.items()}
    model.load_state_dict(P)
    model = model.to(dev).eval()
    pred, loss = model(*batch)
    loss.backward()
    Pr = {k: v.clone() for k, v in P.items()}
    for k, p in Pr.items():
        if k != "embedding.weight":
            p.requires_grad_(True)
    rp, rl = R.umpr_forward(Pr, batch, review_net_only=ronly, aten=True)
    rl.backward()
    check(f"h{h} at{at} pred", pred, rp, atol=1e-4)
    check(f"h{h} at{at} loss", loss, rl, atol=1e-4)
    for k, p in model.named_parameters():
        if p.requires_grad and "vgg16" not in k:
            check(f"h{h} at{at} grad {k}", p.grad, Pr[k].grad, atol=1e-6, rel_to_max=2e-3)
    if ronly:      # one optimiser step through the flat arenas: the scatter / gather sits between them and the kernels
        opt = FusedAdam(model, 1e-3, 1e-3)
        ropt = R.adam_reference(Pr, 1e-3, 1e-3)
        train_step(model, opt, batch)
        _, rl2 = R.umpr_forward(Pr, batch, review_net_only=True, aten=True, train=True)
        ropt.zero_grad()
        rl2.backward()
        ropt.step()
        for k, p in model.named_parameters():
            if p.requires_grad:
                # Adam's first step is lr * sign(g): elements whose gradient is rounding noise around zero move by +-lr on
                # either side at random - compare where the oracle's gradient is above that level
                gr = Pr[k].grad
                big = gr.abs() > 1e-4 * gr.abs().max()
                check(f"h{h} at{at} stepped {k}", p.detach().cpu()[big], Pr[k].detach()[big], atol=2e-5, rtol=1e-4)


@pytest.mark.parametrize("E", [300, 7])
def test_embedding_widths_vs_oracle(dev, E):
    """GloVe-300d (BASELINE.json configs[4] uses it) and an odd width that is not a multiple of the GEMM k-step: the
    embedding gather feeds the GRU input projection for any E; control net included (review_net_only=False needs VGG, so
    the text half is checked through UMPR-R plus the C-Net GRU inside test_control)."""
    from oracle import umpr_ref as R
    from umpr_amd.config import Config
    from umpr_amd.model import UMPR
    from umpr_amd.synthetic import make_batch, make_param_state
    P = make_param_state(65, E, 700, 1, True, m_scale=0.05)
    batch = make_batch(66, 3, 700, review_net_only=True)
    cfg = Config(argv=[])
    cfg.review_net_only = True
    model = UMPR(cfg, P["embedding.weight"].numpy())
    model.load_state_dict(P)
    model = model.to(dev).eval()
    pred, loss = model(*batch)
    loss.backward()
    for k, p in P.items():
        if k != "embedding.weight":
            p.requires_grad_(True)
    rp, rl = R.umpr_forward(P, batch, review_net_only=True, aten=True)
    rl.backward()
    check(f"E{E} pred", pred, rp, atol=1e-4)
    check(f"E{E} loss", loss, rl, atol=1e-4)
    for k, p in model.named_parameters():
        if p.requires_grad:
            check(f"E{E} grad {k}", p.grad, P[k].grad, atol=1e-6, rel_to_max=2e-3)


def test_full_size_properties(L, dev):
    """BASELINE.json sizes (64 images, 1280 sequences): properties that need no CPU reference.
    (1) the VGG16 output of an image does not depend on the batch it is in - bit for bit (tiles straddle images, the
        accumulation order per output pixel is fixed); (2) two runs are bitwise identical (no atomics on the path);
    (3) a GRU sequence's output does not depend on which other sequences share its 64-row tile."""
    from umpr_amd.model import UMPR, VGG16, _EmbedGru
    g = torch.Generator().manual_seed(5)
    vgg = VGG16().to(dev).eval()
    x = torch.rand(64, 3, 224, 224, generator=g).to(dev)
    with torch.no_grad():
        full = vgg(x)
        again = vgg(x)
        part = vgg(x[5:7].contiguous())
    assert torch.isfinite(full).all()
    assert torch.equal(full, again), "VGG16 forward is not run-to-run deterministic"
    assert torch.equal(full[5:7], part), "VGG16 output of an image depends on its batch"
    N, Lm, E = 1280, 20, 50
    emb = (torch.randn(5000, E, generator=g) * 0.4).to(dev)
    ids = torch.randint(3, 5000, (N, Lm), generator=g)
    lengths = torch.randint(1, Lm + 1, (N,), generator=g)
    w = [((torch.rand(s, generator=g) * 2 - 1) / 8).to(dev) for _ in range(2) for s in ((192, E), (192, 64), (192,), (192,))]
    lens, order = UMPR._host_perm(lengths, dev)
    ident = torch.arange(N, dtype=torch.int32, device=dev)
    with torch.no_grad():
        out_sorted = _EmbedGru.apply(ids.to(dev), lens, order, emb, 0, *w)      # tiles grouped by length, permuted rows
    out_ident = torch.empty(N, Lm, 128, device=dev)
    wsb = L.size("umpr_embed_gru_bidir_ws_bytes", N, Lm, E)
    ws = torch.empty(wsb // 4 + 64, device=dev)
    L.call("umpr_embed_gru_bidir_fwd", ids.to(dev), emb, E, *w, lens, ident, ident, N, Lm, out_ident, None, ws,
           ws.numel() * 4, st())                                             # natural grouping, no permutation
    # out_sorted[order[n]] is sequence n (dst_row = sorted_indices); compare with the unpermuted run bit for bit
    assert torch.equal(out_sorted[order.long()], out_ident), "GRU output depends on tile grouping"


def test_full_size_backward_properties(dev):
    """64 images through the VGG16 backward (direct + Winograd dgrad / wgrad, split-K, side streams), no CPU reference:
    (1) linearity - doubling the upstream gradient doubles every parameter gradient bit for bit (every kernel is linear
        in it and scaling by 2 is exact in fp32); (2) additivity over the batch - the gradients of the two half batches
        add up to the full batch's (forward values are batch-independent bit for bit, so only the summation order
        differs); (3) dropout masks are a pure function of (seed, call count)."""
    from umpr_amd.model import VGG16
    g = torch.Generator().manual_seed(9)
    vgg = VGG16().to(dev).eval()
    x = torch.rand(64, 3, 224, 224, generator=g).to(dev)
    gout = torch.randn(64, 1000, generator=g).to(dev)

    def grads(xx, gg):
        vgg.zero_grad(set_to_none=True)
        vgg(xx).backward(gg)
        return {k: p.grad.clone() for k, p in vgg.named_parameters()}

    g1 = grads(x, gout)
    g2 = grads(x, 2 * gout)
    for k in g1:
        assert torch.isfinite(g1[k]).all(), k
        assert torch.equal(g2[k], 2 * g1[k]), f"gradient of {k} is not linear in the upstream gradient"
    ga = grads(x[:32].contiguous(), gout[:32].contiguous())
    gb = grads(x[32:].contiguous(), gout[32:].contiguous())
    for k in g1:
        err = float((ga[k] + gb[k] - g1[k]).double().norm() / (g1[k].double().norm() + 1e-30))
        log(f"full-size additivity {k}: rel L2 {err:.2e}")
        assert err < 2e-5, (k, err)


# ------------------------------------------------------------------------------------------------ R-Net pre-training
class _W2V:
    def __init__(self, P):
        self.embedding = P["embedding.weight"].numpy()
        self.word_dim = self.embedding.shape[1]


def test_bce_head_kernel(L, dev):
    g = torch.Generator().manual_seed(17)
    B, K = 37, 256
    att = torch.randn(B, K, generator=g).requires_grad_(True)
    w = (torch.randn(1, K, generator=g) * 0.2).requires_grad_(True)
    b = torch.randn(1, generator=g).requires_grad_(True)
    att.data[0] *= 40  # saturated rows: p -> 0 / 1, exercises the -100 log clamp and the 1e-12 denominator
    att.data[1] *= -40
    tgt = torch.randint(0, 2, (B,), generator=g).float()
    res_ref = torch.sigmoid(att @ w.t() + b).squeeze(-1)
    loss_ref = F.binary_cross_entropy(res_ref, tgt)
    gl = torch.tensor(0.7)
    loss_ref.backward(gl)
    ad, wd, bd, td = att.detach().to(dev), w.detach().to(dev), b.detach().to(dev), tgt.to(dev)
    res, loss = torch.empty(B, device=dev), torch.empty((), device=dev)
    ws = torch.empty(B, device=dev)
    L.call("umpr_bce_head_fwd", ad, K, wd, bd, td, B, K, res, loss, ws, B * 4, st())
    check("bce result", res, res_ref, atol=1e-6)
    check("bce loss", loss, loss_ref, atol=1e-5, rtol=1e-6)
    d_att, dw, db = torch.empty(B, K, device=dev), torch.empty(1, K, device=dev), torch.empty(1, device=dev)
    L.call("umpr_bce_head_bwd", ad, K, wd, res, td, None, gl.to(dev), B, K, d_att, K, dw, db, ws, B * 4, st())
    check("bce d_att", d_att, att.grad, atol=1e-7, rtol=1e-5)
    check("bce dw", dw, w.grad, atol=1e-5, rtol=1e-5)
    check("bce db", db, b.grad, atol=1e-6, rtol=1e-5)


@pytest.mark.parametrize("tag", ["ragged", "full"])
def test_pretrain_rnet_golden(dev, tag):
    """PretrainRNet (pretrain/pretrain_rnet.py:144-169) through the C ABI against the reference's outputs, gradients
    and 3-step Adam trajectory (pretrain_rnet.py:177-198)."""
    from umpr_amd.optim import FusedAdam
    from umpr_amd.pretrain import PretrainRNet
    from umpr_amd.synthetic import make_pretrain_batch, make_pretrain_state
    g = load_golden("pretrain_rnet_" + tag)
    B, Lmax, ragged, pseed, bseed = [int(v) for v in g["meta"]]
    P = make_pretrain_state(pseed, 50, 300)
    batch = make_pretrain_batch(bseed, B, Lmax, 300, bool(ragged))
    m = PretrainRNet(_W2V(P), 64)
    m.load_state_dict(P)
    m = m.to(dev)
    result, loss = m(*batch)
    check("pretrain result", result, g["result"], atol=1e-5)
    check("pretrain loss", loss, g["loss"], atol=1e-5)
    loss.backward()
    for k, p in m.named_parameters():
        if "grad/" + k in g:
            check("pretrain grad " + k, p.grad, g["grad/" + k], atol=1e-6, rel_to_max=1e-3)
    m.load_state_dict(P)
    opt = FusedAdam(m, 0.01, 1e-3)
    for step in range(3):
        _, l = m(*batch)
        opt.zero_grad()
        l.backward()
        opt.step()
        log(f"pretrain step {step}: loss {l.item():.6f} ref {g['traj_loss'][step]:.6f}")
        assert abs(l.item() - g["traj_loss"][step]) < 1e-4
    for k, p in m.named_parameters():
        if "traj_param/" + k in g and p.requires_grad:
            check("pretrained " + k, p, g["traj_param/" + k], atol=2e-4, rtol=1e-3)


def test_pretrain_rnet_loop_saves_rnet(dev, tmp_path):
    """pretrain_r_net (pretrain_rnet.py:172-205): the loss falls on a learnable toy task and save_r_net writes a
    state_dict that loads into an RNet, i.e. into UMPR's review_net.r_net (the reference only saves the module,
    pretrain_rnet.py:170,205; its main.py has no loader)."""
    from umpr_amd.pretrain import pretrain_r_net
    from umpr_amd.synthetic import make_pretrain_batch, make_pretrain_state
    P = make_pretrain_state(45, 50, 300)
    batches = [make_pretrain_batch(46 + k, 64, 12, 300, True) for k in range(2)]
    lines = []
    path = str(tmp_path / "rnet.pt")
    model = pretrain_r_net(_W2V(P), batches, path, learning_rate=0.01, train_epochs=4, log=lines.append)
    first, last = float(lines[0].split()[-1]), float(lines[-1].split()[-1])
    log(f"pretrain loop loss {first:.4f} -> {last:.4f}")
    assert last < first
    sd = torch.load(path, weights_only=True)
    assert set(sd) == set(model.r_net.state_dict())
    from umpr_amd.model import RNet
    RNet(50, 64).load_state_dict(sd)
