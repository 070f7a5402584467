"""Error behaviour of the boundary (INTEGRATION.md "Error behaviour"): the product path fails loudly - it never falls
back to the CPU or to the oracle."""
import os

import pytest
import torch

from conftest import ROOT


def _cfg(**kw):
    from umpr_amd.config import Config
    cfg = Config(argv=[])
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


def test_model_on_cpu_raises():
    from umpr_amd.model import UMPR
    from umpr_amd.synthetic import make_batch, make_param_state
    P = make_param_state(1, 50, 100, 1, True)
    model = UMPR(_cfg(review_net_only=True), P["embedding.weight"].numpy())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model(*make_batch(2, 2, 100, review_net_only=True))


def test_conv_width_and_sentence_length_limits():
    """Even C-Net kernel sizes and sentences up to 256 tokens construct (round 3); beyond the kernels' range the constructor
    refuses instead of diverging silently."""
    from umpr_amd.model import UMPR
    from umpr_amd.synthetic import make_param_state
    P = make_param_state(1, 50, 100, 1, True)
    UMPR(_cfg(review_net_only=False, kernel_size=2, max_sent_length=200, views=["a"]), P["embedding.weight"].numpy())
    with pytest.raises(AssertionError, match="max_sent_length"):
        UMPR(_cfg(review_net_only=True, max_sent_length=300), P["embedding.weight"].numpy())
    with pytest.raises(AssertionError, match="kernel_size"):
        UMPR(_cfg(review_net_only=False, kernel_size=9, views=["a"]), P["embedding.weight"].numpy())


def test_unsupported_hidden_size_raises():
    from umpr_amd.model import UMPR
    from umpr_amd.synthetic import make_param_state
    P = make_param_state(1, 50, 100, 1, True)
    with pytest.raises(AssertionError, match="gru_size"):
        UMPR(_cfg(review_net_only=True, gru_size=96), P["embedding.weight"].numpy())      # above the kernels' width
    with pytest.raises(AssertionError, match="self_atte_size"):
        UMPR(_cfg(review_net_only=True, self_atte_size=128), P["embedding.weight"].numpy())
    UMPR(_cfg(review_net_only=True, gru_size=32, self_atte_size=48), P["embedding.weight"].numpy())   # below: embedded


def test_missing_library_raises(monkeypatch, tmp_path):
    from umpr_amd import _lib
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libumpr_hip.so"))
    with pytest.raises(_lib.UmprHipError, match="no CPU fallback"):
        _lib._Lib()


def test_optimizer_step_on_cpu_raises():
    from umpr_amd.optim import FusedAdam
    opt = FusedAdam(torch.nn.Linear(3, 2), 1e-3, 1e-3)
    with pytest.raises(RuntimeError, match="cuda"):
        opt.step()


def test_product_package_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    import re
    pat = re.compile(r"^\s*(import\s+oracle|from\s+oracle)\b", re.M)
    files = [os.path.join(ROOT, "main.py")]
    for dirpath, _, names in os.walk(os.path.join(ROOT, "umpr_amd")):
        files += [os.path.join(dirpath, f) for f in names if f.endswith(".py")]
    assert len(files) > 8
    for f in files:
        assert not pat.search(open(f).read()), f
    bench = open(os.path.join(ROOT, "bench.py")).read()
    assert pat.search(bench) is None or "def cpu_baseline" in bench   # bench.py imports it inside cpu_baseline only
    body = bench[bench.index("def cpu_baseline"):]
    body = body[:body.index("\ndef ", 1)]
    assert "oracle" in body and not pat.search(bench.replace(body, ""))


@pytest.mark.gpu
def test_c_abi_reports_errors():
    """A too-small scratch buffer or an unsupported shape comes back as a non-zero return code with a message, surfaced
    as UmprHipError - never as a silent fallback."""
    from umpr_amd._lib import UmprHipError, lib
    L = lib()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    x = torch.zeros(1, 64, 28, 28, device=dev)
    w = torch.zeros(64, 64, 3, 3, device=dev)
    b = torch.zeros(64, device=dev)
    y = torch.empty(1, 64, 28, 28, device=dev)
    dw, db = torch.empty_like(w), torch.empty_like(b)
    tiny = torch.empty(16, device=dev)
    with pytest.raises(UmprHipError, match="workspace"):
        L.call("umpr_conv3x3_bwd_weight", y, x, dw, db, 1, 64, 28, 28, 64, tiny, 64, st)
    with pytest.raises(UmprHipError):
        L.call("umpr_maxpool2_fwd", x, y, 64, 27, 27, st)   # odd extent
    assert "maxpool2" in L.last_error()


@pytest.mark.gpu
def test_second_backward_through_the_gru_raises_instead_of_faulting():
    """the GRU node releases its output buffer after the first backward; a second backward (retain_graph=True) must
    be a Python error, and the C entry point itself refuses NULL tensors instead of handing them to a kernel."""
    from umpr_amd._lib import UmprHipError, lib
    from umpr_amd.model import UMPR, _EmbedGru
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    N, L, E = 5, 6, 50
    emb = torch.randn(30, E, generator=g).to(dev)
    ids = torch.randint(3, 30, (N, L), generator=g).to(dev)
    lens, order = UMPR._host_perm(torch.full((N,), L), dev)
    w = [((torch.rand(s, generator=g) * 2 - 1) / 8).to(dev).requires_grad_(True)
         for _ in range(2) for s in ((192, E), (192, 64), (192,), (192,))]
    out = _EmbedGru.apply(ids, lens, order, emb, 0, *w)
    out.sum().backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="already run"):
        out.sum().backward()
    Lb = lib()
    wsb = Lb.size("umpr_embed_gru_bidir_ws_bytes", N, L, E)
    ws = torch.empty(wsb // 4 + 64, device=dev)
    gr = [torch.empty_like(t) for t in w]
    with pytest.raises(UmprHipError, match="NULL"):
        Lb.call("umpr_embed_gru_bidir_bwd_acc", ids, emb, E, w[1], w[5], lens, order, order, N, L, out.detach(), None, None,
                *gr, 0, ws, ws.numel() * 4, torch.cuda.current_stream().cuda_stream)
