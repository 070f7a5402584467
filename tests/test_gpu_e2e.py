"""End-to-end GPU tests of the host drivers around the kernels: evaluate_mse ("test MSE", src/evaluate.py:6-14), the
CSV / GloVe / photos.json pipeline feeding the model (main.py:64-99), exact checkpoint resume, and the main.py CLI.
The checker is the oracle (oracle/umpr_ref.py) on the same collated batches."""
import os
import shutil
import sys

import pytest
import torch

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu

CORPUS = os.path.join(GOLDEN, "tiny_corpus")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _cfg(**kw):
    from umpr_amd.config import Config
    cfg = Config(argv=[])
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


def test_evaluate_mse_vs_oracle(dev):
    """Test MSE of the full model over two batches of different sizes = the oracle's, to 1e-4 (north_star)."""
    from oracle import umpr_ref as R
    from umpr_amd.model import UMPR
    from umpr_amd.synthetic import make_batch, make_param_state
    from umpr_amd.train import evaluate_mse
    P = make_param_state(81, 50, 600, 1, False, m_scale=0.05)
    batches = [make_batch(82, 2, 600, 1), make_batch(83, 1, 600, 1)]
    model = UMPR(_cfg(views=["unknown"]), P["embedding.weight"].numpy())
    model.load_state_dict(P)
    model = model.to(dev)
    got = evaluate_mse(model, batches)
    se, n = 0.0, 0
    with torch.no_grad():
        for b in batches:
            pred, _ = R.umpr_forward(P, b, review_net_only=False, aten=True)
            se += float(((pred - b[-1]) ** 2).sum())
            n += len(pred)
    assert abs(got - se / n) < 1e-4, (got, se / n)
    assert not model.training  # evaluate_mse leaves the model in eval mode, like the reference


@pytest.mark.parametrize("review_net_only", [True, False])
def test_csv_pipeline_feeds_model(dev, review_net_only):
    """train.csv + glove.txt + photos.json -> Dataset -> DataLoader(batch_loader) -> UMPR on the GPU.  Eval-mode
    (prediction, loss) of every batch equal the oracle's on the same collated tensors; photo files that cannot be read
    become zero images (src/dataset.py:142-143) on both sides."""
    from torch.utils.data import DataLoader
    from oracle import umpr_ref as R
    from umpr_amd.data import Dataset, Word2vec, batch_loader
    from umpr_amd.model import UMPR
    from umpr_amd.synthetic import make_param_state
    cfg = _cfg(max_sent_count=6, min_sent_count=2, max_ui_sent_count=3, max_sent_length=10, photo_count=1,
               views=["food"], review_net_only=review_net_only, batch_size=3)
    w2v = Word2vec(os.path.join(CORPUS, "glove.txt"))
    ds = Dataset(os.path.join(CORPUS, "train.csv"), os.path.join(CORPUS, "photos.json"), os.path.join(CORPUS, "photos"),
                 w2v, cfg)
    assert len(ds) >= 4
    loader = DataLoader(ds, batch_size=cfg.batch_size, collate_fn=lambda x: batch_loader(x, review_net_only))
    E = w2v.embedding.shape[1]
    P = make_param_state(91, E, len(w2v), 1, review_net_only, m_scale=0.05)
    P["embedding.weight"] = torch.tensor(w2v.embedding, dtype=torch.float32)
    model = UMPR(cfg, w2v.embedding)
    model.load_state_dict(P)
    model = model.to(dev).eval()
    n = 0
    for batch in loader:
        if n >= 2:
            break
        with torch.no_grad():
            pred, loss = model(*batch)
            rp, rl = R.umpr_forward(P, batch, review_net_only=review_net_only, aten=True)
        assert torch.allclose(pred.cpu(), rp, atol=1e-4), (pred.cpu(), rp)
        assert abs(float(loss) - float(rl)) < 1e-4
        if not review_net_only:
            assert batch[6].shape[1:] == (1, 1, 3, 224, 224) and float(batch[6].abs().max()) == 0.0
        n += 1
    assert n == 2


def test_checkpoint_resume_is_exact(dev, tmp_path):
    """4 training steps == 2 steps + save + load into a fresh model/optimiser + 2 steps, bit for bit (parameters and
    Adam moments travel; the kernels are deterministic)."""
    from umpr_amd.checkpoint import load_checkpoint, save_checkpoint
    from umpr_amd.model import UMPR
    from umpr_amd.optim import FusedAdam
    from umpr_amd.synthetic import make_batch, make_param_state
    from umpr_amd.train import train_step
    P = make_param_state(101, 50, 500, 1, True, m_scale=0.05)
    cfg = _cfg(review_net_only=True)
    batches = [make_batch(110 + i, 4, 500, review_net_only=True) for i in range(4)]

    def fresh():
        m = UMPR(cfg, P["embedding.weight"].numpy())
        m.load_state_dict(P)
        m = m.to(dev)
        return m, FusedAdam(m, 1e-3, 1e-3, lr_decay=0.99)

    m1, o1 = fresh()
    for b in batches:
        train_step(m1, o1, b)
    m2, o2 = fresh()
    for b in batches[:2]:
        train_step(m2, o2, b)
    path = str(tmp_path / "ck.pt")
    save_checkpoint(path, m2, o2, epoch=0, batch_counter=2)
    m3, o3 = fresh()
    meta = load_checkpoint(path, m3, o3, map_location=dev)
    assert meta["batch_counter"] == 2
    for b in batches[2:]:
        train_step(m3, o3, b)
    for (k, a), (_, c) in zip(m1.state_dict().items(), m3.state_dict().items()):
        assert torch.equal(a, c), k


def test_main_cli_on_csv_corpus(tmp_path, capsys, monkeypatch):
    """main.py --data_dir <reference-layout dir> trains and reports a test MSE (readme.md:70-92).  Run in-process (argv
    patched): a GPU-initialised test process must not fork+exec another GPU program on this pool."""
    import importlib
    d = tmp_path / "music"
    d.mkdir()
    for split in ("train", "valid", "test"):
        shutil.copy(os.path.join(CORPUS, "train.csv"), d / f"{split}.csv")
    shutil.copy(os.path.join(CORPUS, "photos.json"), d / "photos.json")
    (d / "photos").mkdir()
    argv = ["main.py", "--data_dir", str(d), "--word2vec_file", os.path.join(CORPUS, "glove.txt"),
            "--review_net_only", "True", "--train_epochs", "2", "--batch_size", "4", "--max_sent_count", "6",
            "--min_sent_count", "2", "--max_ui_sent_count", "3", "--max_sent_length", "10", "--views", "food",
            "--learning_rate", "1e-3", "--model_path", str(tmp_path / "m.pt")]
    monkeypatch.setattr(sys, "argv", argv)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.chdir(tmp_path)
    main = importlib.import_module("main")
    main.main()
    out = capsys.readouterr().out
    assert "Initial validation mse is" in out and "Test end, test mse is" in out, out[-2000:]
    mse = float(out.strip().split("test mse is")[-1])
    assert mse == mse and mse < 100.0
    # --test_only (main.py:87-99 of the reference): load the saved weights, no training, report the test MSE
    from umpr_amd.checkpoint import save_checkpoint
    from umpr_amd.data import Word2vec
    from umpr_amd.model import UMPR
    from umpr_amd.config import Config
    cfg = Config(argv=argv[1:])
    w2v = Word2vec(os.path.join(CORPUS, "glove.txt"))
    torch.manual_seed(3)
    m = UMPR(cfg, w2v.embedding)
    save_checkpoint(str(tmp_path / "m.pt"), m)
    monkeypatch.setattr(sys, "argv", argv + ["--test_only", "True"])
    main.main()
    out2 = capsys.readouterr().out
    assert "Initial validation mse" not in out2 and "Test end, test mse is" in out2, out2[-2000:]
    mse2 = float(out2.strip().split("test mse is")[-1])
    assert mse2 == mse2 and mse2 != mse      # the freshly initialised weights of the checkpoint, not the trained ones


@pytest.mark.parametrize("review_net_only", [True, False])
def test_training_is_bitwise_reproducible(dev, review_net_only):
    """Two fresh model/optimiser instances driven through the same steps end with identical bits in every parameter
    (no float atomics, fixed-order reductions, dropout mask a pure function of (seed, step, index))."""
    from umpr_amd.model import UMPR
    from umpr_amd.optim import FusedAdam
    from umpr_amd.synthetic import make_batch, make_param_state
    from umpr_amd.train import train_step
    P = make_param_state(121, 50, 500, 1, review_net_only, m_scale=0.05)
    cfg = _cfg(review_net_only=review_net_only, views=["unknown"])
    batches = [make_batch(130 + i, 3, 500, 1, review_net_only=review_net_only) for i in range(3)]
    runs = []
    for _ in range(2):
        torch.manual_seed(5)
        m = UMPR(cfg, P["embedding.weight"].numpy())
        m.load_state_dict(P)
        m = m.to(dev)
        opt = FusedAdam(m, 1e-3, 1e-3)
        losses = [train_step(m, opt, b)[1].item() for b in batches]
        runs.append((losses, {k: v.detach().clone() for k, v in m.state_dict().items()}))
    assert runs[0][0] == runs[1][0]
    for k in runs[0][1]:
        assert torch.equal(runs[0][1][k], runs[1][1][k]), k


def test_in_place_gradients_keep_accumulation_semantics(dev):
    """VGG gradients are written straight into the optimiser's arena on the first backward after zero_grad(); a second
    backward without zero_grad must ADD (torch semantics), and a parameter no backward touched must read as zero."""
    from umpr_amd.model import UMPR
    from umpr_amd.optim import FusedAdam
    from umpr_amd.synthetic import make_batch, make_param_state
    P = make_param_state(141, 50, 400, 1, False, m_scale=0.05)
    model = UMPR(_cfg(views=["unknown"]), P["embedding.weight"].numpy())
    model.load_state_dict(P)
    model = model.to(dev).eval()          # eval: no dropout, two passes are identical
    opt = FusedAdam(model, 1e-3, 1e-3)
    batch = make_batch(142, 2, 400, 1)
    names = ["visual_net.vgg16.0.features.0.weight", "visual_net.vgg16.0.features.28.bias",
             "visual_net.vgg16.0.classifier.0.weight", "visual_net.vgg16.0.classifier.6.bias", "review_net.r_net.M"]
    params = dict(model.named_parameters())
    opt.zero_grad()
    assert params[names[0]]._umpr_fresh
    model(*batch)[1].backward()
    assert not params[names[0]]._umpr_fresh
    once = {k: params[k].grad.clone() for k in names}
    model(*batch)[1].backward()           # no zero_grad in between
    for k in names:
        assert torch.allclose(params[k].grad, 2 * once[k], rtol=1e-5, atol=1e-7 * float(once[k].abs().max())), k
    opt.zero_grad()
    for g in opt.groups:                  # stale contents in the in-place slices, nothing writes them before step()
        for p in g.direct:
            p.grad.fill_(3.0)
    before = params[names[2]].detach().clone()
    opt.step()
    assert float(params[names[2]].grad.abs().max()) == 0.0
    # zero gradient: the first Adam step moves a weight by at most lr * (wd * |w|-driven update) <= lr
    assert float((params[names[2]].detach() - before).abs().max()) <= 1.001e-3


def test_training_driver_validates_and_checkpoints_every_500_batches(dev, tmp_path):
    """umpr_amd.train.training mirrors main.py:16-61: validation before training, again every 500 batches, a checkpoint
    whenever the validation MSE improves on the best so far (which starts at 100), ExponentialLR per epoch."""
    from umpr_amd.checkpoint import load_checkpoint
    from umpr_amd.model import UMPR
    from umpr_amd.synthetic import make_batch, make_param_state
    from umpr_amd.train import evaluate_mse, training
    P = make_param_state(151, 50, 300, 1, True, m_scale=0.05)
    cfg = _cfg(review_net_only=True, train_epochs=2, learning_rate=1e-3, lr_decay=0.5)
    model = UMPR(cfg, P["embedding.weight"].numpy())
    model.load_state_dict(P)
    model = model.to(dev)
    train = [make_batch(160 + i % 7, 2, 300, review_net_only=True, max_sent_count=6, min_sent_count=5) for i in range(260)]
    valid = [make_batch(170, 4, 300, review_net_only=True, max_sent_count=6, min_sent_count=5)]
    path = str(tmp_path / "best.pt")
    lines = []
    opt, saved = training(train, valid, model, cfg, path, logger=type("L", (), {"info": staticmethod(lines.append)}))
    assert saved
    assert lines[0].startswith("Initial validation mse is")
    assert sum("batch   500" in ln for ln in lines) == 1                 # 2 epochs x 260 batches: one 500-batch mark
    assert sum(ln.startswith("Epoch") and "done" in ln for ln in lines) == 2
    assert abs(opt.lr - 1e-3 * 0.25) < 1e-12                             # two ExponentialLR steps
    assert os.path.exists(path)
    meta = load_checkpoint(path, model, opt, map_location=dev)            # the state at batch 500
    assert meta["batch_counter"] == 500 and meta["best_loss"] < 100
    assert meta["epoch"] == 1 and meta["batch_in_epoch"] == 240          # 500 = 260 + 240
    assert abs(evaluate_mse(model, valid) - meta["best_loss"]) < 1e-6


def test_driver_level_resume_is_exact(dev, tmp_path):
    """`--resume` continues a run exactly: a shuffled DataLoader, a checkpoint taken in the middle of epoch
    1, then a fresh process-equivalent (new model, new loader, new generator) resumed from it ends with the same bits as
    the uninterrupted run - shuffle order, position in the epoch, Adam state and learning-rate schedule all restored."""
    from torch.utils.data import DataLoader
    from umpr_amd.model import UMPR
    from umpr_amd.synthetic import make_batch, make_param_state
    from umpr_amd.train import training
    P = make_param_state(181, 50, 300, 1, True, m_scale=0.05)
    samples = [make_batch(190 + i, 2, 300, review_net_only=True, max_sent_count=6, min_sent_count=6, full_pad=True)
               for i in range(10)]

    def run(resume, path, epochs):
        cfg = _cfg(review_net_only=True, train_epochs=epochs, learning_rate=1e-3, lr_decay=0.5)
        Config_extra = dict(valid_every=7, resume=resume)
        for k, v in Config_extra.items():
            setattr(cfg, k, v)
        torch.manual_seed(11)
        m = UMPR(cfg, P["embedding.weight"].numpy())
        m.load_state_dict(P)
        m = m.to(dev)
        g = torch.Generator().manual_seed(0)
        train = DataLoader(samples, batch_size=None, shuffle=True, generator=g)   # each "sample" is a collated batch
        lines = []
        _, saved = training(train, [samples[0]], m, cfg, path, logger=type("L", (), {"info": staticmethod(lines.append)}))
        return m, saved, lines

    full, _, _ = run("", str(tmp_path / "full.pt"), 3)
    # interrupted twin: stop after the checkpoint of batch 14 (epoch 1, 4 batches in) by training 2 epochs only ...
    part, saved, lines = run("", str(tmp_path / "part.pt"), 2)
    assert saved
    from umpr_amd.checkpoint import load_checkpoint
    meta = load_checkpoint(str(tmp_path / "part.pt"), part, map_location=dev)
    assert meta["epoch"] >= 0 and "epoch_rng" in meta and "batch_in_epoch" in meta
    # ... and resume from that checkpoint for the remaining epochs
    res, _, rlines = run(str(tmp_path / "part.pt"), str(tmp_path / "res.pt"), 3)
    assert any(ln.startswith("Resumed from") for ln in rlines)
    for (k, a), (_, c) in zip(full.state_dict().items(), res.state_dict().items()):
        assert torch.equal(a, c), k


def test_two_rank_data_parallel_on_real_kernels():
    """tools/check_dp_gpu.py (started by conftest before this process touched the GPU): two ranks share the box's one GPU
    (gloo transport; RCCL refuses two ranks on one device), drive the full model through train_step's machinery with the
    overlapped GradReducer and in-place VGG gradients; after two optimiser steps the data-parallel parameters equal a
    single-process replay of both shards, and the two-rank evaluate_mse equals the shard-by-shard evaluation."""
    from conftest import DP_CHECK
    p = DP_CHECK["proc"]
    assert p is not None, "the two-rank job was not started (no /dev/kfd, or GPU tests deselected)"
    rc = p.wait(timeout=900)
    out = open(DP_CHECK["log"]).read()
    assert rc == 0, out[-3000:]
    assert "data-parallel parameters equal the sequential replay" in out, out[-3000:]
    assert "two-rank evaluate_mse" in out, out[-3000:]
    assert "short last batch" in out, out[-3000:]


def test_gradient_exchange_on_rccl_at_world_one():
    """tools/check_exchange_world1.py (a job of tools/run_gpu_children.py): one rank on the RCCL backend runs the early slice, the block
    buckets, the remainder and the early Adam step with real NCCL calls - the identity at world size 1 - in the async and
    the in-stream form, fp32 and bf16; parameters and Adam moments after three steps equal the run without exchange
    bit for bit."""
    from conftest import child_result
    rc, out = child_result("exchange_world1_check")
    assert rc == 0, out[-3000:]
    assert "world-1 RCCL exchange check passed" in out, out[-3000:]
    assert out.count("bit-identical") == 4, out[-3000:]


def test_umpr_r_full_size_vs_oracle(dev):
    """BASELINE.json configs[0] at its real size: UMPR-R, batch 32, S = L = 20 fully padded (640 sequences per GRU call),
    GloVe-50d - the oracle finishes a step in well under a second, so this is direct parity, not a property."""
    from oracle import umpr_ref as R
    from umpr_amd.model import UMPR
    from umpr_amd.synthetic import make_batch, make_param_state
    P = make_param_state(211, 50, 5000, 1, True, m_scale=0.05)
    batch = make_batch(212, 32, 5000, review_net_only=True, full_pad=True)
    assert tuple(batch[0].shape) == (32, 20, 20)
    model = UMPR(_cfg(review_net_only=True), P["embedding.weight"].numpy())
    model.load_state_dict(P)
    model = model.to(dev).eval()
    pred, loss = model(*batch)
    loss.backward()
    for k, p in P.items():
        if k != "embedding.weight":
            p.requires_grad_(True)
    rp, rl = R.umpr_forward(P, batch, review_net_only=True, aten=True)
    rl.backward()
    assert float((pred.detach().cpu() - rp.detach()).abs().max()) < 1e-4
    assert abs(float(loss) - float(rl)) < 1e-4
    for k, p in model.named_parameters():
        if p.requires_grad:
            ref = P[k].grad
            assert float((p.grad.cpu() - ref).abs().max()) <= 2e-3 * float(ref.abs().max()) + 1e-6, k


def test_cfg4_full_size_forward_vs_oracle_and_backward_properties(dev):
    """BASELINE.json configs[3] per GPU: 4 views x 32 samples = 128 images through UMPR.forward.  Forward: predictions and
    loss against the oracle at full size (eval mode; ~10 s of CPU).  Backward (no CPU reference at this size): two runs
    are bitwise identical, and scaling the loss by 2 scales every gradient by exactly 2 (every kernel is linear in the
    upstream gradient and x2 is exact in fp32)."""
    from oracle import umpr_ref as R
    from umpr_amd.model import UMPR
    from umpr_amd.synthetic import make_batch, make_param_state
    V = 4
    P = make_param_state(221, 50, 3000, V, False, m_scale=0.05)
    batch = make_batch(222, 32, 3000, V, full_pad=True)
    assert tuple(batch[6].shape) == (32, V, 1, 3, 224, 224)
    model = UMPR(_cfg(views=["food", "inside", "outside", "drink"]), P["embedding.weight"].numpy())
    model.load_state_dict(P)
    model = model.to(dev).eval()

    def grads(scale):
        model.zero_grad(set_to_none=True)
        pred, loss = model(*batch)
        (loss * scale).backward()
        return pred.detach().clone(), loss.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}

    p1, l1, g1 = grads(1.0)
    p2, l2, g2 = grads(1.0)
    p3, l3, g3 = grads(2.0)
    assert torch.equal(p1, p2) and torch.equal(l1, l2)
    for k in g1:
        assert torch.isfinite(g1[k]).all(), k
        assert torch.equal(g1[k], g2[k]), f"{k}: two runs differ"
        assert torch.equal(g3[k], 2 * g1[k]), f"{k}: gradient is not linear in the upstream gradient"
    with torch.no_grad():
        rp, rl = R.umpr_forward(P, batch, review_net_only=False, aten=True)
    assert float((p1.cpu() - rp).abs().max()) < 1e-4, float((p1.cpu() - rp).abs().max())
    assert abs(float(l1) - float(rl)) < 1e-4


def test_graphed_umpr_r_step_equals_eager(dev):
    """umpr_amd/graphs.py: the UMPR-R training step captured once as a hipGraph and replayed on four different batches of one
    geometry leaves parameters and Adam moments BIT-identical to four eager train_step calls from the same start (the Adam
    kernel's per-step scalars come from device memory in both runs' arithmetic), and returns the same losses."""
    from umpr_amd.graphs import GraphedTrainStep
    from umpr_amd.model import UMPR
    from umpr_amd.optim import FusedAdam
    from umpr_amd.synthetic import make_batch, make_param_state
    from umpr_amd.train import train_step
    P = make_param_state(21, 50, 900, 1, True, m_scale=0.05)
    batches = [make_batch(30 + k, 6, 900, review_net_only=True, full_pad=True) for k in range(4)]
    for b in batches[1:]:       # ragged lengths inside a fixed padded geometry: what the graph's index buffer carries
        b[3][:, -2:] = torch.randint(3, 20, b[3][:, -2:].shape)
    res = {}
    def on_dev(b):       # ids / photos / labels on the device, lengths on the host (src/model.py:18)
        return (b[0].to(dev), b[1].to(dev), b[2].to(dev), b[3], b[4], b[5], b[6].to(dev), b[7].to(dev))
    for mode in ("eager", "graph", "graph_dev"):
        model = UMPR(_cfg(review_net_only=True), P["embedding.weight"].numpy())
        model.load_state_dict(P)
        model = model.to(dev)
        opt = FusedAdam(model, 1e-3, 1e-3)
        losses = []
        if mode == "graph":              # host batches: one packed upload per step
            g = GraphedTrainStep(model, opt, batches[0])
            for b in batches:
                losses.append(float(g(b)[1]))
        elif mode == "graph_dev":        # device batches: copies into the step buffer; the last one lives in it (resident())
            g = GraphedTrainStep(model, opt, on_dev(batches[0]))
            for b in batches[:3]:
                losses.append(float(g(on_dev(b))[1]))
            losses.append(float(g(g.resident(on_dev(batches[3])))[1]))
        else:
            for b in batches:
                losses.append(float(train_step(model, opt, b)[1]))
        assert opt.step_count == 4
        res[mode] = (losses, {k: v.detach().clone() for k, v in model.state_dict().items()}, [x.m.clone() for x in opt.groups],
                     [x.v.clone() for x in opt.groups])
    for mode in ("graph", "graph_dev"):
        assert res["eager"][0] == res[mode][0], (mode, res["eager"][0], res[mode][0])
        for k in res["eager"][1]:
            assert torch.equal(res["eager"][1][k], res[mode][1][k]), (mode, k)
        for a, b in zip(res["eager"][2] + res["eager"][3], res[mode][2] + res[mode][3]):
            assert torch.equal(a, b), mode


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_forward_backward_is_reproducible_bit_for_bit(dev, dtype):
    """tools/check_reproducible.py: 30 repetitions of forward + backward of the full model (batch 4, ragged reviews, text path on its
    side stream beside the VGG stack and the library's weight-gradient stream) from the same parameters give bit-identical
    gradients.  Round 3's first merge-backward kernel failed this in bf16 mode (about one repetition in three; text_ops.hip:
    merge_bwd_dx_kernel) - a parity test against the oracle passes such a kernel most of the time, this one does not."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_reproducible", os.path.join(ROOT, "tools", "check_reproducible.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main(["--dtype", dtype, "--reps", "30", "--batch", "4"]) == 0


def test_gradient_consumers_wait_for_side_stream_writes(dev):
    """umpr_amd/streams.py: an event noted after in-place writes on a side stream makes the consumer's stream wait (the hand-off
    autograd does not provide when a backward node returns None for a parameter whose gradient it wrote in place)."""
    from umpr_amd.streams import note_gradients_written, wait_for_gradients
    side = torch.cuda.Stream(dev)
    buf = torch.zeros(1 << 20, device=dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        torch.cuda._sleep(200_000_000)           # ~0.1 s of spinning in front of the write
        buf.fill_(3.0)
        note_gradients_written(dev)
        buf2 = buf * 1.0                          # a later write on the same stream: the one event per stream must cover it too
        note_gradients_written(dev)
    wait_for_gradients(dev)                       # current (default) stream now waits for the side stream's second event
    total = float((buf + buf2).sum())             # issued on the default stream
    assert total == 6.0 * (1 << 20), total
    wait_for_gradients(dev)                       # nothing pending: a no-op
