"""Host input pipeline (umpr_amd/data.py) against fixtures produced by the reference's own src/dataset.py and
src/word2vec.py on a tiny corpus (tests/golden/tiny_corpus, tests/golden/make_golden.py::gen_dataset).  Index work:
bit-exact."""
import json
import os

import numpy as np
import torch

from conftest import GOLDEN
from umpr_amd.data import Dataset, Word2vec, batch_loader, pad_reviews

ROOT = os.path.join(GOLDEN, "tiny_corpus")
G = json.load(open(os.path.join(GOLDEN, "dataset_golden.json")))


class Cfg:
    max_sent_count = 6
    min_sent_count = 2
    max_ui_sent_count = 3
    max_sent_length = 10
    photo_count = 2
    views = ["food"]
    review_level = "sentence"


def w2v():
    return Word2vec(os.path.join(ROOT, "glove.txt"))


def test_word2vec():
    w = w2v()
    probe = ["w1 w2. w3 77 w55", "  w9 w9 w9 ", "12 w0 w59 w3 w4 w5 w6 w7"]
    got = [w.sent2indices(p) for p in probe] + [w.sent2indices(probe[2], 5), w.sent2indices(probe[1], 6)]
    assert got == G["sent2indices"]
    assert len(w) == G["vocab_len"]
    for row, ref in zip((0, 1, 2, 3, 52), G["embedding_rows"]):
        np.testing.assert_allclose(w.embedding[row], ref, rtol=0, atol=0)
    assert not w.embedding[:3].any()  # <PAD>/<UNK>/<NUM> are zero vectors


def _norm(x):
    return json.loads(json.dumps(x))


def test_dataset_and_collate():
    w = w2v()
    for name, views, level in (("amazon", ["food"], "sentence"), ("two_views", ["food", "inside"], "sentence"),
                               ("review_level", ["food"], "review")):
        cfg = Cfg()
        cfg.views, cfg.review_level = views, level
        ds = Dataset(os.path.join(ROOT, "train.csv"), os.path.join(ROOT, "photos.json"), os.path.join(ROOT, "photos"), w, cfg)
        assert [bool(b) for b in ds.retain_idx] == G["retain/" + name], name
        assert _norm([list(x) for x in ds.data]) == G["dataset/" + name], name
        if "batch/" + name in G:
            b = batch_loader([ds[i] for i in range(min(4, len(ds)))], ignore_photos=True)
            ref = G["batch/" + name]
            for t, r in zip(b, ref):
                assert t.tolist() == r
            assert b[0].dtype == torch.int64 and b[3].dtype == torch.int64 and b[7].dtype == torch.float32


def test_pad_reviews_and_missing_photo():
    assert _norm(pad_reviews([[[1, 2, 3], []], [[4]]])) == G["pad_reviews"]
    from umpr_amd.data import get_image
    img = get_image("/nonexistent.jpg")
    assert img.shape == (3, 224, 224) and not img.any()
    # with photos: unreadable paths become zero images, layout [B, V, P, 3, 224, 224]
    w = w2v()
    ds = Dataset(os.path.join(ROOT, "train.csv"), os.path.join(ROOT, "photos.json"), os.path.join(ROOT, "photos"), w, Cfg())
    b = batch_loader([ds[0], ds[1]])
    assert tuple(b[6].shape) == (2, 1, 2, 3, 224, 224) and b[6].dtype == torch.float32


def test_worker_processes_collate_identically():
    """main.py's picklable collate in DataLoader worker processes (--loader_workers) yields the batches of the in-process
    path, tensor for tensor."""
    from torch.utils.data import DataLoader
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from main import _Collate
    w = w2v()
    ds = Dataset(os.path.join(ROOT, "train.csv"), os.path.join(ROOT, "photos.json"), os.path.join(ROOT, "photos"), w, Cfg())
    a = list(DataLoader(ds, batch_size=3, collate_fn=_Collate(True), num_workers=0))
    b = list(DataLoader(ds, batch_size=3, collate_fn=_Collate(True), num_workers=2))
    assert len(a) == len(b) and len(a) >= 2
    for x, y in zip(a, b):
        for t, u in zip(x, y):
            assert torch.equal(t, u)


def test_resize_bilinear_follows_the_cv2_recipe():
    """resize_bilinear_u8 (OpenCV's 8-bit INTER_LINEAR restated): identity at equal size, constants stay constant, a
    horizontal ramp is reproduced to 1 grey level, and the result stays within 1 level of float bilinear sampling at
    half-pixel centres (the fixed-point rounding)."""
    from umpr_amd.data import resize_bilinear_u8
    g = np.random.default_rng(0)
    img = g.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    assert np.array_equal(resize_bilinear_u8(img, (53, 37)), img)
    flat = np.full((40, 60, 3), 77, np.uint8)
    assert np.array_equal(resize_bilinear_u8(flat, (224, 224)), np.full((224, 224, 3), 77, np.uint8))
    big = g.integers(0, 256, (375, 500, 3), dtype=np.uint8)
    out = resize_bilinear_u8(big, (224, 224))
    assert out.shape == (224, 224, 3) and out.dtype == np.uint8

    def ref(src, dw, dh):   # float bilinear, same tap rule
        h, w, _ = src.shape
        fx = (np.arange(dw) + 0.5) * w / dw - 0.5
        fy = (np.arange(dh) + 0.5) * h / dh - 0.5
        x0 = np.floor(fx).astype(int); ax = fx - x0
        ax = np.where((x0 < 0) | (x0 >= w - 1), 0.0, ax)
        x0 = np.clip(x0, 0, w - 1); x1 = np.minimum(x0 + 1, w - 1)
        y0 = np.floor(fy).astype(int); ay = fy - y0
        y1 = np.clip(y0 + 1, 0, h - 1); y0 = np.clip(y0, 0, h - 1)
        s = src.astype(np.float64)
        rows = s[:, x0] * (1 - ax)[None, :, None] + s[:, x1] * ax[None, :, None]
        return rows[y0] * (1 - ay)[:, None, None] + rows[y1] * ay[:, None, None]

    assert np.abs(out.astype(np.float64) - ref(big, 224, 224)).max() <= 1.0
    up = resize_bilinear_u8(img, (224, 224))
    assert np.abs(up.astype(np.float64) - ref(img, 224, 224)).max() <= 1.0
