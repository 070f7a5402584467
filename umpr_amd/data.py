"""Host-side input pipeline with the reference's semantics.

* ``Word2vec``  - GloVe text -> vocabulary + matrix; ids 0/1/2 = <PAD>/<UNK>/<NUM>, all three zero vectors
  (src/word2vec.py:6-36,46-52).  Only the 'glove' source is supported (gensim is not installed).
* ``Dataset``   - CSV -> per-sample (user sentences without the target review, item sentences, ui sentences, photo
  paths, rating) with the reference's filters (src/dataset.py:11-119): sentences of <= 5 tokens dropped, reviews
  without sentences dropped, min/max sentence counts with longest-first *stable* truncation, target review excluded
  from the user/item pools, samples whose item lacks a photo for some view dropped.
* ``pad_reviews`` / ``batch_loader`` - the collate that produces the 8-tensor batch ``UMPR.forward`` consumes
  (src/dataset.py:122-182): batch-wide common (count, length) for user and item, independent padding for ui, pad id 0,
  empty sentences get length 1.
* ``get_image`` - JPEG -> CHW RGB in [0,1], 224x224 (src/dataset.py:134-143).  The reference decodes and resizes with
  cv2, which is not installed here: the file is decoded with PIL and resized by ``resize_bilinear_u8``, a restatement
  of OpenCV's 8-bit INTER_LINEAR (2 taps at half-pixel centres, 11-bit fixed-point weights - PIL's own BILINEAR
  antialiases when shrinking and would give different pictures).  Unpinned against cv2 itself ("parity unpinned" for
  decode + resize); unreadable files become zeros like the reference.

Everything here runs on the host; tensors come out in the layout the kernels expect.
"""
from __future__ import annotations

import os
from collections import defaultdict
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pandas as pd
import torch

PAD, UNK, NUM = '<PAD>', '<UNK>', '<NUM>'


class Word2vec:
    def __init__(self, emb_path, source='glove', vocab_size=0):
        if source != 'glove':
            raise NotImplementedError("only GloVe text files are supported (gensim is not available)")
        self.padding, self.unknown, self.number = PAD, UNK, NUM
        self.vocab = [PAD, UNK, NUM]
        self.word2index = {PAD: 0, UNK: 1, NUM: 2}
        rows = []
        with open(emb_path, encoding='utf-8') as f:
            for line in f:
                tokens = line.split()
                if not tokens:
                    continue
                self.vocab.append(tokens[0])
                self.word2index[tokens[0]] = len(self.word2index)  # duplicates keep the later id, as the reference does
                rows.append(np.asarray(tokens[1:], dtype=np.float64))
        dim = len(rows[0])
        self.embedding = np.zeros((3 + len(rows), dim), dtype=np.float64)  # rows 0..2 stay zero
        self.embedding[3:] = np.stack(rows)
        self.word_dim = dim

    def sent2indices(self, sentence, align_length=0):
        indices = []
        for w in sentence.replace('.', ' ').strip().split():
            if w.isdigit():
                indices.append(2)
            else:
                indices.append(self.word2index.get(w, 1))
            if 0 < align_length <= len(indices):
                break
        if 0 < align_length and len(indices) < align_length:
            indices += [0] * (align_length - len(indices))
        return indices

    def pad(self, sequence, pad_length):
        return (sequence + [0] * (pad_length - len(sequence)))[:pad_length]

    def __len__(self):
        return len(self.embedding)


class Dataset(torch.utils.data.Dataset):
    def __init__(self, data_path, photo_json, photo_dir, word2vec, config):
        self.max_s_count = config.max_sent_count
        self.min_s_count = config.min_sent_count
        self.max_ui_s_count = config.max_ui_sent_count
        self.max_s_length = config.max_sent_length
        self.photo_count = config.photo_count
        self.views = config.views
        df = pd.read_csv(data_path)
        by_sentence = config.review_level == 'sentence'

        def to_sentences(text):
            parts = str(text).strip('. ').split('.') if by_sentence else [str(text)]
            sents = [word2vec.sent2indices(s)[: self.max_s_length] for s in parts]
            return [s for s in sents if len(s) > 5]

        reviews = [to_sentences(x) for x in df['review']]
        keep = [len(r) > 0 for r in reviews]
        photos = self._photo_paths(photo_json, photo_dir, list(df['itemID']), keep)
        users = self._pool(list(df['user_num']), list(df['item_num']), reviews, keep)
        items = self._pool(list(df['item_num']), list(df['user_num']), reviews, keep)
        uis = []
        for i, sents in enumerate(reviews):
            if not keep[i]:
                uis.append(None)
                continue
            if len(sents) > self.max_ui_s_count:
                sents.sort(key=lambda x: -len(x))  # in place, like the reference: later pools see the sorted list
                sents = sents[: self.max_ui_s_count]
            uis.append(sents)
        self.retain_idx = keep
        ratings = list(df['rating'])
        sel = [i for i, k in enumerate(keep) if k]
        self.data = ([users[i] for i in sel], [items[i] for i in sel], [uis[i] for i in sel],
                     [photos[i] for i in sel], [ratings[i] for i in sel])

    def __getitem__(self, idx):
        return tuple(x[idx] for x in self.data)

    def __len__(self):
        return len(self.data[0])

    def _pool(self, lead, costar, reviews, keep):
        """Sentences of every review the lead (user or item) wrote/received except the one about `costar`."""
        groups = defaultdict(list)
        for l, c, r in zip(lead, costar, reviews):
            groups[l].append((c, r))
        out = []
        for i, (l, c) in enumerate(zip(lead, costar)):
            if not keep[i]:
                out.append(None)
                continue
            sents = [s for cid, r in groups[l] if cid != c for s in r]
            if len(sents) < self.min_s_count:
                keep[i] = False
                out.append(None)
                continue
            if len(sents) > self.max_s_count:
                sents.sort(key=lambda x: -len(x))  # stable, longest first
                sents = sents[: self.max_s_count]
            out.append(sents)
        return out

    def _photo_paths(self, photos_json, photo_dir, item_ids, keep):
        photo_df = pd.read_json(photos_json, orient='records', lines=True)
        if 'label' not in photo_df.columns:
            photo_df['label'] = self.views[0]  # Amazon photos carry no view label
        groups = defaultdict(dict)
        for bid, pid, label in zip(photo_df['business_id'], photo_df['photo_id'], photo_df['label']):
            if label in self.views:
                groups[bid].setdefault(label, []).append(pid)
        out = []
        for i, bid in enumerate(item_ids):
            if not keep[i]:
                out.append(None)
                continue
            per_view = []
            for label in self.views:
                pids = groups[bid].get(label, [])
                if len(pids) < 1:
                    keep[i] = False
                    per_view = None
                    break
                paths = [os.path.join(photo_dir, str(p) + '.jpg') for p in pids[: self.photo_count]]
                paths += ['unknown'] * (self.photo_count - len(paths))
                per_view.append(paths)
            out.append(per_view)
        return out


def pad_reviews(reviews, max_count=None, max_len=None, pad=0):
    if max_count is None:
        max_count = max(len(r) for r in reviews)
    reviews = [sents + [[]] * (max_count - len(sents)) for sents in reviews]
    lengths = [[max(1, len(s)) for s in sents] for sents in reviews]
    if max_len is None:
        max_len = max(max(l) for l in lengths)
    padded = [[s + [pad] * (max_len - len(s)) for s in sents] for sents in reviews]
    return padded, lengths


def _linear_coeffs(dst, src):
    """Source index pairs and 11-bit fixed-point weights of OpenCV's 8-bit INTER_LINEAR resize along one axis
    (half-pixel centres: f = (d + 0.5) * src / dst - 0.5; weights cvRound(w * 2048))."""
    f = (np.arange(dst, dtype=np.float64) + 0.5) * (src / dst) - 0.5
    f = f.astype(np.float32)                       # OpenCV keeps the coordinate in a float
    s = np.floor(f).astype(np.int64)
    w = (f - s).astype(np.float32)
    return s, w


def resize_bilinear_u8(img, size):
    """cv2.resize(img, size) for uint8 HxWxC with the default INTER_LINEAR, restated from OpenCV's fixed-point path
    (resize.cpp: HResizeLinear / VResizeLinear for uchar - 2 taps, no antialiasing, coefficients scaled by 2^11):
    horizontally, taps left of column 0 or right of the last column collapse onto the edge pixel with weight 1;
    vertically the row indices are clipped.  cv2 is not installed here, so this is unpinned against cv2 itself; it is
    what the reference's get_image computes by construction, where PIL's own BILINEAR filter would antialias."""
    dw, dh = size
    h, w, _ = img.shape
    sx, fx = _linear_coeffs(dw, w)
    fx = np.where((sx < 0) | (sx >= w - 1), np.float32(0), fx)
    sx = np.clip(sx, 0, w - 1)
    sx1 = np.minimum(sx + 1, w - 1)
    a1 = np.rint(fx * np.float32(2048)).astype(np.int32)
    a0 = np.rint((np.float32(1) - fx) * np.float32(2048)).astype(np.int32)
    sy, fy = _linear_coeffs(dh, h)
    b1 = np.rint(fy * np.float32(2048)).astype(np.int32)
    b0 = np.rint((np.float32(1) - fy) * np.float32(2048)).astype(np.int32)
    y0 = np.clip(sy, 0, h - 1)
    y1 = np.clip(sy + 1, 0, h - 1)
    used = np.unique(np.concatenate([y0, y1]))                     # only the source rows some output row taps
    src = np.ascontiguousarray(np.take(img, used, axis=0).transpose(2, 0, 1)).astype(np.int32)   # [c][rows][w]
    rows = np.take(src, sx, axis=2) * a0 + np.take(src, sx1, axis=2) * a1                         # scaled by 2^11
    rows >>= 4
    r0 = np.take(rows, np.searchsorted(used, y0), axis=1)          # [c][dh][dw]
    r1 = np.take(rows, np.searchsorted(used, y1), axis=1)
    out = (((b0[None, :, None] * r0) >> 16) + ((b1[None, :, None] * r1) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8).transpose(1, 2, 0)


def get_image(path, resize=(224, 224)):
    try:
        from PIL import Image
        with Image.open(path) as im:
            rgb = np.asarray(im.convert('RGB'), dtype=np.uint8)    # cv2.imread + BGR2RGB, resized channel-wise
        a = resize_bilinear_u8(rgb, resize).astype(np.float64).transpose(2, 0, 1) / 255.0
        return a
    except Exception:
        return np.zeros([3] + list(resize))


def batch_loader(batch_list, ignore_photos=False, photo_size=(224, 224), pad=0, shard=None):
    """src/dataset.py:146-182.  `shard=(rank, world)` (data parallel, not in the reference): the review tensors are padded
    to the GLOBAL batch's common (max_count, max_len) exactly as the reference's single collate does before
    DataParallel scatters them (main.py:82), but only this rank's contiguous chunk is kept - and only its photos are
    decoded, so R ranks do 1/R of the JPEG work each instead of all of it.  A ninth element then carries the number
    of ranks with a non-empty chunk (parallel.active_shards)."""
    lo, hi = 0, len(batch_list)
    if shard is not None:
        from .parallel import active_shards, shard_bounds
        lo, hi = shard_bounds(len(batch_list), *shard)
    mine = batch_list[lo:hi]
    users = [s[0] for s in batch_list]
    items = [s[1] for s in batch_list]
    uis = [s[2] for s in batch_list]
    ratings = [s[4] for s in mine]
    photos = []
    if not ignore_photos and mine:
        paths = [p for s in mine for view in s[3] for p in view]
        with ThreadPoolExecutor() as pool:
            imgs = iter(list(pool.map(lambda x: get_image(x, photo_size), paths)))
        photos = [[[next(imgs) for _ in view] for view in s[3]] for s in mine]
    max_count = max(max(len(u), len(i)) for u, i in zip(users, items))
    max_len = max(max(max(len(s) for s in u), max(len(s) for s in i)) for u, i in zip(users, items))
    pu, lu = pad_reviews(users, max_count, max_len, pad)
    pi, li = pad_reviews(items, max_count, max_len, pad)
    pui, lui = pad_reviews(uis, pad=pad)
    long = lambda rows: torch.LongTensor(rows[lo:hi]) if hi > lo else torch.zeros((0,) + tuple(torch.LongTensor(rows[:1]).shape[1:]), dtype=torch.long)
    out = (long(pu), long(pi), long(pui), long(lu), long(li), long(lui),
           torch.from_numpy(np.asarray(photos, dtype=np.float32)) if photos else
           (torch.Tensor([]) if ignore_photos or not batch_list else
            torch.zeros((0, len(batch_list[0][3]), len(batch_list[0][3][0]), 3) + tuple(photo_size))),
           torch.Tensor(ratings))
    if shard is not None:
        out = out + (torch.tensor(active_shards(len(batch_list), shard[1])),)
    return out
