"""Rounding error of F(4x4,3x3) against F(2x2,3x3) and the direct fp32 convolution, measured against float64.
Plain numpy / torch-CPU experiment behind the decision to use the larger tile on the 56x56 and 28x28 layers.
BT4 / G4 / AT4 are the matrices winograd.hip implements since round 3: interpolation points (0, +-3/4, +-3/2, inf)
(tools/wino_points.py builds them and compares point sets); *_STD are the textbook points (0, +-1, +-2, inf) of rounds 1-2."""
import os, sys
from fractions import Fraction as Fr
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from wino_points import cook_toom  # noqa: E402

AT4, G4, BT4 = cook_toom((0, Fr(3, 4), Fr(-3, 4), Fr(3, 2), Fr(-3, 2)), 4)
BT4_STD = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                [0, 4, 0, -5, 0, 1]], dtype=np.float64)
G4_STD = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
               [0, 0, 1]], dtype=np.float64)
AT4_STD = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=np.float64)
BT2 = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
G2 = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)
AT2 = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)


def wino(x, w, BT, G, AT, m):
    """fp32 Winograd: every intermediate rounded to fp32, the channel reduction in fp32 (np.einsum pairwise-ish)."""
    f = np.float32
    BT, G, AT = BT.astype(f), G.astype(f), AT.astype(f)
    N, C, H, W = x.shape
    a = m + 2
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1))).astype(f)
    U = np.einsum('ai,mcij,bj->abmc', G, w.astype(f), G).astype(f)
    y = np.zeros((N, w.shape[0], H, W), f)
    for ty in range(H // m):
        for tx in range(W // m):
            d = xp[:, :, ty * m:ty * m + a, tx * m:tx * m + a]
            V = np.einsum('ai,ncij,bj->abnc', BT, d, BT).astype(f)
            Mx = np.einsum('abmc,abnc->abnm', U, V).astype(f)
            y[:, :, ty * m:ty * m + m, tx * m:tx * m + m] = np.einsum('ia,abnm,jb->nmij', AT, Mx, AT).astype(f)
    return y


def main():
    rng = np.random.default_rng(0)
    for C, H in ((128, 56), (256, 28), (512, 28)):
        x = np.maximum(rng.standard_normal((1, C, H, H)), 0).astype(np.float32)
        w = (rng.standard_normal((32, C, 3, 3)) * np.sqrt(2 / (9 * C))).astype(np.float32)
        ref = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), padding=1).numpy()
        d32 = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), padding=1).numpy()
        y2 = wino(x, w, BT2, G2, AT2, 2)
        y4 = wino(x, w, BT4, G4, AT4, 4)
        y4s = wino(x, w, BT4_STD, G4_STD, AT4_STD, 4)
        s = np.abs(ref).max()
        rms = np.sqrt((ref ** 2).mean())
        for name, y in (('direct fp32', d32), ('F(2x2,3x3)', y2), ('F(4x4,3x3) 3/4,3/2', y4), ('F(4x4,3x3) 1,2', y4s)):
            e = np.abs(y - ref)
            print(f'C={C} H={H} {name:20s} max err / max|y| = {e.max() / s:.2e}   rms err / rms y = {np.sqrt((e ** 2).mean()) / rms:.2e}')


if __name__ == '__main__':
    main()
