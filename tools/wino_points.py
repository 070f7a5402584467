"""Cook-Toom construction of the F(m x m, 3 x 3) Winograd matrices from an arbitrary set of interpolation points, and the
fp32 rounding error of each candidate against the float64 convolution (CPU experiment behind the choice of the points of the
4x4-output tile in winograd.hip).  `python tools/wino_points.py` prints the table."""
import itertools
from fractions import Fraction as Fr

import numpy as np
import torch
import torch.nn.functional as F


def poly_mul(a, b):
    out = [Fr(0)] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            out[i + j] += x * y
    return out


def cook_toom(points, m, r=3, scale=None):
    """(AT [m,n], G [n,r], BT [n,n]) as exact fractions for the finite `points` (n-1 of them) plus infinity.
    y = AT [(G g) .* (BT d)].  `scale`: per-row factors s_j moved from G into BT (G row j / s_j, BT row j * s_j)."""
    n = m + r - 1
    pts = [Fr(p) for p in points]
    assert len(pts) == n - 1 and len(set(pts)) == n - 1
    AT = [[(p ** i if i else Fr(1)) for p in pts] + [Fr(1) if i == m - 1 else Fr(0)] for i in range(m)]
    G, BT = [], []
    for j, p in enumerate(pts):
        N = Fr(1)
        Mj = [Fr(1)]
        for l, q in enumerate(pts):
            if l != j:
                N *= (p - q)
                Mj = poly_mul(Mj, [-q, Fr(1)])
        G.append([(p ** k if k else Fr(1)) / N for k in range(r)])
        BT.append(Mj + [Fr(0)] * (n - len(Mj)))
    Mall = [Fr(1)]
    for q in pts:
        Mall = poly_mul(Mall, [-q, Fr(1)])
    G.append([Fr(0)] * (r - 1) + [Fr(1)])
    BT.append(Mall)
    if scale is not None:
        for j, s in enumerate(scale):
            s = Fr(s)
            G[j] = [v / s for v in G[j]]
            BT[j] = [v * s for v in BT[j]]
    f = lambda M: np.array([[float(v) for v in row] for row in M], dtype=np.float64)
    return f(AT), f(G), f(BT)


def check_identity(AT, G, BT, m):
    rng = np.random.default_rng(0)
    d = rng.standard_normal((m + 2, m + 2))
    g = rng.standard_normal((3, 3))
    y = AT @ ((G @ g @ G.T) * (BT @ d @ BT.T)) @ AT.T
    ref = np.array([[(d[i:i + 3, j:j + 3] * g).sum() for j in range(m)] for i in range(m)])
    return np.abs(y - ref).max()


def wino(x, w, BT, G, AT, m, dtype=np.float32):
    """Winograd convolution with every intermediate rounded to `dtype`; the channel reduction is a float32 dot product whose
    partial sums are kept in float64 chunks of 32 like the kernels' two-level accumulation (chains of 32-144 folded into a total)."""
    f = dtype
    N, C, H, W = x.shape
    a = m + 2
    Hp, Wp = -(-H // m) * m, -(-W // m) * m
    xp = np.zeros((N, C, Hp + 2, Wp + 2), f)
    xp[:, :, 1:H + 1, 1:W + 1] = x
    # 1-D passes one after the other, each rounded (as the kernels do: rows then columns)
    U = np.einsum('ai,mcij->mcaj', G.astype(f), w.astype(f)).astype(f)
    U = np.einsum('mcaj,bj->abmc', U, G.astype(f)).astype(f)
    y = np.zeros((N, w.shape[0], Hp, Wp), f)
    for ty in range(Hp // m):
        for tx in range(Wp // m):
            d = xp[:, :, ty * m:ty * m + a, tx * m:tx * m + a]
            V = np.einsum('ai,ncij->ncaj', BT.astype(f), d).astype(f)
            V = np.einsum('ncaj,bj->abnc', V, BT.astype(f)).astype(f)
            Mx = np.zeros(U.shape[:2] + (N, w.shape[0]), np.float64)
            for c0 in range(0, C, 32):
                Mx += np.einsum('abmc,abnc->abnm', U[..., c0:c0 + 32], V[..., c0:c0 + 32]).astype(f)
            Mx = Mx.astype(f)
            t = np.einsum('ia,abnm->ibnm', AT.astype(f), Mx).astype(f)
            y[:, :, ty * m:ty * m + m, tx * m:tx * m + m] = np.einsum('ibnm,jb->nmij', t, AT.astype(f)).astype(f)
    return y[:, :, :H, :W]


CANDIDATES = {
    "F2 (0,1,-1)": ((0, 1, -1), 2, None),
    "F4 std (0,1,-1,2,-2)": ((0, 1, -1, 2, -2), 4, None),
    "F4 (0,1,-1,1/2,-1/2)": ((0, 1, -1, Fr(1, 2), Fr(-1, 2)), 4, None),
    "F4 (0,1,-1,1/2,-2)": ((0, 1, -1, Fr(1, 2), -2), 4, None),
    "F4 (0,1,-1,2,-1/2)": ((0, 1, -1, 2, Fr(-1, 2)), 4, None),
    "F4 (0,1/2,-1/2,3/2,-3/2)": ((0, Fr(1, 2), Fr(-1, 2), Fr(3, 2), Fr(-3, 2)), 4, None),
    "F4 (0,1,-1,3/2,-3/2)": ((0, 1, -1, Fr(3, 2), Fr(-3, 2)), 4, None),
    "F4 (0,3/4,-3/4,3/2,-3/2)": ((0, Fr(3, 4), Fr(-3, 4), Fr(3, 2), Fr(-3, 2)), 4, None),
    "F4 (0,1/2,-1/2,1,-2)": ((0, Fr(1, 2), Fr(-1, 2), 1, -2), 4, None),
    "F4 (0,1/2,-1/2,2,-2)": ((0, Fr(1, 2), Fr(-1, 2), 2, -2), 4, None),
    "F4 (0,1,-1,1/2,-3/2)": ((0, 1, -1, Fr(1, 2), Fr(-3, 2)), 4, None),
    "F4 (0,2/3,-2/3,4/3,-4/3)": ((0, Fr(2, 3), Fr(-2, 3), Fr(4, 3), Fr(-4, 3)), 4, None),
    "F3 (0,1,-1,2)": ((0, 1, -1, 2), 3, None),
    "F3 (0,1,-1,1/2)": ((0, 1, -1, Fr(1, 2)), 3, None),
    "F3 (0,1,-1,-1/2)": ((0, 1, -1, Fr(-1, 2)), 3, None),
}


def measure(points, m, scale, shapes=((128, 56), (256, 28), (512, 28)), cout=32, seed=0):
    AT, G, BT = cook_toom(points, m, scale=scale)
    ident = check_identity(AT, G, BT, m)
    rng = np.random.default_rng(seed)
    out = []
    for C, H in shapes:
        x = np.maximum(rng.standard_normal((1, C, H, H)), 0).astype(np.float32)
        w = (rng.standard_normal((cout, C, 3, 3)) * np.sqrt(2 / (9 * C))).astype(np.float32)
        ref = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), padding=1).numpy()
        y = wino(x, w, BT, G, AT, m)
        e = np.abs(y - ref)
        out.append((e.max() / np.abs(ref).max(), np.sqrt((e ** 2).mean()) / np.sqrt((ref ** 2).mean())))
    return ident, out


def main():
    for name, (pts, m, scale) in CANDIDATES.items():
        ident, out = measure(pts, m, scale)
        print(f"{name:32s} identity {ident:.1e}  " + "  ".join(f"max/max {a:.2e} rms/rms {b:.2e}" for a, b in out), flush=True)


if __name__ == "__main__":
    main()
