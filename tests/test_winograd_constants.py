"""The transform matrices of the Winograd tiles the HIP kernels implement (winograd.hip: F(2x2,3x3) forward, F(4x4,3x3) data
gradient / inference forward, F(3x3,4x4) weight gradient) satisfy the Winograd identities exactly in float64, and their fp32
rounding ranks as DESIGN.md states (tools/wino43_error.py is the experiment behind keeping the training forward on the 2x2
tile).  CPU only."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import wino43_error as W  # noqa: E402


def test_forward_identities_float64():
    rng = np.random.default_rng(0)
    for BT, G, AT, m in ((W.BT2, W.G2, W.AT2, 2), (W.BT4, W.G4, W.AT4, 4)):
        d = rng.standard_normal((m + 2, m + 2))
        g = rng.standard_normal((3, 3))
        y = AT @ ((G @ g @ G.T) * (BT @ d @ BT.T)) @ AT.T
        ref = np.array([[(d[i:i + 3, j:j + 3] * g).sum() for j in range(m)] for i in range(m)])
        assert np.abs(y - ref).max() < 1e-12, m


def test_weight_gradient_identity_float64():
    """dW = G^T [ (A g A^T) .* (B^T d B) ] G with A = (A^T)^T - the transposition of the forward algorithm that
    wino4_dy_kernel / wino4_wgrad_finish_kernel implement."""
    rng = np.random.default_rng(1)
    for BT, G, AT, m in ((W.BT2, W.G2, W.AT2, 2), (W.BT4, W.G4, W.AT4, 4)):
        d = rng.standard_normal((m + 2, m + 2))
        gy = rng.standard_normal((m, m))
        dw = G.T @ ((AT.T @ gy @ AT) * (BT @ d @ BT.T)) @ G
        ref = np.array([[(d[u:u + m, v:v + m] * gy).sum() for v in range(3)] for u in range(3)])
        assert np.abs(dw - ref).max() < 1e-12, m


def test_fp32_rounding_ranks_as_documented():
    rng = np.random.default_rng(2)
    C, H = 64, 16
    x = np.maximum(rng.standard_normal((1, C, H, H)), 0).astype(np.float32)
    w = (rng.standard_normal((8, C, 3, 3)) * np.sqrt(2 / (9 * C))).astype(np.float32)
    ref = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), padding=1).numpy()
    s = np.abs(ref).max()
    e2 = np.abs(W.wino(x, w, W.BT2, W.G2, W.AT2, 2) - ref).max() / s
    e4 = np.abs(W.wino(x, w, W.BT4, W.G4, W.AT4, 4) - ref).max() / s
    assert e2 < 1e-6 and e4 < 2e-5, (e2, e4)      # both inside test_conv3x3's per-layer bounds
    assert e4 > 3 * e2, (e2, e4)                  # and the larger tile is the noisier one: why forward keeps F(2x2,3x3)
