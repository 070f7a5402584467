"""The transform matrices of the Winograd tiles the HIP kernels implement (winograd.hip: F(2x2,3x3), F(4x4,3x3) on the points
(0, +-3/4, +-3/2, inf) for forward / data gradient, F(3x3,4x4) weight gradient) satisfy the Winograd identities exactly in
float64, the closed forms the kernels evaluate (B^T, G, A^T in terms of a, b) equal the Cook-Toom matrices, and their fp32
rounding ranks as DESIGN.md states (tools/wino43_error.py, tools/wino_points.py: the experiments behind moving the training
forward onto the 4x4 tile with these points).  CPU only."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import wino43_error as W  # noqa: E402


def test_forward_identities_float64():
    rng = np.random.default_rng(0)
    for BT, G, AT, m in ((W.BT2, W.G2, W.AT2, 2), (W.BT4, W.G4, W.AT4, 4), (W.BT4_STD, W.G4_STD, W.AT4_STD, 4)):
        d = rng.standard_normal((m + 2, m + 2))
        g = rng.standard_normal((3, 3))
        y = AT @ ((G @ g @ G.T) * (BT @ d @ BT.T)) @ AT.T
        ref = np.array([[(d[i:i + 3, j:j + 3] * g).sum() for j in range(m)] for i in range(m)])
        assert np.abs(y - ref).max() < 1e-12, m


def test_weight_gradient_identity_float64():
    """dW = G^T [ (A g A^T) .* (B^T d B) ] G with A = (A^T)^T - the transposition of the forward algorithm that
    wino4_dy_kernel / wino4_wgrad_finish_kernel implement."""
    rng = np.random.default_rng(1)
    for BT, G, AT, m in ((W.BT2, W.G2, W.AT2, 2), (W.BT4, W.G4, W.AT4, 4), (W.BT4_STD, W.G4_STD, W.AT4_STD, 4)):
        d = rng.standard_normal((m + 2, m + 2))
        gy = rng.standard_normal((m, m))
        dw = G.T @ ((AT.T @ gy @ AT) * (BT @ d @ BT.T)) @ G
        ref = np.array([[(d[u:u + m, v:v + m] * gy).sum() for v in range(3)] for u in range(3)])
        assert np.abs(dw - ref).max() < 1e-12, m


def _closed_form(a, b):
    """The matrices as winograd.hip's wino4_bt / wino4_g / wino4_at / wino4_a / wino4_gt evaluate them (Wino4C)."""
    p, sm, ab2, a2b = a * a * b * b, a * a + b * b, a * b * b, a * a * b
    ca, cb = 1 / (2 * a * a * (a * a - b * b)), 1 / (2 * b * b * (b * b - a * a))
    BT = np.array([[p, 0, -sm, 0, 1, 0], [0, -ab2, -b * b, a, 1, 0], [0, ab2, -b * b, -a, 1, 0], [0, -a2b, -a * a, b, 1, 0],
                   [0, a2b, -a * a, -b, 1, 0], [0, p, 0, -sm, 0, 1]])
    G = np.array([[1 / p, 0, 0], [ca, ca * a, ca * a * a], [ca, -ca * a, ca * a * a], [cb, cb * b, cb * b * b],
                  [cb, -cb * b, cb * b * b], [0, 0, 1]])
    AT = np.array([[1, 1, 1, 1, 1, 0], [0, a, -a, b, -b, 0], [0, a * a, a * a, b * b, b * b, 0],
                   [0, a ** 3, -a ** 3, b ** 3, -b ** 3, 1]])
    return BT, G, AT


def test_kernel_closed_forms_equal_the_cook_toom_matrices():
    for (a, b), (BT, G, AT) in (((0.75, 1.5), (W.BT4, W.G4, W.AT4)), ((1.0, 2.0), (W.BT4_STD, W.G4_STD, W.AT4_STD))):
        cBT, cG, cAT = _closed_form(a, b)
        assert np.abs(cBT - BT).max() < 1e-14 and np.abs(cG - G).max() < 1e-14 and np.abs(cAT - AT).max() < 1e-14, (a, b)
    # every constant of B^T and A^T for (3/4, 3/2) is dyadic: exact in fp32
    for Mx in (W.BT4, W.AT4):
        assert np.array_equal(Mx.astype(np.float32).astype(np.float64), Mx)


def test_fp32_rounding_ranks_as_documented():
    rng = np.random.default_rng(2)
    C, H = 64, 16
    x = np.maximum(rng.standard_normal((1, C, H, H)), 0).astype(np.float32)
    w = (rng.standard_normal((8, C, 3, 3)) * np.sqrt(2 / (9 * C))).astype(np.float32)
    ref = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), padding=1).numpy()
    rms = lambda y: float(np.sqrt(((y - ref) ** 2).mean()) / np.sqrt((ref ** 2).mean()))
    e2 = rms(W.wino(x, w, W.BT2, W.G2, W.AT2, 2))
    e4 = rms(W.wino(x, w, W.BT4, W.G4, W.AT4, 4))
    e4s = rms(W.wino(x, w, W.BT4_STD, W.G4_STD, W.AT4_STD, 4))
    assert e2 < 5e-7 and e4 < 1.5e-6 and e4s < 5e-6, (e2, e4, e4s)    # all inside test_conv3x3's per-layer bounds
    assert e4 < 0.7 * e4s, (e4, e4s)      # the (3/4, 3/2) points are the quieter 4x4 tile: why round 3 moved to them
    assert e2 < e4, (e2, e4)              # the 2x2 tile stays the quietest
