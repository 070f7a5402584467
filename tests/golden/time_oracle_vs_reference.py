#!/usr/bin/env python3
"""Build-container only (imports /root/reference like make_golden.py): times one training step of the reference's own
UMPR module and of the oracle (oracle/umpr_ref.py, aten=True) on identical inputs and weights, same thread count.
Shows that the CPU baseline bench.py reports (the oracle, kind "port") is not slower than the reference itself
(SURVEY.md 8(d) "CPU reference timing").   usage: python tests/golden/time_oracle_vs_reference.py"""
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402
from oracle import umpr_ref as R  # noqa: E402
from umpr_amd.synthetic import make_batch, make_param_state  # noqa: E402


def time_steps(fn, n):
    fn()
    t = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    return min(t), sorted(t)[len(t) // 2]


def main():
    torch.set_num_threads(8)
    MG._install_import_shims()
    from src import model as refmodel
    for name, ro, B, n in (("UMPR-R (cfg1 shapes), batch 32", True, 32, 5), ("full UMPR V=1, batch 8", False, 8, 2)):
        P = make_param_state(0, 50, 5000, 1, ro, m_scale=0.05)
        batch = make_batch(1, B, 5000, 1, review_net_only=ro, full_pad=True)
        ref = MG.build_reference(refmodel, P, ro, 1)
        ref.train()
        opt = torch.optim.Adam([
            {'params': (p for n_, p in ref.named_parameters() if 'bias' not in n_)},
            {'params': (p for n_, p in ref.named_parameters() if 'bias' in n_), 'weight_decay': 0.}], 1e-6, weight_decay=1e-3)

        def ref_step():
            pred, loss = ref(*batch)
            loss = loss.mean()
            opt.zero_grad()
            loss.backward()
            opt.step()

        Q = {k: v.clone().requires_grad_(k != "embedding.weight") for k, v in P.items()}
        names = [k for k in Q if k != "embedding.weight"]
        oopt = torch.optim.Adam([
            {'params': [Q[k] for k in names if 'bias' not in k]},
            {'params': [Q[k] for k in names if 'bias' in k], 'weight_decay': 0.}], 1e-6, weight_decay=1e-3)

        def ora_step():
            g = torch.Generator().manual_seed(0)
            masks = None if ro else [(torch.rand(B, 4096, generator=g) < 0.5).float() for _ in range(2)]
            pred, loss = R.umpr_forward(Q, batch, review_net_only=ro, train=not ro, dropout_masks=masks, aten=True)
            oopt.zero_grad()
            loss.backward()
            oopt.step()

        rb, rm = time_steps(ref_step, n)
        ob, om = time_steps(ora_step, n)
        print(f"{name}: reference best {rb * 1e3:.0f} ms median {rm * 1e3:.0f} ms ({B / rm:.1f} samples/s) | "
              f"oracle best {ob * 1e3:.0f} ms median {om * 1e3:.0f} ms ({B / om:.1f} samples/s)", flush=True)


if __name__ == "__main__":
    main()
