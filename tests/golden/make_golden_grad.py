#!/usr/bin/env python3
"""Gradient fixtures: autograd through the IMPORTED reference's own cell loop — the reference's
training mechanism (hmm_layer/BaseRNN.py:217-227 over HmmCell.forward, MsaHmmCell.py:73-106).
Same harness as make_golden.py (build container only; writes tests/golden/grad_*.npz).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_grad.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                                                    # noqa: E402  (imports the reference)


def ref_loglik_grads(A, pi, E, w):
    """d(sum_s w_s loglik_s)/d(A, pi, E) by autograd through the reference cell's forward loop."""
    A = A.clone().requires_grad_(True)
    pi = pi.clone().requires_grad_(True)
    E = E.clone().requires_grad_(True)
    cell, _ = mg.make_cells(A, pi)
    B, Ln, q = E.shape
    s = cell.get_initial_state(batch_size=B)
    o, s = cell(E[:, 0], s, init=True)
    for t in range(1, Ln):
        o, s = cell(E[:, t], s)
    loglik = s[1].reshape(B)
    (loglik * w).sum().backward()
    return dict(A=A.detach(), pi=pi.detach(), E=E.detach(), w=w, loglik=loglik.detach(),
                dA=A.grad, dpi=pi.grad.reshape(-1), dE=E.grad)


def main():
    g = torch.Generator().manual_seed(4321)
    out = {}
    q = 5
    A = torch.softmax(2 * torch.randn((q, q), generator=g), -1)
    pi = torch.softmax(torch.randn(q, generator=g), -1)
    E = torch.rand((3, 90, q), generator=g) * 0.9 + 0.05
    w = torch.rand(3, generator=g) + 0.5
    out["grad_q5"] = mg.npy(ref_loglik_grads(A, pi, E, w))
    tr15 = mg.GenePredMultiHMMTransitioner(initial_exon_len=200, initial_intron_len=4500, initial_ir_len=10000,
                                           starting_distribution_init="zeros")
    with torch.no_grad():
        tr15.transition_kernel[tr15.transition_kernel == 0] = 1e-30          # D1
    A15 = tr15.make_A()[0].detach()
    pi15 = tr15.make_initial_distribution().detach().reshape(-1)
    E15 = mg.rand_emissions(g, 2, 130, 15, False)
    w15 = torch.tensor([1.5, -0.75])
    out["grad_q15"] = mg.npy(ref_loglik_grads(A15, pi15, E15, w15))
    for name, d in out.items():
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
        print(name, {k: v.shape for k, v in d.items()})


if __name__ == "__main__":
    main()
