"""CPU study (fp64): which certificate should route a sequence to the exact-clamp recomputation?

On emitter-generated gene-model input (SURVEY section 8(d)'s parity recipe) compare, per sequence,
  Phi  = eps * sum_t 1 / <alpha_hat_t, R_t>                      (round 2's certificate)
  Psi  = sum_t posterior mass on states whose forward or backward prediction sat at the clamp
and the actual deviation from the reference's max-clamp recursion of
  * the floor-free recursion   (no clamp on the state mixture)
  * the floored-linear one     (u + eps instead of max(u, eps): what additive chunk operators compute)
Usage: python tools/experiments/cert_study.py [scale] [L] [b] [pN]
"""
import sys
import os
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import params  # noqa: E402

EPS = 1e-16


def make_input(b, L, scale, pN, seed=0):
    g = torch.Generator().manual_seed(seed)
    cls = torch.softmax(scale * torch.randn((1, b, L, 15), generator=g), -1)
    idx = torch.randint(0, 4, (1, b, L), generator=g)
    isn = torch.rand((1, b, L), generator=g) < pN
    idx = torch.where(isn, torch.full_like(idx, 4), idx)
    nuc = torch.nn.functional.one_hot(idx, 5).float()
    x = torch.cat([cls, nuc], -1)
    kernel = torch.randn((1, 13, 15), generator=g)
    tab = params.codon_table(**params.DEFAULT_CODONS)
    E = params.gene_emissions(x, kernel, tab).numpy()[0].astype(np.float64)
    A = params.intended_A15(200, 4500, 10000).numpy().astype(np.float64)
    pi = np.full(15, 1 / 15)
    return A, pi, E


def recursions(A, pi, E, mode):
    """mode: 'max' (reference), 'free' (no mixture clamp), 'add' (u + eps).  Returns alpha_hat, R, clamp masks."""
    b, L, q = E.shape
    Ec = np.maximum(E, EPS)
    ah = np.empty((b, L, q)); fm = np.zeros((b, L, q), bool)
    st = np.broadcast_to(pi, (b, q)).copy()
    ll = np.zeros(b)
    for t in range(L):
        u = st if t == 0 else st @ A
        if mode == "max":
            fm[:, t] = u < EPS
            r = np.maximum(u, EPS)
        elif mode == "add":
            r = u + EPS if t > 0 else np.maximum(u, EPS)
        else:
            r = u if t > 0 else np.maximum(u, EPS)
        sf = Ec[:, t] * r
        S = sf.sum(-1, keepdims=True)
        ll += np.log(S[:, 0])
        st = sf / S
        ah[:, t] = st
    R = np.empty((b, L, q)); bm = np.zeros((b, L, q), bool)
    st = np.ones((b, q))
    for i, t in enumerate(range(L - 1, -1, -1)):
        u = st if i == 0 else st @ A.T
        if mode == "max":
            bm[:, t] = (u < EPS) if i > 0 else False
            r = np.maximum(u, EPS)
        elif mode == "add":
            r = u + EPS if i > 0 else u
        else:
            r = u
        R[:, t] = r
        sf = Ec[:, t] * r
        st = sf / sf.sum(-1, keepdims=True)
    return ah, R, fm, bm, ll


def main():
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    b = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    pN = float(sys.argv[4]) if len(sys.argv) > 4 else 0.01
    A, pi, E = make_input(b, L, scale, pN)
    print("scale %g  L %d  b %d  pN %g   zeros in E: %.1f %%" % (scale, L, b, pN, 100 * (E == 0).mean()))
    ah, R, fm, bm, ll = recursions(A, pi, E, "max")
    g = ah * R
    Sg = g.sum(-1)
    g /= Sg[..., None]
    # R of the reverse cell is the clamped prediction from a unit-sum bh: <ah, R> is the normaliser the engine sums
    phi = EPS * (1.0 / Sg).sum(-1)
    psi_f = (g * fm).sum((-1, -2))
    psi_b = (g * bm).sum((-1, -2))
    out = {}
    for mode in ("free", "add"):
        ah2, R2, _, _, ll2 = recursions(A, pi, E, mode)
        g2 = ah2 * R2
        g2 /= g2.sum(-1, keepdims=True)
        out[mode] = (np.abs(g2 - g).max((-1, -2)), np.abs(ll2 - ll))
    print("seq      Phi      Psi_fwd   Psi_bwd   |dgamma| free   add     |dll| free    add     min Sg")
    for s in range(b):
        print("%3d  %9.2e %9.2e %9.2e   %9.2e %9.2e   %9.2e %9.2e  %9.2e" % (
            s, phi[s], psi_f[s], psi_b[s], out["free"][0][s], out["add"][0][s], out["free"][1][s], out["add"][1][s],
            Sg[s].min()))


if __name__ == "__main__":
    main()
