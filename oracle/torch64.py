"""float64 torch restatement of the scaled forward-backward with the cell's clamps, so that torch
autograd yields reference-semantics gradients of the POSTERIORS (and of the log-likelihood).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Same recursion as oracle/textbook.py (numpy) and as
the reference's cell step (hmm_layer/MsaHmmCell.py:73-106, forward and reverse direction), written
with differentiable torch ops in fp64: `torch.maximum` against eps routes no gradient through
clamped entries, exactly as autograd through the reference's Python loop does.  Pinned against
oracle/textbook.py and against autograd through oracle/ref_cell.py in tests/test_oracle_golden.py.
"""
import torch

EPS = 1e-16


def posterior(A, pi, E, eps=EPS):
    """A (q,q), pi (q,), E (b,L,q) float64 tensors -> gamma (b,L,q) (rows sum to 1), loglik (b,)."""
    b, L, q = E.shape
    epst = torch.tensor(eps, dtype=E.dtype)
    Ec = torch.maximum(E, epst)
    ah, ll = [], torch.zeros(b, dtype=E.dtype)
    state = pi.expand(b, q)
    for t in range(L):
        R = state if t == 0 else state @ A
        sf = Ec[:, t] * torch.maximum(R, epst)
        S = sf.sum(-1, keepdim=True)
        ll = ll + torch.log(S[:, 0])
        state = sf / S
        ah.append(state)
    Rb = [None] * L
    bh = None
    for t in range(L - 1, -1, -1):
        R = torch.ones((b, q), dtype=E.dtype) if t == L - 1 else torch.maximum(bh @ A.T, epst)
        Rb[t] = R
        sb = Ec[:, t] * R
        bh = sb / sb.sum(-1, keepdim=True)
    g = torch.stack(ah, 1) * torch.stack(Rb, 1)
    return g / g.sum(-1, keepdim=True), ll


def posterior_grad(A, pi, E, grad_out, log=True, eps=EPS, add_loglik=False):
    """d <grad_out, out> / d(A, pi, E) with out = log gamma (log=True) or gamma, plus loglik per
    sequence if add_loglik (the reference's no_loglik=True output); numpy in, numpy out."""
    import numpy as np
    A = torch.as_tensor(np.asarray(A), dtype=torch.float64).clone().requires_grad_(True)
    pi = torch.as_tensor(np.asarray(pi), dtype=torch.float64).clone().requires_grad_(True)
    E = torch.as_tensor(np.asarray(E), dtype=torch.float64).clone().requires_grad_(True)
    G = torch.as_tensor(np.asarray(grad_out), dtype=torch.float64)
    gam, ll = posterior(A, pi, E, eps)
    out = torch.log(gam) if log else gam
    if add_loglik:
        out = out + ll[:, None, None]
    (out * G).sum().backward()
    def grad(t):                      # a length-1 sequence never touches A
        return (torch.zeros_like(t) if t.grad is None else t.grad).numpy()
    return grad(A), grad(pi), grad(E), out.detach().numpy()
