"""fp32 numpy model of the engine's three-phase scan (reduce -> scan -> apply).

Not the product and not the oracle: a CPU executable specification of what the HIP
kernels in hmm_layer_amd/csrc/hmm_engine.hip compute, used by the CPU test-suite to
check the algorithm (chunk operators with per-column power-of-two scaling, the
chunk-level scan in both directions, the in-chunk apply with checkpoints) against the
fp64 textbook oracle.  One sequence at a time, plain loops.
"""
import numpy as np

F = np.float32


def _clamp(x, eps):
    return np.maximum(x, F(eps))


def reduce_chunk(A, Erows, first, eps):
    """Operator of one chunk: X[i,k] = P(obs of chunk, state i at its last step | state k
    just before the chunk), columns scaled by 2^-ex[k].  `first`: the chunk starts at
    t=0, whose step has no transition (hmm_layer/MsaHmmCell.py:78-79)."""
    q = A.shape[0]
    X = np.eye(q, dtype=F)
    ex = np.zeros(q, dtype=np.int64)
    At = A.T.astype(F)
    for t, e in enumerate(Erows):
        e = _clamp(e.astype(F), eps)
        if first and t == 0:
            X = X * e[:, None]
        else:
            X = (At @ X).astype(F) * e[:, None]          # exactly linear: no clamp of the state mixture here
        s = X.sum(axis=0, dtype=F)
        _, xe = np.frexp(s)
        X = np.ldexp(X, -xe[None, :]).astype(F)
        ex += xe
    return X, ex


def scan_forward(ops, exs, pi, eps):
    """prefix[c] = alpha_hat entering chunk c (unit sum; c=0: raw pi), llpre[c]."""
    C = len(ops)
    q = pi.shape[0]
    prefix = [pi.astype(F)]
    llpre = [0.0]
    a = _clamp(pi.astype(F), eps)
    ll = 0.0
    for c in range(C):
        _, ae = np.frexp(a)
        we = np.where(a > 0, ae + exs[c], -10**9)
        emax = we.max()
        w = np.ldexp(a, (exs[c] - emax).clip(-300, 300)).astype(F)
        new = (ops[c] @ w).astype(F)
        S = new.sum(dtype=F)
        a = (new / S).astype(F)
        ll += float(np.log(S)) + float(emax) * np.log(2.0)
        prefix.append(a)
        llpre.append(ll)
    return prefix[:-1], llpre[:-1], ll


def scan_backward(ops, exs, q):
    """suffix[c] = beta at the last position of chunk c up to the factor exp(lsuf[c])."""
    C = len(ops)
    suffix = [None] * C
    lsuf = [0.0] * C
    v = np.ones(q, dtype=F)
    lb = 0.0
    for c in range(C - 1, -1, -1):
        suffix[c] = v
        lsuf[c] = lb
        u = (ops[c].T @ v).astype(F)
        _, ue = np.frexp(u)
        we = np.where(u > 0, ue + exs[c], -10**9)
        emax = we.max()
        v = np.ldexp(u, (exs[c] - emax).clip(-300, 300)).astype(F)
        lb += float(emax) * np.log(2.0)
    return suffix, lsuf


def apply_chunk(A, Erows, prefix, suffix, first, eps):
    """Exact cell semantics inside one chunk: returns alpha_hat (T,q), R (T,q),
    per-step log normalisers of both passes."""
    T, q = Erows.shape
    A = A.astype(F)
    ah = np.empty((T, q), dtype=F)
    lc = np.empty(T)
    x = prefix.astype(F)
    for t in range(T):
        R = x if (first and t == 0) else (x @ A).astype(F)
        sf = _clamp(R, eps) * _clamp(Erows[t].astype(F), eps)
        S = sf.sum(dtype=F)
        x = (sf / S).astype(F)
        ah[t] = x
        lc[t] = np.log(S)
    Rs = np.empty((T, q), dtype=F)
    lb = np.empty(T)
    R = suffix.astype(F)
    acc = 0.0
    for t in range(T - 1, -1, -1):
        Rs[t] = R
        lb[t] = acc
        sf = _clamp(Erows[t].astype(F), eps) * R
        S = sf.sum(dtype=F)
        acc += np.log(S)
        R = _clamp((A @ (sf / S).astype(F)).astype(F), eps)
    return ah, lc, Rs, lb


def posterior(A, pi, E, T, eps=1e-16):
    """E (L,q) one sequence.  Returns gamma (L,q), loglik, log_alpha, log_beta."""
    A = np.asarray(A, dtype=F)
    pi = np.asarray(pi, dtype=F)
    L, q = E.shape
    C = (L + T - 1) // T
    ops, exs = [], []
    for c in range(C):
        X, ex = reduce_chunk(A, E[c * T:(c + 1) * T], c == 0, eps)
        ops.append(X)
        exs.append(ex)
    prefix, llpre, loglik = scan_forward(ops, exs, pi, eps)
    suffix, lsuf = scan_backward(ops, exs, q)
    gam = np.empty((L, q), dtype=F)
    la = np.empty((L, q))
    lbeta = np.empty((L, q))
    for c in range(C):
        rows = E[c * T:(c + 1) * T]
        ah, lc, Rs, lb = apply_chunk(A, rows, prefix[c], suffix[c], c == 0, eps)
        g = ah * Rs
        g = g / g.sum(-1, keepdims=True, dtype=F)
        gam[c * T:c * T + len(rows)] = g
        with np.errstate(divide="ignore"):
            la[c * T:c * T + len(rows)] = np.log(ah) + (llpre[c] + np.cumsum(lc))[:, None]
            lbeta[c * T:c * T + len(rows)] = np.log(Rs) + (lsuf[c] + lb)[:, None]
    return gam, loglik, la, lbeta
