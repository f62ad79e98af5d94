"""Viterbi oracle: max-plus recursion in Q-format fixed point.

TEST INFRASTRUCTURE (see oracle/__init__.py).  The reference has no Viterbi (only a docstring
mention, hmm_layer/MsaHmmCell.py:13, and an unused log_A_dense, :46-47): PARITY UNPINNED — this
file DEFINES the semantics the HIP engine's hmm_viterbi is bit-exact against.

Definition
    Q(x)      = rint(clip(x, -1024, 1024) * 2**16)  as integer      (x = fp32 log-probability;
                -inf and anything below -1024 is the finite "approximately log zero" -1024, the
                reference's convention for absent edges, hmm_layer/Transitioner.py:337)
    d_0[j]    = Q(log pi[j]) + Q(log E_0[j])
    d_t[j]    = max_i( d_{t-1}[i] + Q(log A[i,j]) ) + Q(log E_t[j]);  bp_t[j] = lowest maximising i
    path      = backtrace from the lowest j maximising d_{L-1}[j];   score = d_{L-1}[.] / 2**16
Integer max-plus is exactly associative, so any chunked / scanned evaluation order gives the same
integers, the same argmaxes and therefore the same path as this serial loop.
"""
import numpy as np

FRAC_BITS = 16
CLAMP = 1024.0


def quantise(x):
    x = np.asarray(x, dtype=np.float32)
    with np.errstate(invalid="ignore"):
        y = np.clip(x, np.float32(-CLAMP), np.float32(CLAMP)) * np.float32(1 << FRAC_BITS)
    return np.rint(y).astype(np.int64)


def viterbi(logA, logpi, logE):
    """logA (q,q), logpi (q,), logE (b,L,q) fp32 -> path (b,L) int32, score (b,) float64."""
    a = quantise(logA)
    p0 = quantise(logpi).reshape(-1)
    e = quantise(logE)
    if e.ndim == 2:
        e = e[None]
    b, L, q = e.shape
    d = p0[None, :] + e[:, 0]
    bp = np.zeros((b, L, q), dtype=np.int8)
    for t in range(1, L):
        cand = d[:, :, None] + a[None, :, :]               # (b, i, j)
        bp[:, t] = cand.argmax(axis=1)                     # first (lowest) maximiser
        d = cand.max(axis=1) + e[:, t]
    path = np.zeros((b, L), dtype=np.int32)
    s = d.argmax(axis=1)
    score = d[np.arange(b), s].astype(np.float64) / (1 << FRAC_BITS)
    for t in range(L - 1, -1, -1):
        path[:, t] = s
        if t > 0:
            s = bp[np.arange(b), t, s]
    return path, score


def path_score(logA, logpi, logE, path):
    """Score of a given path under the quantised model (for property tests)."""
    a, p0, e = quantise(logA), quantise(logpi).reshape(-1), quantise(logE)
    tot = p0[path[0]] + e[0, path[0]]
    for t in range(1, len(path)):
        tot += a[path[t - 1], path[t]] + e[t, path[t]]
    return tot / (1 << FRAC_BITS)


def brute_force(logA, logpi, logE):
    """Best path by enumeration (q**L paths): q <= 4, L <= 8.  Ties -> the serial recursion's
    choice is not defined by enumeration order, so only the score is compared."""
    import itertools
    L, q = logE.shape
    best = -np.inf
    for p in itertools.product(range(q), repeat=L):
        best = max(best, path_score(logA, logpi, logE, list(p)))
    return best
