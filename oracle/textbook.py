"""float64 numpy oracle: scaled forward-backward with the reference's clamps.

TEST INFRASTRUCTURE (see oracle/__init__.py).  This is the accuracy yardstick
the fp32 HIP engine and the fp32 reference cell path are both measured against.

Follows the single-step semantics of the reference cell,
hmm_layer/MsaHmmCell.py:73-106:

    R  = state            (first step, init=True)        :78-79
       = state @ A        (forward)  /  state @ A^T (reverse)   :80
    E  = max(E_t, eps);  R = max(R, eps)                 :87-88   eps = 1e-16 (:33)
    sf = E * R;  S = sum(sf);  loglik += log S;  sf /= S :89-92
    forward  output: log sf,  loglik                     :102-103
    reverse  output: log R,   old loglik                 :96-100

so that  log alpha_t = log sf_t + loglik_t  and  log beta_t = log R_t + loglik_{t+1..}.
With ``clamp=False`` it is the plain textbook recursion.
"""
import numpy as np

EPS = 1e-16


def _prep(A, pi, E):
    A = np.asarray(A, dtype=np.float64)
    E = np.asarray(E, dtype=np.float64)
    if E.ndim == 2:
        E = E[None]
    pi = None if pi is None else np.asarray(pi, dtype=np.float64).reshape(-1)
    return A, pi, E


def forward(A, pi, E, eps=EPS, clamp=True):
    """Scaled forward pass.

    A (q,q), pi (q,), E (b,L,q) probabilities.
    Returns alpha_hat (b,L,q) unit-sum rows, cum_loglik (b,L) = sum_{s<=t} log c_s.
    """
    A, pi, E = _prep(A, pi, E)
    b, L, q = E.shape
    ah = np.empty((b, L, q))
    ll = np.empty((b, L))
    state = np.broadcast_to(pi, (b, q))
    acc = np.zeros(b)
    for t in range(L):
        R = state if t == 0 else state @ A
        e = E[:, t]
        if clamp:
            R = np.maximum(R, eps)
            e = np.maximum(e, eps)
        sf = e * R
        S = sf.sum(-1, keepdims=True)
        acc = acc + np.log(S[:, 0])
        state = sf / S
        ah[:, t] = state
        ll[:, t] = acc
    return ah, ll


def backward(A, E, eps=EPS, clamp=True):
    """Scaled backward pass (reverse cell semantics).

    Returns R (b,L,q) = the pre-emission backward vector at t (beta_t up to scale)
    and scale (b,L) with  log beta_t = log R_t + scale_t,  beta_{L-1} = 1.
    """
    A, _, E = _prep(A, None, E)
    b, L, q = E.shape
    At = A.T
    Rs = np.empty((b, L, q))
    sc = np.empty((b, L))
    state = np.ones((b, q))
    acc = np.zeros(b)
    for i, t in enumerate(range(L - 1, -1, -1)):
        R = state if i == 0 else state @ At
        e = E[:, t]
        if clamp:
            R = np.maximum(R, eps)
            e = np.maximum(e, eps)
        Rs[:, t] = R
        sc[:, t] = acc
        sf = e * R
        S = sf.sum(-1, keepdims=True)
        acc = acc + np.log(S[:, 0])
        state = sf / S
    return Rs, sc


def log_alpha(A, pi, E, **kw):
    ah, ll = forward(A, pi, E, **kw)
    with np.errstate(divide="ignore"):
        return np.log(ah) + ll[..., None], ll[:, -1]


def log_beta(A, E, **kw):
    R, sc = backward(A, E, **kw)
    with np.errstate(divide="ignore"):
        return np.log(R) + sc[..., None]


def posterior(A, pi, E, eps=EPS, clamp=True):
    """State posteriors gamma (b,L,q) (rows sum to 1) and loglik (b,).

    gamma_t = alpha_hat_t * R_t / sum(alpha_hat_t * R_t): algebraically equal to
    exp(log alpha + log beta - loglik) (hmm_layer/MsaHMMLayer.py:501-514) but
    free of the large-number cancellation of that formula.
    """
    ah, ll = forward(A, pi, E, eps=eps, clamp=clamp)
    R, _ = backward(A, E, eps=eps, clamp=clamp)
    g = ah * R
    g /= g.sum(-1, keepdims=True)
    return g, ll[:, -1]


def loglik(A, pi, E, **kw):
    return forward(A, pi, E, **kw)[1][:, -1]


def loglik_grad(A, pi, E, grad_loglik=None, eps=EPS, clamp_adjoint=True):
    """d(sum_s w_s loglik_s)/d(A, pi, E) in fp64 -> (dA (q,q), dpi (q,), dE (b,L,q)).

    The adjoint of the reference's forward loop (hmm_layer/BaseRNN.py:217-227 over
    hmm_layer/MsaHmmCell.py:73-106) written as Baum-Welch expectations: dE = gamma / E,
    dA = sum_t xi_t / A, dpi = gamma_0 / pi.  The cell's clamps act as torch.maximum does under
    autograd: clamped E / pi entries get no gradient, and a predicted state (alpha_hat_{t-1} A)[j]
    below eps passes nothing back (it is masked out of the adjoint recursion and of xi_t(., j)) —
    this decides the gradient of absent edges into dead states.  clamp_adjoint=False leaves that
    mask out (plain Baum-Welch): the two versions bracket what the predicted-state clamp can do."""
    A, pi, E = _prep(A, pi, E)
    b, L, q = E.shape
    w = np.ones(b) if grad_loglik is None else np.asarray(grad_loglik, dtype=np.float64).reshape(b)
    ah, _ = forward(A, pi, E, eps=eps)
    Ec = np.maximum(E, eps)
    dA = np.zeros((q, q))
    dE = np.zeros((b, L, q))
    R = np.ones((b, q))                                       # adjoint of the predicted state, up to scale
    for t in range(L - 1, -1, -1):
        g = ah[:, t] * R
        g /= g.sum(-1, keepdims=True)
        dE[:, t] = np.where(E[:, t] > eps, g / Ec[:, t], 0.0) * w[:, None]
        if t == 0:
            dpi = np.where(pi > eps, (g * w[:, None]).sum(0) / np.maximum(pi, eps), 0.0)
            break
        bh = Ec[:, t] * R
        bh /= bh.sum(-1, keepdims=True)
        prev = ah[:, t - 1]
        if clamp_adjoint:
            bh = bh * ((prev @ A) > eps)                      # MsaHmmCell.py:88 clamp active -> no gradient
        R = np.maximum(bh @ A.T, eps)
        norm = (prev * R).sum(-1)
        dA += np.einsum("bi,bj->ij", prev * (w / norm)[:, None], bh)
    return dA, dpi, dE


def brute_force_loglik(A, pi, E):
    """Sum over all q^L paths; for q<=4, L<=8 property tests (no clamps)."""
    A, pi, E = _prep(A, pi, E)
    b, L, q = E.shape
    out = np.zeros(b)
    import itertools
    for n in range(b):
        tot = 0.0
        for path in itertools.product(range(q), repeat=L):
            p = pi[path[0]] * E[n, 0, path[0]]
            for t in range(1, L):
                p *= A[path[t - 1], path[t]] * E[n, t, path[t]]
            tot += p
        out[n] = np.log(tot)
    return out
