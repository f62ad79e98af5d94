"""PyTorch-CPU fp32 restatement of the reference recurrence path ("reference CPU path").

TEST INFRASTRUCTURE (see oracle/__init__.py).  Same ATen ops in the same order
as the reference so that results are bit-identical to the imported reference on
the same torch build (pinned by tests/golden/*.npz, see tests/test_oracle_golden.py).
It is also what ``bench.py`` times as ``cpu_baseline`` (kind "port").

What it follows (reference file:line):
  * one DP step .......................... hmm_layer/MsaHmmCell.py:73-106
  * initial states (pf = 1 and pf > 1) ... hmm_layer/MsaHmmCell.py:108-142
  * time loop ............................ hmm_layer/BaseRNN.py:194-248
  * chunk-summary cell ................... hmm_layer/TotalProbabilityCell.py:30-63
  * forward / backward / posterior drivers and chunk stitching
                                           hmm_layer/MsaHMMLayer.py:227-521
  * transition application ............... hmm_layer/gene_pred_hmm_transitioner.py:114-125

Deliberate differences, all of them workarounds for defects that make the
as-shipped drivers raise or return wrong values:
  D2  the reverse pass uses its own A^T instead of flipping a shared flag;
  D3  exactly one time flip for the backward pass;
  D4  the drivers are plain loops over the step function;
  D6  the last chunk's backward initial matrix is the identity;
  D7  pf == 1 posteriors are assembled without the broken reshape.
"""
import torch

EPS = 1e-16


class HmmParams:
    """A (k,q,q) and pi (1,k,q) fp32 — what ``recurrent_init`` caches
    (hmm_layer/MsaHmmCell.py:41-50, gene_pred_hmm_transitioner.py:66-71)."""

    def __init__(self, A, pi, eps=EPS):
        A = torch.as_tensor(A, dtype=torch.float32)
        pi = torch.as_tensor(pi, dtype=torch.float32)
        if A.dim() == 2:
            A = A.unsqueeze(0)
        self.A = A
        self.At = torch.transpose(A, 1, 2)
        self.k, self.q = A.shape[0], A.shape[-1]
        self.pi = pi.reshape(1, self.k, self.q)
        self.eps = eps


def cell_step(p, emission_row, state, reverse=False, init=False):
    """One recursion step; mirrors hmm_layer/MsaHmmCell.py:73-106 op for op.

    emission_row (k*n, q); state = [scaled (k*n, q | q*q), loglik (k*n, 1 | q)].
    Returns (output (k*n, q+1 | q*q+q), new_state).
    """
    k, q = p.k, p.q
    old_scaled, old_ll = state
    old_scaled = old_scaled.view(k, -1, q)
    if init:
        R = old_scaled
    else:
        R = torch.matmul(old_scaled, p.At if reverse else p.A)
    E = emission_row.view(k, -1, q)
    w = R.shape[1] // E.shape[1]            # 1, or q in chunked (conditional) mode
    R = R.view(k, -1, w, q)
    E = E.view(k, -1, 1, q)
    old_ll = old_ll.view(k, -1, w, 1)
    E = torch.maximum(E, torch.tensor(p.eps))
    R = torch.maximum(R, torch.tensor(p.eps))
    sf = E * R
    S = torch.sum(sf, dim=-1, keepdim=True)
    ll = old_ll + torch.log(S)
    sf /= S
    sf = sf.view(-1, w * q)
    ll = ll.view(-1, w)
    if reverse:
        out = torch.cat([torch.log(R).view(-1, w * q), old_ll.view(-1, w)], dim=-1)
    else:
        out = torch.cat([torch.log(sf), ll], dim=-1)
    return out, [sf, ll]


def initial_state(p, n, reverse=False, parallel_factor=1, chunk_emissions=None):
    """hmm_layer/MsaHmmCell.py:108-142.  n = batch * parallel_factor rows per model."""
    k, q = p.k, p.q
    if parallel_factor == 1:
        if reverse:
            dist = torch.ones((k * n, q), dtype=torch.float32)
        else:
            dist = p.pi.repeat(n, 1, 1).transpose(0, 1).reshape(-1, q)
        return [dist, torch.zeros((k * n, 1), dtype=torch.float32)]
    pf = parallel_factor
    eye = torch.eye(q, dtype=torch.float32).repeat(k * n, 1, 1)        # (k*n, q, q)
    if reverse:
        # conditional start of chunk c = diag(first emission of chunk c+1) @ A^T
        first = chunk_emissions[:, 0, :].view(k, n // pf, pf, q)
        first = torch.roll(first, shifts=-1, dims=2).reshape(k * n, 1, q)
        start = eye * first
    else:
        start = eye
    start = start.view(k, n * q, q)
    moved = torch.matmul(start, p.At if reverse else p.A).view(k, n // pf, pf, q * q)
    edge = torch.zeros((k, n // pf, pf, q * q), dtype=torch.float32)
    if reverse:
        edge[:, :, -1] = 1.0           # last chunk: identity (D6 worked around)
    else:
        edge[:, :, 0] = 1.0            # first chunk: identity
    eye = eye.view(k, n // pf, pf, q * q)
    dist = edge * eye + (1 - edge) * moved
    return [dist.view(k * n, q * q), torch.zeros((k * n, q), dtype=torch.float32)]


def run(p, rows, state, reverse=False):
    """Time loop over rows (N, T, q) in the order given (hmm_layer/BaseRNN.py:216-233)."""
    outs = []
    for t in range(rows.shape[1]):
        o, state = cell_step(p, rows[:, t], state, reverse=reverse)
        outs.append(o)
    if outs:
        return torch.stack(outs, dim=0).transpose(0, 1), state
    return rows.new_zeros((rows.shape[0], 0, 0)), state


# --------------------------------------------------------------------------- pf == 1

def forward_outputs(p, E):
    """E (k,b,L,q) -> per-step outputs (k*b, L, q+1) and the final state."""
    k, b, L, q = E.shape
    rows = E.reshape(k * b, L, q)
    st = initial_state(p, b)
    o1, st = cell_step(p, rows[:, 0], st, init=True)
    rest, st = run(p, rows[:, 1:], st)
    return torch.cat([o1.unsqueeze(1), rest], dim=1) if L > 1 else o1.unsqueeze(1), st


def backward_outputs(p, E):
    """Per-step reverse-cell outputs restored to time order, (k*b, L, q+1)."""
    k, b, L, q = E.shape
    rows = E.reshape(k * b, L, q)
    st = initial_state(p, b, reverse=True)
    o1, st = cell_step(p, rows[:, -1], st, reverse=True, init=True)
    rest, st = run(p, torch.flip(rows[:, :-1], [1]), st, reverse=True)
    out = torch.cat([o1.unsqueeze(1), rest], dim=1) if L > 1 else o1.unsqueeze(1)
    return torch.flip(out, [1]), st


def forward_recursion(p, E):
    """log alpha (k,b,L,q), loglik (k,b)   — hmm_layer/MsaHMMLayer.py:227-282 (pf=1)."""
    k, b, L, q = E.shape
    out, st = forward_outputs(p, E)
    out = out.reshape(k, b, L, -1)
    return out[..., :-1] + out[..., -1:], st[1].reshape(k, b)


def backward_recursion(p, E):
    """log beta (k,b,L,q)   — hmm_layer/MsaHMMLayer.py:322-381 (pf=1)."""
    k, b, L, q = E.shape
    out, _ = backward_outputs(p, E)
    out = out.reshape(k, b, L, -1)
    return out[..., :-1] + out[..., -1:]


def loglik_grad(A, pi, E, grad_loglik=None, eps=EPS):
    """Autograd through the restated time loop — the reference's own training path
    (hmm_layer/MsaHMMLayer.py:180-208 -> forward_recursion -> BaseRNN loop).
    A (k,q,q), pi (k,q), E (k,b,L,q) -> (dA, dpi, dE, loglik (k,b)) in fp32."""
    A = torch.as_tensor(A, dtype=torch.float32).clone().requires_grad_(True)
    pi = torch.as_tensor(pi, dtype=torch.float32).clone().requires_grad_(True)
    E = torch.as_tensor(E, dtype=torch.float32).clone().requires_grad_(True)
    _, ll = forward_recursion(HmmParams(A, pi, eps), E)
    w = torch.ones_like(ll) if grad_loglik is None else torch.as_tensor(grad_loglik, dtype=torch.float32)
    (ll * w).sum().backward()
    return A.grad, pi.grad.reshape(ll.shape[0], -1), E.grad, ll.detach()


def posterior_log_probs(p, E, no_loglik=False):
    """The reference's posterior formula log alpha + log beta - loglik in fp32
    (hmm_layer/MsaHMMLayer.py:501-514).  Cancels catastrophically for |loglik| >~ 1e5."""
    la, ll = forward_recursion(p, E)
    post = la + backward_recursion(p, E)
    if not no_loglik:
        post = post - ll.unsqueeze(-1).unsqueeze(-1)
    return post, ll


def posterior_scaled(p, E):
    """Posteriors from the scaled per-step variables (alpha_hat * R renormalised):
    same cell steps, numerically safe assembly.  Returns gamma (k,b,L,q), loglik."""
    k, b, L, q = E.shape
    fo, st = forward_outputs(p, E)
    bo, _ = backward_outputs(p, E)
    lg = fo[..., :-1] + bo[..., :-1]
    lg = lg - torch.logsumexp(lg, dim=-1, keepdim=True)
    return torch.exp(lg).reshape(k, b, L, q), st[1].reshape(k, b)


# --------------------------------------------------------------------------- pf > 1

def total_probability_step(cond, state):
    """hmm_layer/TotalProbabilityCell.py:30-49: log-space vector x matrix over one
    chunk summary.  cond (n, q*q) rows = conditioning state; state = (f (n,q), _)."""
    f, _ = state
    n = cond.shape[0]
    q = f.shape[-1]
    c = cond.view(n, q, q)
    f = torch.logsumexp(f.unsqueeze(-1) + c, dim=-2)
    return f, (f, torch.logsumexp(f, dim=-1))


def _split(x, q, n, pf, T):
    scaled = x[..., :-q].reshape(n, pf, T, q, -1)
    fac = x[..., -q:].reshape(n, pf, T, q, 1)
    return scaled + fac            # (n, pf, T, q cond, q actual)


def total_forward_from_chunks(p, fwd, b, L, pf):
    """hmm_layer/MsaHMMLayer.py:285-319."""
    k, q = p.k, p.q
    T = L // pf
    chunks = _split(fwd, q, k * b, pf, T)
    last = chunks[:, :, -1].reshape(k * b, pf, q * q)
    init = p.pi.repeat(b, 1, 1).transpose(0, 1).reshape(-1, q)
    st = (torch.log(init), torch.zeros(k * b))
    tot = []
    for c in range(pf):
        o, st = total_probability_step(last[:, c], st)
        tot.append(o)
    tot = torch.stack(tot, dim=1)
    first = torch.log(init + p.eps)
    Tm = torch.cat([first.unsqueeze(1), tot[:, :-1]], dim=1).unsqueeze(2).unsqueeze(4)
    res = torch.logsumexp((chunks + Tm).reshape(k, b, L, q, q), dim=-2)
    return res, st[1].reshape(k, b)


def total_backward_from_chunks(p, bwd, b, L, pf):
    """hmm_layer/MsaHMMLayer.py:384-419 with revert_chunks=False (chunks already in
    time order)."""
    k, q = p.k, p.q
    T = L // pf
    chunks = _split(bwd, q, k * b, pf, T)
    first = chunks[:, :, 0].reshape(k * b, pf, q * q)
    st = (torch.zeros(k * b, q), torch.zeros(k * b))
    tot = [None] * pf
    for c in range(pf - 1, -1, -1):
        o, st = total_probability_step(first[:, c], st)
        tot[c] = o
    tot = torch.stack(tot, dim=1)
    ones = torch.log(torch.ones(k * b, q) + p.eps)
    Tm = torch.cat([tot[:, 1:], ones.unsqueeze(1)], dim=1).unsqueeze(2).unsqueeze(4)
    return torch.logsumexp((chunks + Tm).reshape(k, b, L, q, q), dim=-2)


def chunked_outputs(p, E, pf):
    """Conditional (q x q per position) forward and backward outputs of the
    chunk-parallel mode, both in time order: (k*b*pf, T, q*q+q) each."""
    k, b, L, q = E.shape
    assert L % pf == 0, "parallel_factor must divide the sequence length"
    T = L // pf
    rows = E.reshape(k * b * pf, T, q)
    st = initial_state(p, b * pf, parallel_factor=pf)
    o1, st = cell_step(p, rows[:, 0], st, init=True)
    rest, _ = run(p, rows[:, 1:], st)
    fwd = torch.cat([o1.unsqueeze(1), rest], dim=1)
    st = initial_state(p, b * pf, reverse=True, parallel_factor=pf, chunk_emissions=rows)
    o1, st = cell_step(p, rows[:, -1], st, reverse=True, init=True)
    rest, _ = run(p, torch.flip(rows[:, :-1], [1]), st, reverse=True)
    bwd = torch.flip(torch.cat([o1.unsqueeze(1), rest], dim=1), [1])
    return fwd, bwd


def posterior_log_probs_chunked(p, E, pf, no_loglik=False):
    """hmm_layer/MsaHMMLayer.py:422-521 for parallel_factor > 1."""
    k, b, L, q = E.shape
    fwd, bwd = chunked_outputs(p, E, pf)
    la, ll = total_forward_from_chunks(p, fwd, b, L, pf)
    lb = total_backward_from_chunks(p, bwd, b, L, pf)
    post = la + lb
    if not no_loglik:
        post = post - ll.unsqueeze(-1).unsqueeze(-1)
    return post, ll, la, lb


def aggregate_loglik(loglik, weights=None):
    """hmm_layer/MsaHMMLayer.py:155-164 with aggregate=True."""
    if weights is not None:
        x = loglik * weights
        return torch.mean(torch.sum(x, dim=1) / torch.sum(weights, dim=1))
    return torch.mean(loglik)
