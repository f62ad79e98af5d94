"""hmm_posterior_grad (through the C ABI): gradient of a loss on the state posteriors.  Needs an MI355X.

Oracle: torch autograd in float64 through oracle/torch64.py (the scaled forward-backward with the
cell's clamps; pinned against oracle/textbook.py and against fp32 autograd through the restated
reference formula in tests/test_oracle_golden.py).  Tolerance: |g - g64| <= 3e-4 * max|g64| per
tensor (fp32 serial recursions over L steps; the log mode divides by small posteriors).
"""
import numpy as np
import pytest
import torch

from hmm_layer_amd import engine
from oracle import params, torch64

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev(x):
    return torch.as_tensor(np.asarray(x), dtype=torch.float32, device=DEV)


def rand_model(rng, q, sparse=False):
    A = rng.random((q, q)) ** 2 + 1e-2
    if sparse:
        A *= rng.random((q, q)) < 0.4
        A += np.eye(q) * 0.3
    A /= A.sum(-1, keepdims=True)
    pi = rng.random(q) + 0.1
    pi /= pi.sum()
    return A.astype(np.float32), pi.astype(np.float32)


def check(A, pi, E, G, mode, tag=""):
    """A (k,q,q), pi (k,q), E, G (k,b,L,q)."""
    dA, dpi, dE = [t.cpu().numpy() for t in engine.posterior_grad(dev(A), dev(pi), dev(E), dev(G), mode=mode)]
    for m in range(E.shape[0]):
        rA, rpi, rE, _ = torch64.posterior_grad(A[m], pi[m], E[m], G[m], log=(mode == engine.POST_LOG))
        for got, want, name in ((dA[m], rA, "dA"), (dpi[m], rpi, "dpi"), (dE[m], rE, "dE")):
            assert np.isfinite(got).all(), (tag, name)
            assert np.abs(got - want).max() <= 3e-4 * np.abs(want).max() + 1e-6, (tag, m, name, np.abs(got - want).max(), np.abs(want).max())
    return dA, dpi, dE


@pytest.mark.parametrize("mode", [engine.POST_LOG, engine.POST_PROB])
@pytest.mark.parametrize("q,b,L", [(1, 2, 9), (3, 3, 40), (7, 2, 65), (15, 3, 120), (16, 2, 50), (29, 2, 70), (64, 2, 33)])
def test_against_fp64_autograd(q, b, L, mode):
    rng = np.random.default_rng(q * 100 + L)
    A, pi = rand_model(rng, q)
    E = (rng.random((1, b, L, q)) * 0.9 + 0.05).astype(np.float32)
    G = rng.standard_normal((1, b, L, q)).astype(np.float32)
    check(A[None], pi[None], E, G, mode, "q=%d" % q)


def test_gene_model_sparse_and_clamped_emissions():
    rng = np.random.default_rng(3)
    A = params.intended_A15().numpy().astype(np.float32)
    pi = np.full(15, 1 / 15, dtype=np.float32)
    E = (rng.random((1, 3, 200, 15)) * 0.9 + 0.05).astype(np.float32)
    E[0, :, ::5, 7] = 0.0                                   # clamped emissions: no gradient there
    G = rng.standard_normal(E.shape).astype(np.float32)
    dA, dpi, dE = check(A[None], pi[None], E, G, engine.POST_PROB, "gene15")
    assert np.all(dE[0, :, ::5, 7] == 0.0)


def test_two_models_and_determinism():
    rng = np.random.default_rng(9)
    Ms = [rand_model(rng, 6, sparse=True) for _ in range(2)]
    A = np.stack([m[0] for m in Ms]); pi = np.stack([m[1] for m in Ms])
    E = (rng.random((2, 4, 90, 6)) * 0.9 + 0.05).astype(np.float32)
    G = rng.standard_normal(E.shape).astype(np.float32)
    a = check(A, pi, E, G, engine.POST_LOG, "two models")
    b = [t.cpu().numpy() for t in engine.posterior_grad(dev(A), dev(pi), dev(E), dev(G), mode=engine.POST_LOG)]
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_autograd_node_and_finite_difference():
    """hmm_layer_amd.autograd.posterior: forward = hmm_posterior, backward = hmm_posterior_grad;
    the gradient is the derivative of the engine's own forward value."""
    from hmm_layer_amd import autograd
    rng = np.random.default_rng(5)
    q = 5
    A, pi = rand_model(rng, q)
    E = (rng.random((1, 2, 60, q)) * 0.9 + 0.05).astype(np.float32)
    W = rng.standard_normal(E.shape).astype(np.float32)
    At, pit, Et = dev(A)[None].requires_grad_(True), dev(pi)[None].requires_grad_(True), dev(E).requires_grad_(True)
    out = autograd.posterior(At, pit, Et, mode=engine.POST_PROB)
    loss = (out * dev(W)).sum()
    loss.backward()

    def f(A_, E_):
        o, _ = engine.posterior(dev(A_)[None], dev(pi)[None], dev(E_), mode=engine.POST_PROB)
        return float((o.double() * dev(W).double()).sum())

    h = 2e-3
    for (i, j) in [(0, 1), (3, 3)]:
        Ap, Am = A.copy(), A.copy()
        Ap[i, j] += h; Am[i, j] -= h
        fd = (f(Ap, E) - f(Am, E)) / (Ap[i, j] - Am[i, j])
        assert abs(fd - float(At.grad[0, i, j])) <= 3e-2 * abs(fd) + 3e-2
    for (s, t, j) in [(0, 0, 2), (1, 30, 4)]:
        Ep, Em = E.copy(), E.copy()
        Ep[0, s, t, j] += h; Em[0, s, t, j] -= h
        fd = (f(A, Ep) - f(A, Em)) / (Ep[0, s, t, j] - Em[0, s, t, j])
        assert abs(fd - float(Et.grad[0, s, t, j])) <= 3e-2 * abs(fd) + 3e-2


def test_errors():
    A = torch.eye(70, device=DEV)[None]
    with pytest.raises(ValueError):
        engine.posterior_grad(A, torch.full((1, 70), 1 / 70, device=DEV), torch.rand(1, 2, 8, 70, device=DEV),
                              torch.rand(1, 2, 8, 70, device=DEV))
    with pytest.raises(ValueError):
        engine.posterior_grad(torch.eye(3, device=DEV)[None], torch.ones(1, 3, device=DEV) / 3,
                              torch.rand(1, 2, 8, 3, device=DEV), torch.rand(1, 2, 8, 3, device=DEV),
                              mode=engine.POST_LOG_NO_LL)


def test_long_sequence_with_tiny_posteriors_stays_finite_and_accurate():
    """Gene model over 2 500 positions with emissions of the emitter's magnitude (1/4096 scale,
    30 % zeros in the constrained states): many posteriors are ~1e-30 or exactly clamped.  Log mode."""
    rng = np.random.default_rng(17)
    A = params.intended_A15().numpy().astype(np.float32)
    pi = np.full(15, 1 / 15, dtype=np.float32)
    E = (rng.random((1, 2, 2500, 15)) * 0.9 + 0.05).astype(np.float32) / np.float32(4096)
    dead = rng.random(E.shape) < 0.3
    dead[..., :6] = False
    E[dead] = 0.0
    # upstream gradient of a cross-entropy against a labelling: non-negative weights on log gamma
    G = -(rng.random(E.shape) < 0.1).astype(np.float32)
    dA, dpi, dE = [t.cpu().numpy() for t in engine.posterior_grad(dev(A)[None], dev(pi)[None], dev(E), dev(G), mode=engine.POST_LOG)]
    assert np.isfinite(dA).all() and np.isfinite(dpi).all() and np.isfinite(dE).all()
    rA, rpi, rE, _ = torch64.posterior_grad(A, pi, E[0], G[0], log=True)
    ok = np.isfinite(rE)
    assert ok.mean() > 0.99
    assert np.abs(dA[0] - rA)[A > 0].max() <= 2e-3 * np.abs(rA[A > 0]).max()
    assert np.abs(dE[0] - rE)[ok].max() <= 2e-3 * np.abs(rE[ok]).max()


def test_no_loglik_variant_through_the_autograd_node():
    """out = log gamma + loglik (no_loglik=True): posterior gradient + weighted log-likelihood gradient."""
    from hmm_layer_amd import autograd
    rng = np.random.default_rng(23)
    for q in (6, 29):
        A, pi = rand_model(rng, q, sparse=True)
        E = (rng.random((1, 3, 80, q)) * 0.9 + 0.05).astype(np.float32)
        G = rng.standard_normal(E.shape).astype(np.float32)
        At, pit, Et = dev(A)[None].requires_grad_(True), dev(pi)[None].requires_grad_(True), dev(E).requires_grad_(True)
        out = autograd.posterior(At, pit, Et, mode=engine.POST_LOG_NO_LL)
        (out * dev(G)).sum().backward()
        rA, rpi, rE, rout = torch64.posterior_grad(A, pi, E[0], G[0], log=True, add_loglik=True)
        g64, ll64 = torch64.posterior(torch.tensor(A, dtype=torch.float64), torch.tensor(pi, dtype=torch.float64),
                                      torch.tensor(E[0], dtype=torch.float64))
        m = g64.numpy() > 1e-4          # log space only where eps-clamp paths cannot dominate the value
        assert np.abs(out.detach().cpu().numpy()[0] - rout)[m].max() <= 2e-3
        for got, want in ((At.grad[0], rA), (pit.grad[0], rpi), (Et.grad[0], rE)):
            assert np.abs(got.cpu().numpy() - want).max() <= 3e-4 * np.abs(want).max() + 1e-6
