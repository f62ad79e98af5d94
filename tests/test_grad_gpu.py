"""Parity of hmm_loglik_grad (through the C ABI) with the oracle.  Needs an MI355X.

Oracles: (1) oracle.textbook.loglik_grad — Baum-Welch expectations in fp64; (2)
oracle.ref_cell.loglik_grad — torch autograd through the restated time loop, which is how the
reference itself trains (hmm_layer/BaseRNN.py:217-227).  The two agree to fp32 rounding
(tests/test_oracle_golden.py::test_gradient_oracles_agree).

Tolerance (fp32 engine vs fp64 oracle):  |g - g64| <= 2e-4 * max|g64| per tensor — sums of up
to b*L fp32 terms per entry for dA; dE entries are single ratios, compared at 2e-5 relative to
the tensor's largest entry plus 1e-4 relative per entry.
Exception, stated: dA entries of ABSENT edges (A[i][j] == 0).  Such an entry is "what if mass
entered state j"; when j is a dead state (forward mass at the eps floor) its value is decided by
the steps at which the cell's clamp of the predicted state (MsaHmmCell.py:88) is active, and a
state sitting on the floor crosses it back and forth with E_t[j]/S_t.  The engine follows the
cell's clamp step by step inside a chunk, but its chunk-boundary vectors carry floor-level entries
(~1e-16) with absolute, not relative, accuracy — the property that makes posterior parity a
probability-space statement (tests/test_engine_gpu.py) — so its clamp pattern can differ from the
serial loop's.  For these entries the test bounds the deviation by the size of the clamp effect
itself: |g - g64| <= |g64(no clamp mask) - g64| + 5e-3 * max|g64|.  The reference never uses these
entries: its A is scattered from per-edge parameters
(hmm_layer/gene_pred_hmm_transitioner.py:74-125), structural zeros are constants.
"""
import numpy as np
import pytest
import torch

from hmm_layer_amd import engine
from oracle import params, ref_cell, textbook

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def dev(x, dtype=torch.float32):
    return torch.as_tensor(np.asarray(x), dtype=dtype, device=DEV)


def rand_model(rng, q, sparse=False):
    A = rng.random((q, q)) ** 2 + 1e-2
    if sparse:
        A *= rng.random((q, q)) < 0.35
        A += np.eye(q) * 0.3
    A /= A.sum(-1, keepdims=True)
    pi = rng.random(q) + 0.1
    pi /= pi.sum()
    return A.astype(np.float32), pi.astype(np.float32)


def run_grad(A, pi, E, w=None):
    """A (k,q,q), pi (k,q), E (k,b,L,q), w (k,b) -> numpy dA, dpi, dE, ll."""
    out = engine.loglik_grad(dev(A), dev(pi), dev(E), None if w is None else dev(w))
    torch.cuda.synchronize()
    return [t.cpu().numpy() for t in out]


def check(A, pi, E, w=None, tag=""):
    k = E.shape[0]
    dA, dpi, dE, ll = run_grad(A, pi, E, w)
    for m in range(k):
        rA, rpi, rE = textbook.loglik_grad(A[m], pi[m], E[m], None if w is None else w[m])
        assert np.isfinite(dA[m]).all() and np.isfinite(dE[m]).all(), tag
        rU = textbook.loglik_grad(A[m], pi[m], E[m], None if w is None else w[m], clamp_adjoint=False)[0]
        tolA = np.where(A[m] > 0, 2e-4 * np.abs(rA).max(), np.abs(rU - rA) + 5e-3 * np.abs(rA).max())
        assert np.all(np.abs(dA[m] - rA) <= tolA), (tag, m, np.abs(dA[m] - rA).max(), np.abs(rA).max())
        assert np.abs(dpi[m] - rpi).max() <= 2e-4 * np.abs(rpi).max(), (tag, m)
        assert np.all(np.abs(dE[m] - rE) <= 2e-5 * np.abs(rE).max() + 1e-4 * np.abs(rE)), \
            (tag, m, np.abs(dE[m] - rE).max(), np.abs(rE).max())
        ll64 = textbook.loglik(A[m], pi[m], E[m])
        assert np.all(np.abs(ll[m] - ll64) <= 1e-6 * np.abs(ll64) + 2e-4), tag
    return dA, dpi, dE, ll


@pytest.mark.parametrize("q", [1, 2, 3, 5, 8, 13, 15, 16])
def test_state_counts(q):
    rng = np.random.default_rng(q)
    A, pi = rand_model(rng, q)
    E = (rng.random((1, 3, 70, q)) * 0.9 + 0.05).astype(np.float32)
    w = (rng.random((1, 3)) + 0.5).astype(np.float32)
    check(A[None], pi[None], E, w, "q=%d" % q)


@pytest.mark.parametrize("b,L", [(1, 1), (1, 2), (2, 15), (1, 16), (3, 17), (5, 33), (17, 100), (64, 257),
                                 (3, 1000), (2, 1025), (33, 2049)])
def test_ragged_lengths_and_batches(b, L):
    rng = np.random.default_rng(100 * b + L)
    A, pi = rand_model(rng, 6, sparse=True)
    E = (rng.random((1, b, L, 6)) * 0.9 + 0.05).astype(np.float32)
    check(A[None], pi[None], E, None, "b=%d L=%d" % (b, L))


def test_matches_autograd_through_the_reference_loop():
    """The reference's own mechanism: autograd through the cell loop (fp32)."""
    rng = np.random.default_rng(5)
    A, pi = rand_model(rng, 7, sparse=True)
    E = (rng.random((1, 4, 150, 7)) * 0.9 + 0.05).astype(np.float32)
    w = (rng.random((1, 4)) + 0.5).astype(np.float32)
    dA, dpi, dE, ll = run_grad(A[None], pi[None], E, w)
    gA, gpi, gE, gll = [t.numpy() for t in ref_cell.loglik_grad(A[None], pi[None], E, w)]
    rU = textbook.loglik_grad(A, pi, E[0], w[0], clamp_adjoint=False)[0]
    tol = np.where(A > 0, 3e-4 * np.abs(gA).max(), np.abs(rU - gA[0]) + 5e-3 * np.abs(gA).max())
    assert np.all(np.abs(dA[0] - gA[0]) <= tol)
    assert np.abs(dpi - gpi).max() <= 3e-4 * np.abs(gpi).max()
    assert np.abs(dE - gE).max() <= 3e-4 * np.abs(gE).max()
    assert np.abs(ll - gll).max() <= 3e-4


def test_gene_model_with_emitter_emissions(golden):
    """15-state gene model (23 edges: the sparse reduce kernel), emissions from the emitter fixture."""
    g = golden("cell_q15")
    E = np.tile(g["E"], (1, 6, 1))[None]            # (1, b, 6 L, 15)
    rng = np.random.default_rng(3)
    w = (rng.random((1, E.shape[1])) + 0.5).astype(np.float32)
    check(g["A"][None], g["pi"][None], E, w, "gene15")


def test_multi_model_and_weights():
    rng = np.random.default_rng(11)
    k, b, L, q = 3, 9, 300, 4
    Ms = [rand_model(rng, q) for _ in range(k)]
    A = np.stack([m[0] for m in Ms]); pi = np.stack([m[1] for m in Ms])
    E = (rng.random((k, b, L, q)) * 0.9 + 0.05).astype(np.float32)
    w = (rng.standard_normal((k, b))).astype(np.float32)          # upstream gradients of either sign
    check(A, pi, E, w, "multi")


def test_clamped_emissions_get_no_gradient():
    rng = np.random.default_rng(2)
    A, pi = rand_model(rng, 5)
    E = (rng.random((1, 2, 64, 5)) * 0.9 + 0.05).astype(np.float32)
    E[0, :, ::3, 1] = 0.0
    E[0, 1, 10, :] = 1e-20
    dA, dpi, dE, ll = check(A[None], pi[None], E, None, "clamp")
    assert np.all(dE[0, :, ::3, 1] == 0.0) and np.all(dE[0, 1, 10] == 0.0)


def test_finite_difference_of_engine_loglik():
    """Independent of any oracle: the gradient is the derivative of the engine's own loglik."""
    rng = np.random.default_rng(8)
    q = 5
    A, pi = rand_model(rng, q)
    E = (rng.random((1, 2, 90, q)) * 0.9 + 0.05).astype(np.float32)
    dA, dpi, dE, _ = run_grad(A[None], pi[None], E)

    def f(A_, pi_, E_):
        _, ll = engine.forward(dev(A_)[None], dev(pi_)[None], dev(E_), want_log_alpha=False)
        return ll.sum().item()

    h = 2e-3
    for (i, j) in [(0, 1), (3, 3), (4, 2)]:
        Ap, Am = A.copy(), A.copy()
        Ap[i, j] += h; Am[i, j] -= h
        fd = (f(Ap, pi, E) - f(Am, pi, E)) / (Ap[i, j] - Am[i, j])
        assert abs(fd - dA[0, i, j]) <= 2e-2 * abs(dA[0, i, j]) + 2e-2, (i, j, fd, dA[0, i, j])
    for (s, t, j) in [(0, 0, 2), (1, 50, 4)]:
        Ep, Em = E.copy(), E.copy()
        Ep[0, s, t, j] += h; Em[0, s, t, j] -= h
        fd = (f(A, pi, Ep) - f(A, pi, Em)) / (Ep[0, s, t, j] - Em[0, s, t, j])
        assert abs(fd - dE[0, s, t, j]) <= 2e-2 * abs(dE[0, s, t, j]) + 2e-2


def test_deterministic():
    rng = np.random.default_rng(4)
    A, pi = rand_model(rng, 15, sparse=True)
    E = (rng.random((1, 40, 3000, 15)) * 0.9 + 0.05).astype(np.float32)
    a = run_grad(A[None], pi[None], E)
    b = run_grad(A[None], pi[None], E)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_errors():
    A = torch.eye(70, device=DEV)[None]
    with pytest.raises(ValueError):                      # above hmm_grad_max_states()
        engine.loglik_grad(A, torch.full((1, 70), 1 / 70, device=DEV), torch.rand(1, 2, 8, 70, device=DEV))
    with pytest.raises(engine.EngineError):
        engine.loglik_grad(torch.eye(3)[None], torch.ones(1, 3) / 3, torch.rand(1, 2, 8, 3))
    with pytest.raises(ValueError):
        engine.loglik_grad(torch.eye(3, device=DEV)[None], torch.ones(1, 3, device=DEV) / 3,
                           torch.rand(1, 2, 8, 3, device=DEV), torch.ones(1, 3, device=DEV))


def test_full_size_expected_count_identities():
    """BASELINE config 3 size (b = 1024, L = 100 000, q = 15): size-independent properties of the
    Baum-Welch gradients.  With w = d loss / d loglik per sequence,
        sum_j dE[s,t,j] * E[s,t,j]   = w_s            (posteriors of a position sum to one)
        sum_ij dA[i,j] * A[i,j]      = sum_s w_s (L-1) (one transition per step)
        sum_j dpi[j] * pi[j]         = sum_s w_s
    and the log-likelihood equals hmm_forward's."""
    b, L, q = 1024, 100000, 15
    A = torch.as_tensor(np.asarray(params.intended_A15()), dtype=torch.float32, device=DEV)[None]
    pi = torch.full((1, q), 1.0 / q, device=DEV)
    g = torch.Generator(device=DEV).manual_seed(5)
    E = torch.rand((1, b, L, q), device=DEV, generator=g) * 0.9 + 0.05
    w = torch.rand((1, b), device=DEV, generator=g) + 0.5
    dA, dpi, dE, ll = engine.loglik_grad(A, pi, E, w)
    rows = (dE * E).sum(-1)                                               # (1, b, L)
    del dE
    assert float((rows / w[..., None] - 1).abs().max()) <= 2e-5
    wsum = float(w.double().sum())
    assert abs(float((dA.double() * A.double()).sum()) / (wsum * (L - 1)) - 1) <= 2e-5
    assert abs(float((dpi.double() * pi.double()).sum()) / wsum - 1) <= 2e-5
    _, ll2 = engine.forward(A, pi, E, want_log_alpha=False)
    assert torch.equal(ll, ll2)
    # absent edges aside, structural zeros of A receive finite gradients; present edges positive ones
    assert bool(torch.isfinite(dA).all()) and bool((dA[A > 0] > 0).all())


@pytest.mark.parametrize("q,b,L,sparse", [(17, 3, 60, False), (29, 4, 210, True), (48, 2, 90, True), (64, 3, 75, False)])
def test_mid_size_models_one_wave_per_sequence(q, b, L, sparse):
    """17..64 states (hmm_midq.inc): same oracles and tolerances as the scan path, incl. upstream
    weights of either sign, clamped emissions and two models in one call."""
    rng = np.random.default_rng(q + L)
    Ms = [rand_model(rng, q, sparse=sparse) for _ in range(2)]
    A = np.stack([m[0] for m in Ms]); pi = np.stack([m[1] for m in Ms])
    E = (rng.random((2, b, L, q)) * 0.9 + 0.05).astype(np.float32)
    E[0, :, ::7, 1] = 0.0                                       # clamped emissions: no gradient there
    w = rng.standard_normal((2, b)).astype(np.float32)
    dA, dpi, dE, ll = check(A, pi, E, w, "midq q=%d" % q)
    assert np.all(dE[0, :, ::7, 1] == 0.0)
    a = run_grad(A, pi, E, w)
    for x, y in zip(a, (dA, dpi, dE, ll)):
        assert np.array_equal(x, y)                              # deterministic


@pytest.mark.parametrize("name", ["grad_q5", "grad_q15"])
def test_reference_autograd_fixtures(golden, name):
    """The engine against gradients captured from the imported reference itself (autograd through
    its HmmCell.forward loop, tests/golden/make_golden_grad.py).  Absent edges of the gene model are
    held to the tolerance stated in the module docstring."""
    g = golden(name)
    dA, dpi, dE, ll = run_grad(g["A"][None], g["pi"][None], g["E"][None], g["w"][None])
    assert np.abs(ll[0] - g["loglik"]).max() <= 3e-4
    rU = textbook.loglik_grad(g["A"], g["pi"], g["E"], g["w"], clamp_adjoint=False)[0]
    tol = np.where(g["A"] > 0, 3e-4 * np.abs(g["dA"]).max(), np.abs(rU - g["dA"]) + 5e-3 * np.abs(g["dA"]).max())
    assert np.all(np.abs(dA[0] - g["dA"]) <= tol)
    assert np.abs(dpi[0] - g["dpi"]).max() <= 3e-4 * np.abs(g["dpi"]).max()
    assert np.abs(dE[0] - g["dE"]).max() <= 3e-4 * np.abs(g["dE"]).max()


def two_copy_A29():
    from hmm_layer_amd.gene_pred_hmm_transitioner import GenePredMultiHMMTransitioner
    tr = GenePredMultiHMMTransitioner(k=2, initial_exon_len=200, initial_intron_len=4500, initial_ir_len=10000)
    with torch.no_grad():
        return tr.make_A()[0].numpy().astype(np.float32)


@pytest.mark.parametrize("b,L,chunk", [(2, 700, 0), (3, 333, 16), (1, 2100, 64), (5, 97, 0)])
def test_two_copy_model_29_states_per_chunk(b, L, chunk):
    """The compiled 29-state topology is computed per chunk of the 32-state scan plan (hmm_postgrad_chunked.inc:
    k_pc_values + k_pc_llgrad) instead of by two whole-sequence sweeps; other models of that size, and what the
    device-side routing flags, keep the sweeps.  Both against the fp64 Baum-Welch oracle and each other."""
    rng = np.random.default_rng(29 * L + b)
    q = 29
    A = two_copy_A29()
    pi = (rng.random(q) + 0.1).astype(np.float32); pi /= pi.sum()
    E = (rng.random((1, b, L, q)) * 0.9 + 0.05).astype(np.float32)
    E[0, :, ::9, 20] = 0.0                                   # clamped emissions: no gradient there
    w = (rng.random((1, b)) + 0.5).astype(np.float32)
    got = {}
    with engine.option(engine.OPT_CHUNK, chunk):
        for how in (0, 1):
            with engine.option(engine.OPT_PGCHUNK, how):
                got[how] = check(A[None], pi[None], E, w, "q=29 how=%d" % how)
                assert engine.loglik_grad_serial_count((1, b, L, q)) == (b if how == 0 else 0)
    for s_, c_ in zip(got[0], got[1]):
        assert np.abs(s_ - c_).max() <= 1e-4 * np.abs(s_).max() + 1e-7
    assert np.all(got[1][2][0, :, ::9, 20] == 0.0)
    # two models in one call: the compiled topology (sparse reduce) and a dense 29-state model (dense MFMA reduce),
    # both per chunk of the 32-state scan plan
    Ad, pid = rand_model(rng, q)
    A2, pi2 = np.stack([A, Ad]), np.stack([pi, pid])
    E2 = np.concatenate([E, (rng.random((1, b, L, q)) * 0.9 + 0.05).astype(np.float32)])
    check(A2, pi2, E2, np.concatenate([w, w]), "q=29 two models")
    assert engine.loglik_grad_serial_count((2, b, L, q)) == 0


def test_two_copy_model_floor_decided_sequence_is_redone():
    """A stretch where only one state emits, which the topology leaves after one step: decided by the eps clamps,
    flagged by the certificate and recomputed by the whole-sequence sweeps (bit-identical to them)."""
    rng = np.random.default_rng(2)
    q, b, L = 29, 3, 900
    A = two_copy_A29()
    pi = np.full(q, 1 / q, dtype=np.float32)
    E = (rng.random((1, b, L, q)) * 0.9 + 0.05).astype(np.float32)
    only = int(np.argmax((A > 0).sum(-1) == 1))              # a state with a single successor
    E[0, 1, 400:420] = 0.0
    E[0, 1, 400:420, only] = 0.5
    with engine.option(engine.OPT_PGCHUNK, 1):
        auto = run_grad(A[None], pi[None], E)
        n = engine.loglik_grad_serial_count((1, b, L, q))
    with engine.option(engine.OPT_PGCHUNK, 0):
        serial = run_grad(A[None], pi[None], E)
    assert n >= 1
    assert np.array_equal(auto[2][0, 1], serial[2][0, 1])
    rA, rpi, rE = textbook.loglik_grad(A, pi, E[0], None)
    assert np.abs(auto[2][0] - rE).max() <= 2e-5 * np.abs(rE).max() + 1e-4 * np.abs(rE).max()
    assert np.abs(auto[0][0] - rA)[A > 0].max() <= 2e-4 * np.abs(rA).max()
