"""hmm_posterior_grad, the time-parallel path (csrc/hmm_postgrad_chunked.inc) against the serial sweeps of
the same entry point and against fp64 autograd (oracle/torch64.py).  HMM_OPT_PGCHUNK: 0 = serial sweeps
only, 1 = chunked where it pays, with the certificate deciding per sequence, 2 = chunked for every
sequence the shape allows.  Needs an MI355X."""
import numpy as np
import pytest
import torch

from hmm_layer_amd import engine
from oracle import params, torch64

from test_postgrad_gpu import check, dev, rand_model

pytestmark = pytest.mark.gpu


def grads(A, pi, E, G, mode, how):
    with engine.option(engine.OPT_PGCHUNK, how):
        return [t.cpu().numpy() for t in engine.posterior_grad(dev(A), dev(pi), dev(E), dev(G), mode=mode)]


@pytest.mark.parametrize("mode", [engine.POST_LOG, engine.POST_PROB])
@pytest.mark.parametrize("q,b,L", [(3, 2, 33), (7, 3, 130), (15, 2, 1000), (16, 3, 257), (15, 1, 4001), (2, 5, 64)])
def test_chunked_and_serial_against_fp64(q, b, L, mode):
    rng = np.random.default_rng(q * 100 + L)
    A, pi = rand_model(rng, q)
    if q == 15:
        A, pi = params.intended_A15().numpy().astype(np.float32), np.full(15, 1 / 15, dtype=np.float32)
    E = (rng.random((1, b, L, q)) * 0.9 + 0.05).astype(np.float32)
    G = rng.standard_normal((1, b, L, q)).astype(np.float32)
    got = {}
    for how in (0, 2):
        with engine.option(engine.OPT_PGCHUNK, how):
            got[how] = check(A[None], pi[None], E, G, mode, "q=%d how=%d" % (q, how))
    for s, c in zip(got[0], got[2]):
        assert np.abs(s - c).max() <= 1e-4 * np.abs(s).max() + 1e-7
    # deterministic
    again = grads(A[None], pi[None], E, G, mode, 2)
    for x, y in zip(got[2], again):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("chunk", [16, 48, 512])
def test_chunk_length_does_not_matter(chunk):
    rng = np.random.default_rng(chunk)
    q, b, L = 15, 2, 1300 if chunk == 512 else 700
    A, pi = params.intended_A15().numpy().astype(np.float32), np.full(15, 1 / 15, dtype=np.float32)
    E = (rng.random((1, b, L, q)) * 0.9 + 0.05).astype(np.float32)
    E[0, :, ::7, 9] = 0.0                                   # clamped emissions inside and at the edges of chunks
    G = rng.standard_normal(E.shape).astype(np.float32)
    with engine.option(engine.OPT_CHUNK, chunk), engine.option(engine.OPT_PGCHUNK, 2):
        dA, dpi, dE = check(A[None], pi[None], E, G, engine.POST_PROB, "chunk=%d" % chunk)
    assert np.all(dE[0, :, ::7, 9] == 0.0)


@pytest.mark.parametrize("mode", [engine.POST_LOG, engine.POST_PROB])
def test_rare_emissions_on_the_most_probable_path(mode):
    """A fifth of the constrained states' emissions are 1e-10 (far above eps: nothing is clamped): wherever the
    path has to take one, S_t is ~1e-10 and the adjoint vectors spread over ten orders of magnitude."""
    rng = np.random.default_rng(31)
    q, b, L = 15, 2, 1500
    A, pi = params.intended_A15().numpy().astype(np.float32), np.full(15, 1 / 15, dtype=np.float32)
    E = (rng.random((1, b, L, q)) * 0.9 + 0.05).astype(np.float32)
    rare = rng.random(E.shape) < 0.2
    rare[..., :6] = False
    E[rare] = 1e-10
    gam, _ = engine.posterior(dev(A)[None], dev(pi)[None], dev(E))
    G = -(gam == gam.amax(-1, keepdim=True)).float().cpu().numpy()           # labels = the most probable state
    c = grads(A[None], pi[None], E, G, mode, 1)
    with engine.option(engine.OPT_PGCHUNK, 1):
        assert engine.posterior_grad_serial_count((1, b, L, q)) == 0
    s = grads(A[None], pi[None], E, G, mode, 0)
    for x, y in zip(s, c):
        assert np.abs(x - y).max() <= 5e-5 * np.abs(x).max()
    rA, rpi, rE, _ = torch64.posterior_grad(A, pi, E[0], G[0], log=(mode == engine.POST_LOG))
    assert np.abs(c[0][0] - rA).max() <= 3e-4 * np.abs(rA).max()
    assert np.abs(c[2][0] - rE).max() <= 3e-4 * np.abs(rE).max()


def test_routing_per_model_and_per_sequence(golden):
    """Two models in one call: the intended gene model (chunked) and the as-shipped reducible one (its
    sequences go to the serial sweeps); one sequence of the first is decided by the eps clamps — twenty
    positions in a row emit from state 9 alone, which always leaves after one step — and is flagged by the
    certificate.  Every gradient matches fp64 autograd either way."""
    rng = np.random.default_rng(12)
    q, b, L = 15, 3, 900
    A = np.stack([params.intended_A15().numpy(), golden("transitioner")["A15_as_shipped"]]).astype(np.float32)
    pi = np.full((2, q), 1 / q, dtype=np.float32)
    E = (rng.random((2, b, L, q)) * 0.9 + 0.05).astype(np.float32)
    E[0, 1, 400:420] = 0.0
    E[0, 1, 400:420, 9] = 0.5
    G = rng.standard_normal(E.shape).astype(np.float32)
    auto = grads(A, pi, E, G, engine.POST_PROB, 1)
    with engine.option(engine.OPT_PGCHUNK, 1):
        assert engine.posterior_grad_serial_count((2, b, L, q)) == 1 + b
    serial = grads(A, pi, E, G, engine.POST_PROB, 0)
    with engine.option(engine.OPT_PGCHUNK, 0):
        assert engine.posterior_grad_serial_count((2, b, L, q)) == 2 * b
    for m in range(2):
        rA, rpi, rE, _ = torch64.posterior_grad(A[m], pi[m], E[m], G[m], log=False)
        for got in (auto, serial):
            for g, want in ((got[0][m], rA), (got[1][m], rpi), (got[2][m], rE)):
                assert np.abs(g - want).max() <= 3e-4 * np.abs(want).max() + 1e-6
    # the flagged sequence and the whole second model were computed by the serial sweeps: bit-identical dE
    assert np.array_equal(auto[2][0, 1], serial[2][0, 1])
    assert np.array_equal(auto[2][1], serial[2][1])
    assert not np.array_equal(auto[2][0, 0], serial[2][0, 0])


def test_log_mode_weight_on_negligible_states_is_flagged():
    """Log mode: the same inputs are served per chunk when the upstream gradient sits on states the posterior
    supports, and by the whole-sequence sweeps when it sits on states whose posterior is ~1e-16 (dead
    emissions), where eps-floor paths decide d log gamma."""
    rng = np.random.default_rng(31)
    q, b, L = 15, 2, 1500
    A, pi = params.intended_A15().numpy().astype(np.float32), np.full(15, 1 / 15, dtype=np.float32)
    E = (rng.random((1, b, L, q)) * 0.9 + 0.05).astype(np.float32)
    dead = rng.random(E.shape) < 0.2
    dead[..., :6] = False
    E[dead] = 0.0
    gam, _ = engine.posterior(dev(A)[None], dev(pi)[None], dev(E))
    gam = gam.cpu().numpy()
    G_ok = -(gam == gam.max(-1, keepdims=True)).astype(np.float32)          # labels = the most probable state
    G_bad = G_ok.copy()
    G_bad[0, 1][dead[0, 1]] = -1.0                                            # second sequence: also the dead entries
    for G, want in ((G_ok, 0), (G_bad, 1)):
        got = grads(A[None], pi[None], E, G, engine.POST_LOG, 1)
        with engine.option(engine.OPT_PGCHUNK, 1):
            assert engine.posterior_grad_serial_count((1, b, L, q)) == want
        rA, rpi, rE, _ = torch64.posterior_grad(A, pi, E[0], G[0], log=True)
        ok = np.isfinite(rE)
        assert np.abs(got[0][0] - rA).max() <= 3e-4 * np.abs(rA).max()
        assert np.abs(got[2][0] - rE)[ok].max() <= 3e-4 * np.abs(rE[ok]).max()


def test_training_shape():
    """The reference's own training shape (tests/parallel_rnn_forward.py:19-23: b = 32, L = 9 999), gene model,
    upstream gradient of a cross-entropy on log gamma against the most probable state."""
    rng = np.random.default_rng(4)
    q, b, L = 15, 32, 9999
    A, pi = params.intended_A15().numpy().astype(np.float32), np.full(15, 1 / 15, dtype=np.float32)
    E = (rng.random((1, b, L, q)) * 0.9 + 0.05).astype(np.float32)
    gam, _ = engine.posterior(dev(A)[None], dev(pi)[None], dev(E))
    G = -(gam == gam.amax(-1, keepdim=True)).float().cpu().numpy()
    c = grads(A[None], pi[None], E, G, engine.POST_LOG, 1)
    with engine.option(engine.OPT_PGCHUNK, 1):
        assert engine.posterior_grad_serial_count((1, b, L, q)) == 0
    s = grads(A[None], pi[None], E, G, engine.POST_LOG, 0)
    for x, y in zip(s, c):
        assert np.isfinite(y).all()
        assert np.abs(x - y).max() <= 1e-4 * np.abs(x).max()
    rA, rpi, rE, _ = torch64.posterior_grad(A, pi[None][0], E[0, :2], G[0, :2], log=True)
    assert np.abs(c[2][0, :2] - rE).max() <= 3e-4 * np.abs(rE).max()


def two_copy_A29():
    from hmm_layer_amd.gene_pred_hmm_transitioner import GenePredMultiHMMTransitioner
    tr = GenePredMultiHMMTransitioner(k=2, initial_exon_len=200, initial_intron_len=4500, initial_ir_len=10000)
    with torch.no_grad():
        return tr.make_A()[0].numpy().astype(np.float32)


@pytest.mark.parametrize("mode", [engine.POST_LOG, engine.POST_PROB])
@pytest.mark.parametrize("b,L,chunk", [(2, 700, 0), (3, 333, 16), (1, 2100, 64)])
def test_two_copy_model_29_states(b, L, chunk, mode):
    """The 29-state two-copy gene model (hmm_layer/gene_pred_hmm_transitioner.py:263-308): rows of 32 lanes, two
    chunks per wave, boundaries through the 32 x 32 chunk operators of hmm_scan32.inc."""
    rng = np.random.default_rng(29 * L + b)
    q = 29
    A = two_copy_A29()
    pi = (rng.random(q) + 0.1).astype(np.float32); pi /= pi.sum()
    E = (rng.random((1, b, L, q)) * 0.9 + 0.05).astype(np.float32)
    gam, _ = engine.posterior(dev(A)[None], dev(pi)[None], dev(E))
    G = -(gam == gam.amax(-1, keepdim=True)).float().cpu().numpy() if mode == engine.POST_LOG else \
        rng.standard_normal(E.shape).astype(np.float32)
    got = {}
    with engine.option(engine.OPT_CHUNK, chunk):
        for how in (0, 1):
            with engine.option(engine.OPT_PGCHUNK, how):
                got[how] = check(A[None], pi[None], E, G, mode, "q=29 how=%d" % how)
                assert engine.posterior_grad_serial_count((1, b, L, q)) == (b if how == 0 else 0)
    for s_, c_ in zip(got[0], got[1]):
        assert np.abs(s_ - c_).max() <= 1e-4 * np.abs(s_).max() + 1e-7
    # other primitive 29-state models take the same path with the dense MFMA reduce; reducible ones are handed back
    # to the whole-sequence sweeps on the device
    Ad, pid = rand_model(rng, q)
    check(Ad[None], pid[None], E, G, mode, "q=29 dense")
    assert engine.posterior_grad_serial_count((1, b, L, q)) == 0
    Ar = np.triu(Ad)
    Ar /= Ar.sum(-1, keepdims=True)
    check(Ar[None], pid[None], E, G, mode, "q=29 upper triangular")
    assert engine.posterior_grad_serial_count((1, b, L, q)) == b


def test_randomised_routing_sweep():
    """tests/postgrad_sweep.py: random models (gene, dense, sparse / reducible, degenerate), shapes, chunk lengths,
    dead / rare / tiny emissions, dense and label-like upstream gradients, both modes — the shipped routing
    against the whole-sequence sweeps, 2e-4 of each gradient's largest entry."""
    import postgrad_sweep
    assert postgrad_sweep.run(150, 20261004, verbose=False) == 0
