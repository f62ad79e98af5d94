"""The chunked (time-parallel) scan for 17..32 states: the 29-state two-copy gene model
(GenePredMultiHMMTransitioner(k=2), hmm_layer/gene_pred_hmm_transitioner.py:263-308) through
k_reduce_sparse<TopoGene29> -> k32_scan -> k32_forward / k32_backward, against the serial fp64 oracle at
the tolerances of tests/test_engine_gpu.py.  Other 17..32-state models, and sequences the floor-transition
certificate flags, fall to the one-wave-per-sequence kernels — decided on the device."""
import numpy as np
import pytest
import torch

from hmm_layer_amd import engine
from hmm_layer_amd.gene_pred_hmm_transitioner import GenePredMultiHMMTransitioner
from oracle import build as obuild
from oracle import textbook

from test_engine_gpu import check_all, dev, rand_model

pytestmark = pytest.mark.gpu


def gene29(exon=200, intron=4500, ir=10000):
    tr = GenePredMultiHMMTransitioner(k=2, initial_exon_len=exon, initial_intron_len=intron, initial_ir_len=ir)
    with torch.no_grad():
        return tr.make_A()[0].numpy().copy(), tr.make_initial_distribution().reshape(-1).numpy().copy()


def post(A, pi, E, mode=engine.POST_PROB):
    out, ll = engine.posterior(dev(A).reshape(-1, A.shape[-1], A.shape[-1]), dev(pi), dev(E), mode=mode)
    torch.cuda.synchronize()
    return out.cpu().numpy(), ll.cpu().numpy()


def check(A, pi, E, tag, expect_serial=None):
    """E (b,L,q): posterior (three modes), log-likelihood, loglik-only entry point vs the fp64 oracle."""
    g64, ll64 = obuild.posterior(A, pi, E)
    b, L, q = E.shape
    for mode in (engine.POST_PROB, engine.POST_LOG, engine.POST_LOG_NO_LL):
        out, ll = post(A, pi, E[None], mode)
        if expect_serial is not None:
            assert engine.exact_count(engine.OP_POSTERIOR, (1, b, L, q)) == expect_serial, tag
        got = out[0]
        tol = 2e-5
        if mode == engine.POST_LOG_NO_LL:
            got = got - ll[0][:, None, None]
            tol = 2e-5 + 2.4e-7 * np.abs(ll64).max()
        if mode != engine.POST_PROB:
            got = np.exp(got)
        assert np.isfinite(got).all(), (tag, mode)
        assert np.abs(got - g64).max() <= tol, (tag, mode, np.abs(got - g64).max())
        assert np.all(np.abs(ll[0] - ll64) <= 1e-6 * np.abs(ll64) + 2e-4), (tag, mode)
    _, ll2 = engine.forward(dev(A)[None], dev(pi), dev(E[None]), want_log_alpha=False)
    assert np.all(np.abs(ll2.cpu().numpy()[0] - ll64) <= 1e-6 * np.abs(ll64) + 2e-4), tag
    check_all(A, pi, E, tag)                    # + log alpha (k32_forward<true>), log beta (k32_backward<3>)
    if expect_serial is not None:
        assert engine.exact_count(engine.OP_BACKWARD, (1, b, L, q)) == expect_serial, tag     # check_all's last call
    return out, ll


@pytest.mark.parametrize("b,L", [(1, 1), (2, 3), (1, 16), (3, 17), (5, 100), (2, 1031), (37, 333), (4, 6000), (1, 40000)])
def test_two_copy_gene_model_ragged_shapes(b, L):
    rng = np.random.default_rng(29 * b + L)
    A, pi = gene29()
    assert A.shape == (29, 29) and (A > 0).sum() == 45
    E = (rng.random((b, L, 29)) * 0.9 + 0.05).astype(np.float32) / 4096
    dead = rng.random(E.shape) < 0.4
    dead[..., :13] = False                      # IR, introns and exons always keep mass alive
    E[dead] = 0.0
    check(A, pi, E, "gene29 b=%d L=%d" % (b, L), expect_serial=0)


def test_forced_chunk_lengths_and_determinism():
    rng = np.random.default_rng(5)
    A, pi = gene29(50, 300, 900)
    E = (rng.random((6, 2500, 29)) * 0.9 + 0.05).astype(np.float32)
    ref = None
    for chunk in (16, 48, 128, 512, 0):
        with engine.option(engine.OPT_CHUNK, chunk):
            out, ll = check(A, pi, E, "chunk %d" % chunk, expect_serial=0)
            out2, ll2 = post(A, pi, E[None], engine.POST_LOG_NO_LL)
            assert np.array_equal(out, out2) and np.array_equal(ll, ll2)
        if ref is not None:
            assert np.abs(out - ref).max() <= 2e-3          # log gamma + loglik across chunkings
        ref = out


def test_three_models_in_one_call_sparse_dense_and_reducible(golden):
    """The compiled topology (sparse reduce), a primitive model outside it (dense MFMA reduce) and a reducible one
    (serial kernels) in ONE call, with one clamp-decided sequence in the first model."""
    rng = np.random.default_rng(7)
    q, b, L = 29, 4, 700
    A0, pi0 = gene29()
    A1, pi1 = rand_model(rng, q, dense=False)            # outside the compiled topology, primitive
    A2 = golden("transitioner")["A29_as_shipped"]         # inside it, but reducible (D1)
    E = (rng.random((3, b, L, q)) * 0.9 + 0.05).astype(np.float32)
    E[0, 2, 300:330] = 0.0                                # thirty positions emit from a START state alone:
    E[0, 2, 300:330, 13] = 0.5                            # decided by the eps clamps
    A = np.stack([A0, A1, A2])
    pi = np.stack([pi0, pi1, np.full(q, 1 / q, dtype=np.float32)])
    out, ll = engine.posterior(dev(A), dev(pi), dev(E))
    assert engine.exact_count(engine.OP_POSTERIOR, (3, b, L, q)) == 1 + b       # the flagged sequence + the reducible model
    out, ll = out.cpu().numpy(), ll.cpu().numpy()
    for m in range(3):
        g64, ll64 = textbook.posterior(A[m], pi[m], E[m])
        assert np.abs(out[m] - g64).max() <= 2e-5, m
        assert np.all(np.abs(ll[m] - ll64) <= 1e-6 * np.abs(ll64) + 2e-4), m
    with engine.option(engine.OPT_EXACT, engine.EXACT_OFF):
        scan, _ = engine.posterior(dev(A[:1]), dev(pi[:1]), dev(E[:1]))
    scan = scan.cpu().numpy()
    assert np.array_equal(scan[0][[0, 1, 3]], out[0][[0, 1, 3]])              # unflagged sequences: untouched
    assert np.abs(scan[0][2] - out[0][2]).max() > 1e-4                       # what the routing repaired
    _, ll2 = engine.forward(dev(A), dev(pi), dev(E), want_log_alpha=False)
    ll2 = ll2.cpu().numpy()
    ok = np.ones((3, b), bool)
    ok[0, 2] = False                                       # (the log-likelihood entry point routes per model only)
    assert np.all(np.abs(ll2 - ll)[ok] <= 1e-6 * np.abs(ll)[ok] + 2e-4)


def test_two_copy_model_at_the_reference_test_size():
    """b = 32 x L = 9999 (tests/parallel_rnn_forward.py:19-23) with the two-copy model: whole-output
    properties and the oracle on sampled sequences."""
    torch.manual_seed(3)
    A, pi = gene29()
    b, L, q = 32, 9999, 29
    E = torch.rand((1, b, L, q), device="cuda:0") * 0.9 + 0.05
    out, ll = engine.posterior(dev(A)[None], dev(pi), E)
    assert engine.exact_count(engine.OP_POSTERIOR, (1, b, L, q)) == 0
    assert bool(torch.isfinite(out).all()) and float((out.sum(-1) - 1).abs().max()) <= 2e-5
    out2, ll2 = engine.posterior(dev(A)[None], dev(pi), E)
    assert torch.equal(out, out2) and torch.equal(ll, ll2)
    idx = [0, 15, 31]
    g64, ll64 = obuild.posterior(A, pi, E[0, idx].cpu().numpy())
    assert np.abs(out[0, idx].cpu().numpy() - g64).max() <= 2e-5
    assert np.all(np.abs(ll[0, idx].cpu().numpy() - ll64) <= 1e-6 * np.abs(ll64))


@pytest.mark.parametrize("q", [17, 24, 29, 32])
def test_dense_models_take_the_chunked_scan(q):
    """Any primitive model of 17..32 states — a learned dense A, other topologies — runs the three phases with the
    dense MFMA reduce (k32_reduce_dense: the operator as 2 x 2 tiles, eight block products per step); nothing is
    served by the one-wave-per-sequence kernels.  Ragged shapes, forced chunk lengths, every output, against the
    serial fp64 oracle; the forced-serial result of the same input as a second opinion."""
    rng = np.random.default_rng(400 + q)
    for dense in (True, False):
        A, pi = rand_model(rng, q, dense=dense)
        for (b, L), chunk in (((3, 1), 0), ((2, 17), 16), ((5, 333), 48), ((3, 2600), 0), ((2, 5000), 128)):
            E = (rng.random((b, L, q)) * 0.9 + 0.05).astype(np.float32)
            if dense:
                E[rng.random(E.shape) < 0.2] *= 1e-6          # wide dynamic range: the exponents earn their keep
            with engine.option(engine.OPT_CHUNK, chunk):
                check(A, pi, E, "dense q=%d b=%d L=%d chunk=%d" % (q, b, L, chunk), expect_serial=0)
                out, ll = post(A, pi, E[None])
                with engine.option(engine.OPT_EXACT, engine.EXACT_ALWAYS):
                    ser, lls = post(A, pi, E[None])
                assert engine.exact_count(engine.OP_POSTERIOR, (1, b, L, q)) == b
            assert np.abs(out[0] - ser[0]).max() <= 4e-6 and np.allclose(ll, lls, rtol=1e-7, atol=1e-5)


def test_dense_24_state_model_at_a_training_shape_and_its_gradients():
    """b = 32 x L = 9999 with a dense 24-state model: chunked, deterministic, and both gradients (per chunk of the same
    plan) against the fp64 references on a shorter shape."""
    from oracle import torch64
    torch.manual_seed(5)
    rng = np.random.default_rng(24)
    q = 24
    A, pi = rand_model(rng, q)
    E = torch.rand((1, 32, 9999, q), device="cuda:0") * 0.9 + 0.05
    out, ll = engine.posterior(dev(A)[None], dev(pi), E)
    assert engine.exact_count(engine.OP_POSTERIOR, (1, 32, 9999, q)) == 0
    out2, ll2 = engine.posterior(dev(A)[None], dev(pi), E)
    assert torch.equal(out, out2) and torch.equal(ll, ll2)
    idx = [0, 31]
    g64, ll64 = obuild.posterior(A, pi, E[0, idx].cpu().numpy())
    assert np.abs(out[0, idx].cpu().numpy() - g64).max() <= 2e-5
    assert np.all(np.abs(ll[0, idx].cpu().numpy() - ll64) <= 1e-6 * np.abs(ll64))
    b, L = 6, 400
    En = (rng.random((b, L, q)) * 0.9 + 0.05).astype(np.float32)
    w = (rng.random(b) + 0.5).astype(np.float32)
    dA, dpi, dE, _ = engine.loglik_grad(dev(A)[None], dev(pi)[None], dev(En)[None], dev(w)[None])
    rA, rpi, rE = textbook.loglik_grad(A, pi, En, w)
    assert np.abs(dA.cpu().numpy()[0] - rA).max() <= 3e-4 * np.abs(rA).max()
    assert np.abs(dE.cpu().numpy()[0] - rE).max() <= 3e-4 * np.abs(rE).max()
    G = rng.standard_normal((b, L, q)).astype(np.float32)
    pA, ppi, pE = engine.posterior_grad(dev(A)[None], dev(pi)[None], dev(En)[None], dev(G)[None], mode=engine.POST_LOG)
    qA, qpi, qE, _ = torch64.posterior_grad(A, pi, En, G, log=True)
    assert np.abs(pA.cpu().numpy()[0] - qA).max() <= 3e-4 * np.abs(qA).max()
    assert np.abs(pE.cpu().numpy()[0] - qE).max() <= 3e-4 * np.abs(qE).max()
