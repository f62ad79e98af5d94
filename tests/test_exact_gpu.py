"""The eps-clamp regime on the GPU: inputs on which the cell's clamp of the predicted state MIXTURE
(hmm_layer/MsaHmmCell.py:87-88) decides the answer, which no scan over chunk operators reproduces.
The engine routes them, on the device, to serial kernels with the cell's exact step semantics:

  per model     support of A not primitive (reducible, periodic, states without incoming edges,
                all-zero rows as in the reference's as-shipped matrices, A = I)
  per sequence  (hmm_posterior) the posterior mass of clamp-born paths (psi, hmm_engine.hip::backward_body) above 2e-6: serial
                recomputation in windows around the chunks that carry it, whole sequences as the fallback

Every case is held to the serial fp64 oracle with the reference's clamps (oracle/textbook.py,
oracle/hmm_oracle.c) at the suite's normal tolerances (tests/test_engine_gpu.py docstring).
"""
import numpy as np
import pytest
import torch

from hmm_layer_amd import engine
from oracle import build as obuild
from oracle import params, textbook

from test_engine_gpu import assert_log_close_in_probability_space, check_all, dev, rand_model, run_post
from test_engine_gpu import DEV as DEV_

pytestmark = pytest.mark.gpu


def n_exact(op, shape):
    return engine.exact_count(op, shape)


@pytest.mark.parametrize("name", ["A15_as_shipped", "A7_as_shipped", "A15_single_as_shipped"])
def test_reference_as_shipped_matrices(golden, name):
    """The reference's transition matrices exactly as its constructors produce them (defect D1,
    hmm_layer/Transitioner.py:366-367: zero logits count as absent edges, so rows 7-14 of the
    15-state matrix are all zero and states 0-3 absorb).  Reducible -> serial kernels, every
    entry point, exact cell semantics."""
    A = golden("transitioner")[name]
    q = A.shape[0]
    rng = np.random.default_rng(q)
    pi = np.full(q, 1 / q, dtype=np.float32)
    for scale in (1.0, 1 / 4096):                       # generic and gene-model emission magnitudes
        E = (rng.random((5, 700, q)) * 0.9 + 0.05).astype(np.float32) * scale
        check_all(A, pi, E, name)
        assert n_exact(engine.OP_BACKWARD, (1, 5, 700, q)) == 5        # check_all's last call


@pytest.mark.parametrize("q", [1, 2, 5, 15, 16])
def test_identity_and_cycle(q):
    """A = I (every state absorbs: forward and backward evidence contradict each other by far more
    than 1/eps) and a pure cycle (irreducible but periodic, A^q = I)."""
    rng = np.random.default_rng(100 + q)
    pi = rng.random(q).astype(np.float32) + 0.1
    pi /= pi.sum()
    E = (rng.random((4, 900, q)) * 0.9 + 0.05).astype(np.float32)
    check_all(np.eye(q, dtype=np.float32), pi, E, "identity q=%d" % q)
    if q > 1:
        assert n_exact(engine.OP_BACKWARD, (1, 4, 900, q)) == 4
        check_all(np.roll(np.eye(q, dtype=np.float32), 1, axis=1), pi, E, "cycle q=%d" % q)
        assert n_exact(engine.OP_BACKWARD, (1, 4, 900, q)) == 4


def test_primitive_models_stay_on_the_scan():
    """The routing does not fire on the models the scan is for: the intended gene matrices, dense
    matrices, a cycle with one self loop (primitive with the longest possible index)."""
    rng = np.random.default_rng(5)
    cyc = np.roll(np.eye(16, dtype=np.float32), 1, axis=1)
    cyc[0, 0] = 0.5; cyc[0, 1] = 0.5
    for A in (params.intended_A15().numpy(), rand_model(rng, 16)[0], rand_model(rng, 3)[0], cyc):
        q = A.shape[0]
        pi = np.full(q, 1 / q, dtype=np.float32)
        E = (rng.random((3, 400, q)) * 0.9 + 0.05).astype(np.float32)
        check_all(A, pi, E, "primitive q=%d" % q)
        gam, ll = run_post(A, pi, E[None])
        assert n_exact(engine.OP_POSTERIOR, (1, 3, 400, q)) == 0


def test_deleted_edge_leaves_a_state_without_incoming_edges():
    """k = 3 models in one call, the third the gene topology with edge E1 -> EI1 removed: state 9
    then lives through the eps clamp only (the case round 1 had to take out of the suite)."""
    rng = np.random.default_rng(72)
    q, b, L = 15, 6, 900
    A0 = params.intended_A15().numpy()
    A1, _ = rand_model(rng, q)
    A2 = A0.copy()
    A2[5, 9] = 0.0
    A2[5] /= A2[5].sum()
    A = np.stack([A0, A1, A2])
    pi = np.stack([rand_model(rng, q)[1] for _ in range(3)])
    E = (rng.random((3, b, L, q)) * 0.9 + 0.05).astype(np.float32)
    out, ll = engine.posterior(dev(A), dev(pi), dev(E))
    assert n_exact(engine.OP_POSTERIOR, (3, b, L, q)) == b          # exactly the third model's sequences
    out, ll = out.cpu().numpy(), ll.cpu().numpy()
    la, ll2 = engine.forward(dev(A), dev(pi), dev(E))
    lb = engine.backward(dev(A), dev(E)).cpu().numpy()
    la = la.cpu().numpy()
    for m in range(3):
        g64, ll64 = textbook.posterior(A[m], pi[m], E[m])
        assert np.abs(out[m] - g64).max() <= 2e-5, m
        assert np.all(np.abs(ll[m] - ll64) <= 1e-6 * np.abs(ll64) + 2e-4), m
        la64, _ = textbook.log_alpha(A[m], pi[m], E[m])
        lb64 = textbook.log_beta(A[m], E[m])
        k = la64 > -30
        assert np.all(np.abs(la[m] - la64)[k] <= 3e-4 + 2e-7 * np.abs(la64[k])), m
        k = lb64 > -30
        assert np.all(np.abs(lb[m] - lb64)[k] <= 3e-4 + 2e-7 * np.abs(lb64[k])), m
    assert np.array_equal(ll2.cpu().numpy(), ll)


def test_impossible_stretches_are_recomputed_serially():
    """Primitive model, but a quarter of all emission entries are zero in half of the batch: whole
    stretches are impossible under the model and every path survives through the 1e-16 clamps.  The
    backward kernel's certificate flags exactly such sequences; they are recomputed by the serial
    kernels and then match the serial oracle like everything else.  The other half of the batch is
    left alone (bitwise the scan's result)."""
    rng = np.random.default_rng(12)
    A = params.intended_A15().numpy()
    pi = np.full(15, 1 / 15, dtype=np.float32)
    b, L = 12, 1500
    E = (rng.random((b, L, 15)) * 0.9 + 0.05).astype(np.float32)
    hard = np.arange(b) % 2 == 1
    Eh = E[hard]
    Eh[rng.random(Eh.shape) < 0.25] = 0.0
    E[hard] = Eh
    g64, ll64 = obuild.posterior(A, pi, E)
    for mode in (engine.POST_PROB, engine.POST_LOG, engine.POST_LOG_NO_LL):
        out, ll = run_post(A, pi, E[None], mode)
        nx = n_exact(engine.OP_POSTERIOR, (1, b, L, 15))
        assert hard.sum() * 0.5 <= nx <= hard.sum(), nx           # flagged: (most of) the hard ones, none of the easy ones
        got = out[0]
        if mode == engine.POST_LOG_NO_LL:
            got = got - ll[0][:, None, None]
        if mode != engine.POST_PROB:
            got = np.exp(got)
        assert np.isfinite(got).all()
        # log gamma + loglik is an fp32 number of size |loglik|: its ulp bounds what the sum can resolve
        tol = 2e-5 if mode != engine.POST_LOG_NO_LL else 2e-5 + 2.4e-7 * np.abs(ll64).max()
        assert np.abs(got - g64).max() <= tol, mode
        assert np.all(np.abs(ll[0] - ll64) <= 1e-6 * np.abs(ll64) + 2e-4), mode
    with engine.option(engine.OPT_EXACT, engine.EXACT_OFF):
        scan, _ = run_post(A, pi, E[None], engine.POST_LOG_NO_LL)
    assert np.array_equal(scan[0][~hard], out[0][~hard])
    assert np.abs(np.exp(scan[0] - ll[0][:, None, None]) - g64)[hard].max() > 1e-4     # what the routing repaired
    # the other entry points carry their own certificates (hmm_forward: the clamp-born part of alpha_hat weighed with
    # the chunk scan's suffix vectors; hmm_backward: the mirror image; hmm_loglik_grad: psi's forward half) and route
    # the same kind of sequence: log-likelihood, log alpha, log beta — every component, in probability space — and
    # the gradients at the suite's normal tolerances
    la, ll2 = engine.forward(dev(A)[None], dev(pi), dev(E[None]))
    nf = n_exact(engine.OP_FORWARD, (1, b, L, 15))
    _, ll3 = engine.forward(dev(A)[None], dev(pi), dev(E[None]), want_log_alpha=False)
    nl = n_exact(engine.OP_LOGLIK, (1, b, L, 15))
    lb = engine.backward(dev(A)[None], dev(E[None]))
    nb = n_exact(engine.OP_BACKWARD, (1, b, L, 15))
    for cnt in (nf, nl, nb):
        assert hard.sum() * 0.5 <= cnt <= hard.sum(), (nf, nl, nb)
    la64, _ = textbook.log_alpha(A, pi, E)
    lb64 = textbook.log_beta(A, E)
    assert_log_close_in_probability_space(la.cpu().numpy()[0], la64, "log alpha")
    assert_log_close_in_probability_space(lb.cpu().numpy()[0], lb64, "log beta")
    for v in (ll2, ll3):
        assert np.all(np.abs(v.cpu().numpy()[0] - ll64) <= 1e-6 * np.abs(ll64) + 2e-4)
    w = (rng.random(b) + 0.5).astype(np.float32)
    dA, dpi, dE, llg = engine.loglik_grad(dev(A)[None], dev(pi)[None], dev(E)[None], dev(w)[None])
    ng = engine.loglik_grad_serial_count((1, b, L, 15))
    assert hard.sum() * 0.5 <= ng <= hard.sum(), ng
    rA, rpi, rE = textbook.loglik_grad(A, pi, E, w)
    m = A > 0
    assert np.abs(dA.cpu().numpy()[0] - rA)[m].max() <= 3e-4 * np.abs(rA).max()
    assert np.abs(dE.cpu().numpy()[0] - rE).max() <= 3e-4 * np.abs(rE).max()
    assert np.abs(dpi.cpu().numpy()[0] - rpi).max() <= 3e-4 * np.abs(rpi).max()
    assert np.all(np.abs(llg.cpu().numpy()[0] - ll64) <= 1e-6 * np.abs(ll64) + 2e-4)


def test_serial_kernels_agree_with_the_scan_where_both_apply():
    """EXACT_ALWAYS forces every sequence through the serial kernels: on benign inputs they and the
    chunked scan are two implementations of the same recursion."""
    rng = np.random.default_rng(31)
    for q, b, L in ((15, 37, 3001), (7, 3, 513), (16, 17, 100), (3, 1, 1), (1, 2, 40)):
        A, pi = rand_model(rng, q)
        if q == 15:
            A = params.intended_A15().numpy()
        E = (rng.random((b, L, q)) * 0.9 + 0.05).astype(np.float32)
        res = {}
        for mode in (engine.EXACT_OFF, engine.EXACT_ALWAYS):
            with engine.option(engine.OPT_EXACT, mode):
                gam, ll = run_post(A, pi, E[None])
                la, ll2 = engine.forward(dev(A)[None], dev(pi), dev(E[None]))
                _, ll3 = engine.forward(dev(A)[None], dev(pi), dev(E[None]), want_log_alpha=False)
                lb = engine.backward(dev(A)[None], dev(E[None]))
                res[mode] = (gam, ll, la.cpu().numpy(), lb.cpu().numpy())
                assert np.array_equal(ll2.cpu().numpy(), ll) and np.array_equal(ll3.cpu().numpy(), ll)
        a, c = res[engine.EXACT_OFF], res[engine.EXACT_ALWAYS]
        assert np.abs(a[0] - c[0]).max() <= 2e-6
        assert np.all(np.abs(a[1] - c[1]) <= 1e-7 * np.abs(c[1]) + 1e-5)
        assert np.abs(a[2] - c[2]).max() <= 2e-3 and np.abs(a[3] - c[3]).max() <= 2e-3
        with engine.option(engine.OPT_EXACT, engine.EXACT_ALWAYS):
            check_all(A, pi, E, "always q=%d" % q)


def test_serial_kernels_are_deterministic_and_cover_long_sequences():
    rng = np.random.default_rng(41)
    A = np.eye(15, dtype=np.float32) * 0.9
    A[np.arange(15), (np.arange(15) + 5) % 15] = 0.1            # three closed classes of five states
    pi = np.full(15, 1 / 15, dtype=np.float32)
    E = dev((rng.random((1, 20, 30000, 15)) * 0.9 + 0.05).astype(np.float32))
    a1, l1 = engine.posterior(dev(A)[None], dev(pi), E)
    a2, l2 = engine.posterior(dev(A)[None], dev(pi), E)
    assert torch.equal(a1, a2) and torch.equal(l1, l2)
    assert n_exact(engine.OP_POSTERIOR, (1, 20, 30000, 15)) == 20
    g64, ll64 = obuild.posterior(A, pi, E[0, :3].cpu().numpy())
    assert np.abs(a1[0, :3].cpu().numpy() - g64).max() <= 2e-5
    assert np.all(np.abs(l1[0, :3].cpu().numpy() - ll64) <= 1e-6 * np.abs(ll64))


def test_gradients_of_routed_models():
    """hmm_loglik_grad routes per model; the as-shipped 15-state matrix and A = I against the fp64
    Baum-Welch oracle with the cell's clamps."""
    rng = np.random.default_rng(51)
    from conftest import load_golden
    for A in (load_golden("transitioner")["A15_as_shipped"], np.eye(6, dtype=np.float32)):
        q = A.shape[0]
        pi = np.full(q, 1 / q, dtype=np.float32)
        b, L = 7, 300
        E = (rng.random((b, L, q)) * 0.9 + 0.05).astype(np.float32)
        w = (rng.random(b) + 0.5).astype(np.float32)
        dA, dpi, dE, ll = engine.loglik_grad(dev(A)[None], dev(pi)[None], dev(E)[None], dev(w)[None])
        rA, rpi, rE = textbook.loglik_grad(A, pi, E, w)
        m = A > 0
        assert np.abs(dA.cpu().numpy()[0] - rA)[m].max() <= 3e-4 * np.abs(rA).max()
        assert np.abs(dE.cpu().numpy()[0] - rE).max() <= 3e-4 * np.abs(rE).max()
        assert np.abs(dpi.cpu().numpy()[0] - rpi).max() <= 3e-4 * np.abs(rpi).max()
        assert np.all(np.abs(ll.cpu().numpy()[0] - textbook.loglik(A, pi, E)) <= 2e-4 + 1e-6 * L)


def test_one_sequence_per_wave_layout_of_the_serial_plan():
    """Sequences longer than 2 GB / 16 make the serial plan give every wave a single sequence (32-bit in-wave
    offsets); EXACT_ALWAYS_NARROW forces that layout at a testable size."""
    rng = np.random.default_rng(61)
    A, pi = rand_model(rng, 15)
    E = (rng.random((5, 333, 15)) * 0.9 + 0.05).astype(np.float32)
    with engine.option(engine.OPT_EXACT, engine.EXACT_ALWAYS_NARROW):
        check_all(A, pi, E, "narrow")
        dA, dpi, dE, ll = engine.loglik_grad(dev(A)[None], dev(pi)[None], dev(E)[None])
    rA, rpi, rE = textbook.loglik_grad(A, pi, E)
    assert np.abs(dA.cpu().numpy()[0] - rA).max() <= 3e-4 * np.abs(rA).max()
    assert np.abs(dE.cpu().numpy()[0] - rE).max() <= 3e-4 * np.abs(rE).max()


# ---------------------------------------------------------------- windows: cost proportional to the flagged chunks

def _detail(shape):
    return engine.exact_detail(shape)


def test_local_impossible_stretch_is_recomputed_in_a_window():
    """One short stretch of observations that every path survives only through the clamps, in the middle of long
    sequences: the sequences are flagged, but only a window of chunks around the stretch is walked serially (and
    accepted: the recursion forgets the stretch within the margin); the rest of each sequence and the other
    sequences keep the scan's values bit for bit.  The window's log-likelihood replaces the chunk scan's for its
    span."""
    rng = np.random.default_rng(77)
    A = params.intended_A15().numpy()
    pi = np.full(15, 1 / 15, dtype=np.float32)
    b, L = 9, 24000
    E = (rng.random((b, L, 15)) * 0.9 + 0.05).astype(np.float32)
    hard = {2: 5000, 5: 17003, 7: 23990}                 # sequence -> start of its stretch (one of them near the end)
    for s, t0 in hard.items():
        E[s, t0:t0 + 4, :] = 0.0
        E[s, t0:t0 + 4, 9] = 0.5                         # four positions in a row emitted by state 9 alone (EI1: leaves after one step)
    E[5, 300:303, :] = 0.0                               # a second stretch in the same sequence: a second window
    E[5, 300:303, 12] = 0.7                              # (IE1 alone)
    g64, ll64 = obuild.posterior(A, pi, E)
    for chunk in (0, 64):
        with engine.option(engine.OPT_CHUNK, chunk):
            T = engine.chunk_len(1, b, L, 15)
            for mode in (engine.POST_PROB, engine.POST_LOG, engine.POST_LOG_NO_LL):
                out, ll = run_post(A, pi, E[None], mode)
                det = _detail((1, b, L, 15))
                assert det["routed"] == 3 and det["window_sequences"] == 3 and det["whole"] == 0, det
                assert 4 <= det["windows"] <= 8, det              # one or two per stretch (before / after the flagged chunk)
                assert det["window_chunks"] * T <= 4 * 6000, det              # a few thousand positions, not 3 x 24 000
                got = out[0]
                if mode == engine.POST_LOG_NO_LL:
                    got = got - ll[0][:, None, None]
                if mode != engine.POST_PROB:
                    got = np.exp(got)
                tol = 2e-5 if mode != engine.POST_LOG_NO_LL else 2e-5 + 2.4e-7 * np.abs(ll64).max()
                assert np.abs(got - g64).max() <= tol, (chunk, mode, np.abs(got - g64).max())
                assert np.all(np.abs(ll[0] - ll64) <= 1e-6 * np.abs(ll64) + 2e-4), (chunk, mode)
            # the log-likelihood alone: the same windows, forward half only
            _, ll1 = engine.forward(dev(A)[None], dev(pi), dev(E[None]), want_log_alpha=False)
            assert engine.exact_count(engine.OP_LOGLIK, (1, b, L, 15)) == 3
            assert np.all(np.abs(ll1.cpu().numpy()[0] - ll64) <= 1e-6 * np.abs(ll64) + 2e-4), chunk
            with engine.option(engine.OPT_EXACT, engine.EXACT_OFF):
                scan, sl = run_post(A, pi, E[None], engine.POST_PROB)
            out, ll = run_post(A, pi, E[None], engine.POST_PROB)
            assert np.allclose(ll1.cpu().numpy()[0], ll[0], rtol=1e-9, atol=0)
            # the gradients of the log-likelihood: windows as well (dE of their positions, their share of dA)
            if chunk == 64:
                w = (rng.random(b) + 0.5).astype(np.float32)
                dA, dpi, dE, llg = engine.loglik_grad(dev(A)[None], dev(pi)[None], dev(E)[None], dev(w)[None])
                dg = _detail((1, b, L, 15))
                assert dg["routed"] == 3 and dg["whole"] == 0 and 4 <= dg["windows"] <= 8, dg
                rA, rpi, rE = textbook.loglik_grad(A, pi, E, w)
                m = A > 0
                assert np.abs(dA.cpu().numpy()[0] - rA)[m].max() <= 3e-4 * np.abs(rA).max()
                assert np.abs(dE.cpu().numpy()[0] - rE).max() <= 3e-4 * np.abs(rE).max()
                assert np.abs(dpi.cpu().numpy()[0] - rpi).max() <= 3e-4 * np.abs(rpi).max()
                assert np.all(np.abs(llg.cpu().numpy()[0] - ll64) <= 1e-6 * np.abs(ll64) + 2e-4)
                with engine.option(engine.OPT_EXACT, engine.EXACT_OFF):
                    sA, _, sE, _ = engine.loglik_grad(dev(A)[None], dev(pi)[None], dev(E)[None], dev(w)[None])
                assert np.abs(sE.cpu().numpy()[0] - rE).max() > 3e-3 * np.abs(rE).max()        # what the windows repaired
            # log alpha and log beta: windows as well — the rows after (before) a window move with its log scale
            la, lla = engine.forward(dev(A)[None], dev(pi), dev(E[None]))
            da = engine.exact_detail((1, b, L, 15), op=engine.OP_FORWARD)
            assert da["routed"] == 3 and da["whole"] == 0 and da["window_chunks"] * T <= 4 * 6000, da
            la64, _ = textbook.log_alpha(A, pi, E)
            assert_log_close_in_probability_space(la.cpu().numpy()[0], la64, "log alpha, windows, chunk %d" % chunk)
            assert np.all(np.abs(lla.cpu().numpy()[0] - ll64) <= 1e-6 * np.abs(ll64) + 2e-4), chunk
            lb = engine.backward(dev(A)[None], dev(E[None]))
            db = engine.exact_detail((1, b, L, 15), op=engine.OP_BACKWARD)
            assert 1 <= db["routed"] <= 3 and db["whole"] == 0 and db["window_chunks"] * T <= 4 * 6000, db
            lb64 = textbook.log_beta(A, E)
            assert_log_close_in_probability_space(lb.cpu().numpy()[0], lb64, "log beta, windows, chunk %d" % chunk)
            # sequence 5's second stretch gives the backward certificate nothing (no clamp-born mass to speak of), but
            # its two emission-floor steps in a row take the chunk operator's columns through the denormal range: the
            # reduce marks the chunk, the verdict counts it as flagged (without that log beta is 0.6 off below it)
            wt = engine.window_table((1, b, L, 15), 5, op=engine.OP_BACKWARD)
            assert wt["psi"][17003 // T] == 1.0 and len(wt["windows"]) >= 2, (chunk, wt["windows"])
            with engine.option(engine.OPT_EXACT, engine.EXACT_OFF):
                la_scan, _ = engine.forward(dev(A)[None], dev(pi), dev(E[None]))
            with pytest.raises(AssertionError):                          # what the windows repaired
                assert_log_close_in_probability_space(la_scan.cpu().numpy()[0], la64, "scan alone")
            del la, lb, la_scan
            easy = np.array([s not in hard for s in range(b)])
            assert np.array_equal(scan[0][easy], out[0][easy]) and np.array_equal(sl[0][easy], ll[0][easy])
            # what the windows repaired: the posteriors around the stretch and the log-likelihood
            assert np.abs(scan[0][2] - g64[2]).max() > 1e-4 and np.abs(sl[0][2] - ll64[2]) > 0.05
            # ... and nothing else moved: far from the stretches the flagged sequences are bitwise the scan's
            far = np.ones(L, bool)
            far[max(0, 5000 - 6 * max(T, 192)):5000 + 6 * max(T, 192)] = False
            assert np.array_equal(scan[0][2][far], out[0][2][far])


def test_windows_grow_until_the_recursion_has_forgotten():
    """A model that never forgets: three triangles of states that communicate with probability 1e-15 only, so the
    class a path is in stays what it was.  Where the likelier class becomes impossible the cell's clamps put mass
    into BOTH other classes (the transition matrix only into the next one), and that difference never dies out:
    the window's far-end vectors never meet the scan's and it grows to the ends of the sequence — still the exact
    serial recursion, on exactly the sequences that need it."""
    rng = np.random.default_rng(78)
    q = 9
    A = np.zeros((q, q), dtype=np.float64)
    for i in range(q):
        c = i // 3
        A[i, i] = 0.7
        A[i, 3 * c + (i + 1) % 3] = 0.3
    for c in range(3):
        A[3 * c, 3 * ((c + 1) % 3)] = 1e-15                      # class c -> class c + 1 only, and hardly ever
    A = (A / A.sum(-1, keepdims=True)).astype(np.float32)
    pi = np.full(q, 1 / q, dtype=np.float32)
    b, L = 4, 6000
    E = (rng.random((b, L, q)) * 0.9 + 0.05).astype(np.float32)
    E[1, :2990, 3:] *= 1e-3                                      # class 0 is the likelier one ...
    E[1, 3000:3003, :3] = 0.0                                    # ... until it becomes impossible
    g64, ll64 = obuild.posterior(A, pi, E)
    with engine.option(engine.OPT_CHUNK, 64):
        out, ll = run_post(A, pi, E[None])
        det = _detail((1, b, L, q))
        with engine.option(engine.OPT_EXACT, engine.EXACT_OFF):
            scan, _ = run_post(A, pi, E[None])
    assert det["routed"] >= 1, det
    assert np.abs(out[0] - g64).max() <= 2e-5, (det, np.abs(out[0] - g64).max(axis=(1, 2)))
    assert np.all(np.abs(ll[0] - ll64) <= 1e-6 * np.abs(ll64) + 2e-4)
    assert np.abs(scan[0][1] - g64[1]).max() > 1e-4, det                          # what the routing repaired


def test_emitter_generated_input():
    """The pipeline's own kind of input (reference tests/parallel_rnn_forward.py:19-40; SURVEY section 8(d)'s parity
    recipe): class probabilities softmax(scale * randn), one-hot nucleotides with 1 % N -> hmm_gene_emissions ->
    hmm_posterior.  47 % of the emissions are exact zeros.  For scale 2 nothing is routed; for peaked class
    probabilities (scale 6) a few sequences are, in windows; EVERY sequence matches the serial fp64 oracle."""
    from hmm_layer_amd.gene_pred_hmm_emitter import GenePredHMMEmitter
    em = GenePredHMMEmitter(start_codons=[("ATG", 1.)], stop_codons=[("TAG", .34), ("TAA", .33), ("TGA", .33)],
                            intron_begin_pattern=[("NGT", .99), ("NGC", .005), ("NAT", .005)],
                            intron_end_pattern=[("AGN", .99), ("ACN", .01)])
    b, L = 192, 50000
    em.build((1, b, L, 15))
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        em.emission_kernel.copy_(torch.randn(em.emission_kernel.shape, generator=g))
    em = em.to(DEV_)
    A = params.intended_A15().numpy()
    pi = np.full(15, 1 / 15, dtype=np.float32)
    routed = {}
    for scale in (2.0, 6.0):
        cls = torch.softmax(scale * torch.randn((1, b, L, 15), generator=g), -1)
        idx = torch.where(torch.rand((1, b, L), generator=g) < 0.01, torch.full((1, b, L), 4),
                          torch.randint(0, 4, (1, b, L), generator=g))
        x = torch.cat([cls, torch.nn.functional.one_hot(idx, 5).float()], -1).to(DEV_)
        E = em.forward_fused(x).contiguous()
        assert 0.4 < float((E == 0).float().mean()) < 0.55
        out, ll = engine.posterior(dev(A)[None], dev(pi), E)
        det = _detail((1, b, L, 15))
        routed[scale] = det
        assert det["whole"] == 0, det
        with engine.option(engine.OPT_EXACT, engine.EXACT_OFF):
            off, _ = engine.posterior(dev(A)[None], dev(pi), E)
        changed = (off != out).any(dim=3).any(dim=2)[0].cpu().numpy()
        assert changed.sum() == det["routed"]
        # every routed sequence and a sample of the others against the serial fp64 recursion
        pick = sorted(set(np.nonzero(changed)[0].tolist() + list(range(0, b, 16))))
        g64, ll64 = obuild.posterior(A, pi, E[0, pick].cpu().numpy())
        err = np.abs(out[0, pick].cpu().numpy() - g64).max(axis=(1, 2))
        assert err.max() <= 2e-5, (scale, det, err.max())
        assert np.all(np.abs(ll[0, pick].cpu().numpy() - ll64) <= 1e-6 * np.abs(ll64))
        if det["routed"]:
            assert det["window_chunks"] * engine.chunk_len(1, b, L, 15) <= 0.2 * det["routed"] * L, det
        del x, E, out, off
    assert routed[2.0]["routed"] == 0, routed
    assert 0 < routed[6.0]["routed"] <= b // 4, routed


def test_log_alpha_log_beta_window_sweep():
    """A few cases of tests/logab_sweep.py: random local stretches (a state emitting alone, nothing emitting, dead
    columns), several per sequence, forced chunk lengths — log alpha / log beta and the log-likelihood with the window
    recomputation engaged, against the fp64 serial recursion."""
    import logab_sweep
    assert logab_sweep.run(10, 7, verbose=False) == 0
