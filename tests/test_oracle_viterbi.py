"""Viterbi oracle (parity unpinned: the reference has no Viterbi): numpy definition vs its C twin,
brute-force property tests, and the C posterior twin vs the numpy textbook.  CPU only."""
import numpy as np
import pytest

from oracle import build as obuild
from oracle import textbook, viterbi


def rand_case(rng, b, L, q, scale=3.0):
    logA = np.log(rng.dirichlet(np.ones(q), size=q)).astype(np.float32)
    logpi = np.log(rng.dirichlet(np.ones(q))).astype(np.float32)
    logE = (-scale * rng.random((b, L, q))).astype(np.float32)
    return logA, logpi, logE


@pytest.mark.parametrize("q,L", [(2, 6), (3, 7), (4, 6)])
def test_viterbi_score_is_the_brute_force_maximum(q, L):
    rng = np.random.default_rng(q * 10 + L)
    logA, logpi, logE = rand_case(rng, 3, L, q)
    path, score = viterbi.viterbi(logA, logpi, logE)
    for n in range(3):
        assert abs(score[n] - viterbi.brute_force(logA, logpi, logE[n])) < 1e-9
        assert abs(viterbi.path_score(logA, logpi, logE[n], path[n]) - score[n]) < 1e-9


def test_c_twin_is_bit_exact():
    rng = np.random.default_rng(1)
    for q, b, L in [(3, 4, 50), (7, 3, 300), (15, 5, 2000), (16, 2, 700)]:
        logA, logpi, logE = rand_case(rng, b, L, q)
        logA[rng.random(logA.shape) < 0.4] = -np.inf            # absent edges
        logE[rng.random(logE.shape) < 0.1] = -1000.0
        p1, s1 = viterbi.viterbi(logA, logpi, logE)
        p2, s2 = obuild.viterbi(logA, logpi, logE)
        assert np.array_equal(p1, p2) and np.array_equal(s1, s2)


def test_ties_take_the_lowest_index():
    q, L = 4, 9
    logA = np.zeros((q, q), dtype=np.float32)
    logpi = np.zeros(q, dtype=np.float32)
    logE = np.zeros((1, L, q), dtype=np.float32)
    path, score = viterbi.viterbi(logA, logpi, logE)
    assert (path == 0).all() and score[0] == 0.0
    logE[0, 4, 0] = -1.0                                         # forces a detour at t = 4
    path, _ = viterbi.viterbi(logA, logpi, logE)
    assert list(path[0]) == [0, 0, 0, 0, 1, 0, 0, 0, 0]
    p2, _ = obuild.viterbi(logA, logpi, logE)
    assert np.array_equal(path, p2)


def test_quantisation_rules():
    x = np.array([0.0, -1.0, -np.inf, -2000.0, 5000.0, -36.841362], dtype=np.float32)
    qv = viterbi.quantise(x)
    assert qv[0] == 0 and qv[1] == -65536 and qv[2] == qv[3] == -1024 * 65536 and qv[4] == 1024 * 65536
    assert qv[5] == int(np.rint(np.float32(-36.841362) * np.float32(65536)))


def test_c_posterior_twin_matches_textbook():
    rng = np.random.default_rng(2)
    A = rng.dirichlet(np.ones(7), size=7).astype(np.float32)
    pi = rng.dirichlet(np.ones(7)).astype(np.float32)
    E = (rng.random((3, 400, 7)) * 0.9 + 0.05).astype(np.float32)
    E[rng.random(E.shape) < 0.1] = 0.0
    g1, l1 = textbook.posterior(A, pi, E)
    g2, l2 = obuild.posterior(A, pi, E)
    assert np.abs(g1 - g2).max() < 1e-12 and np.abs(l1 - l2).max() < 1e-9
