"""The three-phase scan algorithm (tests/algo_model.py, the executable spec of the HIP
kernels) against the fp64 textbook oracle.  CPU only."""
import numpy as np
import pytest

from oracle import textbook
from tests import algo_model


def _case(golden, name):
    g = golden(name)
    return g["A"], g["pi"], g["E"]


@pytest.mark.parametrize("name,T", [("cell_q3", 16), ("cell_q3", 48), ("cell_q7", 32),
                                    ("cell_q15", 16), ("cell_q15", 64), ("cell_q15z", 16),
                                    ("cell_q15", 1024)])
def test_scan_matches_textbook(golden, name, T):
    A, pi, E = _case(golden, name)
    gam64, ll64 = textbook.posterior(A, pi, E)
    la64, _ = textbook.log_alpha(A, pi, E)
    lb64 = textbook.log_beta(A, E)
    for n in range(E.shape[0]):
        gam, ll, la, lb = algo_model.posterior(A, pi, E[n], T)
        assert np.abs(gam - gam64[n]).max() < 5e-6
        assert abs(ll - ll64[n]) < 1e-4 * max(1.0, abs(ll64[n]) * 1e-2)
        m = la64[n] > -30
        assert np.abs(la - la64[n])[m].max() < 2e-4
        m = lb64[n] > -30
        assert np.abs(lb - lb64[n])[m].max() < 2e-4


def test_scan_long_sequence_with_tiny_emissions():
    """Gene-model-like magnitudes (E ~ 1e-5, loglik ~ -1e5): the per-column power-of-two
    scaling must keep every chunk operator in range."""
    from oracle import params
    rng = np.random.default_rng(3)
    A = params.intended_A15().numpy()
    pi = np.full(15, 1 / 15, dtype=np.float32)
    L = 6000
    E = (rng.random((L, 15)) * 0.9 + 0.05).astype(np.float32) / 4096
    E[rng.random((L, 15)) < 0.1] = 0.0
    gam64, ll64 = textbook.posterior(A, pi, E)
    gam, ll, _, _ = algo_model.posterior(A, pi, E, 512)
    assert np.isfinite(gam).all()
    assert np.abs(gam - gam64[0]).max() < 2e-5
    assert abs(ll - ll64[0]) < 1e-6 * abs(ll64[0])
