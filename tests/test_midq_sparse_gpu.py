"""The one-wave-per-sequence kernels of 17..64 states with the SPARSE step (hmm_midq.inc: every lane gathers its own
predecessors / successors, k_mq_sp_prep decides per model on the device): the multi-copy gene models
(hmm_layer/gene_pred_hmm_transitioner.py:263-308) and random sparse topologies, every entry point, against the
serial fp64 oracle at the tolerances of tests/test_engine_gpu.py and against the dense step on the same input."""
import numpy as np
import pytest
import torch

from hmm_layer_amd import engine

from test_engine_gpu import dev
from test_scan32_gpu import check, post
from test_scan64_gpu import gene_k

pytestmark = pytest.mark.gpu


def gene_input(rng, k, b, L):
    q = 1 + 14 * k
    E = (rng.random((b, L, q)) * 0.9 + 0.05).astype(np.float32) / 4096
    dead = rng.random(E.shape) < 0.4
    dead[..., :1 + 6 * k] = False                   # IR, introns and exons always keep mass alive
    E[dead] = 0.0
    return E


@pytest.mark.parametrize("k", [2, 3, 4])
def test_multi_copy_gene_models_on_the_serial_kernels(k):
    rng = np.random.default_rng(700 + k)
    A, pi = gene_k(k)
    q = 1 + 14 * k
    for b, L in ((1, 1), (2, 2), (3, 9), (2, 1031), (5, 4000)):
        E = gene_input(rng, k, b, L)
        with engine.option(engine.OPT_EXACT, engine.EXACT_ALWAYS):
            # (33..64 states count routed sequences only where the chunked scan is engaged at all: L >= 256)
            check(A, pi, E, "sparse serial k=%d b=%d L=%d" % (k, b, L), expect_serial=b if (k == 2 or L >= 256) else None)
            out, ll = post(A, pi, E[None])
            with engine.option(engine.OPT_FORCE_DENSE, 1):
                dense, lld = post(A, pi, E[None])
        assert np.abs(out[0] - dense[0]).max() <= 4e-6 and np.allclose(ll, lld, rtol=1e-7, atol=1e-5)


@pytest.mark.parametrize("q,deg", [(17, 2), (32, 4), (40, 6), (64, 8), (50, 9)])
def test_random_sparse_topologies_and_mixed_calls(q, deg):
    """In / out degree up to 4, up to 8, above (dense step); a sparse and a dense model in one call."""
    rng = np.random.default_rng(q * 10 + deg)
    A = np.zeros((q, q), dtype=np.float32)
    for i in range(q):
        A[i, i] = 1.0
        for j in rng.choice(q, size=deg - 1, replace=False):
            A[i, j] = rng.random() + 0.1
    # in-degrees follow the draws: cap them so that the model's class is the intended one
    for j in range(q):
        nz = np.nonzero(A[:, j])[0]
        lim = deg if deg <= 8 else q
        for i in nz[lim:]:
            if i != j:
                A[i, j] = 0.0
    A /= A.sum(-1, keepdims=True)
    pi = np.full(q, 1 / q, dtype=np.float32)
    b, L = 3, 700
    E = (rng.random((b, L, q)) * 0.9 + 0.05).astype(np.float32)
    E[rng.random(E.shape) < 0.1] = 0.0
    with engine.option(engine.OPT_EXACT, engine.EXACT_ALWAYS):
        check(A, pi, E, "random sparse q=%d deg=%d" % (q, deg), expect_serial=b)       # (L = 700, b = 3: the scans are engaged)
        D = rng.dirichlet(np.ones(q), size=q).astype(np.float32)
        A2 = np.stack([A, D]); pi2 = np.stack([pi, pi]); E2 = np.stack([E, E[::-1].copy()])
        out, ll = engine.posterior(dev(A2), dev(pi2), dev(E2))
        out = out.cpu().numpy(); ll = ll.cpu().numpy()
    from oracle import build as obuild
    for m in range(2):
        g64, ll64 = obuild.posterior(A2[m], pi2[m], E2[m])
        assert np.abs(out[m] - g64).max() <= 2e-5, (m, np.abs(out[m] - g64).max())
        assert np.all(np.abs(ll[m] - ll64) <= 1e-6 * np.abs(ll64) + 2e-4)
