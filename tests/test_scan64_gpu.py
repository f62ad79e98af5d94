"""The chunked (time-parallel) scan for 33..64 states (hmm_scan64.inc): GenePredMultiHMMTransitioner(k = 3, 4)
(43 / 57 states, hmm_layer/gene_pred_hmm_transitioner.py:263-308) and dense models, for few long sequences — dense
64-state MFMA reduce, 64-lane chunk scan, apply kernels with four tile rows — against the serial fp64 oracle at the
tolerances of tests/test_engine_gpu.py and against the one-wave-per-sequence kernels it replaces."""
import numpy as np
import pytest
import torch

from hmm_layer_amd import engine
from hmm_layer_amd.gene_pred_hmm_transitioner import GenePredMultiHMMTransitioner
from oracle import build as obuild

from test_engine_gpu import dev, rand_model
from test_scan32_gpu import check, post

pytestmark = pytest.mark.gpu


def gene_k(k):
    tr = GenePredMultiHMMTransitioner(k=k, initial_exon_len=200, initial_intron_len=4500, initial_ir_len=10000)
    with torch.no_grad():
        return tr.make_A()[0].numpy().copy(), tr.make_initial_distribution().reshape(-1).numpy().copy()


@pytest.mark.parametrize("k", [3, 4])
def test_multi_copy_gene_models(k):
    rng = np.random.default_rng(60 + k)
    A, pi = gene_k(k)
    q = 1 + 14 * k
    assert A.shape == (q, q)
    for b, L in ((1, 256), (3, 1031), (2, 6000)):
        E = (rng.random((b, L, q)) * 0.9 + 0.05).astype(np.float32) / 4096
        dead = rng.random(E.shape) < 0.4
        dead[..., :1 + 6 * k] = False                   # IR, introns and exons always keep mass alive
        E[dead] = 0.0
        check(A, pi, E, "gene k=%d b=%d L=%d" % (k, b, L), expect_serial=0)


@pytest.mark.parametrize("q", [33, 48, 64])
def test_dense_models(q):
    rng = np.random.default_rng(600 + q)
    for dense in (True, False):
        A, pi = rand_model(rng, q, dense=dense)
        for (b, L), chunk in (((2, 300), 16), ((5, 777), 48), ((3, 2600), 0)):
            E = (rng.random((b, L, q)) * 0.9 + 0.05).astype(np.float32)
            if dense:
                E[rng.random(E.shape) < 0.2] *= 1e-6
            with engine.option(engine.OPT_CHUNK, chunk):
                check(A, pi, E, "dense q=%d b=%d L=%d chunk=%d" % (q, b, L, chunk), expect_serial=0)
                out, ll = post(A, pi, E[None])
                with engine.option(engine.OPT_EXACT, engine.EXACT_ALWAYS):
                    ser, lls = post(A, pi, E[None])
                assert engine.exact_count(engine.OP_POSTERIOR, (1, b, L, q)) == b
            assert np.abs(out[0] - ser[0]).max() <= 4e-6 and np.allclose(ll, lls, rtol=1e-7, atol=1e-5)


def test_routing_per_model_per_sequence_and_by_batch_size():
    """Two models in one call (primitive / reducible) with one clamp-decided sequence; more than 96 sequences, and
    short sequences, stay on the one-wave-per-sequence kernels altogether."""
    rng = np.random.default_rng(9)
    q, b, L = 43, 3, 900
    # model 0: a cycle with one self loop (primitive, with the longest possible index): mass moves on by one state per
    # step, so thirty positions in a row that only state 20 can emit are survived through the eps clamps alone
    A0 = np.roll(np.eye(q, dtype=np.float32), 1, axis=1)
    A0[0, 0] = 0.5; A0[0, 1] = 0.5
    pi0 = np.full(q, 1 / q, dtype=np.float32)
    A1 = np.triu(rand_model(rng, q)[0])
    A1 /= A1.sum(-1, keepdims=True)
    E = (rng.random((2, b, L, q)) * 0.9 + 0.05).astype(np.float32)
    E[0, 1, 300:330] = 0.0
    E[0, 1, 300:330, 20] = 0.5
    A = np.stack([A0, A1.astype(np.float32)])
    pi = np.stack([pi0, np.full(q, 1 / q, dtype=np.float32)])
    out, ll = engine.posterior(dev(A), dev(pi), dev(E))
    assert engine.exact_count(engine.OP_POSTERIOR, (2, b, L, q)) == 1 + b
    out, ll = out.cpu().numpy(), ll.cpu().numpy()
    for m in range(2):
        g64, ll64 = obuild.posterior(A[m], pi[m], E[m])
        assert np.abs(out[m] - g64).max() <= 2e-5, m
        assert np.all(np.abs(ll[m] - ll64) <= 1e-6 * np.abs(ll64) + 2e-4), m
    with engine.option(engine.OPT_EXACT, engine.EXACT_OFF):
        scan, _ = engine.posterior(dev(A[:1]), dev(pi[:1]), dev(E[:1]))
    scan = scan.cpu().numpy()
    assert np.array_equal(scan[0][[0, 2]], out[0][[0, 2]])
    assert np.abs(scan[0][1] - out[0][1]).max() > 1e-4
    # above the batch limit and below the length limit: the serial kernels, same answers
    A0, pi0 = gene_k(3)
    for bb, LL in ((100, 300), (2, 100)):
        Eb = (rng.random((bb, LL, q)) * 0.9 + 0.05).astype(np.float32)
        o, l = post(A0, pi0, Eb[None])
        g64, ll64 = obuild.posterior(A0, pi0, Eb)
        assert np.abs(o[0] - g64).max() <= 2e-5
        assert np.all(np.abs(l[0] - ll64) <= 1e-6 * np.abs(ll64) + 2e-4)
    # a SPARSE model of more than 56 sequences belongs to the one-wave-per-sequence kernels' sparse step (k64_check);
    # a dense model of the same batch, and the sparse one at 56, stay on the chunked scan
    Ad, pid = rand_model(rng, q, dense=True)
    for Am, pim, bb, want in ((A0, pi0, 60, 60), (A0, pi0, 56, 0), (Ad, pid, 60, 0)):
        Eb = (rng.random((bb, 300, q)) * 0.9 + 0.05).astype(np.float32)
        o, l = post(Am, pim, Eb[None])
        assert engine.exact_count(engine.OP_POSTERIOR, (1, bb, 300, q)) == want, (bb, want)
        g64, ll64 = obuild.posterior(Am, pim, Eb)
        assert np.abs(o[0] - g64).max() <= 2e-5
        assert np.all(np.abs(l[0] - ll64) <= 1e-6 * np.abs(ll64) + 2e-4)
