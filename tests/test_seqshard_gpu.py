"""Sequence-sharded posteriors on ONE GPU: the R time slabs of a batch go through the engine's own
hmm_seqshard_reduce / hmm_seqshard_posterior one after the other (what R ranks would run side by side), the
"all-gather" is seqshard.stack_slab_operators on the same device.  Against the unsharded engine call and the
fp64 oracle."""
import numpy as np
import pytest
import torch

from hmm_layer_amd import engine, seqshard
from oracle import params, textbook

from test_engine_gpu import dev, rand_model

pytestmark = pytest.mark.gpu


def sharded(A, pi, E, cuts, mode=engine.POST_PROB):
    """A (k,q,q), pi (k,q), E (k,b,L,q) numpy; cuts = slab boundaries.  -> out, loglik per rank, phi summed."""
    R = len(cuts) - 1
    Ad, pid = dev(A), dev(pi)
    slabs = [dev(E[:, :, cuts[r]:cuts[r + 1]].copy()) for r in range(R)]
    be = seqshard.EngineBackend()
    # one workspace per "rank": the chunk operators of step 1 have to survive until step 3
    streams = [torch.cuda.Stream() for _ in range(R)]
    ops, exs = [], []
    for r in range(R):
        with torch.cuda.stream(streams[r]):
            o, e = be.reduce(Ad, slabs[r], r == 0, R)
        ops.append(o)
        exs.append(e)
    torch.cuda.synchronize()
    all_ops, all_exps = seqshard.stack_slab_operators(ops, exs)
    torch.cuda.synchronize()                    # (stacked on the default stream, consumed on the ranks' streams)
    outs, lls, phi = [], [], 0
    for r in range(R):
        with torch.cuda.stream(streams[r]):
            o, ll, ph = be.posterior(Ad, pid, slabs[r], all_ops, all_exps, r, mode)
        streams[r].synchronize()
        outs.append(o.cpu().numpy())
        lls.append(ll.cpu().numpy())
        phi = phi + ph.cpu().numpy()
    return np.concatenate(outs, axis=2), lls, phi


@pytest.mark.parametrize("q,b,L,cuts", [(15, 3, 5000, [0, 2500, 5000]), (15, 2, 4001, [0, 1000, 1017, 4001]),
                                        (7, 4, 300, [0, 100, 300]), (16, 1, 20000, [0, 5000, 10000, 15000, 20000]),
                                        (3, 2, 40, [0, 1, 40])])
def test_slabs_reproduce_the_unsharded_result(q, b, L, cuts):
    rng = np.random.default_rng(q * 1000 + L)
    if q == 15:
        A, pi = params.intended_A15().numpy(), np.full(15, 1 / 15, dtype=np.float32)
    else:
        A, pi = rand_model(rng, q)
    E = (rng.random((1, b, L, q)) * 0.9 + 0.05).astype(np.float32)
    got, lls, phi = sharded(A[None], pi[None], E, cuts)
    ref, llref = engine.posterior(dev(A)[None], dev(pi), dev(E))
    g64, ll64 = textbook.posterior(A, pi, E[0])
    assert np.abs(got[0] - g64).max() <= 2e-5
    assert np.abs(got - ref.cpu().numpy()).max() <= 2e-6
    for ll in lls:                                     # every rank holds the whole sequences' log-likelihood
        assert np.all(np.abs(ll[0] - ll64) <= 1e-6 * np.abs(ll64) + 2e-4)
    assert (phi <= seqshard.PHI_LIMIT).all()


def test_log_modes_two_models_and_flags(golden):
    rng = np.random.default_rng(9)
    q, b, L = 15, 3, 1200
    A = np.stack([params.intended_A15().numpy(), golden("transitioner")["A15_as_shipped"]])
    pi = np.full((2, q), 1 / q, dtype=np.float32)
    E = (rng.random((2, b, L, q)) * 0.9 + 0.05).astype(np.float32)
    E[0, 1, 500:520] = 0.0                              # twenty positions in a row emit from state 9 alone, which always
    E[0, 1, 500:520, 9] = 0.5                           # leaves after one step: decided by the eps clamps
    cuts = [0, 400, 800, 1200]
    got, lls, phi = sharded(A, pi, E, cuts, engine.POST_LOG)
    g64, ll64 = textbook.posterior(A[0], pi[0], E[0])
    ok = [0, 2]
    assert np.abs(np.exp(got[0][ok]) - g64[ok]).max() <= 2e-5
    assert np.all(np.abs(lls[1][0][ok] - ll64[ok]) <= 1e-6 * np.abs(ll64[ok]) + 2e-4)
    # flags: the impossible sequence of the primitive model, every sequence of the as-shipped (reducible) model
    flagged = ~(phi <= seqshard.PHI_LIMIT)
    assert flagged[0].tolist() == [False, True, False] and flagged[1].all()
    got2, lls2, _ = sharded(A[:1], pi[:1], E[:1], cuts, engine.POST_LOG_NO_LL)
    assert np.abs(np.exp(got2[0][ok] - lls2[0][0][ok][:, None, None]) - g64[ok]).max() <= 2e-5 + 2.4e-7 * np.abs(ll64).max()
