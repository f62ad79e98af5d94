"""The exchange and recombination of the sequence-sharded posterior on CPU tensors: two gloo ranks, each
with one time slab of every sequence, all-gather of the slab operators, per-rank recombination, all-reduce
of the floor-transition bounds — against the UNSHARDED fp64 oracle.  The local compute comes from
tests/seqshard_ref.py (an fp64 restatement with the engine's operator format); on GPUs the same
hmm_layer_amd/seqshard.py drives the HIP engine (tests/test_seqshard_gpu.py)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hmm_layer_amd import seqshard
from oracle import params, textbook

from seqshard_ref import RefBackend


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _inputs():
    rng = np.random.default_rng(5)
    k, b, L, q = 2, 3, 61, 15
    A = np.stack([params.intended_A15().numpy(), rng.dirichlet(np.ones(q), size=q).astype(np.float32)])
    pi = rng.dirichlet(np.ones(q), size=k).astype(np.float32)
    E = (rng.random((k, b, L, q)) * 0.9 + 0.05).astype(np.float32)
    E[0, 2, 20:24, :] = 0.0                         # one sequence that only the eps clamps survive: four positions
    E[0, 2, 20:24, 9] = 0.5                         # in a row emit from state 9 alone, which always leaves after one
    return A, pi, E


def _worker(rank, world, port, cuts, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    A, pi, E = _inputs()
    lo, hi = cuts[rank], cuts[rank + 1]
    got, ll, flag = seqshard.posterior(torch.from_numpy(A), torch.from_numpy(pi), torch.from_numpy(E[:, :, lo:hi].copy()),
                                       backend=RefBackend())
    got0, ll0 = got.numpy().copy(), ll.numpy().copy()
    # the flagged sequences, recomputed unsharded on the LAST rank and handed back slab by slab (rank 1 is then
    # the root: the point-to-point legs run in both directions)
    got, ll = seqshard.gather_flagged(torch.from_numpy(A[0]) if A.shape[0] == 1 else torch.from_numpy(A),
                                      torch.from_numpy(pi), torch.from_numpy(E[:, :, lo:hi].copy()), got, ll, flag,
                                      root=world - 1, backend=RefBackend())
    out[rank] = (got0, ll0, flag.numpy(), got.numpy(), ll.numpy())
    dist.destroy_process_group()


def test_two_time_slabs_over_gloo_match_the_unsharded_oracle():
    A, pi, E = _inputs()
    k, b, L, q = E.shape
    for cuts in ([0, 30, L], [0, 44, L]):               # equal-ish and unequal slab lengths
        mgr = mp.Manager()
        out = mgr.dict()
        mp.spawn(_worker, args=(2, _free_port(), cuts, out), nprocs=2, join=True)
        got = np.concatenate([out[0][0], out[1][0]], axis=2)
        for m in range(k):
            g64, ll64 = textbook.posterior(A[m], pi[m], E[m])
            benign = ~out[0][2][m]
            for r in range(2):
                assert np.abs(out[r][1][m] - ll64)[benign].max() <= 1e-9 * np.abs(ll64).max()   # every rank: whole-sequence loglik
                assert np.array_equal(out[r][2], out[0][2])                                 # and the same flags
            assert np.abs(got[m][benign] - g64[benign]).max() <= 1e-6                        # (fp32 output tensors)
        flags = out[0][2]
        assert flags[0, 2] and flags.sum() == 1                                             # exactly the clamp-decided sequence
        # after gather_flagged every sequence, the flagged one included, matches the unsharded oracle on every rank
        fixed = np.concatenate([out[0][3], out[1][3]], axis=2)
        for m in range(k):
            g64, ll64 = textbook.posterior(A[m], pi[m], E[m])
            assert np.abs(fixed[m] - g64).max() <= 1e-6
            for r in range(2):
                assert np.abs(out[r][4][m] - ll64).max() <= 1e-9 * np.abs(ll64).max()
        assert np.abs(got[0, 2] - textbook.posterior(A[0], pi[0], E[0])[0][2]).max() > 1e-5    # what it repaired


def test_stacking_layout():
    ops = [torch.full((2, 3, 16, 16), float(r)) for r in range(4)]
    exs = [torch.full((2, 3, 16), r, dtype=torch.int32) for r in range(4)]
    a, e = seqshard.stack_slab_operators(ops, exs)
    assert a.shape == (2, 3, 4, 16, 16) and e.shape == (2, 3, 4, 16) and a.is_contiguous()
    assert float(a[1, 2, 3, 5, 5]) == 3.0 and int(e[0, 1, 2, 7]) == 2
