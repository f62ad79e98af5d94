"""Sequence-sharded state posteriors: every rank owns one contiguous TIME slab of every sequence.

The reference stitches chunk results with TotalProbabilityCell.forward
(hmm_layer/TotalProbabilityCell.py:30-49) inside _get_total_forward_from_chunks /
_get_total_backward_from_chunks (hmm_layer/MsaHMMLayer.py:285-319, 384-419); here the same stitching
runs one level up, across devices, for batches too small to be cut by sequence:

    reduce      local     the slab's operator per sequence, 1.1 KB            (hmm_seqshard_reduce)
    all-gather  RCCL      the ONE collective of this mode: (k,b,16,16) + (k,b,16) per rank
    finish      local     hops over the R slab operators, local chunk scan, apply kernels
                                                                              (hmm_seqshard_posterior)
    all-reduce  RCCL      k*b floats: the sequences' floor-transition bounds, summed over ranks

`backend` is the compute provider of the two local steps: the HIP engine (default), or anything with the
same two methods (tests/seqshard_ref.py drives the exchange on CPU tensors over gloo with an fp64
restatement).  There is no CPU compute in this module.
"""
import torch

from . import engine

PHI_LIMIT = 1e-6        # EXACT_DELTA of the engine: above it a sequence needs the serial exact-clamp kernels


class EngineBackend:
    """The two local steps on the HIP engine (through the C ABI)."""

    def reduce(self, A, E_slab, seq_start, R):
        return engine.seqshard_reduce(A, E_slab, seq_start, R)

    def posterior(self, A, pi, E_slab, all_ops, all_exps, r, mode):
        return engine.seqshard_posterior(A, pi, E_slab, all_ops, all_exps, r, mode=mode)


def stack_slab_operators(ops, exps):
    """Per-rank results in time order, each (k,b,16,16) / (k,b,16) -> the layout the finish step reads:
    (k,b,R,16,16) / (k,b,R,16), contiguous."""
    return torch.stack(list(ops), dim=2).contiguous(), torch.stack(list(exps), dim=2).contiguous()


def posterior(A, pi, E_slab, mode=engine.POST_PROB, group=None, backend=None):
    """-> (out (k,b,Ls,q), loglik (k,b) of the whole sequences, needs_unsharded (k,b) bool).

    Ranks of `group` hold consecutive time slabs in rank order (rank 0 owns position 0); slab lengths
    may differ.  Sequences flagged in `needs_unsharded` (summed floor-transition bound above 1e-6, or a
    model whose support is not primitive) are decided by the eps clamps and have to be recomputed by
    the unsharded call on one device (gather_flagged); everything else is final."""
    import torch.distributed as dist
    backend = backend or EngineBackend()
    R = dist.get_world_size(group)
    r = dist.get_rank(group)
    op, ex = backend.reduce(A, E_slab, r == 0, R)
    ops = [torch.empty_like(op) for _ in range(R)]
    exs = [torch.empty_like(ex) for _ in range(R)]
    dist.all_gather(ops, op.contiguous(), group=group)
    dist.all_gather(exs, ex.contiguous(), group=group)
    all_ops, all_exps = stack_slab_operators(ops, exs)
    out, ll, phi = backend.posterior(A, pi, E_slab, all_ops, all_exps, r, mode)
    phi = phi.clone()
    dist.all_reduce(phi, op=dist.ReduceOp.SUM, group=group)
    return out, ll, ~(phi <= PHI_LIMIT)


def gather_flagged(A, pi, E_slab, out, loglik, needs_unsharded, mode=engine.POST_PROB, group=None, root=0):
    """Recompute the flagged sequences unsharded on rank `root` (all ranks send it their slabs of those
    sequences; it runs engine.posterior and returns every rank its slab of the result).  In place on
    `out` / `loglik`; equal slab lengths on all ranks."""
    import torch.distributed as dist
    R, r = dist.get_world_size(group), dist.get_rank(group)
    idx = needs_unsharded.nonzero(as_tuple=False)
    if idx.numel() == 0:
        return out, loglik
    for m in sorted(set(int(i) for i in idx[:, 0])):
        seqs = idx[idx[:, 0] == m][:, 1]
        mine = E_slab[m, seqs].contiguous()
        parts = [torch.empty_like(mine) for _ in range(R)]
        dist.all_gather(parts, mine, group=group)
        if r == root:
            full = torch.cat(parts, dim=1)[None]
            o, ll = engine.posterior(A[m:m + 1], pi.reshape(-1, pi.shape[-1])[m:m + 1], full, mode=mode)
            res = list(o[0].chunk(R, dim=1))
        else:
            res, ll = [torch.empty_like(mine) for _ in range(R)], torch.empty((1, len(seqs)), dtype=torch.float64,
                                                                             device=mine.device)
        for src in range(R):               # every rank receives its own time slab of the result
            piece = res[src].contiguous()
            dist.broadcast(piece, src=root, group=group)
            if src == r:
                out[m, seqs] = piece
        ll = ll.contiguous()
        dist.broadcast(ll, src=root, group=group)
        loglik[m, seqs] = ll[0]
    return out, loglik
