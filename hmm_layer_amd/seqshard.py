"""Sequence-sharded state posteriors: every rank owns one contiguous TIME slab of every sequence.

The reference stitches chunk results with TotalProbabilityCell.forward
(hmm_layer/TotalProbabilityCell.py:30-49) inside _get_total_forward_from_chunks /
_get_total_backward_from_chunks (hmm_layer/MsaHMMLayer.py:285-319, 384-419); here the same stitching
runs one level up, across devices, for batches too small to be cut by sequence:

    reduce      local     the slab's operator per sequence, 1.1 KB            (hmm_seqshard_reduce)
    all-gather  RCCL      twice: the operators (k,b,16,16) fp32 and their exponents (k,b,16) int32 per rank
    finish      local     hops over the R slab operators, local chunk scan, apply kernels
                                                                              (hmm_seqshard_posterior)
    all-reduce  RCCL      k*b floats: the sequences' clamp-born posterior mass, summed over ranks

Three collectives per call, all of them latency-sized.  Sequences whose summed clamp-born mass exceeds the
engine's limit are decided by the cell's eps clamps, which cannot be cut in time: gather_flagged() sends
their slabs to one rank, runs the unsharded call there and returns every rank its own slab of the result
(point-to-point, nothing is broadcast).

`backend` is the compute provider of the local steps: the HIP engine (default), or anything with the same
three methods (tests/seqshard_ref.py drives the exchange on CPU tensors over gloo with an fp64
restatement).  There is no CPU compute in this module.
"""
import torch

from . import engine

PHI_LIMIT = 2e-6        # EXACT_DELTA of the engine: above it a sequence needs the serial exact-clamp kernels


class EngineBackend:
    """The local steps on the HIP engine (through the C ABI)."""

    def reduce(self, A, E_slab, seq_start, R):
        return engine.seqshard_reduce(A, E_slab, seq_start, R)

    def posterior(self, A, pi, E_slab, all_ops, all_exps, r, mode):
        return engine.seqshard_posterior(A, pi, E_slab, all_ops, all_exps, r, mode=mode)

    def unsharded(self, A, pi, E, mode):
        return engine.posterior(A, pi, E, mode=mode)


def stack_slab_operators(ops, exps):
    """Per-rank results in time order, each (k,b,16,16) / (k,b,16) -> the layout the finish step reads:
    (k,b,R,16,16) / (k,b,R,16), contiguous."""
    return torch.stack(list(ops), dim=2).contiguous(), torch.stack(list(exps), dim=2).contiguous()


def posterior(A, pi, E_slab, mode=engine.POST_PROB, group=None, backend=None):
    """-> (out (k,b,Ls,q), loglik (k,b) of the whole sequences, needs_unsharded (k,b) bool).

    Ranks of `group` hold consecutive time slabs in rank order (rank 0 owns position 0); slab lengths
    may differ.  Sequences flagged in `needs_unsharded` (summed clamp-born posterior mass above 2e-6, or a
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


def _global_rank(group, i):
    import torch.distributed as dist
    return i if group is None else dist.get_global_rank(group, i)


def gather_flagged(A, pi, E_slab, out, loglik, needs_unsharded, mode=engine.POST_PROB, group=None, root=0,
                   backend=None):
    """Recompute the flagged sequences unsharded on group rank `root`: every rank sends it its slabs of those
    sequences (slab lengths may differ), it runs the unsharded call and sends every rank ITS OWN slab of the
    result back; the log-likelihoods (a few doubles) are broadcast.  In place on `out` / `loglik`."""
    import torch.distributed as dist
    backend = backend or EngineBackend()
    R, r = dist.get_world_size(group), dist.get_rank(group)
    flagged = needs_unsharded.nonzero(as_tuple=False).tolist()          # one host sync for the whole call
    if not flagged:
        return out, loglik
    A3 = A if A.dim() == 3 else A.unsqueeze(0)
    q = E_slab.shape[-1]
    pi2 = pi.reshape(-1, q)
    lens = [torch.zeros(1, dtype=torch.int64, device=E_slab.device) for _ in range(R)]
    dist.all_gather(lens, torch.tensor([E_slab.shape[2]], dtype=torch.int64, device=E_slab.device), group=group)
    lens = [int(v) for v in torch.cat(lens).tolist()]
    groot = _global_rank(group, root)
    for m in sorted(set(i[0] for i in flagged)):
        seqs = torch.tensor([i[1] for i in flagged if i[0] == m], dtype=torch.long, device=E_slab.device)
        n = seqs.numel()
        mine = E_slab[m, seqs].contiguous()                             # (n, Ls, q)
        ll = torch.empty((1, n), dtype=torch.float64, device=E_slab.device)
        if r == root:
            parts = []
            for src in range(R):
                if src == r:
                    parts.append(mine)
                else:
                    buf = torch.empty((n, lens[src], q), dtype=mine.dtype, device=mine.device)
                    dist.recv(buf, src=_global_rank(group, src), group=group)
                    parts.append(buf)
            o, ll = backend.unsharded(A3[m:m + 1].contiguous(), pi2[m:m + 1].contiguous(),
                                      torch.cat(parts, dim=1)[None].contiguous(), mode)
            ll = ll.to(torch.float64).reshape(1, n).contiguous()
            off, piece = 0, None
            for dst in range(R):
                sl = o[0][:, off:off + lens[dst]].contiguous()
                off += lens[dst]
                if dst == r:
                    piece = sl
                else:
                    dist.send(sl, dst=_global_rank(group, dst), group=group)
        else:
            dist.send(mine, dst=groot, group=group)
            piece = torch.empty_like(mine)
            dist.recv(piece, src=groot, group=group)
        dist.broadcast(ll, src=groot, group=group)
        out[m, seqs] = piece.to(out.dtype)
        loglik[m, seqs] = ll[0].to(loglik.dtype)
    return out, loglik
