"""The one cross-GPU step of the path: the log-likelihood aggregate of
MsaHmmLayer.apply_sequence_weights(aggregate=True) (reference MsaHMMLayer.py:155-164).

Sequences never interact inside forward / backward / Viterbi, so ranks own disjoint slices of
the batch and exchange nothing but, per model, the pair (sum_b w*loglik, sum_b w): ONE
all-reduce(sum) of 2*k doubles (RCCL over xGMI on GPUs — backend "nccl" — or gloo on CPU).
"""
import torch


def shard_bounds(batch, rank, world):
    """Contiguous batch slice [lo, hi) owned by `rank` (sizes differ by at most one)."""
    base, extra = divmod(batch, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def loglik_partials(loglik, weights=None):
    """(k, b) -> (k, 2) fp64: (sum_b w*loglik, sum_b w).  HIP kernel on GPU tensors
    (hmm_loglik_partials), plain torch on CPU tensors (gloo tests)."""
    if loglik.is_cuda:
        from . import engine
        w = None if weights is None else weights.to(torch.float32)
        return engine.loglik_partials(loglik.to(torch.float64), w)
    ll = loglik.to(torch.float64)
    w = torch.ones_like(ll) if weights is None else weights.to(torch.float64)
    return torch.stack([(w * ll).sum(dim=1), w.sum(dim=1)], dim=1)


def aggregate_loglik(loglik, weights=None, group=None, grad_reduction="sum"):
    """Mean over models of the weighted mean over ALL ranks' sequences.

    With a loglik that carries an autograd graph the result is differentiable: its value is the
    global mean, its gradient reaches this rank's sequences only — each rank then holds the
    parameter gradient of its shard and the data-parallel wrapper combines them (not part of this
    path).  `grad_reduction` names how that wrapper combines:
      "sum"   the wrapper SUMS rank gradients (a plain all-reduce(sum)): d/d loglik[m,s] =
              w / sum_all(w) / k, and the summed parameter gradient is the single-process one;
      "mean"  the wrapper AVERAGES them (torch DistributedDataParallel's default): the local
              gradient is pre-multiplied by the world size so that the average is the
              single-process gradient."""
    if grad_reduction not in ("sum", "mean"):
        raise ValueError("grad_reduction must be 'sum' or 'mean'")
    if weights is not None and weights.shape != loglik.shape:
        weights = torch.broadcast_to(weights, loglik.shape)
    if loglik.requires_grad:
        ll = loglik.to(torch.float64)
        w = torch.ones_like(ll) if weights is None else weights.to(ll.device, torch.float64)
        local = (w * ll).sum(dim=1)
        part = torch.stack([local.detach(), w.sum(dim=1)], dim=1)
        scale = 1.0
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.all_reduce(part, op=torch.distributed.ReduceOp.SUM, group=group)
            if grad_reduction == "mean":
                scale = float(torch.distributed.get_world_size(group))
        graph = local * scale
        total = graph + (part[:, 0] - graph.detach())           # value: all ranks; graph: this rank
        return (total / part[:, 1]).mean()
    part = loglik_partials(loglik, weights)
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.all_reduce(part, op=torch.distributed.ReduceOp.SUM, group=group)
    return (part[:, 0] / part[:, 1]).mean()
