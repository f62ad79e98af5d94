"""Differentiable log-likelihood: the engine's forward pass with the engine's analytic backward.

The reference trains by letting autograd unroll the Python time loop
(hmm_layer/BaseRNN.py:217-227 over HmmCell.forward, hmm_layer/MsaHmmCell.py:73-106), which keeps
every step's tensors alive.  Here the graph holds ONE node: forward = hmm_forward (log-likelihood
only, reads E once), backward = hmm_loglik_grad (one forward-backward pass producing dA, dpi and
dE = w * gamma / E).  Nothing per position is saved between the two.
"""
import torch

from . import engine


class LogLikelihood(torch.autograd.Function):
    """loglik (k,b) fp64 = f(A (k,q,q), pi (k,q), E (k,b,L,q)); all on one HIP device, fp32."""

    @staticmethod
    def forward(ctx, A, pi, E, eps):
        A, pi, E = A.contiguous(), pi.contiguous(), E.contiguous()
        ctx.save_for_backward(A, pi, E)
        ctx.eps = eps
        return engine.forward(A, pi, E, want_log_alpha=False, eps=eps)[1]

    @staticmethod
    def backward(ctx, grad_loglik):
        A, pi, E = ctx.saved_tensors
        dA, dpi, dE, _ = engine.loglik_grad(A, pi, E, grad_loglik.to(torch.float32).contiguous(), eps=ctx.eps)
        need = ctx.needs_input_grad
        return (dA if need[0] else None, dpi.reshape(pi.shape) if need[1] else None,
                dE if need[2] else None, None)


def loglik(A, pi, E, eps=engine.EPS):
    """Differentiable (k,b) fp64 log-likelihoods."""
    return LogLikelihood.apply(A, pi, E, eps)


class Posterior(torch.autograd.Function):
    """State posteriors (k,b,L,q), probabilities or logs, differentiable in A, pi and E.  Forward =
    hmm_posterior (the chunked kernels), backward = hmm_posterior_grad (its four sweeps per chunk of the
    scan plan, or over whole sequences where the device-side routing says so) — the reference gets this gradient by autograd through its forward and backward loops
    (hmm_layer/MsaHMMLayer.py:422-521 with training=True)."""

    @staticmethod
    def forward(ctx, A, pi, E, mode, eps):
        A, pi, E = A.contiguous(), pi.contiguous(), E.contiguous()
        ctx.save_for_backward(A, pi, E)
        ctx.mode, ctx.eps = int(mode), eps
        return engine.posterior(A, pi, E, mode=mode, eps=eps)[0]

    @staticmethod
    def backward(ctx, grad_out):
        A, pi, E = ctx.saved_tensors
        grad_out = grad_out.to(torch.float32).contiguous()
        if ctx.mode == engine.POST_LOG_NO_LL:
            # out = log gamma + loglik (the reference's no_loglik=True): the log-posterior gradient plus the
            # log-likelihood gradient weighted by the per-sequence sum of the upstream gradient
            dA, dpi, dE = engine.posterior_grad(A, pi, E, grad_out, mode=engine.POST_LOG, eps=ctx.eps)
            w = grad_out.sum(dim=(2, 3)).contiguous()
            dA2, dpi2, dE2, _ = engine.loglik_grad(A, pi, E, w, eps=ctx.eps)
            dA, dpi, dE = dA + dA2, dpi + dpi2, dE.add_(dE2)
        else:
            dA, dpi, dE = engine.posterior_grad(A, pi, E, grad_out, mode=ctx.mode, eps=ctx.eps)
        need = ctx.needs_input_grad
        return (dA if need[0] else None, dpi.reshape(pi.shape) if need[1] else None, dE if need[2] else None, None, None)


def posterior(A, pi, E, mode=engine.POST_LOG, eps=engine.EPS):
    """Differentiable state posteriors; mode engine.POST_PROB, POST_LOG or POST_LOG_NO_LL."""
    return Posterior.apply(A, pi, E, mode, eps)
