"""Layer API: log-likelihoods, forward / backward variables and state posteriors of batches of
sequences under k HMMs (drop-in for the reference's hmm_layer/MsaHMMLayer.py).

Every recursion here runs in the HIP engine (hmm_layer_amd.engine -> include/hmm_engine.h):
the layer materialises A, pi and the emission tensor E once (as the reference does,
MsaHMMLayer.py:247-249, 446-452) and makes ONE engine call; the Python time loop, the flips,
stacks and the q x q per-position tensors of the chunked mode do not exist.  Tensors must live
on a HIP device — there is no CPU fallback (the step-at-a-time CPU path is BaseRNN + HmmCell).

Kept from the reference: ``MsaHmmLayer(cell, num_seqs, use_prior, sequence_weights,
parallel_factor)`` with ``build``, ``forward_recursion``, ``backward_recursion``,
``state_posterior_log_probs``, ``apply_sequence_weights``, ``compute_prior``, ``forward``,
``get_config`` / ``from_config``; and the module-level ``_forward_recursion_impl``,
``_backward_recursion_impl``, ``_state_posterior_log_probs_impl`` with their argument lists
(the rnn arguments are accepted and ignored; ``parallel_factor`` is accepted and need not
divide the length — the engine picks its own time chunking).

Training: ``MsaHmmLayer.forward`` is differentiable.  When autograd is recording and a cell
parameter (or the input) requires grad, A, pi and E are built by the cell's torch ops with their
graph and the log-likelihood is ONE autograd node (hmm_layer_amd.autograd.LogLikelihood) whose
backward is the engine's analytic gradient (hmm_loglik_grad) — instead of the reference's
autograd through the unrolled time loop.  ``state_posterior_log_probs`` is differentiable the same
way (hmm_posterior_grad), as training through the posteriors needs, including the ``no_loglik``
variant; forward / backward variables return inference values.
"""
import torch
import torch.nn as nn

from . import autograd, distributed, engine
from .Bidirectional import Bidirectional
from .BaseRNN import BaseRNN
from .TotalProbabilityCell import TotalProbabilityCell


def _engine_inputs(inputs, cell, end_hints, training):
    """cell parameters + raw inputs -> (A (k,q,q), pi (k,q), E (k,b,L,q)) on the inputs' device."""
    cell.recurrent_init()
    with torch.no_grad():
        em = cell.emitter[0] if len(cell.emitter) == 1 else None
        if em is not None and hasattr(em, "forward_fused") and em.can_fuse(inputs):
            E = em.forward_fused(inputs, end_hints=end_hints, training=training)      # HIP kernel
        else:
            E = cell.emission_probs(inputs, end_hints=end_hints, training=training)
        A = cell.A.detach().to(E.device, torch.float32)
        pi = cell.init_dist.detach().to(E.device, torch.float32).reshape(cell.num_models, cell.max_num_states)
    return A.contiguous(), pi.contiguous(), E.detach().to(torch.float32).contiguous()


def _with_prior(cell, result, return_prior):
    if not return_prior:
        return result
    extra = (cell.get_prior_log_density(), cell.get_aux_loss())
    return (*result, *extra) if isinstance(result, tuple) else (result, *extra)


def _forward_recursion_impl(inputs, cell, rnn=None, total_prob_rnn=None, end_hints=None, return_prior=False,
                            training=False, parallel_factor=1):
    """-> log alpha (k,b,L,q), loglik (k,b) [, prior, aux_loss]   (reference MsaHMMLayer.py:227-282)."""
    A, pi, E = _engine_inputs(inputs, cell, end_hints, training)
    log_alpha, loglik = engine.forward(A, pi, E, want_log_alpha=True, eps=cell.epsilon)
    return _with_prior(cell, (log_alpha, loglik.to(torch.float32)), return_prior)


def _backward_recursion_impl(inputs, cell, reverse_cell=None, rnn_backward=None, total_prob_rnn_rev=None,
                             end_hints=None, return_prior=False, training=False, parallel_factor=1):
    """-> log beta (k,b,L,q) [, prior, aux_loss]   (reference MsaHMMLayer.py:322-381)."""
    A, _, E = _engine_inputs(inputs, cell, end_hints, training)
    return _with_prior(cell, engine.backward(A, E, eps=cell.epsilon), return_prior)


def _state_posterior_log_probs_impl(inputs, cell, reverse_cell=None, bidirectional_rnn=None, total_prob_rnn=None,
                                    total_prob_rnn_rev=None, end_hints=None, return_prior=False, training=False,
                                    no_loglik=False, parallel_factor=1):
    """-> log P(state q at position i | inputs), (k,b,L,q) [, prior, aux_loss]
    (reference MsaHMMLayer.py:422-521); with no_loglik the normaliser is left in
    (log alpha + log beta)."""
    if _wants_grad(inputs, cell):
        # training through the posteriors, as the reference's own test does (training=True): one
        # autograd node, analytic backward (hmm_posterior_grad)
        A, pi, E = _graph_inputs(inputs, cell, end_hints, training, what="posterior")
        mode = engine.POST_LOG_NO_LL if no_loglik else engine.POST_LOG
        return _with_prior(cell, autograd.posterior(A, pi, E, mode=mode, eps=cell.epsilon), return_prior)
    A, pi, E = _engine_inputs(inputs, cell, end_hints, training)
    mode = engine.POST_LOG_NO_LL if no_loglik else engine.POST_LOG
    post, _ = engine.posterior(A, pi, E, mode=mode, eps=cell.epsilon)
    return _with_prior(cell, post, return_prior)


def _wants_grad(inputs, cell):
    if not torch.is_grad_enabled():
        return False
    if torch.is_tensor(inputs) and inputs.requires_grad:
        return True
    mods = [cell.transitioner, *cell.emitter]
    return any(p.requires_grad for m in mods if isinstance(m, nn.Module) for p in m.parameters())


def _graph_inputs(inputs, cell, end_hints, training, what="loglik"):
    """A, pi, E built by the cell's torch ops WITH their autograd graph (training)."""
    limit = engine.lib().hmm_grad_max_states() if what == "loglik" else engine.lib().hmm_posterior_grad_max_states()
    if cell.max_num_states > limit:
        # fail before the forward pass, not in backward(): the analytic gradients cover q <= 64
        raise ValueError("training through the HIP engine covers models of at most %d states (got %d); "
                         "wrap inference calls in torch.no_grad()" % (limit, cell.max_num_states))
    cell.recurrent_init()
    E = cell.emission_probs(inputs, end_hints=end_hints, training=training).to(torch.float32)
    if not E.is_cuda:
        raise engine.EngineError("inputs must live on a HIP device (got %s); the engine has no CPU path" % E.device)
    A = cell.A.to(E.device, torch.float32)
    pi = cell.init_dist.to(E.device, torch.float32).reshape(cell.num_models, cell.max_num_states)
    return A, pi, E


def _loglik_impl(inputs, cell, end_hints=None, training=False):
    """loglik (k,b) fp64 only: reads E once, writes nothing per position.  Differentiable when
    autograd is recording and something upstream requires grad."""
    if not _wants_grad(inputs, cell):
        A, pi, E = _engine_inputs(inputs, cell, end_hints, training)
        return engine.forward(A, pi, E, want_log_alpha=False, eps=cell.epsilon)[1]
    A, pi, E = _graph_inputs(inputs, cell, end_hints, training)
    return autograd.loglik(A, pi, E, eps=cell.epsilon)


class MsaHmmLayer(nn.Module):
    def __init__(self, cell, num_seqs=None, use_prior=True, sequence_weights=None, parallel_factor=1):
        super().__init__()
        self.cell = cell
        self.num_seqs = num_seqs
        self.use_prior = use_prior
        self.parallel_factor = parallel_factor
        if sequence_weights is not None:
            w = torch.as_tensor(sequence_weights, dtype=torch.float32)
            self.register_buffer("sequence_weights", w)
            self.register_buffer("weight_sum", w.sum())
        else:
            self.sequence_weights = None
            self.weight_sum = None
        self.reverse_cell = None
        self.rnn = self.rnn_backward = self.bidirectional_rnn = None
        self.total_prob_rnn = self.total_prob_rnn_rev = None
        self.built = False

    def build(self, input_shape=None):
        """Creates the reverse cell and the plumbing-path layers (step-at-a-time drivers over the
        same cells; the engine path does not need them)."""
        if self.built:
            return
        self.reverse_cell = self.cell.make_reverse_direction_offspring()
        self.rnn = BaseRNN(self.cell, batch_first=True, return_sequences=True, return_state=True)
        self.rnn_backward = BaseRNN(self.reverse_cell, batch_first=True, return_sequences=True, return_state=True)
        self.bidirectional_rnn = Bidirectional(self.rnn, merge_mode="concat" if self.parallel_factor > 1 else "sum",
                                               backward_layer=self.rnn_backward)
        if self.parallel_factor > 1:
            self.total_prob_rnn = BaseRNN(TotalProbabilityCell(self.cell), batch_first=True,
                                          return_sequences=True, return_state=True)
            self.total_prob_rnn_rev = BaseRNN(TotalProbabilityCell(self.reverse_cell, reverse=True),
                                              batch_first=True, return_sequences=True, return_state=True,
                                              reverse=True)
        self.built = True

    # -- recursions (engine) --------------------------------------------------------------
    def forward_recursion(self, inputs, end_hints=None, return_prior=False, training=False):
        return _forward_recursion_impl(inputs, self.cell, self.rnn, self.total_prob_rnn, end_hints=end_hints,
                                       return_prior=return_prior, training=training,
                                       parallel_factor=self.parallel_factor)

    def backward_recursion(self, inputs, end_hints=None, return_prior=False, training=False):
        return _backward_recursion_impl(inputs, self.cell, self.reverse_cell, self.rnn_backward,
                                        self.total_prob_rnn_rev, end_hints=end_hints, return_prior=return_prior,
                                        training=training, parallel_factor=self.parallel_factor)

    def state_posterior_log_probs(self, inputs, end_hints=None, return_prior=False, training=False,
                                  no_loglik=False):
        return _state_posterior_log_probs_impl(inputs, self.cell, self.reverse_cell, self.bidirectional_rnn,
                                               self.total_prob_rnn, self.total_prob_rnn_rev, end_hints=end_hints,
                                               return_prior=return_prior, training=training, no_loglik=no_loglik,
                                               parallel_factor=self.parallel_factor)

    def state_posterior_probs(self, inputs, end_hints=None, training=False):
        """Posteriors as probabilities (rows sum to 1) and loglik (k,b) fp64: the engine's native
        output, without the exp/log round trip."""
        if _wants_grad(inputs, self.cell):                  # differentiable, like state_posterior_log_probs
            A, pi, E = _graph_inputs(inputs, self.cell, end_hints, training, what="posterior")
            probs = autograd.posterior(A, pi, E, mode=engine.POST_PROB, eps=self.cell.epsilon)
            return probs, autograd.loglik(A, pi, E, eps=self.cell.epsilon)
        A, pi, E = _engine_inputs(inputs, self.cell, end_hints, training)
        return engine.posterior(A, pi, E, mode=engine.POST_PROB, eps=self.cell.epsilon)

    # -- likelihood aggregation -----------------------------------------------------------
    def apply_sequence_weights(self, loglik, indices, aggregate=False):
        """Weights per sequence; aggregate=True: weighted mean over the batch, then mean over
        models.  Under torch.distributed the means run over ALL ranks' sequences via one
        all-reduce of (sum w*loglik, sum w) per model."""
        weights = None
        if self.sequence_weights is not None:
            weights = self.sequence_weights.to(loglik.device)[indices]
            if not aggregate:
                return loglik * weights
        if not aggregate:
            return loglik
        return distributed.aggregate_loglik(loglik, weights).to(torch.float32)

    def compute_prior(self, scaled=True):
        self.cell.recurrent_init()
        prior = self.cell.get_prior_log_density()
        return self._scale_prior(prior) if scaled else prior

    def _scale_prior(self, prior):
        if self.sequence_weights is not None:
            return prior / self.weight_sum.to(prior.device) if torch.is_tensor(prior) else prior / self.weight_sum
        if self.num_seqs is not None:
            return prior / self.num_seqs
        return prior

    def forward(self, inputs, indices=None, training=False):
        """-> loglik (k,b), aggregated loglik (), [prior (k), aux_loss ()]."""
        inputs = inputs.to(torch.float32)
        loglik64 = _loglik_impl(inputs, self.cell, training=training)
        loglik_mean = torch.squeeze(self.apply_sequence_weights(loglik64, indices, aggregate=True))
        loglik = loglik64.to(torch.float32)
        if self.use_prior:
            prior = self._scale_prior(self.cell.get_prior_log_density())
            return loglik, loglik_mean, prior, self.cell.get_aux_loss()
        return loglik, loglik_mean

    def get_config(self):
        return {"cell": self.cell, "num_seqs": self.num_seqs, "use_prior": self.use_prior,
                "sequence_weights": None if self.sequence_weights is None else self.sequence_weights.cpu().numpy(),
                "parallel_factor": self.parallel_factor}

    @classmethod
    def from_config(cls, config):
        return cls(**config)
