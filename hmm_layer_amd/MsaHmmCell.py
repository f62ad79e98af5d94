"""HMM recurrent cells (drop-in for the reference's hmm_layer/MsaHmmCell.py).

``HmmCell.forward`` is one scaled forward (or backward) DP step written with plain torch ops —
the step-at-a-time API of the reference (MsaHmmCell.py:73-106) that user RNN loops and the
CPU plumbing path (BaseRNN) call.  Whole-sequence recursions do not go through it:
``MsaHmmLayer`` hands A, pi and E to the HIP engine (hmm_layer_amd.engine) instead.

Kept: constructor signature, attributes (num_states, num_models, max_num_states, dim,
emitter (list), transitioner, epsilon = 1e-16, reverse), recurrent_init, emission_probs,
forward, get_initial_state (parallel_factor = 1 and > 1), get_aux_loss,
get_prior_log_density, make_reverse_direction_offspring, reverse_direction.

Changed on purpose (defects in SURVEY.md section 4.3):
  D2  the reverse cell keeps its own direction: it multiplies by A^T itself instead of flipping
      a flag on the transitioner it shares with the forward cell;
  D6  get_initial_state(parallel_factor > 1) of the reverse cell leaves the identity of the
      last chunk intact (the reference scales it by an emission through an aliased view).
"""
import torch
import torch.nn as nn


def get_num_states(lengths):
    """Profile HMM: 2*length + 3 states per model (reference Utility.py:12-14)."""
    return [2 * n + 3 for n in lengths]


class HmmCell(nn.Module):
    def __init__(self, num_states, dim, emitter, transitioner, use_step_counter=False,
                 use_fake_step_counter=False, **kwargs):
        super().__init__(**kwargs)
        self.num_states = num_states
        self.num_models = len(num_states)
        self.max_num_states = max(num_states)
        self.dim = dim
        ems = list(emitter) if isinstance(emitter, (list, tuple, nn.ModuleList)) else [emitter]
        self.emitter = nn.ModuleList(ems) if all(isinstance(e, nn.Module) for e in ems) else ems
        self.transitioner = transitioner
        self.epsilon = 1e-16
        self.reverse = False
        self.use_step_counter = use_step_counter
        self.use_fake_step_counter = use_fake_step_counter
        self.A = self.A_t = self.init_dist = None
        self.recurrent_init()

    # -- per-run setup --------------------------------------------------------------------
    def recurrent_init(self):
        """Re-read the parameters (called before every recursion, like the reference)."""
        self.transitioner.recurrent_init()
        for em in self.emitter:
            em.recurrent_init()
        self.A = self.transitioner.make_A()
        self.A_t = torch.transpose(self.A, 1, 2)
        self.log_A_dense = self.transitioner.make_log_A()
        self.log_A_dense_t = torch.transpose(self.log_A_dense, 1, 2)
        self.init_dist = self.make_initial_distribution()
        if not self.reverse and self.use_step_counter:
            self.step_counter = torch.tensor(-1, dtype=torch.int32)

    def make_initial_distribution(self):
        """(1, num_models, q) start distribution."""
        return self.transitioner.make_initial_distribution()

    def emission_probs(self, inputs, end_hints=None, training=False):
        """Product of all emitters' probabilities, (k, b, L, q)."""
        probs = self.emitter[0](inputs, end_hints=end_hints, training=training)
        for em in self.emitter[1:]:
            probs = probs * em(inputs, end_hints=end_hints, training=training)
        return probs

    # -- one DP step ----------------------------------------------------------------------
    def forward(self, emission_probs, states, training=None, init=False):
        """emission_probs (k*n, q); states = [scaled (k*n, q | q*q), loglik (k*n, 1 | q)].
        Returns (output, new_states); output = [log scaled_forward, loglik] going forward,
        [log R, previous loglik] going backward (R = the vector before the emission)."""
        k, q = self.num_models, self.max_num_states
        scaled, loglik = states
        scaled = scaled.view(k, -1, q)
        R = scaled if init else torch.matmul(scaled, self.A_t if self.reverse else self.A)
        E = emission_probs.view(k, -1, q)
        w = R.shape[1] // E.shape[1]                 # q conditional rows per sequence in chunked mode
        eps = torch.tensor(self.epsilon, dtype=R.dtype, device=R.device)
        R = torch.maximum(R.view(k, -1, w, q), eps)
        E = torch.maximum(E.view(k, -1, 1, q), eps)
        prev = loglik.view(k, -1, w, 1)
        sf = E * R
        S = sf.sum(dim=-1, keepdim=True)
        new_ll = (prev + torch.log(S)).view(-1, w)
        sf = (sf / S).view(-1, w * q)
        if self.reverse:
            out = torch.cat([torch.log(R).view(-1, w * q), prev.view(-1, w)], dim=-1)
        else:
            out = torch.cat([torch.log(sf), new_ll], dim=-1)
        if not self.reverse and self.use_step_counter:
            self.step_counter += 1
        return out, [sf, new_ll]

    def get_initial_state(self, inputs=None, batch_size=None, parallel_factor=1):
        """[start vectors, zero logliks].  parallel_factor > 1: (q x q) conditional starts per
        chunk row — identity for the first (forward) / last (backward) chunk of a sequence,
        A (forward) or diag(first emission of the next chunk) A^T (backward) for the others."""
        k, q = self.num_models, self.max_num_states
        dev, dt = self.A.device, self.A.dtype
        n = batch_size
        if parallel_factor == 1:
            if self.reverse:
                start = torch.ones((k * n, q), dtype=dt, device=dev)
            else:
                start = self.make_initial_distribution().to(dev).repeat(n, 1, 1).transpose(0, 1).reshape(-1, q)
            return [start, torch.zeros((k * n, 1), dtype=dt, device=dev)]
        pf = parallel_factor
        eye = torch.eye(q, dtype=dt, device=dev).expand(k * n, q, q)
        if self.reverse:
            nxt = inputs[:, 0, :].reshape(k, n // pf, pf, q)
            nxt = torch.roll(nxt, shifts=-1, dims=2).reshape(k * n, 1, q)
            moved = torch.matmul((eye * nxt).reshape(k, n * q, q), self.A_t)
        else:
            moved = torch.matmul(eye.reshape(k, n * q, q), self.A)
        moved = moved.reshape(k, n // pf, pf, q * q)
        ident = eye.reshape(k, n // pf, pf, q * q)
        edge = torch.zeros((1, 1, pf, 1), dtype=dt, device=dev)
        edge[:, :, -1 if self.reverse else 0] = 1.0
        start = (edge * ident + (1 - edge) * moved).reshape(k * n, q * q)
        return [start, torch.zeros((k * n, q), dtype=dt, device=dev)]

    def get_aux_loss(self):
        return sum(em.get_aux_loss() for em in self.emitter)

    def get_prior_log_density(self):
        em = [torch.sum(e.get_prior_log_density(), dim=1) for e in self.emitter]
        tr = self.transitioner.get_prior_log_densities()
        return sum(em) + sum(tr.values())

    def make_reverse_direction_offspring(self):
        """A cell over the same parameters that runs the backward recursion."""
        twin = HmmCell(self.num_states, self.dim, list(self.emitter), self.transitioner)
        twin.reverse_direction()
        twin.recurrent_init()
        return twin

    def reverse_direction(self, reverse=True):
        self.reverse = reverse


class MsaHmmCell(HmmCell):
    """Profile-HMM flavour: num_states = 2*length + 3 per model (reference MsaHmmCell.py:164-182).
    The reference's default ProfileHMMEmitter / ProfileHMMTransitioner cannot be constructed
    (SURVEY.md defect D8); pass interface-conforming objects explicitly."""

    def __init__(self, length, dim=24, emitter=None, transitioner=None, **kwargs):
        if emitter is None or transitioner is None:
            raise ValueError("MsaHmmCell needs an explicit emitter and transitioner: the profile-HMM "
                             "parameter producers are outside this engine's scope")
        self.length = [length] if not isinstance(length, (list, tuple)) else list(length)
        super().__init__(get_num_states(self.length), dim, emitter, transitioner, **kwargs)
        for em in self.emitter:
            if hasattr(em, "set_lengths"):
                em.set_lengths(self.length)
        if hasattr(self.transitioner, "set_lengths"):
            self.transitioner.set_lengths(self.length)
