"""Transition-matrix producers of the gene-prediction HMMs (drop-in for the reference's
hmm_layer/gene_pred_hmm_transitioner.py).

Interface kept from the reference (gene_pred_hmm_transitioner.py:66-129): ``recurrent_init``,
``make_A`` -> (k,q,q), ``make_log_A``, ``make_initial_distribution`` -> (1,k,q),
``forward(x)`` = x @ A (or x @ A^T when ``reverse``), ``get_prior_log_densities``,
``get_config`` / ``from_config``; parameters ``transition_kernel`` (1, edges) and
``starting_distribution_kernel`` (1,1,q).

Differences, on purpose:
  * the dense matrix is built directly on the parameters' device with one scatter + masked
    softmax (no sparse tensors, no host round trip), so it can feed the HIP engine without a sync;
  * an explicit 0.0 logit is a real edge.  The reference drops such edges (its dense
    conversion cannot tell 0.0 from "absent", Transitioner.py:366-367 — defect D1 in SURVEY.md),
    which leaves rows 7-14 of the freshly initialised 15-state matrix empty;
    ``zero_logit_is_absent=True`` reproduces that behaviour bit for bit;
  * ``make_log_A`` returns log A (absent edges = -1000) as Viterbi needs it; the reference
    applies a sparse log-softmax to probabilities there (defect D10).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

APPROX_LOG_ZERO = -1000.0


def dense_transition_matrix(src, dst, logits, num_states, zero_logit_is_absent=False):
    """Edge logits -> row-stochastic (q,q) matrix: softmax over each state's outgoing edges
    (the reference's make_transition_matrix_from_indices, Transitioner.py:337-380)."""
    logits = torch.clamp_min(logits.reshape(-1), APPROX_LOG_ZERO + 1.0)
    dense = torch.full((num_states, num_states), APPROX_LOG_ZERO, dtype=logits.dtype, device=logits.device)
    dense = dense.index_put((src, dst), logits)
    if zero_logit_is_absent:
        dense = torch.where(dense == 0, torch.full_like(dense, APPROX_LOG_ZERO), dense)
    mask = (dense > APPROX_LOG_ZERO).to(logits.dtype)
    probs = (F.softmax(dense, dim=-1) + 1e-16) * mask
    return probs / (probs.sum(dim=-1, keepdim=True) + 1e-16)


class SimpleGenePredHMMTransitioner(nn.Module):
    """7 states Ir, I0, I1, I2, E0, E1, E2 and 15 edges."""

    def __init__(self, num_models=1, initial_exon_len=100, initial_intron_len=10000, initial_ir_len=10000,
                 init=None, starting_distribution_init="zeros", starting_distribution_trainable=True,
                 transitions_trainable=True, init_component_sd=0, zero_logit_is_absent=False, **kwargs):
        super().__init__(**kwargs)
        self.num_models = num_models
        self.initial_exon_len = initial_exon_len
        self.initial_intron_len = initial_intron_len
        self.initial_ir_len = initial_ir_len
        if not hasattr(self, "num_states"):
            self.num_states = 7
        if not hasattr(self, "k"):
            self.k = 1
        self.zero_logit_is_absent = zero_logit_is_absent
        self.indices = self.make_transition_indices()
        self.num_transitions = len(self.indices)
        self.starting_distribution_init = starting_distribution_init
        self.starting_distribution_trainable = starting_distribution_trainable
        self.transitions_trainable = transitions_trainable
        self.reverse = False
        self.init = self.make_transition_init(1, init_component_sd) if init is None else init
        self.register_buffer("_src", torch.as_tensor(self.indices[:, 1]), persistent=False)
        self.register_buffer("_dst", torch.as_tensor(self.indices[:, 2]), persistent=False)
        self.transition_kernel = nn.Parameter(
            torch.as_tensor(np.asarray(self.init), dtype=torch.float32).reshape(1, -1),
            requires_grad=transitions_trainable)
        start = torch.zeros if starting_distribution_init == "zeros" else torch.ones
        self.starting_distribution_kernel = nn.Parameter(start(1, 1, self.num_states),
                                                         requires_grad=starting_distribution_trainable)
        self.A = None
        self.A_transposed = None

    # -- topology ---------------------------------------------------------------------
    def make_transition_indices(self, model_index=0):
        """(model, from, to) triples, reference order (gene_pred_hmm_transitioner.py:132-148)."""
        intron, exon = [1, 2, 3], [4, 5, 6]
        edges = [(0, 0), (0, exon[0]), (exon[2], 0)]
        for c in range(3):
            nxt = exon[(c + 1) % 3]
            edges += [(exon[c], nxt), (exon[c], intron[c]), (intron[c], intron[c]), (intron[c], nxt)]
        return _with_model(edges, model_index, 15)

    def edge_class(self, edge, k=1):
        """'ir_loop' | 'intron_loop' | 'exon_next' | 'exon1_out' | 'ir_out' | 'other'
        (the predicates of gene_pred_hmm_transitioner.py:44-64)."""
        _, u, v = (int(x) for x in edge)
        first_exon = 1 + 3 * k
        if u == v == 0:
            return "ir_loop"
        if u == v and 0 < u < first_exon:
            return "intron_loop"
        if first_exon <= u < first_exon + 3 * k and v - first_exon == (u - first_exon + k) % (3 * k):
            return "exon_next"
        if 1 + 4 * k <= u < 1 + 5 * k and u != v:
            return "exon1_out"
        if u == 0 and v != 0:
            return "ir_out"
        return "other"

    def make_transition_init(self, k=1, sd=0.05):
        """Logits from expected segment lengths (gene_pred_hmm_transitioner.py:150-170)."""
        def stay(length):
            p = 1.0 - 1.0 / length
            return -np.log(1.0 / p - 1.0)
        table = {"ir_loop": lambda: stay(self.initial_ir_len),
                 "intron_loop": lambda: stay(self.initial_intron_len),
                 "exon_next": lambda: stay(self.initial_exon_len),
                 "exon1_out": lambda: np.log(0.5),
                 "ir_out": lambda: np.log(1.0 / k) + np.random.normal(0.0, sd),
                 "other": lambda: 0.0}
        return np.array([table[self.edge_class(e, k)]() for e in self.indices])

    # -- matrices ---------------------------------------------------------------------
    def recurrent_init(self):
        self.A = self.make_A()
        self.A_transposed = torch.transpose(self.A, 1, 2)

    def make_A(self):
        A = dense_transition_matrix(self._src, self._dst, self.transition_kernel, self.num_states,
                                    self.zero_logit_is_absent)
        return A.unsqueeze(0).repeat(self.num_models, 1, 1)

    def make_A_sparse(self, values=None):
        """Sparse COO view (1,q,q) of the edge probabilities, row-major edge order."""
        vals = self.transition_kernel.reshape(-1) if values is None else values.reshape(-1)
        A = dense_transition_matrix(self._src, self._dst, vals, self.num_states, self.zero_logit_is_absent)
        order = np.argsort(self.indices[:, 1] * self.num_states + self.indices[:, 2], kind="stable")
        idx = torch.as_tensor(self.indices[order].T, device=A.device)
        return torch.sparse_coo_tensor(idx, A[idx[1], idx[2]], size=(1, self.num_states, self.num_states))

    def make_log_A(self):
        A = self.make_A()
        return torch.where(A > 0, torch.log(A.clamp_min(1e-45)), torch.full_like(A, APPROX_LOG_ZERO))

    def make_initial_distribution(self):
        return F.softmax(self.starting_distribution_kernel, dim=-1).repeat(1, self.num_models, 1)

    def forward(self, inputs):
        """(k, b, q) -> (k, b, q): one transition step (the reference's per-step matmul)."""
        if self.A is None:
            self.recurrent_init()
        return torch.matmul(inputs, self.A_transposed if self.reverse else self.A)

    def get_prior_log_densities(self):
        return {"none": 0.0}

    def get_config(self):
        return {"num_models": self.num_models,
                "initial_exon_len": self.initial_exon_len, "initial_intron_len": self.initial_intron_len,
                "initial_ir_len": self.initial_ir_len,
                "starting_distribution_init": self.starting_distribution_init,
                "starting_distribution_trainable": self.starting_distribution_trainable,
                "transitions_trainable": self.transitions_trainable,
                # D1 compatibility switch: without it a from_config round trip would silently turn a
                # bug-compatible model into one with the intended semantics
                "zero_logit_is_absent": self.zero_logit_is_absent}

    @classmethod
    def from_config(cls, config):
        return cls(**config)


class GenePredHMMTransitioner(SimpleGenePredHMMTransitioner):
    """15 states: the simple model plus START, EI0-2, IE0-2, STOP; 23 edges."""

    def __init__(self, use_experimental_prior=False, **kwargs):
        if not hasattr(self, "num_states"):
            self.num_states = 15
        if not hasattr(self, "k"):
            self.k = 1
        self._sd = kwargs.get("init_component_sd", 0)
        super().__init__(**kwargs)
        self.use_experimental_prior = use_experimental_prior
        if use_experimental_prior:
            self.alpha = self.make_prior_alpha()

    def make_transition_indices(self, model_index=0):
        intron, exon, start, ei, ie, stop = [1, 2, 3], [4, 5, 6], 7, [8, 9, 10], [11, 12, 13], 14
        edges = [(0, 0), (0, start), (stop, 0), (start, exon[1]), (exon[1], stop)]
        for c in range(3):
            edges += [(exon[c], exon[(c + 1) % 3]), (exon[c], ei[c]), (ei[c], intron[c]),
                      (intron[c], intron[c]), (intron[c], ie[c]), (ie[c], exon[c])]
        return _with_model(edges, model_index, 23)

    def gather_binary_probs_for_prior(self, A):
        """(stay, leave) pairs of the looping states and (next exon, leave) of the exon states."""
        m = 1 + 3 * self.k
        diag = torch.diagonal(A[:m, :m])
        loops = torch.stack([diag, A[:m].sum(-1) - diag], dim=1)
        rows = []
        for i in range(3):
            for j in range(self.k):
                e = 1 + (i + 3) * self.k + j
                nxt = 1 + 3 * self.k + ((i + 1) % 3) * self.k + j
                rows.append(torch.stack([A[e, nxt], A[e].sum() - A[e, nxt]]))
        return torch.cat([loops, torch.stack(rows)], dim=0)

    def make_prior_alpha(self, n=1e3):
        logits = torch.as_tensor(self.make_transition_init(self.k, self._sd), dtype=torch.float32,
                                 device=self.transition_kernel.device)
        A0 = dense_transition_matrix(self._src, self._dst, logits, self.num_states, self.zero_logit_is_absent)
        return self.gather_binary_probs_for_prior(A0) * n

    def get_prior_log_densities(self):
        if not self.use_experimental_prior:
            return {"none": 0.0}
        if self.A is None:
            self.recurrent_init()
        probs = self.gather_binary_probs_for_prior(self.A[0])
        pri = torch.sum((self.alpha.to(probs.device) - 1) * torch.log(probs), dim=-1)
        return {i: pri[i].item() for i in range(1 + 6 * self.k)}


class GenePredMultiHMMTransitioner(GenePredHMMTransitioner):
    """k copies of the 14 gene states sharing one intergenic state: 1 + 14k states, 1 + 22k edges."""

    def __init__(self, k=1, init_component_sd=0.2, **kwargs):
        self.k = k
        self.num_states = 1 + 14 * k
        self.init_component_sd = init_component_sd
        super().__init__(**kwargs)
        self.init = self.make_transition_init(k, init_component_sd)

    def make_transition_indices(self, model_index=0):
        k = self.k
        block = lambda first, n: list(range(first, first + n))          # noqa: E731
        intron, exon = block(1, 3 * k), block(1 + 3 * k, 3 * k)
        start, ei = block(1 + 6 * k, k), block(1 + 7 * k, 3 * k)
        ie, stop = block(1 + 10 * k, 3 * k), block(1 + 13 * k, k)
        edges = [(0, 0)]
        for h in range(k):
            edges += [(0, start[h]), (stop[h], 0), (start[h], exon[k + h]), (exon[k + h], stop[h])]
            for c in range(3):
                s = k * c + h
                edges += [(exon[s], exon[k * ((c + 1) % 3) + h]), (exon[s], ei[s]), (ei[s], intron[s]),
                          (intron[s], intron[s]), (intron[s], ie[s]), (ie[s], exon[s])]
        return _with_model(edges, model_index, 1 + 22 * k)

    def get_config(self):
        config = super().get_config()
        config.update({"k": self.k})
        return config


def _with_model(edges, model_index, expected):
    assert len(edges) == expected
    col = np.full((len(edges), 1), model_index, dtype=np.int64)
    return np.concatenate([col, np.asarray(edges, dtype=np.int64)], axis=1)
