"""Emission-probability producers of the gene-prediction HMMs (drop-in for the reference's
hmm_layer/gene_pred_hmm_emitter.py): class predictions (and optionally nucleotides) in,
E (k, b, L, q) probabilities out — the tensor the HIP engine consumes.

Interface kept (gene_pred_hmm_emitter.py:61-128, 231-277): ``build(input_shape)``,
``recurrent_init()``, ``make_B()``, ``forward(inputs, end_hints=None, training=False)``,
``get_prior_log_density()``, ``get_aux_loss()``, ``get_config()/from_config()``, parameter
``emission_kernel`` (k, rows, s).

Differences, on purpose: everything is created on / follows the parameters' device; the k-mer
helper does not mutate its input (defect D5 — ``n_mass_compat=True`` reproduces the as-shipped
doubling of N mass in the right-pivot 3-mers); embedding emissions (``emit_embeddings``) are
not provided (they depend on the reference's MvnMixture, upstream of the hot path).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import kmer


class SimpleGenePredHMMEmitter(nn.Module):
    """Class-probability emissions for the 1 + 6*copies state model (Ir, I0-2, E0-2)."""

    def __init__(self, num_models=1, num_copies=1, init=0.0, trainable_emissions=True, emit_embeddings=False,
                 embedding_dim=None, full_covariance=False, embedding_kernel_init="random_normal",
                 initial_variance=1.0, temperature=1.0, share_intron_parameters=True, **kwargs):
        super().__init__(**kwargs)
        if emit_embeddings:
            raise NotImplementedError("embedding emissions (MvnMixture) are outside the engine's scope")
        assert embedding_dim is None, "embedding_dim requires emit_embeddings=True"
        self.num_models = num_models
        self.num_copies = num_copies
        self.num_states = 1 + 6 * num_copies
        self.init = init
        self.trainable_emissions = trainable_emissions
        self.emit_embeddings = False
        self.embedding_dim = None
        self.full_covariance = full_covariance
        self.embedding_kernel_init = embedding_kernel_init
        self.initial_variance = initial_variance
        self.temperature = temperature
        self.share_intron_parameters = share_intron_parameters
        self.emission_kernel = None
        self.B = None
        self.built = False

    def kernel_rows(self):
        return self.num_states - 2 * self.num_copies * int(self.share_intron_parameters)

    def build(self, input_shape):
        if self.built:
            return
        s = input_shape[-1]
        if torch.is_tensor(self.init):
            start = self.init.detach().clone().to(torch.float32).reshape(self.num_models, self.kernel_rows(), s)
        else:
            start = torch.full((self.num_models, self.kernel_rows(), s), float(self.init))
        self.emission_kernel = nn.Parameter(start, requires_grad=self.trainable_emissions)
        self.built = True

    def recurrent_init(self):
        self.B = self.make_B()

    def make_B(self):
        return F.softmax(self.emission_kernel, dim=-1)

    def class_emissions(self, inputs):
        """(k, b, L, s) class probabilities -> (k, b, L, q)."""
        if self.B is None:
            self.recurrent_init()
        emit = torch.einsum("...s,kqs->k...q", inputs[0], self.B)
        if self.share_intron_parameters:
            c = self.num_copies
            emit = torch.cat([emit[..., :1 + c], emit[..., 1:1 + c], emit[..., 1:1 + c], emit[..., 1 + c:]], dim=-1)
        return emit

    def apply_end_hints(self, emit, end_hints):
        if end_hints is None:
            return emit
        left = end_hints[..., :1, :] * emit[..., :1, :]
        right = end_hints[..., 1:, :] * emit[..., -1:, :]
        return torch.cat([left, emit[..., 1:-1, :], right], dim=-2)

    def forward(self, inputs, end_hints=None, training=False):
        return self.apply_end_hints(self.class_emissions(inputs), end_hints)

    def get_prior_log_density(self):
        dev = self.emission_kernel.device if self.emission_kernel is not None else None
        return torch.zeros((1, 1), device=dev)

    def get_aux_loss(self):
        return 0.0

    def get_config(self):
        return {"num_models": self.num_models, "num_copies": self.num_copies, "init": self.init,
                "trainable_emissions": self.trainable_emissions, "emit_embeddings": self.emit_embeddings,
                "embedding_dim": self.embedding_dim, "full_covariance": self.full_covariance,
                "embedding_kernel_init": self.embedding_kernel_init, "initial_variance": self.initial_variance,
                "temperature": self.temperature, "share_intron_parameters": self.share_intron_parameters}

    @classmethod
    def from_config(cls, config):
        return cls(**config)


def assert_codons(codons):
    assert sum(p for _, p in codons) == 1, "codon probabilities must sum to 1: %s" % (codons,)
    for triplet, prob in codons:
        assert len(triplet) == 3 and 0 <= prob <= 1, "bad codon entry: %s" % (codons,)


def make_codon_probs(codons, pivot_left):
    """[(triplet, prob)] -> (1, 1, 64) distribution over 3-mer classes."""
    assert_codons(codons)
    acc = sum(prob * kmer.encode_kmer_string(tri, pivot_left) for tri, prob in codons)
    return acc.reshape(1, 1, 64)


class GenePredHMMEmitter(SimpleGenePredHMMEmitter):
    """1 + 14*copies states: adds START, EI0-2, IE0-2, STOP and 3-mer (codon / splice-site)
    constraints on states E2 .. STOP (gene_pred_hmm_emitter.py:198-217)."""

    def __init__(self, start_codons, stop_codons, intron_begin_pattern, intron_end_pattern, l2_lambda=0.01,
                 nucleotide_kernel_init=None, trainable_nucleotides_at_exons=False, n_mass_compat=False,
                 **kwargs):
        super().__init__(**kwargs)
        self.num_states = 1 + 14 * self.num_copies
        self.start_codons, self.stop_codons = start_codons, stop_codons
        self.intron_begin_pattern, self.intron_end_pattern = intron_begin_pattern, intron_end_pattern
        self.l2_lambda = l2_lambda
        self.nucleotide_kernel_init = nucleotide_kernel_init
        self.trainable_nucleotides_at_exons = trainable_nucleotides_at_exons
        self.n_mass_compat = n_mass_compat
        start = make_codon_probs(start_codons, True)
        stop = make_codon_probs(stop_codons, False)
        ibeg = make_codon_probs(intron_begin_pattern, True)
        iend = make_codon_probs(intron_end_pattern, False)
        anyc = make_codon_probs([("NNN", 1.0)], False)
        not_stop = anyc * (stop == 0).float()
        not_stop = not_stop / not_stop.sum()
        # constrained states in order: E2, START, EI0, EI1, EI2, IE0, IE1, IE2, STOP
        left = [anyc, start, ibeg, ibeg, ibeg, anyc, anyc, anyc, anyc]
        right = [not_stop, anyc, anyc, not_stop, anyc, iend, iend, iend, stop]
        self.start_codon_probs, self.stop_codon_probs = start, stop
        self.intron_begin_codon_probs, self.intron_end_codon_probs = ibeg, iend
        self.any_codon_probs, self.not_stop_codon_probs = anyc, not_stop
        self.register_buffer("codon_probs", torch.cat([torch.cat(left, dim=1), torch.cat(right, dim=1)], dim=0),
                             persistent=False)           # (2, 9, 64)
        self.nuc_emission_kernel = None

    def build(self, input_shape):
        if self.built:
            return
        super().build(input_shape)
        if self.trainable_nucleotides_at_exons:
            assert self.num_models == 1, "trainable nucleotide emissions support one model"
            self.nuc_emission_kernel = nn.Parameter(torch.zeros(self.num_models, 3 * self.num_copies, 4))

    def get_nucleotide_probs(self):
        return torch.softmax(self.nuc_emission_kernel, dim=-1)

    def codon_emissions(self, nucleotides):
        """(k, b, L, 5) one-hot nucleotides -> (k, b, L, q) factor: 1/4096 for the first
        1 + 5*copies states, left x right 3-mer compatibility for the constrained ones."""
        k, b, L = nucleotides.shape[:3]
        flat = nucleotides.reshape(-1, L, 5)
        left = kmer.make_k_mers(flat, 3, True).reshape(k, b, L, 64)
        right = kmer.make_k_mers(flat, 3, False, n_mass=2 if self.n_mass_compat else 1).reshape(k, b, L, 64)
        tab = self.codon_probs.to(nucleotides.dtype)
        cod = torch.einsum("kbls,qs->kblq", left, tab[0]) * torch.einsum("kbls,qs->kblq", right, tab[1])
        if self.num_copies > 1:
            cod = cod.repeat_interleave(self.num_copies, dim=-1)
        free = torch.full_like(cod[..., :1], 1.0 / 4096.0).expand(*cod.shape[:-1], 1 + 5 * self.num_copies)
        return torch.cat([free, cod], dim=-1)

    def forward(self, inputs, end_hints=None, training=False):
        """(k, b, L, s + 5): class probabilities then one-hot nucleotides -> E (k, b, L, q)."""
        nucleotides, classes = inputs[..., -5:], inputs[..., :-5]
        emit = super().forward(classes, end_hints=end_hints, training=training)
        cod = self.codon_emissions(nucleotides)
        if training:
            cod = cod + 1e-7
        full = emit * cod
        if self.trainable_nucleotides_at_exons:
            acgt = nucleotides[..., :4] + nucleotides[..., 4:] / 4
            c = self.num_copies
            nuc = torch.einsum("k...s,kqs->k...q", acgt, self.get_nucleotide_probs())
            quarter = torch.full_like(full[..., :1], 0.25)
            nuc = torch.cat([quarter.expand(*full.shape[:-1], 1 + 3 * c), nuc,
                             quarter.expand(*full.shape[:-1], full.shape[-1] - 1 - 6 * c)], dim=-1)
            full = full * nuc
        return full

    # -- fused inference path (HIP kernel hmm_gene_emissions) ---------------------------------
    def can_fuse(self, inputs):
        return (inputs.is_cuda and inputs.shape[0] == 1 and self.num_models == 1
                and not self.trainable_nucleotides_at_exons and self.built)

    def state_tables(self, device):
        """(state -> kernel row, state -> codon-table row or -1) as int32 tensors."""
        c = self.num_copies
        n_free = 1 + 5 * c
        if self.share_intron_parameters:      # rows: Ir, I(c), E0..E2 .., states: Ir, I0, I1, I2, rest
            rows = list(range(1 + c)) + list(range(1, 1 + c)) * 2 + list(range(1 + c, self.kernel_rows()))
        else:
            rows = list(range(self.num_states))
        cod = [-1] * n_free + [i // c for i in range(self.num_states - n_free)]
        return (torch.tensor(rows, dtype=torch.int32, device=device),
                torch.tensor(cod, dtype=torch.int32, device=device))

    def forward_fused(self, inputs, end_hints=None, training=False):
        """Same values as forward() (inference, one model) without the (b,L,64) 3-mer tensors:
        one HIP kernel from class probabilities + nucleotides to E."""
        from . import engine
        with torch.no_grad():
            if self.B is None:
                self.recurrent_init()
            row, cod = self.state_tables(inputs.device)
            E = engine.gene_emissions(inputs[0].to(torch.float32).contiguous(), self.B[0].to(torch.float32).contiguous(),
                                      row, self.codon_probs.to(inputs.device, torch.float32).contiguous(), cod,
                                      add=1e-7 if training else 0.0, n_mass=2 if self.n_mass_compat else 1)
            return self.apply_end_hints(E.unsqueeze(0), end_hints)

    def get_config(self):
        config = super().get_config()
        config.update({"start_codons": self.start_codons, "stop_codons": self.stop_codons,
                       "intron_begin_pattern": self.intron_begin_pattern,
                       "intron_end_pattern": self.intron_end_pattern, "l2_lambda": self.l2_lambda,
                       "nucleotide_kernel_init": self.nucleotide_kernel_init,
                       "trainable_nucleotides_at_exons": self.trainable_nucleotides_at_exons,
                       "n_mass_compat": self.n_mass_compat})          # D5 compatibility switch
        return config
