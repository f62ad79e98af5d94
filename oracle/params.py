"""CPU restatement of the boundary producers of the gene-prediction HMM:
transition matrix A, start distribution pi, emission tensor E, k-mer encoding.

TEST INFRASTRUCTURE (see oracle/__init__.py).  fp32 torch-CPU, pinned against
fixtures captured from the imported reference and against the k-mer outputs the
reference's tests/test_tf.ipynb records.

Follows:
  * edge lists ............ hmm_layer/gene_pred_hmm_transitioner.py:132-148, 200-221, 279-303
  * logit initialisation .. hmm_layer/gene_pred_hmm_transitioner.py:150-170
  * logits -> dense A ..... hmm_layer/Transitioner.py:337-380
  * pi .................... hmm_layer/gene_pred_hmm_transitioner.py:111-112
  * class emissions ....... hmm_layer/gene_pred_hmm_emitter.py:87-121
  * codon constraints ..... hmm_layer/gene_pred_hmm_emitter.py:154-217, 231-277
  * k-mers ................ hmm_layer/kmer.py:3-65
"""
import numpy as np
import torch

APPROX_LOG_ZERO = -1000.0


# ------------------------------------------------------------------ topology

def edges_simple():
    """7 states Ir,I0,I1,I2,E0,E1,E2; 15 edges, reference order."""
    Ir, I, E = 0, [1, 2, 3], [4, 5, 6]
    ed = [(Ir, Ir), (Ir, E[0]), (E[2], Ir)]
    for c in range(3):
        ed += [(E[c], E[(c + 1) % 3]), (E[c], I[c]), (I[c], I[c]), (I[c], E[(c + 1) % 3])]
    return np.asarray(ed, dtype=np.int64)


def edges_multi(k=1):
    """1+14k states Ir, I*3k, E*3k, START*k, EI*3k, IE*3k, STOP*k; 1+22k edges."""
    Ir = 0
    I = list(range(1, 1 + 3 * k))
    E = list(range(1 + 3 * k, 1 + 6 * k))
    START = list(range(1 + 6 * k, 1 + 7 * k))
    EI = list(range(1 + 7 * k, 1 + 10 * k))
    IE = list(range(1 + 10 * k, 1 + 13 * k))
    STOP = list(range(1 + 13 * k, 1 + 14 * k))
    ed = [(Ir, Ir)]
    for h in range(k):
        ed += [(Ir, START[h]), (STOP[h], Ir), (START[h], E[k + h]), (E[k + h], STOP[h])]
        for c in range(3):
            s = k * c + h
            ed += [(E[s], E[k * ((c + 1) % 3) + h]), (E[s], EI[s]), (EI[s], I[s]),
                   (I[s], I[s]), (I[s], IE[s]), (IE[s], E[s])]
    return np.asarray(ed, dtype=np.int64)


def edges_15():
    """15-state single-copy model in GenePredHMMTransitioner's own edge order."""
    Ir, I, E, START, EI, IE, STOP = 0, [1, 2, 3], [4, 5, 6], 7, [8, 9, 10], [11, 12, 13], 14
    ed = [(Ir, Ir), (Ir, START), (STOP, Ir), (START, E[1]), (E[1], STOP)]
    for c in range(3):
        ed += [(E[c], E[(c + 1) % 3]), (E[c], EI[c]), (EI[c], I[c]),
               (I[c], I[c]), (I[c], IE[c]), (IE[c], E[c])]
    return np.asarray(ed, dtype=np.int64)


def init_logits(edges, k=1, exon_len=100, intron_len=10000, ir_len=10000, sd=0.0, rng=None):
    """Length-based logit initialisation per edge class."""
    rng = np.random if rng is None else rng
    ex0 = 1 + 3 * k
    out = []
    for (u, v) in edges:
        if u == v and u == 0:
            p = 1 - 1.0 / ir_len
            out.append(-np.log(1 / p - 1))
        elif u == v and 0 < u < 1 + 3 * k:
            p = 1 - 1.0 / intron_len
            out.append(-np.log(1 / p - 1))
        elif ex0 <= u < ex0 + 3 * k and v - ex0 == (u - ex0 + k) % (3 * k):
            p = 1 - 1.0 / exon_len
            out.append(-np.log(1 / p - 1))
        elif 1 + 4 * k <= u < 1 + 5 * k and u != v:
            out.append(np.log(1.0 / 2))
        elif u == 0 and v != 0:
            out.append(np.log(1.0 / k) + rng.normal(0.0, sd))
        else:
            out.append(0.0)
    return np.asarray(out)


def dense_A(edges, logits, q, zero_logit_is_absent=False):
    """Edge logits -> row-stochastic dense A (q,q), fp32.

    softmax over a dense logit matrix whose absent entries are -1000, then
    +1e-16, mask, renormalise.  ``zero_logit_is_absent=True`` reproduces the
    as-shipped behaviour in which an explicit 0.0 logit is indistinguishable from
    a missing edge (defect D1)."""
    logits = torch.as_tensor(logits, dtype=torch.float32).reshape(-1)
    order = np.argsort([u * q + v for u, v in edges], kind="stable")
    ed = edges[order]
    val = torch.maximum(logits[order], torch.tensor(APPROX_LOG_ZERO + 1.0))
    dense = torch.full((q, q), APPROX_LOG_ZERO, dtype=torch.float32)
    present = torch.zeros((q, q), dtype=torch.bool)
    dense[ed[:, 0], ed[:, 1]] = val
    present[ed[:, 0], ed[:, 1]] = True
    if zero_logit_is_absent:
        gone = present & (dense == 0)
        dense[gone] = APPROX_LOG_ZERO
    probs = torch.nn.functional.softmax(dense, dim=-1)
    mask = (dense > APPROX_LOG_ZERO).float()
    probs = probs + 1e-16
    probs = probs * mask
    probs = probs / (torch.sum(probs, dim=-1, keepdim=True) + 1e-16)
    return probs


def start_distribution(kernel):
    return torch.nn.functional.softmax(torch.as_tensor(kernel, dtype=torch.float32), dim=-1)


# ------------------------------------------------------------------ k-mers

def make_k_mers(seq, k, pivot_left=True, n_mass=1.0):
    """One-hot (b,L,5) nucleotides (last class = N) -> (b,L,4**(k-1),4) k-mer
    probabilities; N spreads uniformly; k-mers crossing the border use 1/4 padding.
    Does not modify its input.  The reference mutates it (defect D5), so a second
    call on the same tensor sees N rows carrying twice the mass: ``n_mass=2``
    reproduces that."""
    seq = torch.as_tensor(seq)
    L = seq.shape[-2]
    n = torch.tensor(seq.shape[-1] - 1, dtype=seq.dtype)
    is_n = (seq[..., -1:] == 1).to(seq.dtype)
    acgt = seq[..., :-1] + (1 / n) * is_n
    for _ in range(int(n_mass) - 1):
        acgt = acgt + (1 / n) * is_n
    pad = torch.ones_like(acgt[:, :k - 1, :]) / n
    if pivot_left:
        padded = torch.cat([acgt, pad], dim=-2)
        km = padded[:, :L, None, :]
        steps = range(1, k)
    else:
        padded = torch.cat([pad, acgt], dim=-2)
        km = padded[:, k - 1:L + k - 1, None, :]
        steps = range(k - 2, -1, -1)
    for i in steps:
        nxt = padded[:, i:L + i, None, :, None]
        km = km[..., None, :] * nxt
        shape = [4 ** i, 4] if pivot_left else [4 ** (k - i - 1), 4]
        km = km.reshape(list(km.shape[:-3]) + shape)
    return km


def encode_kmer_string(s, pivot_left=True, alphabet="ACGT"):
    idx = torch.tensor([(alphabet + "N").index(c) for c in s])
    oh = torch.nn.functional.one_hot(idx, num_classes=len(alphabet) + 1).to(torch.float32)
    enc = make_k_mers(oh.unsqueeze(0), k=len(s), pivot_left=pivot_left).squeeze(0)
    return enc[0] if pivot_left else enc[-1]


# ------------------------------------------------------------------ emissions

def codon_table(start_codons, stop_codons, intron_begin, intron_end):
    """(2, 9, 64): left / right 3-mer constraints for states E2, START, EI0-2, IE0-2, STOP."""
    def probs(codons, left):
        v = sum(pr * encode_kmer_string(tri, left) for tri, pr in codons)
        return v.reshape(64).unsqueeze(0).unsqueeze(0)
    start = probs(start_codons, True)
    stop = probs(stop_codons, False)
    ibeg = probs(intron_begin, True)
    iend = probs(intron_end, False)
    anyc = probs([("NNN", 1.0)], False)
    notstop = anyc * (stop == 0).float()
    notstop = notstop / notstop.sum()
    left = torch.cat([anyc, start] + [ibeg] * 3 + [anyc] * 4, dim=1)
    right = torch.cat([notstop] + [anyc] * 2 + [notstop, anyc] + [iend] * 3 + [stop], dim=1)
    return torch.cat([left, right], dim=0)


def class_emissions(x, kernel, copies=1, share_intron=True):
    """x (k,b,L,s) class probabilities, kernel (k, rows, s) -> (k,b,L,q)."""
    B = torch.nn.functional.softmax(torch.as_tensor(kernel, dtype=torch.float32), dim=-1)
    emit = torch.einsum("...s,kqs->k...q", x[0], B)
    if share_intron:
        emit = torch.cat([emit[..., :1 + copies]] + [emit[..., 1:1 + copies]] * 2
                         + [emit[..., 1 + copies:]], dim=-1)
    return emit


def gene_emissions(x, kernel, table, copies=1, share_intron=True, training=False, d5_compat=False):
    """x (k,b,L,s+5): class probabilities followed by one-hot nucleotides.
    Returns E (k,b,L,1+14*copies).  ``d5_compat`` reproduces the as-shipped doubling of
    N mass in the right-pivot 3-mers (defect D5: the left-pivot call mutates the input)."""
    nuc = x[..., -5:]
    emit = class_emissions(x[..., :-5], kernel, copies, share_intron)
    k, b, L = nuc.shape[:3]
    nuc = nuc.reshape(-1, L, 5)
    left = make_k_mers(nuc, 3, True).reshape(k, b, L, 64)
    right = make_k_mers(nuc, 3, False, n_mass=2.0 if d5_compat else 1.0).reshape(k, b, L, 64)
    both = torch.stack([left, right], dim=-2)
    cod = torch.einsum("k...rs,rqs->k...rq", both, table).prod(dim=-2)
    if copies > 1:
        cod = cod.repeat_interleave(torch.tensor([copies] * cod.shape[-1]), dim=-1)
    cod = torch.cat([torch.ones_like(cod[..., :1 + 5 * copies]) / 4096.0, cod], dim=-1)
    if training:
        cod = cod + 1e-7
    return emit * cod


DEFAULT_CODONS = dict(
    start_codons=[("ATG", 1.0)],
    stop_codons=[("TAG", 0.34), ("TAA", 0.33), ("TGA", 0.33)],
    intron_begin=[("NGT", 0.99), ("NGC", 0.005), ("NAT", 0.005)],
    intron_end=[("AGN", 0.99), ("ACN", 0.01)],
)


def intended_A15(exon_len=200, intron_len=4500, ir_len=10000):
    """The 23-edge 15-state matrix of SURVEY.md section 8(c) (tests/parallel_rnn_forward.py:35-40
    kwargs, init_component_sd irrelevant for k=1 only through the Ir->START noise,
    which softmax over a single non-loop edge... is not: use sd=0 here)."""
    ed = edges_multi(1)
    lg = init_logits(ed, 1, exon_len, intron_len, ir_len, sd=0.0)
    lg = np.where(lg == 0, 1e-30, lg)
    return dense_A(ed, lg, 15)
