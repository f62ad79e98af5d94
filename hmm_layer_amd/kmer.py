"""k-mer encodings of one-hot nucleotide sequences (drop-in for the reference's
hmm_layer/kmer.py:3-65; semantics of the TF original recorded in tests/test_tf.ipynb).

``make_k_mers`` never modifies its argument.  The reference's torch port adds the N mass to
a view of the caller's tensor (kmer.py:23-25, defect D5 in SURVEY.md), so its emitter's
second call sees doubled N mass; ``n_mass=2`` reproduces that for bug-compatibility tests.
"""
import torch
import torch.nn.functional as F


def make_k_mers(sequences, k, pivot_left=True, n_mass=1):
    """(b, L, 5) one-hot nucleotides, last class = N  ->  (b, L, 4**(k-1), 4).

    The last axis is the left-most (pivot_left) or right-most nucleotide of the k-mer that
    starts (pivot_left) or ends at each position; the other k-1 nucleotides index axis -2.
    N spreads its mass uniformly over A, C, G, T; k-mers reaching over the sequence border
    see uniform padding."""
    L = sequences.shape[-2]
    n = sequences.shape[-1] - 1
    is_n = (sequences[..., -1:] == 1).to(sequences.dtype)
    acgt = sequences[..., :-1] + (n_mass / n) * is_n
    pad = torch.full_like(acgt[:, :k - 1, :], 1.0 / n)
    if pivot_left:
        ext = torch.cat([acgt, pad], dim=-2)
        offsets = range(1, k)
        out = ext[:, :L, None, :]
    else:
        ext = torch.cat([pad, acgt], dim=-2)
        offsets = range(k - 2, -1, -1)
        out = ext[:, k - 1:L + k - 1, None, :]
    for step, i in enumerate(offsets, start=1):
        nxt = ext[:, i:L + i, None, :, None]
        out = (out[..., None, :] * nxt).reshape(*out.shape[:-2], 4 ** step, 4)
    return out


def encode_kmer_string(kmer, pivot_left=True, alphabet="ACGT"):
    """'ACG', 'NGT', ... -> (4**(k-1), 4) class probabilities (N = uniform)."""
    symbols = alphabet + "N"
    idx = torch.tensor([symbols.index(c) for c in kmer])
    one_hot = F.one_hot(idx, num_classes=len(symbols)).to(torch.float32)
    enc = make_k_mers(one_hot.unsqueeze(0), k=len(kmer), pivot_left=pivot_left)[0]
    return enc[0] if pivot_left else enc[-1]
