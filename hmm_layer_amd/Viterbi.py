"""Most probable state paths for an HmmCell's parameters (the reference only mentions Viterbi in a
docstring, hmm_layer/MsaHmmCell.py:13; learnMSA, which it ports, has a Viterbi module of this name).

``viterbi(inputs, cell)`` materialises log A, log pi and log E exactly as the layer does for the
forward-backward engine and makes one ``hmm_viterbi`` call (include/hmm_engine.h)."""
import torch

from . import engine


def viterbi(inputs, cell, end_hints=None, training=False):
    """inputs (k, b, L, s) on a HIP device -> (path (k,b,L) int32, score (k,b) fp64).

    Emissions are clamped at the cell's epsilon before the log, like the forward recursion does
    (hmm_layer/MsaHmmCell.py:87); absent edges (A == 0) become -inf, which the engine treats as
    its "approximately log zero" (-1024)."""
    from .MsaHMMLayer import _engine_inputs
    A, pi, E = _engine_inputs(inputs, cell, end_hints, training)        # fused emitter when it qualifies
    with torch.no_grad():
        logE = torch.log(torch.clamp_min(E, cell.epsilon))
        del E
        logA = torch.log(A)
        logpi = torch.log(torch.clamp_min(pi, cell.epsilon))
    return engine.viterbi(logA.contiguous(), logpi.contiguous(), logE)
