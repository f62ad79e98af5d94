"""Randomised parity sweep on the GPU: posterior, log-likelihood, Viterbi (bit-exact) and
log-likelihood gradients against the C / numpy oracles over random models (gene topology, dense,
sparse-irreducible), shapes and forced chunk lengths — many chunk / group combinations of the
two-level scan and ragged tails that the hand-picked cases do not enumerate."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def test_random_shapes_models_and_chunk_lengths():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stress_sweep.py")
    spec = importlib.util.spec_from_file_location("hmm_stress", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(50, 2024, verbose=False) == 0
