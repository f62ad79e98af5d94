"""hmm_layer_amd — MI355X-native HMM forward / backward / posterior / Viterbi engine behind
the module API of sukui-genomics-cn/hmm_layer (MsaHmmLayer / HmmCell / Emitter /
Transitioner / Bidirectional / TotalProbabilityCell).

``hmm_layer_amd.engine`` is the thin ctypes binding of the C ABI in include/hmm_engine.h;
the recursion itself runs in hand-written HIP (hmm_layer_amd/csrc/hmm_engine.hip).
"""
from . import engine  # noqa: F401

__all__ = ["engine"]
