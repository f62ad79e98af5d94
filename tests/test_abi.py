"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports exactly
what include/hmm_engine.h declares; host-side argument checking; no compute calls."""
import ctypes
import os
import re

import pytest
import torch

from hmm_layer_amd import build as hbuild
from hmm_layer_amd import engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    hbuild.build()
    return engine.lib()


def declared_functions():
    text = open(os.path.join(ROOT, "include", "hmm_engine.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hmm_[a-z_]+)\s*\(", text)))


def test_header_symbols_are_exported(lib):
    names = declared_functions()
    assert {"hmm_forward", "hmm_backward", "hmm_posterior", "hmm_workspace_bytes"} <= set(names)
    raw = ctypes.CDLL(engine.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), "symbol %s declared in hmm_engine.h is not exported" % n


def test_version_and_errors(lib):
    assert lib.hmm_abi_version() == engine.ABI_VERSION == 3
    assert lib.hmm_max_states() == 4096 and lib.hmm_scan_max_states() == 16
    assert lib.hmm_strerror(0) == b"ok"
    assert b"states" in lib.hmm_strerror(-2)


def test_plan_queries(lib):
    # chunk length is a multiple of 16, at most 1024, and the workspace grows with the op
    for dims in [(1, 4, 128, 3), (1, 256, 10000, 15), (1, 1024, 100000, 15), (2, 3, 17, 7)]:
        T = lib.hmm_chunk_len(*dims)
        assert T % 16 == 0 and 16 <= T <= 512
        w0 = lib.hmm_workspace_bytes(engine.OP_LOGLIK, *dims)
        w3 = lib.hmm_workspace_bytes(engine.OP_POSTERIOR, *dims)
        assert 0 < w0 <= w3
    assert lib.hmm_workspace_bytes(engine.OP_POSTERIOR, 1, 1024, 100000, 15) < 2 << 30
    assert lib.hmm_workspace_bytes(engine.OP_POSTERIOR, 1, 4, 128, 5000) == 0   # q unsupported
    assert lib.hmm_chunk_len(1, 4, 128, 5000) == -2
    assert lib.hmm_chunk_len(1, 4, 128, 1027) == 0                              # serial large-q path
    assert lib.hmm_workspace_bytes(engine.OP_POSTERIOR, 1, 4, 128, 1027) >= 3 * 4 * 1027 * 4
    assert lib.hmm_viterbi_workspace_bytes(1, 4, 128, 17) >= 4 * 128 * 64            # one wave per sequence
    assert lib.hmm_viterbi_workspace_bytes(1, 4, 128, 65) == 0 and lib.hmm_viterbi_max_states() == 64
    assert lib.hmm_chunk_len(1, 0, 128, 3) == -1


def test_null_and_shape_errors_without_device(lib):
    # argument validation happens before any HIP call
    assert lib.hmm_forward(None, None, None, 1, 1, 16, 3, 1e-16, None, None, None, 0, None) == -3
    assert lib.hmm_forward(None, None, None, 1, 1, 0, 3, 1e-16, None, None, None, 0, None) == -1
    assert lib.hmm_posterior(None, None, None, 1, 1, 16, 5000, 1e-16, 0, None, None, None, 0, None) == -2


def test_host_wrapper_rejects_cpu_tensors(lib):
    A = torch.eye(3).unsqueeze(0)
    pi = torch.ones(3) / 3
    E = torch.rand(1, 2, 8, 3)
    with pytest.raises(engine.EngineError, match="HIP device"):
        engine.posterior(A, pi, E)
    with pytest.raises(engine.EngineError, match="HIP device"):
        engine.forward(A, pi, E)


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(engine, "_lib", None)
    monkeypatch.setattr(engine, "LIB_PATH", "/nonexistent/libhmm_engine.so")
    with pytest.raises(engine.EngineError, match="no CPU fallback"):
        engine.lib()
