"""Builds oracle/_build/liboracle.so (gcc) — the C twin of the numpy oracle.  Test infrastructure."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "hmm_oracle.c")
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "liboracle.so")
_lib = None


def build(force=False):
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    os.makedirs(OUT_DIR, exist_ok=True)
    subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", SRC, "-o", LIB, "-lm"], check=True)
    return LIB


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _f32(x):
    return np.ascontiguousarray(x, dtype=np.float32)


def posterior(A, pi, E, eps=1e-16):
    """float64 posteriors (b,L,q) and loglik (b,) — same definition as oracle.textbook.posterior."""
    A, pi, E = _f32(A), _f32(pi).reshape(-1), _f32(E)
    b, L, q = E.shape
    gam = np.empty((b, L, q), dtype=np.float64)
    ll = np.empty(b, dtype=np.float64)
    lib().oracle_posterior(A.ctypes, pi.ctypes, E.ctypes, b, L, q, ctypes.c_double(eps), gam.ctypes, ll.ctypes)
    return gam, ll


def viterbi(logA, logpi, logE):
    """Q16 fixed-point Viterbi — same definition as oracle.viterbi.viterbi."""
    logA, logpi, logE = _f32(logA), _f32(logpi).reshape(-1), _f32(logE)
    b, L, q = logE.shape
    path = np.empty((b, L), dtype=np.int32)
    score = np.empty(b, dtype=np.float64)
    lib().oracle_viterbi(logA.ctypes, logpi.ctypes, logE.ctypes, b, L, q, path.ctypes, score.ctypes)
    return path, score
