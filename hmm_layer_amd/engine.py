"""ctypes binding of the HIP engine's C ABI (include/hmm_engine.h).

PyTorch is used only for device memory and the current HIP stream.  There is no CPU
fallback: if the library is missing or a tensor is not on a HIP device these functions
raise.  Shapes follow the reference (k models, b sequences, L positions, q states):
A (k,q,q), pi (k,q) or (1,k,q), E (k,b,L,q).
"""
import ctypes
import os

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libhmm_engine.so")

OP_LOGLIK, OP_FORWARD, OP_BACKWARD, OP_POSTERIOR, OP_VITERBI = 0, 1, 2, 3, 4
POST_PROB, POST_LOG, POST_LOG_NO_LL = 0, 1, 2
EPS = 1e-16
ABI_VERSION = 3
# tuning / test options (include/hmm_engine.h: HMM_OPT_*, HMM_EXACT_*)
OPT_CHUNK, OPT_FORCE_DENSE, OPT_SCAN2, OPT_GROUPS, OPT_EXACT, OPT_PGCHUNK, OPT_VGROUPS = 0, 1, 2, 3, 4, 5, 6
EXACT_AUTO, EXACT_OFF, EXACT_ALWAYS, EXACT_ALWAYS_NARROW = 0, 1, 2, 3

_lib = None
_workspaces = {}


class EngineError(RuntimeError):
    pass


def lib():
    """The loaded shared library (loaded once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineError(
            "HIP engine library %s is missing: build it with `python -m hmm_layer_amd.build` "
            "(there is no CPU fallback)" % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    c_p, c_i, c_f, c_sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
    L.hmm_strerror.restype = ctypes.c_char_p
    L.hmm_strerror.argtypes = [c_i]
    L.hmm_abi_version.restype = c_i
    L.hmm_set_option.restype = c_i
    L.hmm_set_option.argtypes = [c_i, c_i]
    L.hmm_get_option.restype = c_i
    L.hmm_get_option.argtypes = [c_i]
    L.hmm_exact_count.restype = ctypes.c_longlong
    L.hmm_exact_count.argtypes = [c_i, c_i, c_i, c_i, c_i, c_p, c_sz]
    L.hmm_exact_detail.restype = c_i
    L.hmm_exact_detail.argtypes = [c_i, c_i, c_i, c_i, c_p, c_sz, c_p]
    if hasattr(L, "hmm_exact_detail_op"):           # (diagnostics added within ABI version 3: an older build lacks them)
        L.hmm_exact_detail_op.restype = c_i
        L.hmm_exact_detail_op.argtypes = [c_i, c_i, c_i, c_i, c_i, c_p, c_sz, c_p]
        L.hmm_window_table.restype = c_i
        L.hmm_window_table.argtypes = [c_i, c_i, c_i, c_i, c_i, c_p, c_sz, c_i, c_p, c_p, c_p, c_i]
    L.hmm_max_states.restype = c_i
    L.hmm_scan_max_states.restype = c_i
    L.hmm_viterbi_max_states.restype = c_i
    L.hmm_grad_max_states.restype = c_i
    L.hmm_posterior_grad_max_states.restype = c_i
    L.hmm_posterior_grad_workspace_bytes.restype = c_sz
    L.hmm_posterior_grad_workspace_bytes.argtypes = [c_i] * 4
    L.hmm_loglik_grad_serial_count.restype = ctypes.c_longlong
    L.hmm_loglik_grad_serial_count.argtypes = [c_i, c_i, c_i, c_i, c_p, c_sz]
    L.hmm_posterior_grad_serial_count.restype = ctypes.c_longlong
    L.hmm_posterior_grad_serial_count.argtypes = [c_i, c_i, c_i, c_i, c_p, c_sz]
    L.hmm_posterior_grad.restype = c_i
    L.hmm_posterior_grad.argtypes = [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_i, c_p, c_p, c_p, c_p, c_p, c_sz, c_p]
    L.hmm_largeq_tile_cols.restype = c_i
    L.hmm_largeq_tile_cols.argtypes = [c_i, c_i]
    L.hmm_chunk_len.restype = c_i
    L.hmm_chunk_len.argtypes = [c_i] * 4
    L.hmm_workspace_bytes.restype = c_sz
    L.hmm_workspace_bytes.argtypes = [c_i] * 5
    L.hmm_forward.restype = c_i
    L.hmm_forward.argtypes = [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_p, c_p, c_p, c_sz, c_p]
    L.hmm_backward.restype = c_i
    L.hmm_backward.argtypes = [c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_p, c_p, c_sz, c_p]
    L.hmm_posterior.restype = c_i
    L.hmm_posterior.argtypes = [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_i, c_p, c_p, c_p, c_sz, c_p]
    L.hmm_viterbi_workspace_bytes.restype = c_sz
    L.hmm_viterbi_workspace_bytes.argtypes = [c_i] * 4
    L.hmm_viterbi.restype = c_i
    L.hmm_viterbi.argtypes = [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_sz, c_p]
    L.hmm_gene_emissions.restype = c_i
    L.hmm_gene_emissions.argtypes = [c_p, c_i, c_i, c_i, c_p, c_i, c_p, c_p, c_i, c_p, c_i, c_f, c_f, c_i, c_p, c_p]
    L.hmm_profile_create.restype = c_p
    L.hmm_profile_destroy.argtypes = [c_p]
    L.hmm_posterior_profiled.restype = c_i
    L.hmm_posterior_profiled.argtypes = L.hmm_posterior.argtypes + [c_p]
    L.hmm_profile_read.restype = c_i
    L.hmm_profile_read.argtypes = [c_p, c_p, c_p]
    L.hmm_loglik_partials.restype = c_i
    L.hmm_loglik_partials.argtypes = [c_p, c_p, c_i, c_i, c_p, c_p]
    L.hmm_loglik_allreduce.restype = c_i
    L.hmm_loglik_allreduce.argtypes = [c_p, c_p, c_i, c_p]
    L.hmm_seqshard_workspace_bytes.restype = c_sz
    L.hmm_seqshard_workspace_bytes.argtypes = [c_i] * 5
    L.hmm_seqshard_reduce.restype = c_i
    L.hmm_seqshard_reduce.argtypes = [c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_i, c_i, c_p, c_p, c_p, c_sz, c_p]
    L.hmm_seqshard_posterior.restype = c_i
    L.hmm_seqshard_posterior.argtypes = [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_i, c_p, c_p, c_i, c_i, c_i,
                                         c_p, c_p, c_p, c_p, c_sz, c_p]
    L.hmm_loglik_grad_workspace_bytes.restype = c_sz
    L.hmm_loglik_grad_workspace_bytes.argtypes = [c_i] * 4
    L.hmm_loglik_grad.restype = c_i
    L.hmm_loglik_grad.argtypes = [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_sz, c_p]
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise EngineError("hmm_engine: %s (code %d)" % (lib().hmm_strerror(rc).decode(), rc))


def _dev(t, name, dtype=torch.float32):
    if not torch.is_tensor(t):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise EngineError("%s must live on a HIP device (got %s); the engine has no CPU path" % (name, t.device))
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    return t.contiguous()


def _shapes(A, E, pi=None):
    if E.dim() != 4:
        raise ValueError("E must have shape (k, b, L, q), got %s" % (tuple(E.shape),))
    k, b, L, q = E.shape
    if A.dim() == 2:
        A = A.unsqueeze(0)
    if tuple(A.shape) != (k, q, q):
        raise ValueError("A must have shape (k=%d, q=%d, q=%d), got %s" % (k, q, q, tuple(A.shape)))
    if pi is not None:
        if pi.numel() != k * q:
            raise ValueError("pi must hold k*q = %d values, got %s" % (k * q, tuple(pi.shape)))
        pi = pi.reshape(k, q)
    if min(k, b, L, q) < 1:
        raise ValueError("empty input: (k, b, L, q) = %s" % ((k, b, L, q),))
    if q > lib().hmm_max_states():
        raise ValueError("q = %d states exceeds the scan kernels' limit of %d" % (q, lib().hmm_max_states()))
    return A, pi, (k, b, L, q)


def _workspace(op, dims, device, need=None):
    if need is None:
        need = lib().hmm_workspace_bytes(op, *dims)
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def release_workspaces():
    _workspaces.clear()


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def chunk_len(k, b, L, q):
    return lib().hmm_chunk_len(k, b, L, q)


def largeq_tile_cols(b, q):
    """Column width of the GEMM tile serving (b sequences per model, q > 64 states); 0 otherwise."""
    return lib().hmm_largeq_tile_cols(int(b), int(q))


def set_option(option, value):
    """Sets a process-wide tuning / test option (OPT_*); returns the previous value."""
    if not 0 <= int(option) <= OPT_VGROUPS:
        raise ValueError("unknown option %r" % (option,))
    return lib().hmm_set_option(int(option), int(value))


def get_option(option):
    return lib().hmm_get_option(int(option))


class option:
    """Context manager: `with engine.option(engine.OPT_CHUNK, 64): ...` (tests and A/B scripts)."""

    def __init__(self, opt, value):
        self.opt, self.value = opt, value

    def __enter__(self):
        self.old = set_option(self.opt, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.opt, self.old)
        return False


def exact_count(op, dims, device=None):
    """How many of the k*b sequences of the LAST q <= 16 call of kind `op` with shape `dims` on this
    device and stream were served by the serial exact-clamp kernels (synchronises)."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    with torch.cuda.device(device):
        key = (device.index, torch.cuda.current_stream(device).cuda_stream)
        ws = _workspaces.get(key)
        if ws is None:
            raise EngineError("no call has run on this device / stream yet")
        torch.cuda.current_stream(device).synchronize()
        n = lib().hmm_exact_count(int(op), *dims, ws.data_ptr(), ws.numel())
    if n < 0:
        _check(int(n))
    return int(n)


def exact_detail(dims, device=None, op=OP_POSTERIOR):
    """Routing of the LAST posterior() call (q <= 16) with shape `dims` on this device and stream ->
    dict(routed=sequences that left the scan, window_sequences=..., windows=..., whole=sequences redone whole,
    window_chunks=chunks the windows walked).  Synchronises.  op: OP_LOGLIK / OP_FORWARD (forward() without / with
    log alpha) or OP_BACKWARD for the last call of those entry points instead."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    with torch.cuda.device(device):
        key = (device.index, torch.cuda.current_stream(device).cuda_stream)
        ws = _workspaces.get(key)
        if ws is None:
            raise EngineError("no call has run on this device / stream yet")
        torch.cuda.current_stream(device).synchronize()
        d = (ctypes.c_longlong * 5)()
        _check(lib().hmm_exact_detail_op(int(op), *[int(x) for x in dims], ws.data_ptr(), ws.numel(), d))
    return dict(routed=int(d[0]), window_sequences=int(d[1]), windows=int(d[2]), whole=int(d[3]), window_chunks=int(d[4]))


def window_table(dims, seq, op=OP_POSTERIOR, device=None):
    """Diagnostics (q <= 16): the windows of sequence `seq` (index into k*b) after the LAST call of `op` with shape `dims`
    on this device and stream -> dict(windows=[(first chunk, chunks), ...], shifts=[log-scale shift per window]
    (OP_FORWARD / OP_BACKWARD), psi=numpy array of the per-chunk certificate sums; chunks the reduce marked for having
    gone through the denormal range read 1.0).  Synchronises."""
    import numpy as np
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    with torch.cuda.device(device):
        key = (device.index, torch.cuda.current_stream(device).cuda_stream)
        ws = _workspaces.get(key)
        if ws is None:
            raise EngineError("no call has run on this device / stream yet")
        torch.cuda.current_stream(device).synchronize()
        k, b, L, q = (int(x) for x in dims)
        C = (L + chunk_len(k, b, L, q) - 1) // chunk_len(k, b, L, q)
        tab = (ctypes.c_int * 34)()
        sh = (ctypes.c_double * 24)()
        ps = (ctypes.c_float * C)()
        rc = lib().hmm_window_table(int(op), k, b, L, q, ws.data_ptr(), ws.numel(), int(seq), tab, sh, ps, C)
        if rc < 0:
            _check(rc)
    n = max(0, min(int(tab[0]), 16))
    return dict(windows=[(int(tab[2 + 2 * i]), int(tab[3 + 2 * i]) & 0xFFFFFF) for i in range(n)],
                shifts=[float(sh[i]) for i in range(n)], psi=np.array(ps[:], dtype=np.float32))


def loglik_grad_serial_count(dims, device=None):
    """The same for the LAST loglik_grad call (17..64 states)."""
    return posterior_grad_serial_count(dims, device, _fn="hmm_loglik_grad_serial_count")


def posterior_grad_serial_count(dims, device=None, _fn="hmm_posterior_grad_serial_count"):
    """How many of the k*b sequences of the LAST posterior_grad call with shape `dims` on this device and
    stream were served by the whole-sequence sweeps rather than per chunk (synchronises)."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    with torch.cuda.device(device):
        key = (device.index, torch.cuda.current_stream(device).cuda_stream)
        ws = _workspaces.get(key)
        if ws is None:
            raise EngineError("no call has run on this device / stream yet")
        torch.cuda.current_stream(device).synchronize()
        n = getattr(lib(), _fn)(*[int(d) for d in dims], ws.data_ptr(), ws.numel())
    if n < 0:
        _check(int(n))
    return int(n)


def forward(A, pi, E, want_log_alpha=True, eps=EPS):
    """-> (log_alpha (k,b,L,q) fp32 or None, loglik (k,b) fp64)."""
    A, pi, E = _dev(A, "A"), _dev(pi, "pi"), _dev(E, "E")
    A, pi, dims = _shapes(A, E, pi)
    with torch.cuda.device(E.device):
        ws = _workspace(OP_FORWARD if want_log_alpha else OP_LOGLIK, dims, E.device)
        la = torch.empty_like(E) if want_log_alpha else None
        ll = torch.empty(dims[:2], dtype=torch.float64, device=E.device)
        _check(lib().hmm_forward(A.data_ptr(), pi.data_ptr(), E.data_ptr(), *dims, eps,
                                 la.data_ptr() if want_log_alpha else None, ll.data_ptr(),
                                 ws.data_ptr(), ws.numel(), _stream(E.device)))
    return la, ll


def backward(A, E, eps=EPS):
    """-> log_beta (k,b,L,q) fp32."""
    A, E = _dev(A, "A"), _dev(E, "E")
    A, _, dims = _shapes(A, E)
    with torch.cuda.device(E.device):
        ws = _workspace(OP_BACKWARD, dims, E.device)
        lb = torch.empty_like(E)
        _check(lib().hmm_backward(A.data_ptr(), E.data_ptr(), *dims, eps, lb.data_ptr(),
                                  ws.data_ptr(), ws.numel(), _stream(E.device)))
    return lb


KERNELS = ("reduce", "scan", "forward", "backward", "exact")


class Profile:
    """Per-kernel HIP-event timing of posterior() calls (bench.py's roofline leg)."""

    def __init__(self):
        self.handle = ctypes.c_void_p(lib().hmm_profile_create())

    def read(self):
        """-> {kernel: (total ms, launches)} since the last read; waits for the events."""
        ms = (ctypes.c_double * len(KERNELS))()
        n = (ctypes.c_longlong * len(KERNELS))()
        _check(lib().hmm_profile_read(self.handle, ms, n))
        return {name: (ms[i], n[i]) for i, name in enumerate(KERNELS)}

    def close(self):
        if self.handle:
            lib().hmm_profile_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def posterior(A, pi, E, mode=POST_PROB, eps=EPS, out=None, profile=None):
    """-> (posterior (k,b,L,q) fp32 per `mode`, loglik (k,b) fp64)."""
    A, pi, E = _dev(A, "A"), _dev(pi, "pi"), _dev(E, "E")
    A, pi, dims = _shapes(A, E, pi)
    with torch.cuda.device(E.device):
        ws = _workspace(OP_POSTERIOR, dims, E.device)
        if out is None:
            out = torch.empty_like(E)
        elif (out.shape != E.shape or out.dtype != torch.float32 or not out.is_contiguous()
              or out.device != E.device):
            raise ValueError("out must be a contiguous fp32 tensor shaped like E on E's device")
        ll = torch.empty(dims[:2], dtype=torch.float64, device=E.device)
        args = (A.data_ptr(), pi.data_ptr(), E.data_ptr(), *dims, eps, int(mode),
                out.data_ptr(), ll.data_ptr(), ws.data_ptr(), ws.numel(), _stream(E.device))
        if profile is None:
            _check(lib().hmm_posterior(*args))
        else:
            _check(lib().hmm_posterior_profiled(*args, profile.handle))
    return out, ll


def gene_emissions(x, B, state_row, codon, state_codon, free_value=1.0 / 4096.0, add=0.0, n_mass=1):
    """Fused GenePredHMMEmitter.forward for one model: x (b,L,s+5) -> E (b,L,q) fp32.
    B (rows,s) fp32, state_row (q) int32, codon (2,nc,64) fp32, state_codon (q) int32."""
    x, B, codon = _dev(x, "x"), _dev(B, "B"), _dev(codon, "codon")
    state_row = _dev(state_row, "state_row", torch.int32)
    state_codon = _dev(state_codon, "state_codon", torch.int32)
    if x.dim() != 3:
        raise ValueError("x must have shape (b, L, s+5), got %s" % (tuple(x.shape),))
    b, L, w = x.shape
    s = w - 5
    rows, q, nc = B.shape[0], state_row.numel(), codon.shape[1]
    if B.shape[1] != s or tuple(codon.shape) != (2, nc, 64) or state_codon.numel() != q:
        raise ValueError("inconsistent emitter tables")
    with torch.cuda.device(x.device):
        E = torch.empty((b, L, q), dtype=torch.float32, device=x.device)
        _check(lib().hmm_gene_emissions(x.data_ptr(), b, L, s, B.data_ptr(), rows, state_row.data_ptr(),
                                        codon.data_ptr(), nc, state_codon.data_ptr(), q, float(free_value),
                                        float(add), int(n_mass), E.data_ptr(), _stream(x.device)))
    return E


def viterbi(logA, logpi, logE):
    """Most probable state paths.  logA (k,q,q), logpi (k,q), logE (k,b,L,q) fp32 log-probabilities
    (-inf allowed: anything below -1024 counts as -1024).  -> (path (k,b,L) int32, score (k,b) fp64).
    Scores are Q16 fixed point, so the result is bit-identical to the serial recursion
    (oracle/viterbi.py); ties take the lowest state index."""
    logA, logpi, logE = _dev(logA, "logA"), _dev(logpi, "logpi"), _dev(logE, "logE")
    logA, logpi, dims = _shapes(logA, logE, logpi)
    k, b, L, q = dims
    if q > lib().hmm_viterbi_max_states():
        raise ValueError("viterbi covers q <= %d states, got %d" % (lib().hmm_viterbi_max_states(), q))
    with torch.cuda.device(logE.device):
        need = lib().hmm_viterbi_workspace_bytes(*dims)
        key = (logE.device.index, torch.cuda.current_stream(logE.device).cuda_stream, "viterbi")
        ws = _workspaces.get(key)
        if ws is None or ws.numel() < need:
            ws = torch.empty(max(need, 1 << 20), dtype=torch.uint8, device=logE.device)
            _workspaces[key] = ws
        path = torch.empty((k, b, L), dtype=torch.int32, device=logE.device)
        score = torch.empty((k, b), dtype=torch.float64, device=logE.device)
        _check(lib().hmm_viterbi(logA.data_ptr(), logpi.data_ptr(), logE.data_ptr(), *dims,
                                 path.data_ptr(), score.data_ptr(), ws.data_ptr(), ws.numel(),
                                 _stream(logE.device)))
    return path, score


def loglik_partials(loglik, weights=None):
    """(k,b) fp64 loglik [, (k,b) fp32 weights] -> (k,2) fp64: (sum w*loglik, sum w) per model."""
    loglik = _dev(loglik, "loglik", torch.float64)
    k, b = loglik.shape
    if weights is not None:
        weights = _dev(weights, "weights")
        if tuple(weights.shape) != (k, b):
            raise ValueError("weights must have shape %s" % ((k, b),))
    with torch.cuda.device(loglik.device):
        part = torch.empty((k, 2), dtype=torch.float64, device=loglik.device)
        _check(lib().hmm_loglik_partials(loglik.data_ptr(),
                                         weights.data_ptr() if weights is not None else None,
                                         k, b, part.data_ptr(), _stream(loglik.device)))
    return part


def _seqshard_ws(dims, R, device):
    need = lib().hmm_seqshard_workspace_bytes(*dims, int(R))
    if need == 0:
        raise ValueError("sequence-sharded calls cover q <= %d states" % lib().hmm_scan_max_states())
    key = (device.index, torch.cuda.current_stream(device).cuda_stream, "seqshard")
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def seqshard_reduce(A, E_slab, seq_start, R, eps=EPS):
    """Step 1 of the sequence-sharded posterior (include/hmm_engine.h): this rank's time slab E_slab
    (k,b,Ls,q) -> its operator per sequence, (k,b,16,16) fp32 and (k,b,16) int32 exponents.  The chunk
    operators stay in this device's "seqshard" workspace for seqshard_posterior."""
    A, E = _dev(A, "A"), _dev(E_slab, "E_slab")
    A, _, dims = _shapes(A, E)
    k, b = dims[:2]
    with torch.cuda.device(E.device):
        ws = _seqshard_ws(dims, R, E.device)
        op = torch.empty((k, b, 16, 16), dtype=torch.float32, device=E.device)
        ex = torch.empty((k, b, 16), dtype=torch.int32, device=E.device)
        _check(lib().hmm_seqshard_reduce(A.data_ptr(), E.data_ptr(), *dims, eps, int(bool(seq_start)), int(R),
                                         op.data_ptr(), ex.data_ptr(), ws.data_ptr(), ws.numel(), _stream(E.device)))
    return op, ex


def seqshard_posterior(A, pi, E_slab, all_ops, all_exps, r, mode=POST_PROB, eps=EPS):
    """Step 3: all_ops (k,b,R,16,16) / all_exps (k,b,R,16) = every rank's slab operators in time order;
    -> (out (k,b,Ls,q), loglik (k,b) fp64 of the whole sequences, phi (k,b) fp32: this slab's share of
    the floor-transition bound)."""
    A, pi, E = _dev(A, "A"), _dev(pi, "pi"), _dev(E_slab, "E_slab")
    all_ops, all_exps = _dev(all_ops, "all_ops"), _dev(all_exps, "all_exps", torch.int32)
    A, pi, dims = _shapes(A, E, pi)
    k, b = dims[:2]
    R = all_ops.shape[2]
    if tuple(all_ops.shape) != (k, b, R, 16, 16) or tuple(all_exps.shape) != (k, b, R, 16):
        raise ValueError("all_ops / all_exps must have shapes (k,b,R,16,16) / (k,b,R,16)")
    with torch.cuda.device(E.device):
        ws = _seqshard_ws(dims, R, E.device)
        out = torch.empty_like(E)
        ll = torch.empty((k, b), dtype=torch.float64, device=E.device)
        phi = torch.empty((k, b), dtype=torch.float32, device=E.device)
        _check(lib().hmm_seqshard_posterior(A.data_ptr(), pi.data_ptr(), E.data_ptr(), *dims, eps, int(r == 0),
                                            all_ops.data_ptr(), all_exps.data_ptr(), int(R), int(r), int(mode),
                                            out.data_ptr(), ll.data_ptr(), phi.data_ptr(), ws.data_ptr(), ws.numel(),
                                            _stream(E.device)))
    return out, ll, phi


def loglik_allreduce(comm, partial):
    """In-place all-reduce(sum) of the (k,2) fp64 partials over a raw RCCL communicator (an ncclComm_t as an
    integer / ctypes pointer) created by the caller — the C-ABI route for hosts without
    torch.distributed; hmm_layer_amd.distributed uses torch's process group instead."""
    partial = _dev(partial, "partial", torch.float64)
    if partial.dim() != 2 or partial.shape[1] != 2:
        raise ValueError("partial must have shape (k, 2)")
    with torch.cuda.device(partial.device):
        _check(lib().hmm_loglik_allreduce(ctypes.c_void_p(int(comm) if not isinstance(comm, ctypes.c_void_p) else comm.value),
                                          partial.data_ptr(), partial.shape[0], _stream(partial.device)))
    return partial


def loglik_grad(A, pi, E, grad_loglik=None, eps=EPS):
    """Gradients of sum_{m,s} grad_loglik[m,s] * loglik[m,s] -> (dA (k,q,q), dpi (k,q), dE (k,b,L,q), loglik (k,b) fp64).

    What autograd through the reference's time loop (hmm_layer/BaseRNN.py:217-227) computes, from
    one forward-backward pass."""
    A, pi, E = _dev(A, "A"), _dev(pi, "pi"), _dev(E, "E")
    A, pi, dims = _shapes(A, E, pi)
    k, b, L, q = dims
    if q > lib().hmm_grad_max_states():
        raise ValueError("loglik_grad covers q <= %d states" % lib().hmm_grad_max_states())
    if grad_loglik is not None:
        grad_loglik = _dev(grad_loglik, "grad_loglik")
        if tuple(grad_loglik.shape) != (k, b):
            raise ValueError("grad_loglik must have shape %s" % ((k, b),))
    with torch.cuda.device(E.device):
        ws = _workspace(None, dims, E.device, need=lib().hmm_loglik_grad_workspace_bytes(*dims))
        dA = torch.empty((k, q, q), dtype=torch.float32, device=E.device)
        dpi = torch.empty((k, q), dtype=torch.float32, device=E.device)
        dE = torch.empty_like(E)
        ll = torch.empty((k, b), dtype=torch.float64, device=E.device)
        _check(lib().hmm_loglik_grad(A.data_ptr(), pi.data_ptr(), E.data_ptr(), *dims, eps,
                                     grad_loglik.data_ptr() if grad_loglik is not None else None,
                                     dA.data_ptr(), dpi.data_ptr(), dE.data_ptr(), ll.data_ptr(),
                                     ws.data_ptr(), ws.numel(), _stream(E.device)))
    return dA, dpi, dE, ll


def posterior_grad(A, pi, E, grad_out, mode=POST_LOG, eps=EPS):
    """Gradients of <grad_out, out> with out = posterior(A, pi, E, mode) -> (dA (k,q,q), dpi (k,q), dE (k,b,L,q)).

    What autograd through the reference's _state_posterior_log_probs_impl loops computes
    (hmm_layer/MsaHMMLayer.py:422-521); mode POST_PROB or POST_LOG."""
    A, pi, E, grad_out = _dev(A, "A"), _dev(pi, "pi"), _dev(E, "E"), _dev(grad_out, "grad_out")
    A, pi, dims = _shapes(A, E, pi)
    k, b, L, q = dims
    if q > lib().hmm_posterior_grad_max_states():
        raise ValueError("posterior_grad covers q <= %d states" % lib().hmm_posterior_grad_max_states())
    if grad_out.shape != E.shape:
        raise ValueError("grad_out must be shaped like E")
    if int(mode) not in (POST_PROB, POST_LOG):
        raise ValueError("posterior_grad supports mode POST_PROB or POST_LOG")
    with torch.cuda.device(E.device):
        ws = _workspace(None, dims, E.device, need=lib().hmm_posterior_grad_workspace_bytes(*dims))
        dA = torch.empty((k, q, q), dtype=torch.float32, device=E.device)
        dpi = torch.empty((k, q), dtype=torch.float32, device=E.device)
        dE = torch.empty_like(E)
        _check(lib().hmm_posterior_grad(A.data_ptr(), pi.data_ptr(), E.data_ptr(), *dims, eps, int(mode),
                                        grad_out.data_ptr(), dA.data_ptr(), dpi.data_ptr(), dE.data_ptr(),
                                        ws.data_ptr(), ws.numel(), _stream(E.device)))
    return dA, dpi, dE
