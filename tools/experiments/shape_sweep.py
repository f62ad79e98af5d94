"""Posterior / log-likelihood / Viterbi time across batch x length shapes (15-state gene model)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from hmm_layer_amd import engine
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _model import gene15

dev = "cuda:0"
q = 15
A, pi = gene15(dev)
logA = torch.log(A.clamp_min(1e-30)); logpi = torch.log(pi)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


print("%8s %9s %5s | %10s %10s | %10s %10s | %10s %10s" % ("b", "L", "T", "post ms", "Gcell/s", "loglik ms", "Gcell/s", "vit ms", "Gcell/s"))
for b, L in ((4, 128), (32, 9999), (256, 10000), (1, 1000000), (8, 1000000), (16384, 1000), (1024, 100000), (4096, 100000)):
    E = torch.rand((1, b, L, q), device=dev) * 0.9 + 0.05
    cells = b * L * q
    tp = timed(lambda: engine.posterior(A, pi, E))
    tl = timed(lambda: engine.forward(A, pi, E, want_log_alpha=False))
    logE = torch.log(E)
    tv = timed(lambda: engine.viterbi(logA, logpi, logE), reps=3)
    print("%8d %9d %5d | %10.3f %10.1f | %10.3f %10.1f | %10.3f %10.1f" % (
        b, L, engine.chunk_len(1, b, L, q), tp * 1e3, cells / tp / 1e9, tl * 1e3, cells / tl / 1e9, tv * 1e3, cells / tv / 1e9), flush=True)
    del E, logE
    engine.release_workspaces()
    torch.cuda.empty_cache()
