"""Few long sequences: posterior / log-likelihood / Viterbi time against the chunk length (two-level scans)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from hmm_layer_amd import engine
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _model import gene15
dev = "cuda:0"
A, pi = gene15(dev)
logA = torch.log(A.clamp_min(1e-30)); logpi = torch.log(pi)

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

for b, L in ((1, 1000000), (8, 1000000), (2, 100000), (64, 100000)):
    E = torch.rand((1, b, L, 15), device=dev) * 0.9 + 0.05
    logE = torch.log(E)
    for T in (0, 64, 128, 192, 256, 384, 512):
        with engine.option(engine.OPT_CHUNK, T):
            print("b=%d L=%d T=%3d(%3d): posterior %.3f  loglik %.3f  viterbi %.3f ms" % (
                b, L, T, engine.chunk_len(1, b, L, 15), timed(lambda: engine.posterior(A, pi, E)),
                timed(lambda: engine.forward(A, pi, E, want_log_alpha=False)), timed(lambda: engine.viterbi(logA, logpi, logE))), flush=True)
    del E, logE
