"""Times hmm_loglik_grad against hmm_posterior on BASELINE config 3 sizes (needs an MI355X)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from hmm_layer_amd import engine
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _model import gene15

dev = "cuda:0"
b, L, q = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, int(sys.argv[2]) if len(sys.argv) > 2 else 100000, 15
A, pi = gene15(dev)
E = torch.rand((1, b, L, q), device=dev) * 0.9 + 0.05
w = torch.rand((1, b), device=dev) + 0.5
for name, fn in (("posterior", lambda: engine.posterior(A, pi, E)),
                 ("loglik_grad", lambda: engine.loglik_grad(A, pi, E, w)),
                 ("loglik", lambda: engine.forward(A, pi, E, want_log_alpha=False))):
    out = fn(); torch.cuda.synchronize()
    del out
    t0 = time.perf_counter()
    for _ in range(5):
        out = fn()
        del out
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print("%-12s %8.3f ms  %.3e cells/s" % (name, ms, b * L * q / ms * 1e3), flush=True)
