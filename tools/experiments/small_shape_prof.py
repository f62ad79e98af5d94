"""Posterior and log-likelihood at the reference's training shape (b = 32, L = 9 999) and at BASELINE config 2
(b = 256, L = 10 000): a few hundred calls for rocprofv3 --kernel-trace --stats (where do 0.12 ms go?)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hmm_layer_amd import engine
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _model import gene15
dev = 'cuda:0'
A, pi = gene15(dev)
for b, L in ((32, 9999), (256, 10000)):
    E = torch.rand((1, b, L, 15), device=dev) * 0.9 + 0.05
    out = torch.empty_like(E)
    for name, fn in (("posterior", lambda: engine.posterior(A, pi, E, out=out)), ("loglik", lambda: engine.forward(A, pi, E, want_log_alpha=False))):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
        print("b=%d L=%d %s: %.1f us per call" % (b, L, name, dt * 1e6), flush=True)
