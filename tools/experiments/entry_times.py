"""ms per call of every q <= 16 entry point at the headline shape (uniform emissions), routing auto vs off."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from hmm_layer_amd import engine
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _model import gene15
dev = torch.device("cuda:0")
b, L, q = 1024, 100000, 15
A, pi = gene15(dev)
E = torch.rand((1, b, L, q), device=dev) * 0.9 + 0.05
out = torch.empty_like(E)


def timed(fn, reps=5):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for name, mode in (("auto", engine.EXACT_AUTO), ("off", engine.EXACT_OFF)):
    with engine.option(engine.OPT_EXACT, mode):
        r = {"posterior": timed(lambda: engine.posterior(A, pi, E, out=out)),
             "loglik": timed(lambda: engine.forward(A, pi, E, want_log_alpha=False)),
             "log_alpha": timed(lambda: engine.forward(A, pi, E)),
             "log_beta": timed(lambda: engine.backward(A, E)),
             "loglik_grad": timed(lambda: engine.loglik_grad(A, pi, E))}
    print(name, {k: round(v, 3) for k, v in r.items()}, flush=True)
