"""Time the 17..64-state one-wave-per-sequence kernels of two builds of the engine side by side:
   python ab_lib.py OLD.so NEW.so q b L"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hmm_layer_amd import engine
dev = 'cuda:0'
libs = sys.argv[1:3]
q, b, L = (int(v) for v in sys.argv[3:6])
torch.manual_seed(0)
A = torch.rand((1, q, q), device=dev) ** 4; A = A / A.sum(-1, keepdim=True)
pi = torch.full((1, q), 1 / q, device=dev)
E = torch.rand((1, b, L, q), device=dev) * 0.9 + 0.05
out = torch.empty_like(E)
def timed(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for rnd in range(2):
    for path in libs:
        engine._lib = None; engine.LIB_PATH = os.path.abspath(path); engine.release_workspaces()
        with engine.option(engine.OPT_EXACT, 2):          # every sequence on the one-wave-per-sequence kernels
            r = {"loglik": timed(lambda: engine.forward(A, pi, E, want_log_alpha=False)),
                 "log_alpha": timed(lambda: engine.forward(A, pi, E, want_log_alpha=True)),
                 "posterior": timed(lambda: engine.posterior(A, pi, E, out=out)),
                 "post_log": timed(lambda: engine.posterior(A, pi, E, out=out, mode=engine.POST_LOG)),
                 "grad": timed(lambda: engine.loglik_grad(A, pi, E))}
        print(os.path.basename(path), "q=%d b=%d L=%d" % (q, b, L), {k: round(v, 2) for k, v in r.items()}, flush=True)
