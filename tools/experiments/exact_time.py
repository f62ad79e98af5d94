"""Cost of the exact-clamp routing on BASELINE config 3 (and small shapes): scan only (EXACT_OFF), default
(AUTO: certificate + empty serial launches), everything serial (EXACT_ALWAYS)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hmm_layer_amd import engine
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _model import gene15
dev = 'cuda:0'
A, pi = gene15(dev)
for (b, L) in ((1024, 100000), (256, 10000), (32, 9999), (1, 1000000)):
    E = torch.rand((1, b, L, 15), device=dev) * 0.9 + 0.05
    out = torch.empty_like(E)
    for name, mode in (("off", engine.EXACT_OFF), ("auto", engine.EXACT_AUTO), ("always", engine.EXACT_ALWAYS)):
        with engine.option(engine.OPT_EXACT, mode):
            res = []
            for fn in (lambda: engine.posterior(A, pi, E, out=out), lambda: engine.forward(A, pi, E, want_log_alpha=False)):
                fn(); torch.cuda.synchronize()
                n = 5 if mode != engine.EXACT_ALWAYS else 2
                t0 = time.perf_counter()
                for _ in range(n): fn()
                torch.cuda.synchronize()
                res.append((time.perf_counter() - t0) / n * 1e3)
        print("b=%d L=%d exact=%-6s posterior %.3f ms  loglik %.3f ms" % (b, L, name, res[0], res[1]), flush=True)
    del E, out
