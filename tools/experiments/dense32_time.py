"""Dense 17..32-state models: the chunked scan (dense MFMA reduce) vs the one-wave-per-sequence kernels."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from hmm_layer_amd import engine
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
for q in (24, 32):
    A = rng.random((q, q)).astype(np.float32) ** 3 + 1e-3
    A /= A.sum(-1, keepdims=True)
    A = torch.tensor(A, device=dev)[None]
    pi = torch.full((1, q), 1.0 / q, device=dev)
    for b, L in ((16, 100000), (128, 100000), (1024, 100000), (32, 9999)):
        E = torch.rand((1, b, L, q), device=dev) * 0.9 + 0.05
        res = []
        for mode in (engine.EXACT_AUTO, engine.EXACT_ALWAYS):
            with engine.option(engine.OPT_EXACT, mode):
                engine.posterior(A, pi, E); torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(3): engine.posterior(A, pi, E)
                torch.cuda.synchronize(); tp = (time.perf_counter() - t0) / 3
                engine.forward(A, pi, E, want_log_alpha=False); torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(3): engine.forward(A, pi, E, want_log_alpha=False)
                torch.cuda.synchronize(); tl = (time.perf_counter() - t0) / 3
            res.append((tp * 1e3, tl * 1e3))
        print("q=%d b=%4d L=%6d: chunked posterior %.2f ms loglik %.2f ms | one wave per sequence %.2f / %.2f ms" % (
            q, b, L, res[0][0], res[0][1], res[1][0], res[1][1]), flush=True)
        del E
