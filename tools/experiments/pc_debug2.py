"""Chunked vs whole-sequence posterior gradients: which inputs make them differ (debug aid)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from hmm_layer_amd import engine
from oracle import params
dev = lambda x: torch.as_tensor(np.asarray(x), dtype=torch.float32, device="cuda:0")
q, b, L = 15, 2, 1500
A, pi = params.intended_A15().numpy().astype(np.float32), np.full(15, 1 / 15, dtype=np.float32)

def run(E, G, mode, chunk):
    out = {}
    for how in (0, 2):
        with engine.option(engine.OPT_PGCHUNK, how), engine.option(engine.OPT_CHUNK, chunk):
            out[how] = [t.cpu().numpy() for t in engine.posterior_grad(dev(A)[None], dev(pi)[None], dev(E), dev(G), mode=mode)]
    return [np.abs(s - c).max() / np.abs(s).max() for s, c in zip(out[0], out[2])]

for frac, fill in ((0.0, 0.0), (0.2, 0.0), (0.2, 1e-10), (0.2, 1e-5), (0.02, 0.0), (0.2, 1e-3)):
    rng = np.random.default_rng(31)
    E = (rng.random((1, b, L, q)) * 0.9 + 0.05).astype(np.float32)
    dead = rng.random(E.shape) < frac
    dead[..., :6] = False
    E[dead] = fill
    gam, _ = engine.posterior(dev(A)[None], dev(pi)[None], dev(E))
    gam = gam.cpu().numpy()
    G = -(gam == gam.max(-1, keepdims=True)).astype(np.float32)
    Gr = rng.standard_normal(E.shape).astype(np.float32)
    for mode, nm in ((engine.POST_LOG, "log "), (engine.POST_PROB, "prob")):
        for chunk in (16, 64, 512):
            print("dead %.2f fill %g  %s chunk %3d  labels: dA %.2e dpi %.2e dE %.2e   randn: dA %.2e dpi %.2e dE %.2e" %
                  ((frac, fill, nm, chunk) + tuple(run(E, G, mode, chunk)) + tuple(run(E, Gr, mode, chunk))), flush=True)
