import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from hmm_layer_amd import engine
from oracle import params, textbook
dev = "cuda:0"
t = lambda x: torch.as_tensor(np.asarray(x), dtype=torch.float32, device=dev)
rng = np.random.default_rng(12)
A = params.intended_A15().numpy(); pi = np.full(15, 1 / 15, dtype=np.float32)
b, L = 12, 1500
E = (rng.random((b, L, 15)) * 0.9 + 0.05).astype(np.float32)
hard = np.arange(b) % 2 == 1
Eh = E[hard]; Eh[rng.random(Eh.shape) < 0.25] = 0.0; E[hard] = Eh
w = (rng.random(b) + 0.5).astype(np.float32)
for chunk in (0, 16, 64):
    with engine.option(engine.OPT_CHUNK, chunk):
        dA, dpi, dE, ll = engine.loglik_grad(t(A)[None], t(pi)[None], t(E)[None], t(w)[None])
        det = engine.exact_detail((1, b, L, 15))
        T = engine.chunk_len(1, b, L, 15)
    d = dE.cpu().numpy()[0]
    bad = ~np.isfinite(d).all(-1)
    print("chunk", chunk, "T", T, det, "nan dA", int(np.isnan(dA.cpu().numpy()).sum()))
    for s in range(b):
        if bad[s].any():
            idx = np.nonzero(bad[s])[0]
            print("  seq", s, "hard" if hard[s] else "easy", "bad rows", len(idx), "first", idx[:6], "last", idx[-3:], "chunks", sorted(set((idx // T).tolist()))[:20])
