import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hmm_layer_amd import engine
from oracle import textbook, params
dev = 'cuda:0'
def t(x, dt=torch.float32): return torch.as_tensor(np.asarray(x), dtype=dt, device=dev)
rng = np.random.default_rng(11)
A = params.intended_A15().numpy(); pi = np.full(15, 1/15, dtype=np.float32)
E = (rng.random((4, 600, 15)) * 0.9 + 0.05).astype(np.float32) / 4096
dead = rng.random(E.shape) < 0.6; dead[..., :6] = False; E[dead] = 0.0
g64, ll64 = textbook.posterior(A, pi, E)
for mode in ("0", "1"):
    os.environ["HMM_ENGINE_FORCE_DENSE"] = mode
    g, ll = engine.posterior(t(A)[None], t(pi), t(E)[None])
    g = g.cpu().numpy()[0]; ll = ll.cpu().numpy()[0]
    err = np.abs(g - g64).max(-1)
    print("force_dense", mode, "max err", err.max(), "ll err", np.abs(ll - ll64))
    bad = np.argwhere(err > 1e-4)
    print("  bad positions:", len(bad), bad[:10].tolist())
    if len(bad):
        n, tt = bad[0]
        print("  first bad", n, tt, "chunk", tt // 16, "\n  got", g[n, tt].round(4), "\n  ref", g64[n, tt].round(4))
# does any step have near-dead total?
ah, cum = textbook.forward(A, pi, E)
step = np.diff(np.concatenate([np.zeros((4,1)), cum], 1), axis=1)
print("min log c_t", step.min(1))
