import sys, numpy as np, torch
sys.path.insert(0, '.')
from hmm_layer_amd import engine
from oracle import textbook
np.set_printoptions(precision=5, suppress=True, linewidth=200)
dev='cuda:0'
def t(x, dt=torch.float32): return torch.as_tensor(np.asarray(x), dtype=dt, device=dev)
for q, b, L in [(3,1,4), (16,2,20), (15,2,40)]:
    rng = np.random.default_rng(q)
    A = rng.random((q,q)); A/=A.sum(-1,keepdims=True); pi = rng.random(q); pi/=pi.sum()
    E = rng.random((b,L,q))*0.9+0.05
    A=A.astype(np.float32); pi=pi.astype(np.float32); E=E.astype(np.float32)
    g64, ll64 = textbook.posterior(A,pi,E); la64,_=textbook.log_alpha(A,pi,E); lb64=textbook.log_beta(A,E)
    la, ll = engine.forward(t(A)[None], t(pi), t(E)[None])
    print("q",q,"ll", ll.cpu().numpy(), ll64)
    print(" la err", np.abs(la.cpu().numpy()[0]-la64).max())
    lb = engine.backward(t(A)[None], t(E)[None]).cpu().numpy()[0]
    print(" lb err", np.abs(lb-lb64).max())
    out = torch.full((1,b,L,q), -7.0, device=dev)
    g, _ = engine.posterior(t(A)[None], t(pi), t(E)[None], out=out)
    g = g.cpu().numpy()[0]
    print(" gam err", np.abs(g-g64).max())
    if q==3:
        print(la.cpu().numpy()[0,0], la64[0]); print(lb[0], lb64[0]); print(g[0], g64[0])
