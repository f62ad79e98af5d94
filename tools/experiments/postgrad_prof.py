"""hmm_posterior_grad per chunk at one shape, a few calls (for rocprofv3 --kernel-trace --stats): python postgrad_prof.py [b] [L]"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hmm_layer_amd import engine
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _model import gene15
dev = 'cuda:0'
b = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
L = int(sys.argv[2]) if len(sys.argv) > 2 else 9999
q = 15
A, pi = gene15(dev)
E = torch.rand((1, b, L, q), device=dev) * 0.9 + 0.05
gam, _ = engine.posterior(A, pi, E, mode=engine.POST_PROB)
lab = torch.multinomial(gam.reshape(-1, q).clamp_min(0) + 1e-30, 1).reshape(1, b, L, 1)
G = torch.zeros((1, b, L, q), device=dev).scatter_(3, lab, -1.0)
del gam, lab
for _ in range(2): engine.posterior_grad(A, pi, E, G, mode=engine.POST_LOG)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): engine.posterior_grad(A, pi, E, G, mode=engine.POST_LOG)
torch.cuda.synchronize()
print("b=%d L=%d: %.2f ms" % (b, L, (time.perf_counter() - t0) / 5 * 1e3))
