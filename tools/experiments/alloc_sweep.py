"""k_forward per allocation strategy of E inside ONE process (is the 1.37 / 1.53 ms spread a matter of how the
input was allocated?)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hmm_layer_amd import engine
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _model import gene15
dev = torch.device('cuda:0')
b, L, q = 1024, 100000, 15
A, pi = gene15(dev)

def measure(E, tag):
    out = torch.empty_like(E)
    prof = engine.Profile()
    engine.posterior(A, pi, E, out=out)
    for _ in range(5): engine.posterior(A, pi, E, out=out, profile=prof)
    torch.cuda.synchronize()
    k = prof.read(); prof.close()
    ms = {n: round(v[0] / max(v[1], 1), 3) for n, v in k.items()}
    print("%-44s E %#x out %#x  %s" % (tag, E.data_ptr(), out.data_ptr(), ms), flush=True)
    del out

E = torch.rand((1, b, L, q), device=dev) * 0.9 + 0.05
measure(E, "rand * 0.9 + 0.05 (temporaries)")
del E; engine.release_workspaces(); torch.cuda.empty_cache()
E = torch.empty((1, b, L, q), device=dev).uniform_(0.05, 0.95)
measure(E, "empty().uniform_ after empty_cache")
E2 = torch.empty((1, b, L, q), device=dev).uniform_(0.05, 0.95)
measure(E2, "second tensor, first still alive")
del E, E2; engine.release_workspaces(); torch.cuda.empty_cache()
pad = torch.empty(3 << 30, dtype=torch.uint8, device=dev)
E = torch.empty((1, b, L, q), device=dev).uniform_(0.05, 0.95)
measure(E, "after a 3 GB pad allocation")
del E, pad; engine.release_workspaces(); torch.cuda.empty_cache()
E = torch.rand((1, b, L, q), device=dev) * 0.9 + 0.05
measure(E, "rand * 0.9 + 0.05 again")
