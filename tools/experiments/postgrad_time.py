"""hmm_posterior_grad: time per call at training-like and BASELINE shapes."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hmm_layer_amd import engine
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _model import gene15
dev = 'cuda:0'
q = 15
A, pi = gene15(dev)
for b, L in ((32, 9999), (256, 9999), (1024, 9999), (1024, 100000)):
    E = torch.rand((1, b, L, q), device=dev) * 0.9 + 0.05
    G = torch.randn((1, b, L, q), device=dev)
    fn = lambda: engine.posterior_grad(A, pi, E, G, mode=engine.POST_LOG)
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    fw = lambda: engine.posterior(A, pi, E, mode=engine.POST_LOG)
    fw(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): fw()
    torch.cuda.synchronize(); df = (time.perf_counter() - t0) / 3
    print("b=%5d L=%6d: posterior %.2f ms, posterior_grad %.2f ms  (%.3g cells/s)" % (b, L, df * 1e3, dt * 1e3, b * L * q / dt), flush=True)
    del E, G
    engine.release_workspaces(); torch.cuda.empty_cache()
