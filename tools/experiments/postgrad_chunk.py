"""hmm_posterior_grad per chunk: time against the chunk length (HMM_OPT_CHUNK) at a few shapes."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hmm_layer_amd import engine
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _model import gene15
dev = 'cuda:0'
q = 15
A, pi = gene15(dev)
for b, L in ((32, 9999), (8, 9999), (128, 9999), (512, 9999), (2, 100000)):
    E = torch.rand((1, b, L, q), device=dev) * 0.9 + 0.05
    gam, _ = engine.posterior(A, pi, E, mode=engine.POST_PROB)
    lab = torch.multinomial(gam.reshape(-1, q).clamp_min(0) + 1e-30, 1).reshape(1, b, L, 1)
    G = torch.zeros((1, b, L, q), device=dev).scatter_(3, lab, -1.0)
    del gam, lab
    res = []
    for chunk in (0, 16, 32, 48, 64, 128):
        engine.set_option(engine.OPT_CHUNK, chunk)
        fn = lambda: engine.posterior_grad(A, pi, E, G, mode=engine.POST_LOG)
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize(); res.append("%s: %.2f" % (chunk or "auto(%d)" % engine.chunk_len(1, b, L, q), (time.perf_counter() - t0) / 5 * 1e3))
    engine.set_option(engine.OPT_CHUNK, 0)
    print("b=%4d L=%6d  ms by chunk length  " % (b, L) + "  ".join(res), flush=True)
