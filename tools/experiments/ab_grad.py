"""A/B of compile-time variants of hmm_loglik_grad on BASELINE config 3:  python ab_grad.py "" "HMM_GRAD_AB=1" ..."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hmm_layer_amd import build as hb, engine
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _model import gene15
dev = torch.device('cuda:0')
b, L, q = 1024, 100000, 15
variants = sys.argv[1:] or [""]
paths = []
for i, defs in enumerate(variants):
    path = "/tmp/libhmm_g%d.so" % i
    hb.build(out=path, defines=[x for x in defs.split(";") if x])
    paths.append(path)
A, pi = gene15(dev)
E = torch.rand((1, b, L, q), device=dev) * 0.9 + 0.05
w = torch.rand((1, b), device=dev) + 0.5
for rnd in range(2):
    for defs, path in zip(variants, paths):
        engine._lib = None; engine.LIB_PATH = path; engine.release_workspaces()
        o = engine.loglik_grad(A, pi, E, w); torch.cuda.synchronize(); chk = float(o[0].double().abs().sum()); del o
        t0 = time.perf_counter()
        for r in range(5):
            o = engine.loglik_grad(A, pi, E, w); del o
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print("%-24s loglik_grad %.3f ms   |dA| %.6g" % (defs or "(default)", dt * 1e3, chk), flush=True)
