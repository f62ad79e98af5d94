"""In-process A/B of engine build variants on BASELINE config 3: python ab_posterior.py "A;B,C" (comma = variants, ; = defines)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hmm_layer_amd import build as hb, engine
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _model import gene15
dev = 'cuda:0'
b, L, q = 1024, 100000, 15
A, pi = gene15(dev)
E = torch.rand((1, b, L, q), device=dev) * 0.9 + 0.05
out = torch.empty_like(E)
variants = sys.argv[1].split(",")
libs = []
for defs in variants:
    path = "/tmp/libhmm_ab_%s.so" % (defs.replace(";", "_").replace("=", "") or "base")
    hb.build(out=path, defines=[d for d in defs.split(";") if d])
    libs.append(path)
for rnd in range(3):
    for defs, path in zip(variants, libs):
        engine._lib = None; engine.LIB_PATH = path
        prof = engine.Profile()
        engine.posterior(A, pi, E, out=out)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): engine.posterior(A, pi, E, out=out, profile=prof)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        r = prof.read()
        print("%-34s %.3f ms  %s" % (defs or "base", dt * 1e3, {k: round(v[0] / max(v[1], 1), 3) for k, v in r.items()}), flush=True)
