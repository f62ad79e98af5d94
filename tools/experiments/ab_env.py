"""In-process A/B of an engine option on BASELINE config 3: python ab_env.py SCAN2 0 1  (CHUNK, FORCE_DENSE, SCAN2, GROUPS, EXACT)"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hmm_layer_amd import engine
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _model import gene15
dev = 'cuda:0'
b, L, q = 1024, 100000, 15
A, pi = gene15(dev)
E = torch.rand((1, b, L, q), device=dev) * 0.9 + 0.05
out = torch.empty_like(E)
name, vals = sys.argv[1], sys.argv[2:]
for rnd in range(3):
    for v in vals:
        engine.set_option(getattr(engine, 'OPT_' + name.replace('HMM_ENGINE_', '')), int(v))
        prof = engine.Profile()
        engine.posterior(A, pi, E, out=out)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): engine.posterior(A, pi, E, out=out, profile=prof)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        r = prof.read()
        print("%s=%s  %.3f ms  %s" % (name, v, dt * 1e3, {k: round(x[0] / max(x[1], 1), 3) for k, x in r.items()}), flush=True)
