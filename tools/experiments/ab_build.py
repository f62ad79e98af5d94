"""A/B of compile-time variants of the engine in one process on BASELINE config 3:
   python ab_build.py "" "HMM_RS_WPE=4" "HMM_SUB=16;HMM_RS_WPE=4" ...
Each argument is a ';'-separated list of -D defines; per-kernel times come from the HIP-event profile."""
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
    path = "/tmp/libhmm_v%d.so" % i
    hb.build(out=path, defines=[x for x in defs.split(";") if x])
    paths.append(path)
A, pi = gene15(dev)
E = torch.rand((1, b, L, q), device=dev) * 0.9 + 0.05
out = torch.empty_like(E)
GROUPS = [int(x) for x in os.environ.get("AB_GROUPS", "1").split(",")]
CHUNK = int(os.environ.get("AB_CHUNK", "0"))
for rnd in range(2):
  for ngr in GROUPS:
    for defs, path in zip(variants, paths):
        engine._lib = None; engine.LIB_PATH = path; engine.release_workspaces()
        engine.set_option(engine.OPT_GROUPS, ngr)
        engine.set_option(engine.OPT_CHUNK, CHUNK)
        defs = "%s groups=%d" % (defs, ngr)
        prof = engine.Profile()
        engine.posterior(A, pi, E, out=out)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for r in range(5):
            engine.posterior(A, pi, E, out=out, profile=prof)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        k = prof.read()
        ms = {n: v[0] / max(v[1], 1) for n, v in k.items()}
        t1 = time.perf_counter()
        for r in range(5): engine.forward(A, pi, E, want_log_alpha=False)
        torch.cuda.synchronize(); dl = (time.perf_counter() - t1) / 5
        print("%-32s" % (defs or "(default)"), {n: round(v, 3) for n, v in ms.items()}, "pass %.3f ms  loglik %.3f ms" % (dt * 1e3, dl * 1e3),
              "chk %.6f" % float(out[0, ::97, ::997].double().sum()), flush=True)
        prof.close()
