"""k_forward for two builds (coalesced loader on / off) over successive allocations of the same tensors in ONE
process: is the run-to-run spread a property of the coalesced load pattern?"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hmm_layer_amd import build as hb, engine
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _model import gene15
dev = torch.device('cuda:0')
b, L, q = 1024, 100000, 15
variants = sys.argv[1:] or ["", "HMM_COALESCE_F=0"]
paths = []
for i, defs in enumerate(variants):
    path = "/tmp/libhmm_al%d.so" % i
    hb.build(out=path, defines=[x for x in defs.split(";") if x])
    paths.append(path)
A, pi = gene15(dev)
for trial in range(6):
    E = torch.empty((1, b, L, q), device=dev).uniform_(0.05, 0.95)
    out = torch.empty_like(E)
    res = []
    for defs, path in zip(variants, paths):
        engine._lib = None; engine.LIB_PATH = path; engine.release_workspaces()
        prof = engine.Profile()
        engine.posterior(A, pi, E, out=out)
        for _ in range(4): engine.posterior(A, pi, E, out=out, profile=prof)
        torch.cuda.synchronize()
        k = prof.read(); prof.close()
        res.append("%s fwd %.3f bwd %.3f" % (defs or "(default)", k["forward"][0] / k["forward"][1], k["backward"][0] / k["backward"][1]))
    print("allocation %d: " % trial + "   ".join(res), flush=True)
    del E, out
    engine.release_workspaces(); torch.cuda.empty_cache()
    pad = torch.empty((trial + 1) << 28, dtype=torch.uint8, device=dev)     # shift what the next allocation gets
