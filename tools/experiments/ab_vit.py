"""A/B of compile-time variants of hmm_viterbi x batch-group settings on BASELINE config 4:
   python ab_vit.py "" "HMM_VRS_WPE=2" ..."""
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
    path = "/tmp/libhmm_vt%d.so" % i
    hb.build(out=path, defines=[x for x in defs.split(";") if x])
    paths.append(path)
A, pi = gene15(dev)
logA = torch.log(A); logpi = torch.log(pi)
logE = torch.log(torch.rand((1, b, L, q), device=dev) * 0.9 + 0.05)
for defs, path in zip(variants, paths):
    engine._lib = None; engine.LIB_PATH = path; engine.release_workspaces()
    for n in (0,):
        engine.set_option(engine.OPT_VGROUPS, n); engine.release_workspaces()
        for _ in range(2): engine.viterbi(logA, logpi, logE)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): p, s = engine.viterbi(logA, logpi, logE)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print("%-20s groups %d: %.3f ms" % (defs or "(default)", n, dt * 1e3), flush=True)
