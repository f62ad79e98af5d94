"""hmm_viterbi at BASELINE config 4 (and a few other shapes) for every batch-group setting (HMM_OPT_VGROUPS);
paths and scores must be identical whatever the grouping."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hmm_layer_amd import engine
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _model import gene15
dev = 'cuda:0'
q = 15
A, pi = gene15(dev)
logA = torch.log(A); logpi = torch.log(pi)
for b, L in ((1024, 100000), (256, 100000), (32, 9999)):
    logE = torch.log(torch.rand((1, b, L, q), device=dev) * 0.9 + 0.05)
    ref = None
    for n in (1, 2, 3, 4, 6, 8, 0):
        engine.set_option(engine.OPT_VGROUPS, n)
        engine.release_workspaces()
        for _ in range(2): engine.viterbi(logA, logpi, logE)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): p, s = engine.viterbi(logA, logpi, logE)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        if ref is None: ref = (p.clone(), s.clone())
        same = torch.equal(p, ref[0]) and torch.equal(s, ref[1])
        print("b=%d L=%d groups %d: %.3f ms/pass  %.3g cells/s  ws %.0f MB  identical: %s" % (b, L, n, dt * 1e3, b * L * q / dt,
              engine.lib().hmm_viterbi_workspace_bytes(1, b, L, q) / 1e6, same), flush=True)
    del logE, ref, p, s
    torch.cuda.empty_cache()
