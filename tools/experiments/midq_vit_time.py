"""Viterbi of the multi-copy gene models (29 / 43 / 57 states), one wave per sequence: the sparse loop against the
all-candidates loop.  python midq_vit_time.py [k b L]"""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hmm_layer_amd import engine
from hmm_layer_amd.gene_pred_hmm_transitioner import GenePredMultiHMMTransitioner
dev = 'cuda:0'
k, b, L = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (2, 1024, 100000)
q = 1 + 14 * k
tr = GenePredMultiHMMTransitioner(k=k, initial_exon_len=200, initial_intron_len=4500, initial_ir_len=10000)
with torch.no_grad():
    A = tr.make_A()[:1].to(dev); pi = tr.make_initial_distribution().reshape(1, q).to(dev)
logA = torch.log(A); logpi = torch.log(pi)
torch.manual_seed(0)
logE = torch.log(torch.rand((1, b, L, q), device=dev) * 0.9 + 0.05)
res = {}
for name, force in (("sparse", 0), ("all candidates", 1)):
    with engine.option(engine.OPT_FORCE_DENSE, force):
        fn = lambda: engine.viterbi(logA, logpi, logE)
        res[name] = fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print("viterbi %-14s q=%d b=%d L=%d: %.2f ms  (%.3f us/step)" % (name, q, b, L, dt * 1e3, dt / L * 1e6), flush=True)
print("same paths:", bool((res["sparse"][0] == res["all candidates"][0]).all()), " same scores:", bool((res["sparse"][1] == res["all candidates"][1]).all()))
