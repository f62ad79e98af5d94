"""33..64 states: the chunked scan vs one wave per sequence."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from hmm_layer_amd import engine
from hmm_layer_amd.gene_pred_hmm_transitioner import GenePredMultiHMMTransitioner
dev = torch.device("cuda:0")
tr = GenePredMultiHMMTransitioner(k=3, initial_exon_len=200, initial_intron_len=4500, initial_ir_len=10000).to(dev)
with torch.no_grad():
    A = tr.make_A().contiguous(); pi = tr.make_initial_distribution().reshape(1, -1).contiguous()
q = A.shape[-1]
for b, L in ((1, 100000), (8, 100000), (32, 100000), (96, 100000), (32, 9999), (1, 1000000)):
    E = torch.rand((1, b, L, q), device=dev) * 0.9 + 0.05
    res = []
    for mode in (engine.EXACT_AUTO, engine.EXACT_ALWAYS):
        with engine.option(engine.OPT_EXACT, mode):
            engine.posterior(A, pi, E); torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(2): engine.posterior(A, pi, E)
            torch.cuda.synchronize(); tp = (time.perf_counter() - t0) / 2
            engine.forward(A, pi, E, want_log_alpha=False); torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(2): engine.forward(A, pi, E, want_log_alpha=False)
            torch.cuda.synchronize(); tl = (time.perf_counter() - t0) / 2
        res.append((tp * 1e3, tl * 1e3))
    print("q=%d b=%3d L=%7d: chunked posterior %.2f ms loglik %.2f ms | one wave per sequence %.2f / %.2f ms" % (
        q, b, L, res[0][0], res[0][1], res[1][0], res[1][1]), flush=True)
    del E
