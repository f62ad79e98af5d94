"""29-state two-copy gene model at config-3 size: posterior / log-likelihood through the chunked 32-state scan."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hmm_layer_amd import engine
from hmm_layer_amd.gene_pred_hmm_transitioner import GenePredMultiHMMTransitioner
dev = torch.device("cuda:0")
tr = GenePredMultiHMMTransitioner(k=2, initial_exon_len=200, initial_intron_len=4500, initial_ir_len=10000).to(dev)
with torch.no_grad():
    A = tr.make_A().contiguous(); pi = tr.make_initial_distribution().reshape(1, -1).contiguous()
for (b, L) in ((1024, 100000), (32, 9999), (128, 100000)):
    E = torch.rand((1, b, L, 29), device=dev) * 0.9 + 0.05
    out = torch.empty_like(E)
    for name, mode in (("auto", engine.EXACT_AUTO), ("serial", engine.EXACT_ALWAYS)):
        with engine.option(engine.OPT_EXACT, mode):
            res = []
            for fn in (lambda: engine.posterior(A, pi, E, out=out), lambda: engine.forward(A, pi, E, want_log_alpha=False)):
                fn(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3): fn()
                torch.cuda.synchronize()
                res.append((time.perf_counter() - t0) / 3 * 1e3)
        print("b=%d L=%d %-6s posterior %.3f ms  loglik %.3f ms  (%.3g cells/s)" % (b, L, name, res[0], res[1], b * L * 29 / res[0] * 1e3), flush=True)
    del E, out
