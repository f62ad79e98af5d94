"""The multi-copy gene models (29 / 43 / 57 states) on the one-wave-per-sequence kernels: sparse step against the
dense step, and against the default routing (chunked scan where it applies).  python midq_gene_time.py [k b L]"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hmm_layer_amd import engine
from hmm_layer_amd.gene_pred_hmm_transitioner import GenePredMultiHMMTransitioner
dev = 'cuda:0'
k, b, L = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (3, 1024, 100000)
q = 1 + 14 * k
tr = GenePredMultiHMMTransitioner(k=k, initial_exon_len=200, initial_intron_len=4500, initial_ir_len=10000)
with torch.no_grad():
    A = tr.make_A()[:1].to(dev).contiguous(); pi = tr.make_initial_distribution().reshape(1, q).to(dev)
torch.manual_seed(0)
E = torch.rand((1, b, L, q), device=dev) * 0.9 + 0.05
out = torch.empty_like(E)
def timed(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def row():
    return {"loglik": timed(lambda: engine.forward(A, pi, E, want_log_alpha=False)),
            "log_alpha": timed(lambda: engine.forward(A, pi, E, want_log_alpha=True)),
            "posterior": timed(lambda: engine.posterior(A, pi, E, out=out)),
            "post_log": timed(lambda: engine.posterior(A, pi, E, out=out, mode=engine.POST_LOG))}
print("k=%d q=%d b=%d L=%d" % (k, q, b, L))
print("default routing      ", {n: round(v, 2) for n, v in row().items()}, flush=True)
with engine.option(engine.OPT_EXACT, engine.EXACT_ALWAYS):
    print("serial, sparse step  ", {n: round(v, 2) for n, v in row().items()}, flush=True)
    with engine.option(engine.OPT_FORCE_DENSE, 1):
        print("serial, dense step   ", {n: round(v, 2) for n, v in row().items()}, flush=True)
