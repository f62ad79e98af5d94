"""Posterior gradients of the multi-copy gene models on the whole-sequence sweeps (33..64 states have no chunked
path): python midq_postgrad_time.py [k b L]"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hmm_layer_amd import engine
from hmm_layer_amd.gene_pred_hmm_transitioner import GenePredMultiHMMTransitioner
dev = 'cuda:0'
k, b, L = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (3, 32, 9999)
q = 1 + 14 * k
tr = GenePredMultiHMMTransitioner(k=k, initial_exon_len=200, initial_intron_len=4500, initial_ir_len=10000)
with torch.no_grad():
    A = tr.make_A()[:1].to(dev).contiguous(); pi = tr.make_initial_distribution().reshape(1, q).to(dev)
torch.manual_seed(0)
E = torch.rand((1, b, L, q), device=dev) * 0.9 + 0.05
G = -torch.nn.functional.one_hot(torch.randint(0, q, (1, b, L), device=dev), q).float()
def timed(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("k=%d q=%d b=%d L=%d" % (k, q, b, L),
      {"posterior_grad": round(timed(lambda: engine.posterior_grad(A, pi, E, G, mode=engine.POST_LOG)), 2),
       "loglik_grad": round(timed(lambda: engine.loglik_grad(A, pi, E)), 2),
       "posterior": round(timed(lambda: engine.posterior(A, pi, E)), 2)}, flush=True)
