"""A/B of compile-time variants of the one-wave-per-sequence Viterbi (17..64 states) on the k-copy gene model:
   python ab_mqvit.py K B L "" "MQ_PF=16" ...   (each argument: ';'-separated -D defines)"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hmm_layer_amd import build as hb, engine
from hmm_layer_amd.gene_pred_hmm_transitioner import GenePredMultiHMMTransitioner
dev = 'cuda:0'
k, b, L = (int(v) for v in sys.argv[1:4])
variants = sys.argv[4:] or [""]
paths = []
for i, defs in enumerate(variants):
    path = "/tmp/libhmm_mq%d.so" % i
    hb.build(out=path, defines=[x for x in defs.split(";") if x])
    paths.append(path)
q = 1 + 14 * k
tr = GenePredMultiHMMTransitioner(k=k, initial_exon_len=200, initial_intron_len=4500, initial_ir_len=10000)
with torch.no_grad():
    A = tr.make_A()[:1].to(dev); pi = tr.make_initial_distribution().reshape(1, q).to(dev)
logA = torch.log(A); logpi = torch.log(pi)
torch.manual_seed(0)
logE = torch.log(torch.rand((1, b, L, q), device=dev) * 0.9 + 0.05)
for rnd in range(2):
    for defs, path in zip(variants, paths):
        engine._lib = None; engine.LIB_PATH = path; engine.release_workspaces()
        fn = lambda: engine.viterbi(logA, logpi, logE)
        r = fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        print("%-28s q=%d b=%d L=%d: %.2f ms  (%.3f us/step)  chk %d" % (defs or "(default)", q, b, L, dt * 1e3, dt / L * 1e6,
              int(r[0][0, ::7, ::101].sum())), flush=True)
