"""hmm_loglik_grad for the 29-state two-copy model: per chunk against the two whole-sequence sweeps."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hmm_layer_amd import engine
from hmm_layer_amd.gene_pred_hmm_transitioner import GenePredMultiHMMTransitioner
dev = 'cuda:0'
tr = GenePredMultiHMMTransitioner(k=2, initial_exon_len=200, initial_intron_len=4500, initial_ir_len=10000)
with torch.no_grad():
    A = tr.make_A().to(dev).float()
q = 29
pi = torch.full((1, q), 1 / q, device=dev)
for b, L in ((32, 9999), (128, 9999), (512, 9999)):
    E = torch.rand((1, b, L, q), device=dev) * 0.9 + 0.05
    w = torch.rand((1, b), device=dev) + 0.5
    ts = []
    for how in (0, 1):
        engine.set_option(engine.OPT_PGCHUNK, how)
        fn = lambda: engine.loglik_grad(A, pi, E, w)
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): fn()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 3)
    n = engine.loglik_grad_serial_count((1, b, L, q))
    print("q=29 b=%4d L=%5d: loglik_grad whole-sequence sweeps %.2f ms, per chunk %.2f ms (%d sequences redone)" % (b, L, ts[0] * 1e3, ts[1] * 1e3, n), flush=True)
