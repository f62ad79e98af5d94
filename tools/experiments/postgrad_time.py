"""hmm_posterior_grad: time per call at training-like and BASELINE shapes."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hmm_layer_amd import engine
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _model import gene15
dev = 'cuda:0'
A15, pi15 = gene15(dev)


def two_copy():
    from hmm_layer_amd.gene_pred_hmm_transitioner import GenePredMultiHMMTransitioner
    tr = GenePredMultiHMMTransitioner(k=2, initial_exon_len=200, initial_intron_len=4500, initial_ir_len=10000)
    with torch.no_grad():
        return tr.make_A().to(dev).float(), torch.full((1, 29), 1 / 29, device=dev)


A29, pi29 = two_copy()
SHAPES = [(15, 32, 9999), (15, 128, 9999), (15, 512, 9999), (15, 1024, 9999), (15, 2048, 9999), (15, 2, 100000),
          (15, 1024, 100000), (29, 32, 9999), (29, 128, 9999), (29, 512, 9999)]
if len(sys.argv) > 1:
    SHAPES = [s for s in SHAPES if s[0] == int(sys.argv[1])]
for q, b, L in SHAPES:
    A, pi = (A15, pi15) if q == 15 else (A29, pi29)
    E = torch.rand((1, b, L, q), device=dev) * 0.9 + 0.05
    # upstream gradient of a cross-entropy on log gamma against a labelling drawn from the posterior itself
    gam, _ = engine.posterior(A, pi, E, mode=engine.POST_PROB)
    lab = torch.multinomial(gam.reshape(-1, q).clamp_min(0) + 1e-30, 1).reshape(1, b, L, 1)
    G = torch.zeros((1, b, L, q), device=dev).scatter_(3, lab, -1.0)
    del gam, lab
    fn = lambda: engine.posterior_grad(A, pi, E, G, mode=engine.POST_LOG)
    ts = []
    for how in (0, 2, 1):
        engine.set_option(engine.OPT_PGCHUNK, how)
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): fn()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 3)
    nser = engine.posterior_grad_serial_count((1, b, L, q))
    dt = ts[1]
    fw = lambda: engine.posterior(A, pi, E, mode=engine.POST_LOG)
    fw(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): fw()
    torch.cuda.synchronize(); df = (time.perf_counter() - t0) / 3
    print("q=%d b=%5d L=%6d: posterior %.2f ms, posterior_grad serial %.2f ms, chunked %.2f ms  (%.3g cells/s); as shipped %.2f ms, %d sequences redone serially" % (q, b, L, df * 1e3, ts[0] * 1e3, dt * 1e3, b * L * q / dt, ts[2] * 1e3, nser), flush=True)
    del E, G
    engine.release_workspaces(); torch.cuda.empty_cache()
