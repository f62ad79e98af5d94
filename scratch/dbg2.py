import sys, time, os, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hmm_layer_amd import engine
from hmm_layer_amd.gene_pred_hmm_transitioner import GenePredMultiHMMTransitioner
dev = torch.device('cuda:0')
print("cpus", os.cpu_count(), flush=True)
tr = GenePredMultiHMMTransitioner(initial_exon_len=200, initial_intron_len=4500, initial_ir_len=10000).to(dev)
with torch.no_grad():
    A = tr.make_A().contiguous(); pi = tr.make_initial_distribution().reshape(1, -1).contiguous()
for b, L in [(64, 10000), (256, 10000), (1024, 10000), (1024, 100000)]:
    E = torch.rand((1, b, L, 15), device=dev) * 0.9 + 0.05
    out = torch.empty_like(E)
    prof = engine.Profile()
    torch.cuda.synchronize(); t0 = time.time()
    engine.posterior(A, pi, E, out=out, profile=prof)
    torch.cuda.synchronize(); t1 = time.time()
    print(b, L, "T", engine.chunk_len(1, b, L, 15), "first call %.1f ms" % ((t1 - t0) * 1e3), {k: round(v[0], 3) for k, v in prof.read().items()}, flush=True)
    for _ in range(3):
        engine.posterior(A, pi, E, out=out, profile=prof)
    torch.cuda.synchronize()
    r = prof.read()
    print("   avg ms", {k: round(v[0] / max(v[1], 1), 3) for k, v in r.items()}, "rowsum err", float((out.sum(-1) - 1).abs().max()), flush=True)
    del E, out
