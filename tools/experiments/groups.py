import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hmm_layer_amd import engine
from hmm_layer_amd.gene_pred_hmm_transitioner import GenePredMultiHMMTransitioner
dev = torch.device('cuda:0')
tr = GenePredMultiHMMTransitioner(initial_exon_len=200, initial_intron_len=4500, initial_ir_len=10000).to(dev)
with torch.no_grad():
    A = tr.make_A().contiguous(); pi = tr.make_initial_distribution().reshape(1, -1).contiguous()
b, L, q = 1024, 100000, 15
E = torch.rand((1, b, L, q), device=dev) * 0.9 + 0.05
out = torch.empty_like(E)
ref = None
for G in sys.argv[1].split(","):
    engine.set_option(engine.OPT_GROUPS, int(G))
    engine.release_workspaces()
    for _ in range(2): engine.posterior(A, pi, E, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): o, ll = engine.posterior(A, pi, E, out=out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    if ref is None: ref = (out.clone(), ll.clone())
    print("groups", G, "%.3f ms/pass" % (dt * 1e3), "%.3g cells/s" % (b * L * q / dt), "same:", torch.equal(out, ref[0]), torch.equal(ll, ref[1]), flush=True)
