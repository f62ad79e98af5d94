import sys, os, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hmm_layer_amd import engine
from hmm_layer_amd.gene_pred_hmm_transitioner import GenePredMultiHMMTransitioner
dev = torch.device("cuda:0")
tr = GenePredMultiHMMTransitioner(k=2, initial_exon_len=200, initial_intron_len=4500, initial_ir_len=10000).to(dev)
with torch.no_grad():
    A = tr.make_A().contiguous(); pi = tr.make_initial_distribution().reshape(1, -1).contiguous()
b, L = 256, 100000
torch.manual_seed(0)
E = torch.rand((1, b, L, 29), device=dev) * 0.9 + 0.05
with engine.option(engine.OPT_EXACT, engine.EXACT_OFF):
    o1, l1 = engine.posterior(A, pi, E)
o2, l2 = engine.posterior(A, pi, E)
print("serial count", engine.exact_count(engine.OP_POSTERIOR, (1, b, L, 29)))
d = (o1 - o2).abs().amax(dim=(0, 2, 3))
bad = torch.nonzero(d > 0).flatten().tolist()
print("sequences recomputed:", bad, "max diff", [float(d[i]) for i in bad], "ll diff", [float((l1 - l2)[0, i]) for i in bad])
# locate where along the sequence the scan and serial results differ
for i in bad[:3]:
    dd = (o1[0, i] - o2[0, i]).abs().amax(-1)
    nz = torch.nonzero(dd > 1e-6).flatten()
    print(i, "positions differing > 1e-6:", nz.numel(), nz[:5].tolist(), nz[-5:].tolist() if nz.numel() else None)
    # certificate by hand from the serial posterior: Sg small where?
