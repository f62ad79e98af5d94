import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hmm_layer_amd import engine
from hmm_layer_amd.gene_pred_hmm_transitioner import GenePredMultiHMMTransitioner
dev = torch.device('cuda:0')
b = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
L = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
n = int(sys.argv[3]) if len(sys.argv) > 3 else 2
tr = GenePredMultiHMMTransitioner(initial_exon_len=200, initial_intron_len=4500, initial_ir_len=10000).to(dev)
with torch.no_grad():
    A = tr.make_A().contiguous(); pi = tr.make_initial_distribution().reshape(1, -1).contiguous()
E = torch.rand((1, b, L, 15), device=dev) * 0.9 + 0.05
out = torch.empty_like(E)
for _ in range(n):
    engine.posterior(A, pi, E, out=out)
torch.cuda.synchronize()
print("done", float(out[0,0,0].sum()))
