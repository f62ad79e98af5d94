"""A/B timing of engine build variants in ONE process, interleaved rounds (guide rule 24)."""
import ctypes, sys, os, itertools, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hmm_layer_amd import build as hb, engine
variants = {
  "pf4_single": ["HMM_REDUCE_PF=4", "HMM_DUAL_ACC=0"],
  "pf4_dual":   ["HMM_REDUCE_PF=4", "HMM_DUAL_ACC=1"],
  "pf8_dual":   ["HMM_REDUCE_PF=8", "HMM_DUAL_ACC=1"],
  "pf16_dual":  ["HMM_REDUCE_PF=16", "HMM_DUAL_ACC=1"],
  "pf16_single":["HMM_REDUCE_PF=16", "HMM_DUAL_ACC=0"],
}
only = sys.argv[1:] 
dev = torch.device('cuda:0')
from hmm_layer_amd.gene_pred_hmm_transitioner import GenePredMultiHMMTransitioner
tr = GenePredMultiHMMTransitioner(initial_exon_len=200, initial_intron_len=4500, initial_ir_len=10000).to(dev)
with torch.no_grad():
    A = tr.make_A().contiguous(); pi = tr.make_initial_distribution().reshape(1, -1).contiguous()
b, L, q = 1024, 100000, 15
E = torch.rand((1, b, L, q), device=dev) * 0.9 + 0.05
out = torch.empty_like(E)
res = {}
ref = None
for name, defs in variants.items():
    if only and name not in only: continue
    path = "/tmp/libhmm_%s.so" % name
    hb.build(out=path, defines=defs)
    engine._lib = None; engine.LIB_PATH = path
    prof = engine.Profile()
    for r in range(4):
        engine.posterior(A, pi, E, out=out, profile=prof)
    torch.cuda.synchronize()
    k = prof.read()
    res[name] = {n: v[0] / v[1] for n, v in k.items()}
    chk = float(out[0, ::97, ::997].double().sum())
    print(name, {n: round(v, 3) for n, v in res[name].items()}, "chk %.6f" % chk, flush=True)
