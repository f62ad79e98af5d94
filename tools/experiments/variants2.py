"""A/B of apply-kernel variants (SUB, chunk length) in one process."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hmm_layer_amd import build as hb, engine
from hmm_layer_amd.gene_pred_hmm_transitioner import GenePredMultiHMMTransitioner
dev = torch.device('cuda:0')
tr = GenePredMultiHMMTransitioner(initial_exon_len=200, initial_intron_len=4500, initial_ir_len=10000).to(dev)
with torch.no_grad():
    A = tr.make_A().contiguous(); pi = tr.make_initial_distribution().reshape(1, -1).contiguous()
b, L, q = 1024, 100000, 15
E = torch.rand((1, b, L, q), device=dev) * 0.9 + 0.05
out = torch.empty_like(E)
configs = [(d, c) for d in sys.argv[1].split(",") for c in sys.argv[2].split(",")]
built = {}
for defs, chunk in configs:
    path = "/tmp/libhmm_%s.so" % defs.replace("=", "").replace(";", "_")
    if path not in built:
        hb.build(out=path, defines=[x for x in defs.split(";") if x])
        built[path] = 1
    engine._lib = None; engine.LIB_PATH = path; engine.release_workspaces()
    engine.lib(); engine.set_option(engine.OPT_CHUNK, int(chunk))
    prof = engine.Profile()
    for r in range(4):
        engine.posterior(A, pi, E, out=out, profile=prof)
    torch.cuda.synchronize()
    k = prof.read()
    ms = {n: v[0] / v[1] for n, v in k.items()}
    print("%-28s T=%-5s" % (defs, chunk), {n: round(v, 3) for n, v in ms.items()}, "total %.3f" % sum(ms.values()),
          "chk %.6f" % float(out[0, ::97, ::997].double().sum()), flush=True)
