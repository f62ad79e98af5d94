"""The two-copy (29-state) gene model on emitter-generated input: ms per posterior pass and routed sequences."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from hmm_layer_amd import engine
from hmm_layer_amd.gene_pred_hmm_emitter import GenePredHMMEmitter
from hmm_layer_amd.gene_pred_hmm_transitioner import GenePredMultiHMMTransitioner
from pipeline_input import gene_x
dev = torch.device("cuda:0")
b, L = int(sys.argv[1]) if len(sys.argv) > 1 else 256, int(sys.argv[2]) if len(sys.argv) > 2 else 100000
em = GenePredHMMEmitter(start_codons=[("ATG", 1.)], stop_codons=[("TAG", .34), ("TAA", .33), ("TGA", .33)],
                        intron_begin_pattern=[("NGT", .99), ("NGC", .005), ("NAT", .005)],
                        intron_end_pattern=[("AGN", .99), ("ACN", .01)], num_copies=2)
em.build((1, b, L, 15))
g = torch.Generator().manual_seed(0)
with torch.no_grad():
    em.emission_kernel.copy_(torch.randn(em.emission_kernel.shape, generator=g))
em = em.to(dev); em.recurrent_init()
tr = GenePredMultiHMMTransitioner(k=2, initial_exon_len=200, initial_intron_len=4500, initial_ir_len=10000).to(dev)
with torch.no_grad():
    A = tr.make_A().contiguous(); pi = tr.make_initial_distribution().reshape(1, -1).contiguous()
q = A.shape[-1]
for scale in (2.0, 6.0):
    x = gene_x(b, L, scale, 0.01, dev)
    E = em.forward_fused(x.unsqueeze(0)).reshape(1, b, L, q).contiguous()
    del x
    for name, mode in (("auto", engine.EXACT_AUTO), ("off", engine.EXACT_OFF)):
        with engine.option(engine.OPT_EXACT, mode):
            engine.posterior(A, pi, E); torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3): engine.posterior(A, pi, E)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
            n = engine.exact_count(engine.OP_POSTERIOR, (1, b, L, q))
        print("q=%d scale %g b %d L %d routing %-4s: %.2f ms, %d sequences redone serially" % (q, scale, b, L, name, dt * 1e3, n), flush=True)
    del E
