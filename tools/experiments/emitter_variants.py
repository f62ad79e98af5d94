import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hmm_layer_amd import build as hb, engine
from hmm_layer_amd.gene_pred_hmm_emitter import GenePredHMMEmitter
dev = 'cuda:0'
CODONS = dict(start_codons=[("ATG", 1.)], stop_codons=[("TAG", .34), ("TAA", .33), ("TGA", .33)],
              intron_begin_pattern=[("NGT", .99), ("NGC", .005), ("NAT", .005)], intron_end_pattern=[("AGN", .99), ("ACN", .01)])
b, L = 1024, 100000
em = GenePredHMMEmitter(**CODONS); em.build((1, b, L, 15)); em = em.to(dev); em.recurrent_init()
x = torch.empty((1, b, L, 20), device=dev)
x[..., :15] = torch.softmax(torch.randn((1, b, L, 15), device=dev), -1)
idx = torch.where(torch.rand((1, b, L), device=dev) < 0.01, torch.full((1, b, L), 4, device=dev), torch.randint(0, 4, (1, b, L), device=dev))
x[..., 15:] = torch.nn.functional.one_hot(idx, 5).float()
for defs in sys.argv[1].split(","):
    path = "/tmp/libhmm_em_%s.so" % defs.replace(";", "_")
    hb.build(out=path, defines=[d for d in defs.split(";") if d])
    engine._lib = None; engine.LIB_PATH = path
    for _ in range(2): E = em.forward_fused(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): E = em.forward_fused(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print("%-32s %.2f ms" % (defs, dt * 1e3), flush=True)
