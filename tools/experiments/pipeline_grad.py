"""loglik / loglik_grad on emitter-generated input (peaked class probabilities): ms and routing, auto vs off."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from hmm_layer_amd import engine
from pipeline_input import gene_x
from oracle import params
import bench
dev = torch.device("cuda:0")
b, L = 1024, 100000
em = bench.gene_emitter(dev, b, L)
A = params.intended_A15(200, 4500, 10000).to(dev).unsqueeze(0)
pi = torch.full((1, 15), 1 / 15, device=dev)
x = gene_x(b, L, 6.0, 0.01, dev)
E = em.forward_fused(x.unsqueeze(0)).reshape(1, b, L, 15).contiguous()
del x


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for name, mode in (("auto", engine.EXACT_AUTO), ("off", engine.EXACT_OFF)):
    with engine.option(engine.OPT_EXACT, mode):
        t1 = timed(lambda: engine.forward(A, pi, E, want_log_alpha=False))
        n1 = engine.exact_count(engine.OP_LOGLIK, (1, b, L, 15))
        t2 = timed(lambda: engine.loglik_grad(A, pi, E))
        d2 = engine.exact_detail((1, b, L, 15))
        t3 = timed(lambda: engine.forward(A, pi, E))
        n3 = engine.exact_count(engine.OP_FORWARD, (1, b, L, 15))
    print(name, "loglik %.2f ms (%d routed)  loglik_grad %.2f ms %s  log_alpha %.2f ms (%d routed)" % (t1, n1, t2, d2, t3, n3), flush=True)
