"""hmm_forward (log-likelihood, log alpha) and hmm_backward on emitter-generated gene-model input (peaked class
probabilities: the input on which the clamp certificates fire): ms per call with routing auto / off, and the error of
log alpha of sampled sequences against the fp64 serial recursion, compared as tests/test_engine_gpu.py does.
    python tools/experiments/pipeline_entries.py [b] [L] [scale]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from hmm_layer_amd import engine  # noqa: E402
from oracle import params, textbook  # noqa: E402
from pipeline_input import gene_x  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
L = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 6.0
dev = torch.device("cuda:0")
from hmm_layer_amd.gene_pred_hmm_emitter import GenePredHMMEmitter
em = GenePredHMMEmitter(start_codons=[("ATG", 1.)], stop_codons=[("TAG", .34), ("TAA", .33), ("TGA", .33)],
                        intron_begin_pattern=[("NGT", .99), ("NGC", .005), ("NAT", .005)],
                        intron_end_pattern=[("AGN", .99), ("ACN", .01)])
em.build((1, b, L, 15))
g = torch.Generator().manual_seed(0)
with torch.no_grad():
    em.emission_kernel.copy_(torch.randn(em.emission_kernel.shape, generator=g))
em = em.to(dev)
A = params.intended_A15(200, 4500, 10000).to(dev).unsqueeze(0)
pi = torch.full((1, 15), 1 / 15, device=dev)
x = gene_x(b, L, scale, 0.01, dev)
with torch.no_grad():
    em.recurrent_init()
    E = em.forward_fused(x.unsqueeze(0)).reshape(1, b, L, 15).contiguous()
del x


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, r


def excess_over_tolerance(la, idx):
    la64, ll64 = textbook.log_alpha(A[0].cpu().numpy(), pi[0].cpu().numpy(), E[0, idx].cpu().numpy())
    got = la[0, idx].cpu().numpy()
    ref = la64.max(-1, keepdims=True)
    with np.errstate(over="ignore", under="ignore"):
        p, p64 = np.exp(np.minimum(got - ref, 50.0)), np.exp(la64 - ref)
    return float((np.abs(p - p64) - (2e-5 + p64 * (3e-4 + 2e-7 * np.abs(ref)))).max()), ll64


res = {}
for name, mode in (("off", engine.EXACT_OFF), ("auto", engine.EXACT_AUTO)):
    with engine.option(engine.OPT_EXACT, mode):
        t_lb, _ = timed(lambda: engine.backward(A, E))          # (first: the 6 GB results kept below crowd the allocator)
        detb = engine.exact_detail((1, b, L, 15), op=engine.OP_BACKWARD)
        torch.cuda.empty_cache()
        t_ll, _ = timed(lambda: engine.forward(A, pi, E, want_log_alpha=False))
        t_la, (la, ll) = timed(lambda: engine.forward(A, pi, E))
        det = engine.exact_detail((1, b, L, 15), op=engine.OP_FORWARD)
    res[name] = la
    if name == "auto":
        # the sequences the routing changed most, and three fixed ones
        moved = (res["auto"][0] - res["off"][0]).abs().amax(dim=(1, 2))
        idx = sorted(set([0, b // 3, b - 1] + [int(i) for i in torch.topk(moved, min(3, b)).indices.tolist()]))
    else:
        idx = sorted(set([0, b // 3, b - 1]))
    excess, ll64 = excess_over_tolerance(la, idx)
    print("scale %g b %d L %d routing %-4s: loglik %.2f ms  log alpha %.2f ms %s  log beta %.2f ms %s   sequences %s: log alpha error over tolerance %.2e (<= 0 passes)  max|dll| %.2e" % (
        scale, b, L, name, t_ll, t_la, det, t_lb, detb, idx, excess, float(np.abs(ll[0, idx].cpu().numpy() - ll64).max())), flush=True)
excess_off, _ = excess_over_tolerance(res["off"], idx)
print("the same sequences with routing off: error over tolerance %.2e" % excess_off)
