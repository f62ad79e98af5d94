"""17..64-state models (one wave per sequence): time per pass.  python midq_time.py [q b L]"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hmm_layer_amd import engine
dev = 'cuda:0'
q, b, L = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (29, 1024, 100000)
torch.manual_seed(0)
A = torch.rand((1, q, q), device=dev) ** 4; A = A / A.sum(-1, keepdim=True)
pi = torch.full((1, q), 1 / q, device=dev)
E = torch.rand((1, b, L, q), device=dev) * 0.9 + 0.05
out = torch.empty_like(E)
for name, fn in (("loglik", lambda: engine.forward(A, pi, E, want_log_alpha=False)), ("posterior", lambda: engine.posterior(A, pi, E, out=out))):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print("%s q=%d b=%d L=%d: %.2f ms  %.3g cells/s  (%.3f us/step)" % (name, q, b, L, dt * 1e3, b * L * q / dt, dt / L * 1e6), flush=True)
logA = torch.log(A.clamp_min(1e-30)); logpi = torch.log(pi)
del out
logE = torch.log(E)
fn = lambda: engine.viterbi(logA, logpi, logE)
fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): fn()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
print("viterbi q=%d b=%d L=%d: %.2f ms  %.3g cells/s  (%.3f us/step)" % (q, b, L, dt * 1e3, b * L * q / dt, dt / L * 1e6), flush=True)
