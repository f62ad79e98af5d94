import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hmm_layer_amd import engine
dev = 'cuda:0'
q, b, L = 1027, 1024, 256
torch.manual_seed(0)
A = torch.rand((1, q, q), device=dev) ** 4; A = A / A.sum(-1, keepdim=True)
pi = torch.full((1, q), 1 / q, device=dev)
E = torch.rand((1, b, L, q), device=dev) * 0.9 + 0.05
for name, fn in (("loglik", lambda: engine.forward(A, pi, E, want_log_alpha=False)), ("posterior", lambda: engine.posterior(A, pi, E))):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    flops = 2.0 * b * q * q * L * (1 if name == "loglik" else 2)
    print("%s q=%d b=%d L=%d: %.2f ms  %.3g cells/s  %.1f TFLOP/s  (%.1f us/step)" % (name, q, b, L, dt * 1e3, b * L * q / dt, flops / dt / 1e12, dt / L * 1e6))
