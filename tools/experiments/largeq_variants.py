"""A/B of large-q GEMM compile-time variants (-D defines, comma separated; "" = product build) in one process.
The ablation macros used for DESIGN.md section 9 (LQ_ABL_NOMFMA / NOLOAD / MFMAONLY) lived in the kernel only for
those runs; add your own #ifdef to test a variant."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hmm_layer_amd import build as hb, engine
dev = 'cuda:0'
q, b, L = 1027, 1024, 128
torch.manual_seed(0)
A = torch.rand((1, q, q), device=dev) ** 4; A = A / A.sum(-1, keepdim=True)
pi = torch.full((1, q), 1 / q, device=dev)
E = torch.rand((1, b, L, q), device=dev) * 0.9 + 0.05
for defs in sys.argv[1].split(","):
    path = "/tmp/libhmm_lq_%s.so" % defs.replace(";", "_").replace("=", "")
    hb.build(out=path, defines=[d for d in defs.split(";") if d])
    engine._lib = None; engine.LIB_PATH = path
    fn = lambda: engine.forward(A, pi, E, want_log_alpha=False)
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print("%-40s %.1f us/step  %.1f TFLOP/s" % (defs or "base", dt / L * 1e6, 2.0 * b * q * q * L / dt / 1e12), flush=True)
