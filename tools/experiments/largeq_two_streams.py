"""Do two independent large-q recursions overlap on one GPU?  (q = 1027, b = 1024: each position is one
208-workgroup GEMM launch; two chains on two streams vs one after the other.)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hmm_layer_amd import engine
dev = "cuda:0"
q, b, L = 1027, 1024, 64
g = torch.Generator(device=dev).manual_seed(0)
A = torch.rand((1, q, q), device=dev, generator=g); A /= A.sum(-1, keepdim=True)
pi = torch.full((1, q), 1.0 / q, device=dev)
E1 = torch.rand((1, b, L, q), device=dev, generator=g) * 0.9 + 0.05
E2 = torch.rand((1, b, L, q), device=dev, generator=g) * 0.9 + 0.05
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def one(E, s):
    with torch.cuda.stream(s):
        return engine.forward(A, pi, E, want_log_alpha=False)

for _ in range(2):
    one(E1, s1); one(E2, s2)
torch.cuda.synchronize()
t0 = time.perf_counter(); one(E1, s1); torch.cuda.synchronize(); t1 = time.perf_counter()
print("one chain: %.2f ms (%.1f us per position)" % ((t1 - t0) * 1e3, (t1 - t0) * 1e6 / L))
t0 = time.perf_counter(); one(E1, s1); one(E2, s2); torch.cuda.synchronize(); t1 = time.perf_counter()
print("two chains on two streams: %.2f ms (%.1f us per position pair)" % ((t1 - t0) * 1e3, (t1 - t0) * 1e6 / L))
t0 = time.perf_counter(); one(E1, s1); one(E2, s1); torch.cuda.synchronize(); t1 = time.perf_counter()
print("two chains on one stream: %.2f ms" % ((t1 - t0) * 1e3))
