import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hmm_layer_amd import engine
from oracle import params
dev = 'cuda:0'
b, L, q = 1024, 100000, 15
A = params.intended_A15().to(dev)
logA = torch.log(A)[None]; logpi = torch.log(torch.full((1, q), 1 / q, device=dev))
logE = torch.log(torch.rand((1, b, L, q), device=dev) * 0.9 + 0.05)
for _ in range(2): engine.viterbi(logA, logpi, logE)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): p, s = engine.viterbi(logA, logpi, logE)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print("viterbi %.3f ms/pass  %.3g cells/s" % (dt * 1e3, b * L * q / dt), "ws MB", engine.lib().hmm_viterbi_workspace_bytes(1, b, L, q) / 1e6)
