"""hmm_viterbi at BASELINE config 4 with the shipped settings, a few calls (for rocprofv3 --kernel-trace --stats)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hmm_layer_amd import engine
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _model import gene15
dev = 'cuda:0'
b, L, q = 1024, 100000, 15
A, pi = gene15(dev)
logA = torch.log(A); logpi = torch.log(pi)
logE = torch.log(torch.rand((1, b, L, q), device=dev) * 0.9 + 0.05)
for _ in range(2): engine.viterbi(logA, logpi, logE)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): p, s = engine.viterbi(logA, logpi, logE)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print("viterbi b=%d L=%d: %.3f ms/pass  %.3g cells/s" % (b, L, dt * 1e3, b * L * q / dt))
