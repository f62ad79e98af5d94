import sys, os, torch
sys.path.insert(0, "/root/repo")
sys.path.insert(0, "/root/repo/tools/experiments")
from hmm_layer_amd import engine
from _model import gene15
dev = "cuda:0"
A, pi = gene15(dev)
logA, logpi = torch.log(A.clamp_min(1e-30)), torch.log(pi)
for (b, L) in ((1, 1000000), (8, 1000000)):
    logE = torch.log(torch.rand((1, b, L, 15), device=dev) * 0.9 + 0.05)
    for _ in range(3): engine.viterbi(logA, logpi, logE)
    torch.cuda.synchronize()
