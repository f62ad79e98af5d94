import time, torch, os, sys
sys.path.insert(0,'.')
from oracle import ref_cell, params
A = params.intended_A15(); q=15
p = ref_cell.HmmParams(A, torch.full((q,), 1.0/q))
E = torch.rand((1,1024,1000,q))*0.9+0.05
for th in (1, 4, 8, 16, 32):
    torch.set_num_threads(th)
    ref_cell.posterior_scaled(p, E[:, :, :50])
    t0=time.perf_counter(); ref_cell.posterior_scaled(p, E); dt=time.perf_counter()-t0
    print(th, "%.2fs"%dt, "%.3g cells/s"%(1024*1000*q/dt), flush=True)
