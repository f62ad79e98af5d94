"""Does any call write past its workspace or its output?  Guard bytes behind both, checked after the call."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hmm_layer_amd import engine
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _model import gene15
DEV = torch.device("cuda:0")
A, pi = gene15(DEV)
GUARD = 1 << 20
for (b, L, groups) in ((256, 70000, 1), (256, 70000, 2), (256, 70000, 4), (37, 1031, 1), (1024, 100000, 1)):
    E = torch.rand((1, b, L, 15), device=DEV) * 0.9 + 0.05
    with engine.option(engine.OPT_GROUPS, groups):
        need = engine.lib().hmm_workspace_bytes(engine.OP_POSTERIOR, 1, b, L, 15)
        ws = torch.full((need + GUARD,), 0xAB, dtype=torch.uint8, device=DEV)
        key = (DEV.index, torch.cuda.current_stream(DEV).cuda_stream)
        engine._workspaces[key] = ws[:need]                      # the engine sees exactly `need` bytes
        big = torch.full((E.numel() + GUARD // 4,), 7.0, device=DEV)
        out = big[:E.numel()].view_as(E)
        engine.posterior(A, pi, E, out=out)
        torch.cuda.synchronize()
    ok_ws = bool((ws[need:] == 0xAB).all())
    ok_out = bool((big[E.numel():] == 7.0).all())
    print("b=%d L=%d groups=%d need=%.1f MB  workspace guard intact: %s  output guard intact: %s" % (b, L, groups, need / 1e6, ok_ws, ok_out), flush=True)
    engine.release_workspaces()
    del E, ws, big, out
