"""Do results depend on what the workspace held before the call?  (uninitialised reads)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hmm_layer_amd import engine
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _model import gene15
DEV = torch.device("cuda:0")
A, pi = gene15(DEV)
torch.manual_seed(5)
for (b, L) in ((256, 70000), (37, 1031)):
    E = torch.rand((1, b, L, 15), device=DEV) * 0.9 + 0.05
    for groups in (1, 2, 4):
        with engine.option(engine.OPT_GROUPS, groups):
            need = engine.lib().hmm_workspace_bytes(engine.OP_POSTERIOR, 1, b, L, 15)
            res = []
            for fill in (0, 0xAB, 0xFF, 0x7F):
                ws = torch.full((need,), fill, dtype=torch.uint8, device=DEV)
                engine._workspaces[(DEV.index, torch.cuda.current_stream(DEV).cuda_stream)] = ws
                out, ll = engine.posterior(A, pi, E)
                torch.cuda.synchronize()
                res.append((out.clone(), ll.clone(), engine.exact_count(engine.OP_POSTERIOR, (1, b, L, 15))))
            print("b=%d L=%d groups=%d" % (b, L, groups), [(bool(torch.equal(r[0], res[0][0])), bool(torch.equal(r[1], res[0][1])), r[2]) for r in res], flush=True)
    del E
