"""Repeats the grouped-pipeline comparison (OPT_GROUPS 1 / 2 / 4 bitwise equal?) and reports where it differs."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hmm_layer_amd import engine
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _model import gene15
DEV = "cuda:0"
torch.manual_seed(5)
b, L, q = 256, 70000, 15
A, pi = gene15(DEV)
E = torch.rand((1, b, L, q), device=DEV) * 0.9 + 0.05
for rep in range(6):
    outs = []
    for groups in (1, 2, 4):
        engine.release_workspaces()
        s = torch.cuda.Stream()
        with engine.option(engine.OPT_GROUPS, groups), torch.cuda.stream(s):
            E2 = E * 1.0
            out, ll = engine.posterior(A, pi, E2)
            chk = out.sum(-1)
        s.synchronize()
        outs.append((out, ll, float((chk - 1).abs().max())))
    for gi, (out, ll, c) in enumerate(outs):
        d = (out - outs[0][0]).abs()
        bad = torch.nonzero(d.amax(dim=(0, 2, 3)) > 0).flatten().tolist()
        msg = ""
        if bad:
            i = bad[0]
            pos = torch.nonzero(d[0, i].amax(-1) > 0).flatten()
            msg = " first bad seq %d: %d positions from %d to %d, max %.3g" % (i, pos.numel(), int(pos[0]), int(pos[-1]), float(d[0, i].max()))
        print("rep", rep, "groups", (1, 2, 4)[gi], "rowsum err %.2g" % c, "ll equal", bool(torch.equal(ll, outs[0][1])), "bad seqs", len(bad), msg, flush=True)
