"""Randomised check of hmm_posterior_grad's routing: for random models (gene topology, dense, sparse incl.
reducible, degenerate; the 29-state two-copy model), shapes, chunk lengths, emissions with dead / rare entries and upstream gradients
(dense random, labels the posterior supports, labels on arbitrary states) the shipped setting (per chunk where
the device-side rules allow, whole-sequence sweeps otherwise) must agree with the whole-sequence sweeps alone
(relative to the largest entry of each gradient, or to 1e-3 of the largest upstream weight where the gradient
vanishes identically, as for q = 1).
Test infrastructure: tests/test_postgrad_chunked_gpu.py runs 150 cases of it; for a longer run on the GPU box, from
the repo root:  python tests/postgrad_sweep.py [cases] [seed] [only this case | -1] [1 shipped | 2 no routing]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from hmm_layer_amd import engine
from oracle import params

dev = "cuda:0"
A15 = params.intended_A15().numpy().astype(np.float32)


def _two_copy():
    from hmm_layer_amd.gene_pred_hmm_transitioner import GenePredMultiHMMTransitioner
    tr = GenePredMultiHMMTransitioner(k=2, initial_exon_len=200, initial_intron_len=4500, initial_ir_len=10000)
    with torch.no_grad():
        return tr.make_A()[0].numpy().astype(np.float32)


A29 = _two_copy()
t = lambda x: torch.as_tensor(np.asarray(x), dtype=torch.float32, device=dev)


def run(ncase, seed, verbose=True, tol=2e-4, only=None, shipped=1):
    rng = np.random.default_rng(seed)
    bad = 0
    for case in range(ncase):
        kind = int(rng.integers(0, 5))
        if kind == 0:
            q = 15; A = A15.copy()
        elif kind == 4:                  # the 29-state two-copy gene model: rows of 32 lanes
            q = 29; A = A29.copy()
        else:
            q = int(rng.integers(1, 17))
            A = rng.random((q, q)).astype(np.float32) ** 3 + 1e-3
            if kind == 2:
                A *= rng.random((q, q)) < 0.4
                A += np.eye(q, dtype=np.float32) * 0.3
            if kind == 3:
                sub = rng.integers(0, 4)
                if sub == 0: A = np.eye(q, dtype=np.float32)
                elif sub == 1: A = np.roll(np.eye(q, dtype=np.float32), 1, axis=1)
                elif sub == 2: A[: q // 2, q // 2:] = 0; A[q // 2:, : q // 2] = 0
                else: A[rng.random(q) < 0.3] = 0
            A /= np.maximum(A.sum(-1, keepdims=True), 1e-30)
        pi = rng.random(q).astype(np.float32) + 0.1; pi /= pi.sum()
        b = int(rng.integers(1, 12)); L = int(rng.choice([2, 17, 100, 333, 999, 2500, 6001]))
        chunk = int(rng.choice([0, 16, 32, 64, 128]))
        E = (rng.random((b, L, q)) * 0.9 + 0.05).astype(np.float32)
        emis = int(rng.integers(0, 4))
        if emis == 1: E[rng.random(E.shape) < 0.1] = 0.0
        if emis == 2: E[rng.random(E.shape) < 0.15] = 1e-10
        if emis == 3: E *= np.float32(1.0 / 4096)
        mode = engine.POST_LOG if rng.random() < 0.5 else engine.POST_PROB
        gk = int(rng.integers(0, 3))
        Gd = rng.standard_normal(E.shape).astype(np.float32)
        Gl = -(rng.random(E.shape) < 0.1).astype(np.float32)
        if only is not None and case != only:
            continue
        with engine.option(engine.OPT_CHUNK, chunk):
            if gk == 0:
                G = Gd
            else:
                gam, _ = engine.posterior(t(A)[None], t(pi)[None], t(E)[None])
                gam = gam[0].cpu().numpy()
                if gk == 1: G = -(gam == gam.max(-1, keepdims=True)).astype(np.float32)
                else: G = Gl
            res = {}
            for how in (0, shipped):
                with engine.option(engine.OPT_PGCHUNK, how):
                    res[how] = [x.cpu().numpy() for x in engine.posterior_grad(t(A)[None], t(pi)[None], t(E)[None], t(G)[None], mode=mode)]
                    if how == shipped: nser = engine.posterior_grad_serial_count((1, b, L, q))
        errs = []
        absent_err = 0.0
        for idx, (s, c) in enumerate(zip(res[0], res[1])):
            fin = np.isfinite(s)
            same_nonfinite = np.array_equal(fin, np.isfinite(c))
            scale = max(np.abs(s[fin]).max() if fin.any() else 1.0, 1e-3 * np.abs(G).max(), 1e-30)
            d = np.where(fin, np.abs(np.where(fin, s, 0) - np.where(fin, c, 0)), 0.0) / scale
            if idx == 0:
                # d loss / d A of ABSENT edges (A = 0) weighs the adjoint of states that are improbable where the
                # edge would lead to them; across chunk boundaries that adjoint travels through the chunk
                # operators, whose eps floors are additive — with emissions of ~1e-10 on the way it can be off
                # by a percent of the entry.  The reference never reads these entries (A is scattered from
                # per-edge parameters); held to 2e-2 of the largest entry, present edges to `tol`.
                absent_err = float(d[0][A == 0].max()) if (A == 0).any() else 0.0
                d = d[0][A > 0] if (A > 0).any() else np.zeros(1)
            errs.append(float(d.max()) if same_nonfinite else np.inf)
        if q == 1:                    # gamma = 1 identically: every gradient is rounding noise around zero (times 1 / E)
            errs, absent_err = [0.0, 0.0, 0.0], 0.0
        ok = max(errs) <= tol and absent_err <= 2e-2
        if only is not None:
            from oracle import torch64
            for sq in range(b):
                rA, rpi, rE, _ = torch64.posterior_grad(A, pi, E[sq:sq + 1], G[sq:sq + 1], log=(mode == engine.POST_LOG))
                print("  sequence", sq, "max |dE| %.3g" % np.abs(rE).max(),
                      "serial dE err %.2e" % (np.abs(res[0][2][0, sq] - rE[0]).max() / np.abs(rE).max()),
                      "chunked dE err %.2e" % (np.abs(res[1][2][0, sq] - rE[0]).max() / np.abs(rE).max()))
            rA, rpi, rE, _ = torch64.posterior_grad(A, pi, E, G, log=(mode == engine.POST_LOG))
            for how in (0, 1):
                err = np.abs(res[how][0][0] - rA)
                i, j = np.unravel_index(err.argmax(), err.shape)
                print("  how", how, "dA err %.2e of max %.3g; worst entry (%d,%d) A=%.3g got %.6g want %.6g" % (err.max() / np.abs(rA).max(), np.abs(rA).max(), i, j, A[i, j], res[how][0][0][i, j], rA[i, j]))
        bad += not ok
        if verbose or not ok:
            print("case %3d kind %d q %2d b %2d L %4d chunk %3d emis %d G %d mode %d: redone serially %2d/%2d  dA %.1e (absent edges %.1e) dpi %.1e dE %.1e %s"
                  % (case, kind, q, b, L, chunk, emis, gk, mode, nser, b, errs[0], absent_err, errs[1], errs[2], "" if ok else "FAIL"), flush=True)
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    only = int(sys.argv[3]) if len(sys.argv) > 3 and int(sys.argv[3]) >= 0 else None
    shipped = int(sys.argv[4]) if len(sys.argv) > 4 else 1        # 2: per chunk for every sequence (no routing)
    bad = run(n, seed, only=only, shipped=shipped)
    print("%d cases, %d failures" % (n, bad))
    sys.exit(1 if bad else 0)
