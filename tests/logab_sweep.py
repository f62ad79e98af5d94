"""Randomised parity sweep of log alpha / log beta with the window recomputation engaged: gene-model input with local
stretches that every path survives only through the clamps or at the emission floor (states emitting alone for a few
positions, all-zero rows, dead columns), several per sequence, forced chunk lengths — hmm_forward / hmm_backward
against the fp64 serial recursion, every component, in probability space (tests/test_engine_gpu.py).  Test
infrastructure: tests/test_exact_gpu.py runs a few cases; for a longer run:  python tests/logab_sweep.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from hmm_layer_amd import engine
from oracle import params, textbook

dev = "cuda:0"
EXACT_MODE = int(os.environ.get("LOGAB_EXACT", "0"))           # 0 auto, 2 always (every sequence on the serial kernels)
if os.environ.get("LOGAB_LIB"):                                # another build of the engine (A/B of a change)
    engine.LIB_PATH = os.path.abspath(os.environ["LOGAB_LIB"])
def DETAIL(dims, op):
    try:
        return engine.exact_detail(dims, op=op)
    except Exception:
        return {"window_sequences": -1, "windows": -1, "whole": -1}
A15 = params.intended_A15().numpy().astype(np.float32)
if os.environ.get("LOGAB_K"):                                  # the k-copy gene model instead (29 / 43 states: other code paths)
    from hmm_layer_amd.gene_pred_hmm_transitioner import GenePredMultiHMMTransitioner
    _tr = GenePredMultiHMMTransitioner(k=int(os.environ["LOGAB_K"]), initial_exon_len=200, initial_intron_len=4500, initial_ir_len=10000)
    with torch.no_grad():
        A15 = _tr.make_A()[0].numpy().astype(np.float32)
Q = A15.shape[0]


def t(x):
    return torch.as_tensor(np.asarray(x), dtype=torch.float32, device=dev)


def close(x, x64):
    ref = x64.max(-1, keepdims=True)
    with np.errstate(over="ignore", under="ignore"):
        p, p64 = np.exp(np.minimum(x - ref, 50.0)), np.exp(x64 - ref)
    return float((np.abs(p - p64) - (2e-5 + p64 * (3e-4 + 2e-7 * np.abs(ref)))).max())


def run(ncase, seed, verbose=True):
    rng = np.random.default_rng(seed)
    bad = 0
    for case in range(ncase):
        b = int(rng.integers(1, 7))
        L = int(rng.choice([700, 3000, 9000, 20000]))
        chunk = int(rng.choice([0, 0, 16, 48, 128]))
        E = (rng.random((b, L, Q)) * 0.9 + 0.05).astype(np.float32)
        if rng.random() < 0.5:
            E /= 4096
        nst = 0
        for s in range(b):
            for _ in range(int(rng.integers(0, 5))):
                t0 = int(rng.integers(1, L - 8)); n = int(rng.integers(1, 7)); kind = rng.integers(0, 3)
                if kind == 0:                                  # one state emits alone
                    j = int(rng.integers(0, Q)); v = E[s, t0:t0 + n, j].copy(); E[s, t0:t0 + n] = 0.0; E[s, t0:t0 + n, j] = v
                elif kind == 1:                                # nothing emits at all
                    E[s, t0:t0 + n] = 0.0
                else:                                          # the intergenic / intron / exon states are dead
                    E[s, t0:t0 + n, :Q // 2] = 0.0
                nst += 1
        pi = np.full(Q, 1 / Q, dtype=np.float32)
        la64, ll64 = textbook.log_alpha(A15, pi, E)
        lb64 = textbook.log_beta(A15, E)
        with engine.option(engine.OPT_CHUNK, chunk), engine.option(engine.OPT_EXACT, EXACT_MODE):
            la, ll = engine.forward(t(A15)[None], t(pi), t(E[None]))
            da = DETAIL((1, b, L, Q), engine.OP_FORWARD)
            lb = engine.backward(t(A15)[None], t(E[None]))
            db = DETAIL((1, b, L, Q), engine.OP_BACKWARD)
        if os.environ.get("LOGAB_CASE") and int(os.environ["LOGAB_CASE"]) == case:
            T = engine.lib().hmm_chunk_len(1, b, L, 15) if chunk == 0 else chunk
            C = (L + T - 1) // T
            for name, arr, ref, op in (("log beta", lb.cpu().numpy()[0], lb64, engine.OP_BACKWARD), ("log alpha", la.cpu().numpy()[0], la64, engine.OP_FORWARD)):
                if op == engine.OP_FORWARD:
                    with engine.option(engine.OPT_CHUNK, chunk):
                        engine.forward(t(A15)[None], t(pi), t(E[None]))
                for sq in range(b):
                    d = arr[sq] - ref[sq]
                    m = ref[sq] > ref[sq].max(-1, keepdims=True) - 20
                    med = np.nanmedian(np.where(m, d, np.nan), axis=-1)
                    per = [round(float(np.median(med[c * T:(c + 1) * T])), 3) for c in range(C)]
                    chg = [(c, per[c]) for c in range(C) if c == 0 or abs(per[c] - per[c - 1]) > 3e-3]
                    with engine.option(engine.OPT_CHUNK, chunk):
                        wt = engine.window_table((1, b, L, 15), sq, op=op)
                    hot = [(c, float(v)) for c, v in enumerate(wt["psi"]) if v > 1e-7]
                    zer = sorted(set((np.nonzero((E[sq] == 0).sum(-1) >= 8)[0] // T).tolist()))
                    refm = ref[sq].max(-1, keepdims=True)
                    with np.errstate(over="ignore", under="ignore"):
                        pp, p64 = np.exp(np.minimum(arr[sq] - refm, 50.0)), np.exp(ref[sq] - refm)
                    ex = (np.abs(pp - p64) - (2e-5 + p64 * (3e-4 + 2e-7 * np.abs(refm)))).max(-1)
                    badt = np.nonzero(ex > 0)[0]
                    if len(badt):
                        print("  ", name, "seq", sq, "rows over tolerance:", len(badt), "first", badt[:8].tolist(), "chunks", sorted(set((badt // T).tolist()))[:8])
                        for tt in badt[:2]:
                            print("     t", tt, "E row", np.round(E[sq, tt], 3).tolist())
                            print("     ours", np.round(arr[sq, tt] - refm[tt], 2).tolist())
                            print("     ref ", np.round(ref[sq, tt] - refm[tt], 2).tolist())
                    print(name, "seq", sq, "T", T, "offset changes", chg[:12], "| windows", wt["windows"], "shifts", [round(v, 3) for v in wt["shifts"]], "| hot", hot[:12], "| chunks with stretches", zer)
        ea, eb = close(la.cpu().numpy()[0], la64), close(lb.cpu().numpy()[0], lb64)
        el = float(np.max(np.abs(ll.cpu().numpy()[0] - ll64) - (1e-6 * np.abs(ll64) + 2e-4)))
        ok = ea <= 0 and eb <= 0 and el <= 0
        bad += not ok
        if verbose or not ok:
            print("%s case %d b=%d L=%d chunk=%d stretches=%d  log alpha %+.1e %s  log beta %+.1e %s  loglik %+.1e" % (
                "ok  " if ok else "FAIL", case, b, L, chunk, nst, ea, {k: da[k] for k in ("window_sequences", "windows", "whole")},
                eb, {k: db[k] for k in ("window_sequences", "windows", "whole")}, el), flush=True)
    return bad


if __name__ == "__main__":
    nbad = run(int(sys.argv[1]) if len(sys.argv) > 1 else 30, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print("failures:", nbad)
