"""Randomised parity sweep (engine vs the C / numpy / torch-fp64 oracles) over shapes, chunk lengths
and models (gene topology, dense, sparse incl. reducible chains, degenerate: identity / cycle / block
diagonal / zero rows).  Test infrastructure: tests/test_stress_gpu.py runs 50 cases of it; for a longer run on
the GPU box, from the repo root:  python tests/stress_sweep.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from hmm_layer_amd import engine
from oracle import build as obuild, params, textbook, torch64

dev = "cuda:0"
A15 = params.intended_A15().numpy().astype(np.float32)


def _two_copy():
    from hmm_layer_amd.gene_pred_hmm_transitioner import GenePredMultiHMMTransitioner
    tr = GenePredMultiHMMTransitioner(k=2, initial_exon_len=200, initial_intron_len=4500, initial_ir_len=10000)
    with torch.no_grad():
        return tr.make_A()[0].numpy().astype(np.float32)


A29 = _two_copy()


def t(x):
    return torch.as_tensor(np.asarray(x), dtype=torch.float32, device=dev)


def run(ncase, seed, verbose=True):
  """-> number of failing cases (also used by tests/test_stress_gpu.py)."""
  rng = np.random.default_rng(seed)
  bad = 0
  old_chunk = engine.get_option(engine.OPT_CHUNK)
  for case in range(ncase):
      kind = rng.integers(0, 5)
      if kind == 0:
          q = 15; A = A15.copy(); pi = np.full(15, 1 / 15, np.float32)
      elif kind == 4:                  # the 29-state two-copy gene model: the chunked 32-state scan
          q = 29; A = A29.copy(); pi = rng.random(q).astype(np.float32) + 0.1; pi /= pi.sum()
      else:
          q = int(rng.integers(1, 17)) if rng.random() < 0.7 else int(rng.integers(17, 65))      # 17..64: one wave per sequence
          A = rng.random((q, q)).astype(np.float32) ** 3 + 1e-3
          if kind == 2:
              # no ring: reducible chains, states without incoming edges and the like are drawn too; the
              # engine finds them on the device and serves them with the serial exact-clamp kernels
              A *= rng.random((q, q)) < 0.4
              A += np.eye(q, dtype=np.float32) * 0.3
          if kind == 3:
              # the degenerate ends: identity, a pure cycle (irreducible but periodic), block diagonal,
              # absorbing / all-zero rows as in the reference's as-shipped matrices
              sub = rng.integers(0, 4)
              if sub == 0: A = np.eye(q, dtype=np.float32)
              elif sub == 1: A = np.roll(np.eye(q, dtype=np.float32), 1, axis=1)
              elif sub == 2: A[: q // 2, q // 2:] = 0; A[q // 2:, : q // 2] = 0
              else: A[rng.random(q) < 0.3] = 0
          A /= np.maximum(A.sum(-1, keepdims=True), 1e-30)
          pi = rng.random(q).astype(np.float32) + 0.1; pi /= pi.sum()
      b = int(rng.integers(1, 30)); L = int(rng.choice([1, 2, 17, 100, 999, 2500, 6001]))
      chunk = int(rng.choice([0, 16, 32, 48, 64, 128]))
      engine.set_option(engine.OPT_CHUNK, chunk)
      E = (rng.random((b, L, q)) * 0.9 + 0.05).astype(np.float32)
      clampy = rng.random() < 0.3
      if clampy:                      # zero emissions: whole stretches survive only through the eps clamps
          E[rng.random(E.shape) < 0.1] = 0.0
          E[..., 0] = np.maximum(E[..., 0], 0.05)
      g64, ll64 = obuild.posterior(A, pi, E)
      gam, ll = engine.posterior(t(A)[None], t(pi)[None], t(E)[None])
      e1 = np.abs(gam.cpu().numpy()[0] - g64).max()
      e2 = np.max(np.abs(ll.cpu().numpy()[0] - ll64) / (1e-6 * np.abs(ll64) + 2e-4))
      with np.errstate(divide="ignore"):
          logA, logpi, logE = np.log(A), np.log(pi), np.log(np.maximum(E, 1e-16)).astype(np.float32)
      wp, ws = obuild.viterbi(logA, logpi, logE)
      path, score = engine.viterbi(t(logA)[None], t(logpi)[None], t(logE)[None])
      vit_ok = np.array_equal(path.cpu().numpy()[0], wp) and np.array_equal(score.cpu().numpy()[0], ws)
      e3 = 0.0
      if L <= 999:
          w = (rng.random(b) + 0.5).astype(np.float32)
          dA, dpi, dE, _ = engine.loglik_grad(t(A)[None], t(pi)[None], t(E)[None], t(w)[None])
          rA, rpi, rE = textbook.loglik_grad(A, pi, E, w)
          m = A > 0
          e3 = max((np.abs(dA.cpu().numpy()[0] - rA)[m].max() if m.any() else 0.0) / max(np.abs(rA).max(), 1e-30),
                   np.abs(dE.cpu().numpy()[0] - rE).max() / max(np.abs(rE).max(), 1e-30))
      if L <= 100 and not clampy:                       # gradients of the posteriors against fp64 autograd
          Gup = rng.standard_normal((b, L, q)).astype(np.float32)
          mode = engine.POST_LOG if rng.random() < 0.5 else engine.POST_PROB
          pA, ppi, pE = engine.posterior_grad(t(A)[None], t(pi)[None], t(E)[None], t(Gup)[None], mode=mode)
          qA, qpi, qE, _ = torch64.posterior_grad(A, pi, E, Gup, log=(mode == engine.POST_LOG))
          # (with q = 1 the posterior is identically 1 and its gradient 0: absolute floor on the scale: fp32 noise of a few 1e-6 against upstream gradients of order 1)
          e3 = max(e3, np.abs(pA.cpu().numpy()[0] - qA).max() / max(np.abs(qA).max(), 1e-2),
                   np.abs(pE.cpu().numpy()[0] - qE).max() / max(np.abs(qE).max(), 1e-2),
                   np.abs(ppi.cpu().numpy()[0] - qpi).max() / max(np.abs(qpi).max(), 1e-2))
      # posteriors, log-likelihoods and gradients hold the normal tolerances on every input: eps-dominated
      # sequences are found on the device by every entry point and recomputed serially
      ok = e1 <= 2e-5 and e2 <= 1.0 and vit_ok and e3 <= 3e-4
      bad += (not ok)
      if verbose or not ok: print("%3d kind=%d%s q=%2d b=%2d L=%4d chunk=%3d  post %.1e  ll %.2f  vit %s  grad %.1e  %s" % (
          case, kind, "z" if clampy else " ", q, b, L, chunk, e1, e2, vit_ok, e3, "ok" if ok else "FAIL"), flush=True)
  engine.set_option(engine.OPT_CHUNK, old_chunk)
  return bad


if __name__ == "__main__":
    nbad = run(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print("failures:", nbad)
    sys.exit(1 if nbad else 0)
