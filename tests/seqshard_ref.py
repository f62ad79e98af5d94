"""fp64 numpy restatement of the two LOCAL steps of the sequence-sharded posterior (test infrastructure:
drives hmm_layer_amd/seqshard.py's exchange on CPU tensors over gloo, tests/test_seqshard_cpu.py).
Same interface and operator format as the HIP engine's hmm_seqshard_* entry points
(include/hmm_engine.h): op[i][k] = X[i][k] * 2^-exp[k], X = product of the slab's step matrices,
i = state at the slab's last position, k = state just before the slab."""
import numpy as np
import torch

EPS = 1e-16


def _rescale(X, ex):
    s = X.sum(0)
    e = np.where(s > 0, np.frexp(np.where(s > 0, s, 1.0))[1], 0)
    return X * np.ldexp(1.0, -e)[None, :], ex + e


class RefBackend:
    def reduce(self, A, E_slab, seq_start, R):
        A, E = A.numpy().astype(np.float64), E_slab.numpy().astype(np.float64)
        k, b, L, q = E.shape
        op = np.zeros((k, b, 16, 16), np.float32)
        ex = np.zeros((k, b, 16), np.int32)
        for m in range(k):
            for s in range(b):
                X, e = np.eye(q), np.zeros(q, np.int64)
                for t in range(L):
                    em = np.maximum(E[m, s, t], EPS)
                    if t == 0 and seq_start:
                        X = em[:, None] * X
                    else:
                        X = em[:, None] * (A[m].T @ X)
                    X, e = _rescale(X, e)
                op[m, s, :q, :q] = X
                ex[m, s, :q] = e
        return torch.from_numpy(op), torch.from_numpy(ex)

    def posterior(self, A, pi, E_slab, all_ops, all_exps, r, mode):
        assert mode == 0
        A, E = A.numpy().astype(np.float64), E_slab.numpy().astype(np.float64)
        pi = pi.numpy().astype(np.float64).reshape(A.shape[0], -1)
        ops, exs = all_ops.numpy().astype(np.float64), all_exps.numpy().astype(np.int64)
        k, b, L, q = E.shape
        R = ops.shape[2]
        out = np.zeros((k, b, L, q), np.float32)
        ll = np.zeros((k, b))
        phi = np.zeros((k, b), np.float32)
        for m in range(k):
            for s in range(b):
                X = [ops[m, s, j, :q, :q] * np.ldexp(1.0, exs[m, s, j, :q])[None, :] for j in range(R)]
                # alpha_hat entering slab r, log-likelihood of the whole sequence
                a, tot = np.maximum(pi[m], EPS), 0.0
                ent = None
                for j in range(R):
                    if j == r:
                        ent = a.copy()
                    a = X[j] @ a
                    tot += np.log(a.sum())
                    a = a / a.sum()
                # beta leaving slab r
                v = np.ones(q)
                for j in range(R - 1, r, -1):
                    v = X[j].T @ v
                    v = v / v.max()
                # serial cell recursion on the slab
                ah = np.zeros((L, q))
                fm = np.zeros((L, q), bool)                              # forward prediction sat at the clamp
                x = ent
                for t in range(L):
                    em = np.maximum(E[m, s, t], EPS)
                    first = t == 0 and r == 0
                    pred = x if first else x @ A[m]
                    fm[t] = (pred <= EPS) & (not first)
                    sf = em * np.maximum(pred, EPS)
                    x = sf / sf.sum()
                    ah[t] = x
                Rv, acc = v, 0.0
                bm = np.zeros(q, bool)
                for t in range(L - 1, -1, -1):
                    g = ah[t] * Rv
                    g = g / g.sum()
                    acc += g[fm[t]].sum() + g[bm].sum()                  # psi: posterior mass on clamp-born components
                    out[m, s, t] = g
                    bh = np.maximum(E[m, s, t], EPS) * Rv
                    u = A[m] @ (bh / bh.sum())
                    bm = u <= EPS
                    Rv = np.maximum(u, EPS)
                ll[m, s] = tot
                phi[m, s] = acc
        return torch.from_numpy(out), torch.from_numpy(ll), torch.from_numpy(phi)

    def unsharded(self, A, pi, E, mode):
        """The unsharded call of gather_flagged: the serial fp64 recursion with the cell's clamps."""
        assert mode == 0
        from oracle import textbook
        g, ll = textbook.posterior(A[0].numpy(), pi[0].numpy(), E[0].numpy())
        return torch.from_numpy(g.astype(np.float32))[None], torch.from_numpy(ll)[None]
