import sys, os, ctypes, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from hmm_layer_amd import engine
from oracle import params, textbook
dev = 'cuda:0'
rng = np.random.default_rng(77)
A = params.intended_A15().numpy(); pi = np.full(15, 1 / 15, dtype=np.float32)
b, L = 9, 24000
E = (rng.random((b, L, 15)) * 0.9 + 0.05).astype(np.float32)
hard = {2: 5000, 5: 17003, 7: 23990}
for s, t0 in hard.items():
    E[s, t0:t0 + 4, :] = 0.0; E[s, t0:t0 + 4, 9] = 0.5
E[5, 300:303, :] = 0.0; E[5, 300:303, 12] = 0.7
t_ = lambda x: torch.as_tensor(x, device=dev)
lb = engine.backward(t_(A)[None], t_(E[None]))
torch.cuda.synchronize()
print(engine.exact_detail((1, b, L, 15), op=engine.OP_BACKWARD))
T = engine.lib().hmm_chunk_len(1, b, L, 15); C = (L + T - 1) // T
lbn = lb.cpu().numpy()[0]
lb64 = textbook.log_beta(A, E)
L_ = engine.lib()
L_.hmm_window_table.restype = ctypes.c_int
L_.hmm_window_table.argtypes = [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
ws = engine._workspaces[(0, torch.cuda.current_stream().cuda_stream)]
print("T", T, "C", C)
for s in range(b):
    d = lbn[s] - lb64[s]
    m = lb64[s] > lb64[s].max(-1, keepdims=True) - 20
    med = np.nanmedian(np.where(m, d, np.nan), axis=-1)
    per = [round(float(np.median(med[c * T:(c + 1) * T])), 3) for c in range(C)]
    chg = [(c, per[c]) for c in range(C) if c == 0 or abs(per[c] - per[c - 1]) > 1e-2]
    tab = (ctypes.c_int * 34)(); sh = (ctypes.c_double * 24)(); ps = (ctypes.c_float * 2048)()
    L_.hmm_window_table(engine.OP_BACKWARD, 1, b, L, 15, ws.data_ptr(), ws.numel(), s, tab, sh, ps, 2048)
    ints = (ctypes.c_int * 16).from_buffer(sh, 16 * 8)
    hot = [(c, float(ps[c])) for c in range(C) if ps[c] > 1e-7]
    print("seq", s, "offset changes at chunks", chg, "| table", list(tab)[:2 + 2 * max(tab[0], 0)] if 0 <= tab[0] <= 16 else "?", "shifts", [round(v, 3) for v in list(sh)[:max(min(tab[0], 16), 0)]], "lo", list(ints)[:max(min(tab[0], 16), 0)], "hot", hot[:6])

with engine.option(engine.OPT_EXACT, engine.EXACT_OFF):
    lbo = engine.backward(t_(A)[None], t_(E[None])).cpu().numpy()[0]
for s in (2, 5, 7):
    d = lbo[s] - lb64[s]
    m = lb64[s] > lb64[s].max(-1, keepdims=True) - 20
    med = np.nanmedian(np.where(m, d, np.nan), axis=-1)
    per = [round(float(np.median(med[c * T:(c + 1) * T])), 3) for c in range(C)]
    chg = [(c, per[c]) for c in range(C) if c == 0 or abs(per[c] - per[c - 1]) > 1e-2]
    print("routing off: seq", s, "offset changes at chunks", chg)
    t0 = hard[s]
    print("   rows around the stretch:", [(t, round(float(med[t]), 3)) for t in range(t0 - 3, t0 + 6)])

# fp64 model of CERT3 over chunk 354 of seq 5: R and its clamp-born part G, from the exact R at the chunk's top
EPS = 1e-16
s5 = 5
Ec = np.maximum(E[s5].astype(np.float64), EPS)
A64 = A.astype(np.float64)
# exact R for the whole sequence (normalised bh convention)
Rv = np.ones(15); Rs = np.empty((L, 15)); 
for t in range(L - 1, -1, -1):
    Rs[t] = Rv
    sf = Ec[t] * Rv; bh = sf / sf.sum()
    Rv = np.maximum(A64 @ bh, EPS)
for c in (354, 104):
    sq = 5 if c == 354 else 2
    Ec2 = np.maximum(E[sq].astype(np.float64), EPS)
    Rv = np.ones(15); Rl = {}
    for t in range(L - 1, -1, -1):
        Rl[t] = Rv
        sf = Ec2[t] * Rv; bh = sf / sf.sum(); Rv = np.maximum(A64 @ bh, EPS)
    top = min(L, (c + 1) * T) - 1
    R = Rl[top].copy(); G = np.zeros(15)
    for t in range(top, c * T - 1, -1):
        sf = Ec2[t] * R; S = sf.sum(); U = A64 @ (sf / S); Ug = A64 @ (Ec2[t] * G / S)
        G = np.where(U > EPS, Ug, EPS); R = np.maximum(U, EPS)
    print("fp64 model: seq", sq, "chunk", c, "clamp-born share of R at the chunk's bottom:", G.sum() / R.sum())
ps = (ctypes.c_float * 2048)(); tab = (ctypes.c_int * 34)()
L_.hmm_window_table(engine.OP_BACKWARD, 1, b, L, 15, ws.data_ptr(), ws.numel(), 5, tab, None, ps, 2048)
print("engine psi seq 5 chunks 350..357:", [float(ps[c]) for c in range(350, 358)])
