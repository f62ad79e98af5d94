import sys, os, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from hmm_layer_amd import engine
from oracle import params, textbook
dev = 'cuda:0'
rng = np.random.default_rng(12)
A = params.intended_A15().numpy(); pi = np.full(15, 1 / 15, dtype=np.float32)
b, L = 12, 1500
E = (rng.random((b, L, 15)) * 0.9 + 0.05).astype(np.float32)
hard = np.arange(b) % 2 == 1
Eh = E[hard]; Eh[rng.random(Eh.shape) < 0.25] = 0.0; E[hard] = Eh
t = lambda x: torch.as_tensor(x, device=dev)
t_ = t
lb = engine.backward(t(A)[None], t(E[None])).cpu().numpy()[0]
det = engine.exact_detail((1, b, L, 15), op=engine.OP_BACKWARD)
print(det, "T =", engine.lib().hmm_chunk_len(1, b, L, 15))
lb64 = textbook.log_beta(A, E)
T = engine.lib().hmm_chunk_len(1, b, L, 15)
for s in range(b):
    d = (lb[s] - lb64[s])
    m = lb64[s] > lb64[s].max(-1, keepdims=True) - 20          # components that carry weight
    err = np.where(m, np.abs(d), 0).max(-1)                     # per position
    badpos = np.nonzero(err > 1e-2)[0]
    if len(badpos):
        ch = sorted(set((badpos // T).tolist()))
        off = np.where(m, d, np.nan)
        med = np.nanmedian(off, axis=-1)
        print("seq", s, "bad positions", len(badpos), "chunk range", ch[0], ch[-1], "of", (L + T - 1) // T)
        print("   median offset per chunk:", [round(float(np.median(med[c * T:(c + 1) * T])), 2) for c in range(0, (L + T - 1) // T)])
    else:
        print("seq", s, "ok")

s = 1
d = lb[s] - lb64[s]
m = lb64[s] > lb64[s].max(-1, keepdims=True) - 20
off = np.where(m, d, np.nan)
for t in range(73 * T - 4, 75 * T):
    print(t, t // T, np.round(np.nanmedian(off[t]), 3), "zeros in E row:", int((E[s, t] == 0).sum()), "row max lb64", round(float(lb64[s, t].max()), 2))
with engine.option(engine.OPT_EXACT, engine.EXACT_ALWAYS):
    lbw = engine.backward(t_(A)[None], t_(E[None])).cpu().numpy()[0]
print("whole-sequence routing: max |offset| seq 1:", float(np.nanmax(np.abs(np.where(m, lbw[1] - lb64[1], np.nan)))))

import ctypes
L_ = engine.lib()
L_.hmm_window_table.restype = ctypes.c_int
L_.hmm_window_table.argtypes = [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
lb = engine.backward(t_(A)[None], t_(E[None]))
torch.cuda.synchronize()
key = (0, torch.cuda.current_stream().cuda_stream)
ws = engine._workspaces[key]
for sq in (1, 7):
    tab = (ctypes.c_int * 34)(); sh = (ctypes.c_double * 24)(); ps = (ctypes.c_float * 128)()
    C = L_.hmm_window_table(engine.OP_BACKWARD, 1, b, L, 15, ws.data_ptr(), ws.numel(), sq, tab, sh, ps, 128)
    ints = (ctypes.c_int * 16).from_buffer(sh, 16 * 8)
    print("seq", sq, "C", C, "table", list(tab)[:2 + 2 * tab[0]], "shifts", [round(v, 3) for v in list(sh)[:tab[0]]], "lo", list(ints)[:tab[0]])
    print("   psi > 1e-7:", [(c, float(ps[c])) for c in range(C) if ps[c] > 1e-7])

for name, arr in (("windows", lb.cpu().numpy()[0]), ("whole", lbw)):
    d = arr[1] - lb64[1]
    m = lb64[1] > lb64[1].max(-1, keepdims=True) - 20
    err = np.where(m, np.abs(d), 0).max(-1)
    bp = np.nonzero(err > 1e-2)[0]
    print(name, "seq 1 bad positions", bp.tolist(), [round(float(err[t]), 3) for t in bp])
    for t in bp[:2]:
        print("   t", t, "ours", np.round(arr[1, t], 2).tolist()); print("   t", t, "ref ", np.round(lb64[1, t], 2).tolist())
